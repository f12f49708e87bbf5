"""Share of a rocprofv3 kernel_stats.csv's GPU time spent in torch's own kernels (at::native / rocclr copies), weight-initialisation kernels
(normal distribution, the float -> bf16 copies and scalings behind it) excluded. usage: python tools/native_share.py FILE"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t = lambda r: float(r["TotalDurationNs"]) / 1e6
init = [r for r in rows if "distribution" in r["Name"] or "bfloat16_copy_kernel_cuda" in r["Name"] or ("AUnaryFunctor<float, float, float" in r["Name"] and "MulFunctor" in r["Name"])]
nat = [r for r in rows if ("at::native" in r["Name"] or "rocclr" in r["Name"]) and r not in init]
tot = sum(t(r) for r in rows) - sum(t(r) for r in init)
print(f"GPU time {tot:.1f} ms (weight init {sum(t(r) for r in init):.1f} ms excluded); torch-native kernels {sum(t(r) for r in nat):.1f} ms = {sum(t(r) for r in nat) / tot:.2%}")
for r in sorted(nat, key=t, reverse=True)[:8]:
    print(f"  {t(r):8.1f} ms {r['Calls']:>6s}  {r['Name'][:110]}")
