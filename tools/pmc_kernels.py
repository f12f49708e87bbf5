"""Per-kernel sums of rocprofv3 counter_collection CSVs: python tools/pmc_kernels.py FILTER CSV [CSV ...]
Prints, per kernel whose name contains FILTER, launches, average duration and every counter averaged per launch (counters of all CSVs merged)."""
import csv, re, sys
from collections import defaultdict

flt = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))
for path in sys.argv[2:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if flt not in n: continue
            m = re.search(r"(\w+)<([^>]*)>", n)
            short = f"{m.group(1)}<{m.group(2)}>" if m else n[:60]
            a = acc[short][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc):
    any_c = next(iter(acc[k].values()))
    print(f"{k}: launches {any_c[1]}, avg {any_c[2] / any_c[1] / 1e3:.1f} us")
    for c in sorted(acc[k]):
        s, n, _ = acc[k][c]
        print(f"    {c:32s} {s / n:16.1f}")
