"""Energy ledger: joules per algorithmic FLOP and held shader clock of each kernel class under the package power cap.

    python tools/energy_ledger.py [--seconds 2.5] [--out profiles/r05_energy.json] [--only gemm attn vendor pwg2]

The forward runs at the 1400 W package cap (bench.py `power`), so what a kernel structure costs is joules, not idle cycles: a structure that
does the same FLOPs for fewer joules holds a higher clock under the cap. For each class this tool runs a SUSTAINED loop (>= --seconds of
back-to-back launches, no host sync inside, the buffers of the cfg2 / cfg5 forward's own shapes, >= 1 s of the same loop first so the
clock has settled) and samples the amdgpu hwmon files of the device around it on a side thread (power1_input every 10 ms; energy from the
trapezoid of those samples - this hwmon has no energy counter file; freq1_input = shader clock). Reported per class:

    tflops            algorithmic FLOPs / wall seconds of the loop (HIP events around it)
    watts             mean package power inside the loop;  watts_over_idle = minus the idle reading taken at the start
    pj_per_flop       watts / (FLOP/s) * 1e12 (whole package: HBM, fabric and idle included - what the cap meters)
    pj_per_flop_net   the same with the idle power subtracted
    sclk_mhz          median shader clock inside the loop

Classes: the product GEMM (ug_gemm_bf16, gemm256_kernel) on the cfg2 launches per epilogue class, hipBLASLt (torch F.linear, bias epilogue) on the
same shapes, the probe library's one-wave-per-SIMD kernel gemm_pwg2 (UG_GEMM_PWG=4, bias epilogue; needs tools/probe/libunigen_hip_probe.so
and runs in a child process, because the product library is loaded once per process), flash_attn_kernel<128> (cfg2: B=4, H=24, L=4608) and
flash_attn_kernel<64> (cfg5: B=16, H=24, L=4429 joint / 4096 self), and the bare MFMA probe (ug_probe_mfma_bf16: the matrix pipe alone).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


class Hwmon:
    def __init__(self, dev):
        self.dir = None
        pr = torch.cuda.get_device_properties(dev)
        want = (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
        base = "/sys/class/drm"
        for card in sorted(os.listdir(base)):
            if not card.startswith("card") or "-" in card:
                continue
            real = os.path.realpath(os.path.join(base, card, "device"))
            parts = os.path.basename(real).replace(".", ":").split(":")
            if len(parts) != 4 or (int(parts[0], 16), int(parts[1], 16), int(parts[2], 16)) != want:
                continue
            hw = os.path.join(real, "hwmon")
            for h in sorted(os.listdir(hw)):
                if os.path.exists(os.path.join(hw, h, "power1_input")):
                    self.dir = os.path.join(hw, h)
        if self.dir is None:
            raise SystemExit("no amdgpu hwmon directory for this device: nothing to meter")
        self.files = sorted(os.listdir(self.dir))

    def read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def sample(self):
        return time.perf_counter(), self.read("power1_input") / 1e6, self.read("freq1_input") / 1e6


class Meter:
    def __init__(self, hw: Hwmon, period=0.01):
        self.hw, self.period, self.samples, self._stop = hw, period, [], threading.Event()
        self._t = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append(self.hw.sample())
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=2)

    def window(self, t0, t1):
        s = [x for x in self.samples if t0 <= x[0] <= t1]
        if len(s) < 3:
            return None
        joules = sum(0.5 * (a[1] + b[1]) * (b[0] - a[0]) for a, b in zip(s, s[1:]))
        span = s[-1][0] - s[0][0]
        f = sorted(x[2] for x in s)
        return dict(watts=joules / span, sclk_mhz=f[len(f) // 2], samples=len(s), watts_max=max(x[1] for x in s))


def sustained(fn, flops_per_call, seconds, hw, idle_w):
    """>= 1 s settle + >= `seconds` measured; returns the ledger row."""
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    per = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
    n_settle, n = max(1, int(1.0 / per)), max(2, int(seconds / per))
    for _ in range(n_settle):
        fn()
    torch.cuda.synchronize()
    with Meter(hw) as m:
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        t1 = time.perf_counter()
    wall = e0.elapsed_time(e1) * 1e-3
    w = m.window(t0 + 0.1 * (t1 - t0), t1)          # the first tenth still holds launch ramp / the settle loop's tail
    rate = flops_per_call * n / wall
    row = dict(calls=n, seconds=wall, tflops=rate / 1e12)
    if w:
        row.update(watts=w["watts"], watts_over_idle=w["watts"] - idle_w, sclk_mhz=w["sclk_mhz"], power_samples=w["samples"], watts_max=w["watts_max"],
                   pj_per_flop=w["watts"] / rate * 1e12, pj_per_flop_net=(w["watts"] - idle_w) / rate * 1e12)
    return row


B, NI, T, D = 4, 4096, 512, 3072
GEMM_SHAPES = [      # label, M, N, K, epilogue class, launches per cfg2 forward (share of the GEMM FLOPs follows)
    ("single qkv+mlp (q/k rope | v | gelu)", B * (NI + T), 7 * D, D, "qkrope", 76),
    ("single out K=15360 res_gate", B * (NI + T), D, 5 * D, "res_gate", 76),
    ("ff up gelu", B * NI, 4 * D, D, "gelu", 40),
    ("ff down K=12288 res_gate", B * NI, D, 4 * D, "res_gate", 40),
    ("qkv image (q/k rope | v)", B * NI, 3 * D, D, "qkrope3", 40),
    ("attn out res_gate", B * NI, D, D, "res_gate", 40),
    ("zero-res res_scale", B * (NI + T), D, D, "res_scale", 38),
]


def gemm_rows(which, seconds, hw, idle_w, only_bias=False):
    from unigen_amd import lib as L, ops
    from unigen_amd.ops import QkRope
    dev, BF = torch.device("cuda:0"), torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)
    rows = {}
    for label, M, N, K, epi, count in GEMM_SHAPES:
        a, w, b = rn(M, K), rn(N, K, sc=0.03), rn(N, sc=0.1)
        out = torch.empty(M, N, device=dev, dtype=BF)
        rps = M // B
        if which == "vendor":
            fn = lambda: torch.addmm(b, a, w.t(), out=out)
            name = "hipBLASLt via torch.addmm (bias epilogue)"
        else:
            kw = dict(M=M)
            if only_bias:
                pass
            elif epi in ("qkrope", "qkrope3"):
                cs = torch.rand(rps, 64, 2, generator=g, device=dev) * 2 - 1
                kw.update(qk_rope=QkRope(rn(128) + 1, rn(128) + 1, cs.contiguous(), rps, 0, 2 * D, 1e-6, 128))
                if epi == "qkrope":
                    kw.update(gelu_from_n=3 * D)
            elif epi == "gelu":
                kw.update(epilogue=L.EPI_BIAS_GELU)
            elif epi == "res_gate":
                kw.update(epilogue=L.EPI_RES_GATE, residual=rn(M, N), alpha=0.5, gate=rn(B, N), gate_ld=N, rows_per_sample=rps)
            elif epi == "res_scale":
                kw.update(epilogue=L.EPI_RES_SCALE, residual=rn(M, N), alpha=0.5)
            fn = lambda: ops.gemm(a, w, b, out, **kw)
            name = "ug_gemm_bf16" + (" (bias epilogue)" if only_bias else f" ({epi})")
        r = sustained(fn, 2.0 * M * N * K, seconds, hw, idle_w)
        r.update(kernel=name, shape=f"{M}x{N}x{K}", launches_per_forward=count, flops_per_forward=2.0 * M * N * K * count)
        rows[label] = r
        print(f"[{which}{' bias' if only_bias else ''}] {label:40s} {r['tflops']:7.1f} TFLOP/s  {r.get('watts', 0):6.0f} W  {r.get('pj_per_flop', 0):.3f} pJ/FLOP  "
              f"{r.get('sclk_mhz', 0):.0f} MHz", flush=True)
        del a, w, b, out
    tot = sum(r["flops_per_forward"] for r in rows.values())
    if all("pj_per_flop" in r for r in rows.values()):
        mix = dict(pj_per_flop=sum(r["pj_per_flop"] * r["flops_per_forward"] for r in rows.values()) / tot,
                   pj_per_flop_net=sum(r["pj_per_flop_net"] * r["flops_per_forward"] for r in rows.values()) / tot,
                   tflops=tot / sum(r["flops_per_forward"] / r["tflops"] for r in rows.values()),
                   note="weighted by each class's FLOPs in one cfg2 forward (launch counts above)")
    else:
        mix = None
    return dict(classes=rows, forward_mix=mix)


def attn_rows(seconds, hw, idle_w):
    from unigen_amd import ops
    dev, BF = torch.device("cuda:0"), torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(1)
    rows = {}
    for label, Bq, H, dh, Lq, Lkv in (("flash_attn_kernel<128> cfg2 joint 4608", 4, 24, 128, 4608, 4608),
                                      ("flash_attn_kernel<128> cfg2 shared_expert[1] 8192x8704", 4, 24, 128, 8192, 8704),
                                      ("flash_attn_kernel<64> cfg5 joint 4429", 16, 24, 64, 4429, 4429),
                                      ("flash_attn_kernel<64> cfg5 attn2 4096", 16, 24, 64, 4096, 4096)):
        Dm = H * dh
        Lj = max(Lq, Lkv)
        qkv = (torch.randn(Bq * Lj, 3 * Dm, generator=g, device=dev) * 0.5).to(BF)
        out = torch.empty(Bq * Lq, Dm, device=dev, dtype=BF)
        st = (3 * Dm, Lj * 3 * Dm)
        fn = lambda: ops.flash_attn(qkv[Lj - Lq:], qkv[0, Dm:], qkv[0, 2 * Dm:], out, batches=Bq, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st,
                                    v_strides=st, o_strides=(Dm, Lq * Dm))
        r = sustained(fn, 4.0 * Bq * H * Lq * Lkv * dh, seconds, hw, idle_w)
        r.update(kernel=label, shape=f"B={Bq} H={H} dh={dh} Lq={Lq} Lkv={Lkv}")
        rows[label] = r
        print(f"[attn] {label:55s} {r['tflops']:7.1f} TFLOP/s  {r.get('watts', 0):6.0f} W  {r.get('pj_per_flop', 0):.3f} pJ/FLOP  {r.get('sclk_mhz', 0):.0f} MHz", flush=True)
        del qkv, out
    return dict(classes=rows)


def mfma_row(seconds, hw, idle_w):
    from unigen_amd import ops
    dev = torch.device("cuda:0")
    rows = {}
    for shape, nm in ((1, "16x16x32"), (0, "32x32x16")):
        # the probe reports its own rate; meter the power around repeated calls
        rate = ops.probe_mfma_peak(dev, shape=shape)
        with Meter(hw) as m:
            t0 = time.perf_counter()
            rates = []
            while time.perf_counter() - t0 < seconds + 1.0:
                rates.append(ops.probe_mfma_peak(dev, shape=shape))
            t1 = time.perf_counter()
        w = m.window(t0 + 1.0, t1)
        rate = sorted(rates)[len(rates) // 2]
        r = dict(kernel=f"ug_probe_mfma_bf16 {nm} (register operands, no memory traffic)", tflops=rate)
        if w:
            r.update(watts=w["watts"], watts_over_idle=w["watts"] - idle_w, sclk_mhz=w["sclk_mhz"], pj_per_flop=w["watts"] / (rate * 1e12) * 1e12,
                     pj_per_flop_net=(w["watts"] - idle_w) / (rate * 1e12) * 1e12,
                     note="the probe syncs between its launches: the power window includes those gaps, so pJ/FLOP here is an upper bound for the bare pipe")
        rows[nm] = r
        print(f"[mfma] {nm} {rate:7.1f} TFLOP/s {r.get('watts', 0):6.0f} W {r.get('pj_per_flop', 0):.3f} pJ/FLOP {r.get('sclk_mhz', 0):.0f} MHz", flush=True)
    return dict(classes=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_energy.json"))
    ap.add_argument("--only", nargs="*", default=["gemm", "gemm_bias", "vendor", "pwg2", "attn", "mfma"])
    ap.add_argument("--child-pwg2", action="store_true", help="internal: this process loaded the probe library with UG_GEMM_PWG=4")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    hw = Hwmon(dev)
    torch.zeros(1, device=dev); torch.cuda.synchronize()
    time.sleep(1.0)
    idle = [hw.sample() for _ in range(20) if not time.sleep(0.05)]
    idle_w = sum(x[1] for x in idle) / len(idle)
    cap = None
    try:
        cap = hw.read("power1_cap") / 1e6
    except Exception:
        pass
    if a.child_pwg2:
        res = gemm_rows("pwg2", a.seconds, hw, idle_w, only_bias=True)
        print("PWG2_JSON " + json.dumps(res), flush=True)
        return
    doc = dict(device=torch.cuda.get_device_name(dev), hwmon_files=hw.files, idle_watts=idle_w, cap_watts=cap, seconds_per_class=a.seconds,
               collected_utc=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
               method="sustained loops of back-to-back launches; power1_input / freq1_input every 10 ms on a side thread; energy = trapezoid of the power samples "
                      "over the last 90 % of the loop; pJ/FLOP = mean watts / algorithmic FLOP rate")
    print(f"idle {idle_w:.0f} W, cap {cap} W, hwmon files: {hw.files}", flush=True)
    if "gemm" in a.only:
        doc["product_gemm"] = gemm_rows("product", a.seconds, hw, idle_w)
    if "gemm_bias" in a.only:
        doc["product_gemm_bias_only"] = gemm_rows("product", a.seconds, hw, idle_w, only_bias=True)
    if "vendor" in a.only:
        doc["hipblaslt"] = gemm_rows("vendor", a.seconds, hw, idle_w)
    if "attn" in a.only:
        doc["attention"] = attn_rows(a.seconds, hw, idle_w)
    if "mfma" in a.only:
        doc["bare_mfma"] = mfma_row(a.seconds, hw, idle_w)
    if "pwg2" in a.only:
        probe = os.path.join(ROOT, "tools", "probe", "libunigen_hip_probe.so")
        if os.path.exists(probe):
            env = dict(os.environ, UG_LIB_PATH=probe, UG_GEMM_PWG="4")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-pwg2", "--seconds", str(a.seconds)], env=env, capture_output=True, text=True)
            sys.stdout.write("".join(ln + "\n" for ln in r.stdout.splitlines() if not ln.startswith("PWG2_JSON")))
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("PWG2_JSON ")]
            doc["probe_pwg2_bias_only"] = json.loads(js[0][len("PWG2_JSON "):]) if js else dict(error=r.stderr[-2000:])
        else:
            doc["probe_pwg2_bias_only"] = dict(error="tools/probe/libunigen_hip_probe.so not built (python -m unigen_amd.build --probe)")
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
