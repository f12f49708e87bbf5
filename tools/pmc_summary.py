"""Summarise rocprofv3 --pmc passes of bench.py into profiles/<tag>_pmc.json.

usage: python tools/pmc_summary.py TAG FETCH_CSV WRITE_CSV BUSY_CSV [TCC_CSV]
Each CSV is a *_counter_collection.csv of one pass (`rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py
--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timer`; counters in separate passes as MI355X_MICROARCH.md prescribes).
HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-counts wide coalesced reads by 2x; KiB units).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)  [busy cycles are summed over SIMDs];
effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time."""
import csv, json, sys
from collections import defaultdict


def classify(name: str):
    if "gemm256_kernel" in name: return "gemm256"
    if "gemm_pwg_kernel" in name: return "gemm_pwg"
    if "gemm128_kernel" in name: return "gemm128"
    if "flash_attn" in name: return "attn"
    return None


def load(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))     # class -> counter -> [sum, launches, ns]
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            c = classify(r["Kernel_Name"])
            if c is None: continue
            a = acc[c][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc


def main():
    tag, fcsv, wcsv, bcsv = sys.argv[1:5]
    F, W, Bz = load(fcsv), load(wcsv), load(bcsv)
    T = load(sys.argv[5]) if len(sys.argv) > 5 else {}
    import time
    out = {"collected_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "tag": tag,
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timer",
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction; KiB units); counters beyond the XCD L2 include Infinity-Cache hits",
           "kernels": {}}
    tot_b, tot_n = 0.0, 0
    for c in ("gemm256", "gemm128", "attn"):
        if c not in F: continue
        fs, fn, _ = F[c]["FETCH_SIZE"]; ws, wn, _ = W[c]["WRITE_SIZE"]
        k = {"launches": fn, "fetch_bytes_per_launch_corrected": 2 * fs * 1024 / fn, "write_bytes_per_launch": ws * 1024 / wn}
        k["hbm_bytes_per_launch"] = k["fetch_bytes_per_launch_corrected"] + k["write_bytes_per_launch"]
        if c in Bz and "SQ_VALU_MFMA_BUSY_CYCLES" in Bz[c]:
            bs, bn, bns = Bz[c]["SQ_VALU_MFMA_BUSY_CYCLES"]; gs, gn, _ = Bz[c]["GRBM_GUI_ACTIVE"]
            cyc = gs / 8.0                                             # per-XCD active cycles summed over launches
            k["mfma_busy_frac"] = bs / (cyc * 256 * 4) * 8 if False else bs / (gs / 8.0 * 1024)
            k["effective_clock_ghz"] = cyc / bns
            k["avg_launch_us_profiled"] = bns / bn / 1e3
        if c in T and "TCC_HIT_sum" in T[c]:
            hs, _, _ = T[c]["TCC_HIT_sum"]; ms, _, _ = T[c]["TCC_MISS_sum"]
            k["l2_hit_rate"] = hs / max(hs + ms, 1.0)
            k["l2_requests_per_launch"] = (hs + ms) / fn
        out["kernels"][c] = k
        if c.startswith("gemm"):
            tot_b += k["hbm_bytes_per_launch"] * fn; tot_n += fn
    out["kernels"]["gemm_all"] = {"launches": tot_n, "hbm_bytes_per_launch": tot_b / max(tot_n, 1)}
    path = f"profiles/{tag}_pmc.json"
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
