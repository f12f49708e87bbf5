"""MFMA-busy summary of one training step from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 tools/train_bench.py --layers 4 8 --steps 1
   python tools/pmc_train.py DIR/*/*counter_collection.csv"""
import csv, json, sys
from collections import defaultdict


def classify(n):
    for key, c in (("attn_bwd_kernel<128, 0", "attn_bwd_lse"), ("attn_bwd_kernel<128, 1", "attn_bwd_dq"), ("attn_bwd_kernel<128, 2", "attn_bwd_dk"),
                   ("attn_bwd_kernel<128, 3", "attn_bwd_dv"), ("flash_attn_kernel", "attn_fwd"), ("gemm256_kernel", "gemm256"), ("gemm128_kernel", "gemm128"),
                   ("transpose", "transpose"), ("colsum", "colsum")):
        if key in n:
            return c
    return None


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        c = classify(r["Kernel_Name"])
        if c is None:
            continue
        a = acc[c][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for c, d in acc.items():
    if "GRBM_GUI_ACTIVE" not in d:
        continue
    bs = d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0, 0, 0.0])[0]
    gs, gn, ns = d["GRBM_GUI_ACTIVE"]
    out[c] = dict(launches=gn, total_ms=round(ns / 1e6, 2), avg_us=round(ns / gn / 1e3, 1), mfma_busy=round(bs / (gs / 8.0 * 1024), 3), clock_ghz=round(gs / 8.0 / ns, 2))
print("PMC_TRAIN", json.dumps(out))
