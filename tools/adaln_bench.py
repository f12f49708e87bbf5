"""ug_adaln_modulate at the cfg2 sizes: the generic kernel (UG_ADALN_FAST=0) vs the branch-free one, interleaved in one process. TB/s = (read + write) / time."""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unigen_amd import ops
dev, BF = torch.device("cuda:0"), torch.bfloat16
for rows, D, rps in [(16384, 3072, 4096), (18432, 3072, 4608), (32768, 3072, 8192), (65536, 1536, 4096)]:
    x = torch.randn(rows, D, device=dev).to(BF); emb = torch.randn(rows // rps, 6 * D, device=dev).to(BF)
    out = torch.empty(rows, D, device=dev, dtype=BF)
    res = {}
    for rnd in range(5):
        for fast in ("0", "1"):
            os.environ["UG_ADALN_FAST"] = fast
            run = lambda: ops.adaln_modulate(x, emb[:, 3 * D:], emb[:, 4 * D:], out, rows=rows, D=D, rows_per_sample=rps, mod_ld=6 * D)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); e1.synchronize()
            res.setdefault(fast, []).append(e0.elapsed_time(e1) / 10)
    gb = 2 * rows * D * 2 / 1e9
    print(f"adaln rows={rows} D={D}: generic {min(res['0']) * 1e3:.1f} us ({gb / min(res['0']):.2f} TB/s)   fast {min(res['1']) * 1e3:.1f} us ({gb / min(res['1']):.2f} TB/s)", flush=True)
