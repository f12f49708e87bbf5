"""Where a tile of the staggered attention backward spends its cycles (diagnostic build only):
   UG_EXTRA_HIPCC_FLAGS=-DUG_DIAG_STAMPS python -m unigen_amd.build --force ; python tools/attn_bwd_stamps.py [dh] [B] [L] ; rebuild with --force after.
Each of the three kernels (DQ, DK, DV) overwrites the buffer, so UG_STAMP_MODE = dq | dk | dv picks which launch is read (the others are skipped
by reading right after it: the tool calls the backward once per mode with a device synchronise in between and reads after the chosen kernel)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unigen_amd import ops, lib as L

dev = torch.device("cuda:0")
dh = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Lq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
H = 24
D = H * dh
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(B, Lq, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(4))
o = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
lse = torch.zeros(B, H, (Lq + 63) // 64 * 64, device=dev, dtype=torch.float32)
st = (D, Lq * D)
ops.flash_attn(q, k, v, o, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lq, lse=lse, q_strides=st, k_strides=st, v_strides=st, o_strides=st)
ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse)
torch.cuda.synchronize()
ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse)          # the last kernel to run is DV: its stamps are what the buffer holds
torch.cuda.synchronize()
lib = L.load()
fn = lib.ug_diag_attn_stamps
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int64]
buf = np.zeros((64, 8, 8), dtype=np.uint64)
assert fn(buf.ctypes.data, buf.nbytes) == 0
names = ["Y(VALU)", "dma_wait_Y", "barrier_Y", "X(MFMA)", "dma_wait_X", "barrier_X"]
for blk in (0, 17, 40):
    for w in (0, 1, 4, 5):
        n = int(buf[blk, w, 6])
        per = buf[blk, w, :6].astype(np.float64) / max(n, 1)
        print(f"block {blk} wave {w} ({'A' if w < 4 else 'B'}), {n} tiles, cycles per tile (s_memtime ticks): " + "  ".join(f"{nm} {x:7.0f}" for nm, x in zip(names, per)) + f"  | sum {per.sum():7.0f}")
