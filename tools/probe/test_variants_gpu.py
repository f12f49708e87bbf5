"""Tests of the measured-and-dropped kernel variants. They exist only in the PROBE library (python -m unigen_amd.build --probe ->
tools/probe/libunigen_hip_probe.so, built with -DUG_PROBE_BUILD), where the UG_* switches select them. Not part of the product test suite
(tests/); run on a GPU box with

    python -m unigen_amd.build --probe && UG_LIB_PATH=tools/probe/libunigen_hip_probe.so python -m pytest tools/probe/test_variants_gpu.py -q -m gpu

Every variant must return the product kernel's bits (or, where the summation order differs, agree to rounding): that is what made the A/B
timings in DESIGN.md comparisons of like with like."""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("UG_LIB_PATH", os.path.join(ROOT, "tools", "probe", "libunigen_hip_probe.so"))

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from unigen_amd import lib
    assert lib.LIB_PATH.endswith("libunigen_hip_probe.so"), lib.LIB_PATH
    lib.load()
    return torch.device("cuda:0")


_PWG_SNIPPET = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, {root!r})
from unigen_amd import ops
gpu, BF = torch.device("cuda:0"), torch.bfloat16
import os
def run(B, H, Lq, Lkv, spike):
    dh = int(os.environ.get("UG_TEST_DH", "128")); D = H * dh
    g = torch.Generator().manual_seed(Lq + 7 * Lkv)
    qkv = (torch.randn(B, Lkv, 3 * D, generator=g) * (0.5 if spike else 1.0)).to(BF)
    if spike:
        qkv[0, 200, D:D + dh] = qkv[0, Lkv - Lq + 17, :dh] * 8.0     # tile 3: far past the lazy-rescale threshold
        qkv[0, Lkv - 5, D:D + dh] = qkv[0, Lkv - Lq + min(99, Lq - 1), :dh] * 12.0
    d = qkv.to(gpu)
    out = torch.zeros(B, Lq, D, device=gpu, dtype=BF)
    st = (3 * D, Lkv * 3 * D)
    ops.flash_attn(d[0, Lkv - Lq:], d[0, 0, D:], d[0, 0, 2 * D:], out, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, q_strides=st, k_strides=st, v_strides=st, o_strides=(D, Lq * D))
    q = qkv[:, Lkv - Lq:, :D].view(B, Lq, H, dh).transpose(1, 2).float()
    k = qkv[:, :, D:2 * D].view(B, Lkv, H, dh).transpose(1, 2).float()
    v = qkv[:, :, 2 * D:].view(B, Lkv, H, dh).transpose(1, 2).float()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Lq, D)
    rel = ((out.float().cpu() - ref).norm() / ref.norm()).item()
    print("pwg", B, H, Lq, Lkv, spike, rel)
    return rel
worst = max(run(1, 2, 256, 256, False), run(2, 3, 300, 333, False), run(1, 2, 512, 1024, True), run(1, 1, 64, 64, False), run(1, 2, 70, 700, True))
sys.exit(0 if worst <= 4e-3 else 1)
"""


@pytest.mark.parametrize("env", [{"UG_ATTN_M16": "1"}, {"UG_ATTN_M16": "1", "UG_ATTN_PRIO": "0"}, {"UG_ATTN_M16": "1", "UG_TEST_DH": "64"}, {"UG_ATTN_PWG": "1"}, {"UG_ATTN_STAGGER": "0"}, {"UG_ATTN_STAGGER": "0", "UG_TEST_DH": "64"},
                                 {"UG_ATTN_WAVES": "4"}, {"UG_ATTN_PRIO": "1", "UG_ATTN_WIDE": "0"}, {"UG_ATTN_PRIO": "2", "UG_ATTN_WIDE": "1", "UG_TEST_DH": "64"},
                                 {"UG_ATTN_DMA": "1"}, {"UG_ATTN_DMA": "1", "UG_TEST_DH": "64"}, {"UG_ATTN_DMA": "0"}, {"UG_ATTN_DMA": "0", "UG_TEST_DH": "64"},
                                 {"UG_ATTN_PRIO": "0"}, {"UG_ATTN_PRIO": "3", "UG_TEST_DH": "64"},
                                 {"UG_ATTN_KV64": "64", "UG_TEST_DH": "64"}, {"UG_ATTN_KV64": "464", "UG_TEST_DH": "64"}, {"UG_ATTN_KV64": "128", "UG_TEST_DH": "64"}],
                         ids=["mfma-16x16x32-stagger", "mfma-16x16x32-stagger-no-prio", "mfma-16x16x32-stagger-dh64", "one-wave-per-simd", "lock-step-dh128", "lock-step-dh64", "four-wave-workgroups", "r1-default-prio1-narrow", "static-prio-wide-dh64",
                              "lds-dma-dh128", "lds-dma-dh64", "register-staged-dh128", "register-staged-dh64", "no-priority-dh128", "softmax-priority-dh64",
                              "dh64-64-key-tiles", "dh64-two-workgroups-per-cu", "dh64-128-key-tiles"])
def test_flash_attn_selectable_variants(gpu, env):
    """The non-default attention kernels: UG_ATTN_PWG=1 (4 waves x 64 rows, 512 registers, software-pipelined in the wave), the lock-step
    loop at both head dims and the 4-wave workgroups (the default is the X|Y stagger). The switches are read once per process, so each variant runs
    in a child: ragged Lq / Lkv, one tile, and keys that force the lazy rescale in a late tile."""
    import os, subprocess, sys
    root = ROOT
    r = subprocess.run([sys.executable, "-c", _PWG_SNIPPET.format(root=root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gemm_one_wave_per_simd_variants_match_default(gpu):
    """gemm_pwg.hip (UG_GEMM_PWG=1: 4 waves x 128x128, 512 registers; =2: 8 waves x 128x64): selectable alternatives to the 8-phase 256^2
    kernel, kept because they are the measurement that located the GEMM's bound (the L2 -> LDS intake). Same MFMA
    shape and K order -> bit-identical results, all epilogues, ragged edges, row maps, grouped, column split. UG_ENV_DYNAMIC re-reads the switch."""
    import os, subprocess, sys
    root = ROOT
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K, epi, grp) in [(600, 520, 192, L.EPI_BIAS, 1), (512, 768, 3072, L.EPI_BIAS_GELU, 1), (1000, 256, 320, L.EPI_RES_GATE, 1), (300, 512, 64, L.EPI_RES_SCALE, 1),
                            (260, 256, 128, L.EPI_BIAS, 3), (512, 1024, 256, "split", 1)]:
    a, w, b = rn(grp, M, K), rn(grp, N, K) * 0.1, rn(grp, N)
    r, gate = rn(grp, M, N), rn(grp, (M + 99) // 100, N)
    outs = []
    for mode in ("0", "1", "2"):
        os.environ["UG_GEMM_PWG"] = mode
        out = torch.zeros(grp, M, N + 64, device=dev, dtype=torch.bfloat16)
        kw = dict(M=M, groups=grp, a_gstride=M * K, w_gstride=N * K, bias_gstride=N, c_gstride=M * (N + 64), ldc=N + 64)
        if epi == "split":
            kw.update(epilogue=L.EPI_BIAS_GELU, gelu_from_n=512, c_shift_from_n=512, c_shift=64)
        else:
            kw.update(epilogue=epi)
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=r, r_gstride=M * N, alpha=0.7)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=100, gate_gstride=gate.shape[1] * N)
        ops.gemm(a, w, b, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        if not torch.equal(o, outs[0]):
            bad += 1
            print("MISMATCH", M, N, K, epi, grp, float((o.float() - outs[0].float()).abs().max()))
sys.exit(1 if bad else 0)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gemm_whole_line_one_wave_per_simd_kernel_matches_default(gpu):
    """Round 3: gemm_pwg64_kernel (UG_GEMM_PWG=3) - one wave per SIMD, 4 x (128 x 128), 64-deep ring units of whole 128-byte
    lines, one barrier per K-tile. A measurement build of that main loop (whole 256^2 tiles only): same MFMA shape and K order as the 8-phase
    kernel -> bit-identical, on every epilogue it carries, K = 192 (shorter than the ring) .. 3072, a row-mapped A, several tiles per workgroup."""
    import os, subprocess, sys
    root = ROOT
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
from unigen_amd.ops import RowMap
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K, epi, mapped) in [(512, 512, 192, L.EPI_BIAS, False), (1024, 768, 3072, L.EPI_BIAS_GELU, False), (768, 256, 320, L.EPI_RES_GATE, False),
                               (512, 512, 256, L.EPI_RES_SCALE, True), (8192, 3072, 1024, L.EPI_BIAS, False), (256, 256, 576, L.EPI_BIAS, False)]:
    rows = M // 2 if mapped else 0
    a = rn(M + (128 if mapped else 0), K)                 # mapped: two batches of M / 2 rows at a stride of M / 2 + 64 rows
    w, b = rn(N, K) * 0.1, rn(N)
    r, gate = rn(M, N), rn(M // 256, N)
    outs = []
    for mode in ("0", "3"):
        os.environ["UG_GEMM_PWG"] = mode
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        kw = dict(M=M, epilogue=epi)
        if mapped:
            kw.update(a_map=RowMap(rows, rows + 64))
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=r, alpha=0.7)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=256)
        ops.gemm(a, w, b, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    ref = (a.float()[:M] if not mapped else torch.cat([a.float()[:rows], a.float()[rows + 64:2 * rows + 64]])) @ w.float().t()
    if not torch.equal(outs[0], outs[1]) or not torch.isfinite(outs[1].float()).all() or (epi == L.EPI_BIAS and float(((outs[1].float() - b.float()) - ref).norm() / ref.norm()) > 1e-2):
        bad += 1
        print("MISMATCH", M, N, K, epi, mapped, float((outs[1].float() - outs[0].float()).abs().max()))
sys.exit(1 if bad else 0)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gemm_cross_tile_stream_matches_default(gpu):
    """Round 3: UG_GEMM_XTILE=1 - the last two K-tiles of a full tile stage the NEXT tile's first two K-tiles into the ring slots that would idle
    (gemm.hip next_src), so the ring never drains at a tile boundary. Same MFMAs in the same order -> bit-identical. Shapes with 2-6 tiles per
    workgroup, K-tile counts 4 .. 48 (even: streamed) and 5 (odd: falls back per tile), ragged M / N rims, a row-mapped A, every epilogue."""
    import os, subprocess, sys
    root = ROOT
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
from unigen_amd.ops import RowMap
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K, epi, mapped) in [(8192, 4096, 256, L.EPI_BIAS, False), (8192, 6144, 512, L.EPI_BIAS_GELU, False), (8192, 6144, 1024, L.EPI_RES_GATE, False),
                               (8192, 4096, 256, L.EPI_RES_SCALE, True), (16384, 3072, 3072, L.EPI_RES_GATE, False), (8000, 4100 - 4, 384, L.EPI_BIAS, False),
                               (8192, 4096, 320, L.EPI_BIAS, False), (8192, 8192, 640, L.EPI_F32, False)]:
    rows = M // 2 if mapped else 0
    a = rn(M + (128 if mapped else 0), K)
    w, b = rn(N, K) * 0.1, rn(N)
    r, gate = rn(M, N), rn((M + 255) // 256, N)
    outs = []
    for mode in ("0", "1"):
        os.environ["UG_GEMM_XTILE"] = mode
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == L.EPI_F32 else torch.bfloat16)
        kw = dict(M=M, epilogue=epi)
        if mapped:
            kw.update(a_map=RowMap(rows, rows + 64))
        if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
            kw.update(residual=r, alpha=0.7)
        if epi == L.EPI_RES_GATE:
            kw.update(gate=gate, gate_ld=N, rows_per_sample=256)
        ops.gemm(a, w, None if epi == L.EPI_F32 else b, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    ref = (a.float()[:M] if not mapped else torch.cat([a.float()[:rows], a.float()[rows + 64:2 * rows + 64]])) @ w.float().t()
    plain = epi in (L.EPI_BIAS, L.EPI_F32)
    if not torch.equal(outs[0], outs[1]) or not torch.isfinite(outs[1].float()).all() or \
            (plain and float(((outs[1].float() - (0 if epi == L.EPI_F32 else b.float())) - ref).norm() / ref.norm()) > 1e-2):
        bad += 1
        print("MISMATCH", M, N, K, epi, mapped, float((outs[1].float() - outs[0].float()).abs().max()))
sys.exit(1 if bad else 0)
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


_BWD_VARIANT_SNIPPET = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, {root!r})
import torch
from unigen_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for dh, H, B, Lq, Lkv in [(64, 3, 2, 257, 300), (64, 2, 1, 64, 64), (64, 2, 2, 1000, 1003), (128, 2, 2, 300, 333), (128, 2, 1, 640, 512)]:
    D = H * dh
    q, do = (torch.randn(B, Lq, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(2))
    k, v = (torch.randn(B, Lkv, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(2))
    o = torch.empty(B, Lq, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, (Lq + 63) // 64 * 64, device=dev, dtype=torch.float32)
    ops.flash_attn(q, k, v, o, batches=B, heads=H, dh=dh, Lq=Lq, Lkv=Lkv, lse=lse, q_strides=(D, Lq * D), k_strides=(D, Lkv * D),
                   v_strides=(D, Lkv * D), o_strides=(D, Lq * D))
    outs = []
    # variant 0: pair-scheme dQ kernel (forced: by default it takes head width 128 from 2048 queries) + fused dK / dV kernel; 1: separate DK / DV modes; 2: the 256-query DQ mode; 3: register staging (all modes)
    for env in ({{"UG_ATTN_BWD_DMA": "1", "UG_ATTN_BWD_FUSE_DKV": "1", "UG_ATTN_BWD_PAIR_DQ": "2"}}, {{"UG_ATTN_BWD_FUSE_DKV": "0"}},
                {{"UG_ATTN_BWD_FUSE_DKV": "1", "UG_ATTN_BWD_PAIR_DQ": "0"}}, {{"UG_ATTN_BWD_DMA": "0"}}):
        os.environ.update(env)
        outs.append(ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=lse))
        outs.append(ops.flash_attn_bwd(q, k, v, o, do, heads=H, lse=None))          # statistics recomputed by the LSE mode
    torch.cuda.synchronize()
    close = lambda a, b: bool(((a.float() - b.float()).norm() / b.float().norm()) < 2e-3)
    for i, got in enumerate(outs[1:], 1):
        for a, b, nm in zip(got, outs[0], ("dq", "dk", "dv")):
            # given statistics: dk / dv bitwise everywhere; dq bitwise among the pair-scheme runs (variants 0, 1), to fp32 rounding against the DQ mode
            # (two partial sums over the keys instead of one); recomputed statistics: to rounding
            exact = i % 2 == 0 and (nm != "dq" or i // 2 == 1)
            same = torch.equal(a, b) if exact else close(a, b)
            if not same:
                bad += 1
                print("MISMATCH", dh, Lq, Lkv, "variant", i, nm)
    if not (torch.equal(outs[4][0], outs[6][0])):                                  # the DQ mode itself: LDS-DMA and register staging agree bitwise
        bad += 1
        print("MISMATCH", dh, Lq, Lkv, "dq of the DQ mode, DMA vs registers")
sys.exit(1 if bad else 0)
"""


def test_flash_attention_backward_selectable_variants(gpu):
    """The fused dK / dV kernel (default; round 3: a 32-key block shared by the two waves of a pair, 5 product units instead of 8) against the separate DK and
    DV modes (UG_ATTN_BWD_FUSE_DKV=0): same products in the same order -> the same bits. Register staging instead of LDS-DMA (UG_ATTN_BWD_DMA=0; statistics
    of streamed queries by global loads; never fused) returns the default's bits; the pair-scheme dQ kernel (default; 128 queries per workgroup, each wave of a pair
    sums over half of the keys) agrees with the 256-query DQ mode (UG_ATTN_BWD_PAIR_DQ=0) to fp32 rounding of the sum over keys; with
    lse=None (statistics recomputed by the LSE mode rather than taken from the forward) the gradients agree to rounding of the statistics.
    Ragged lengths, one tile, several tiles, both head widths."""
    import os, subprocess, sys
    root = ROOT
    r = subprocess.run([sys.executable, "-c", _BWD_VARIANT_SNIPPET.format(root=root)], env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())


def gemm_tn(a, b):
    """ug_gemm_tn_bf16 (tools/probe/unigen_hip_probe.h; probe library only): a [R, I], b [R, J] row-major -> a^T b [I, J]."""
    import ctypes as C
    from unigen_amd import lib as L
    cdll = L.load()
    fn = cdll.ug_gemm_tn_bf16
    fn.restype, fn.argtypes = C.c_int32, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    R_, I = a.shape
    J = b.shape[1]
    out = torch.empty(I, J, device=a.device, dtype=BF)
    L.check(fn(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), J, R_, I, J, torch.cuda.current_stream().cuda_stream), "ug_gemm_tn_bf16")
    return out


@pytest.mark.parametrize("R_,I,J", [(4608, 3072, 3072), (1000, 192, 64), (130, 64, 256), (64, 8, 8), (2500, 320, 1544)])
def test_gemm_tn_weight_gradient(gpu, R_, I, J):
    """ug_gemm_tn_bf16 = A^T B (dW = dY^T X) from row-major operands: against fp32 torch on the CPU and against the transposes + ug_gemm_bf16 route."""
    from unigen_amd import ops
    g = torch.Generator().manual_seed(R_ + I)
    a, b = (torch.randn(R_, I, generator=g) * 0.5).to(BF), (torch.randn(R_, J, generator=g) * 0.5).to(BF)
    ref = a.float().t() @ b.float()
    out = gemm_tn(a.to(gpu), b.to(gpu))
    e = rel(out, ref)
    Rp = (R_ + 63) // 64 * 64
    alt = torch.empty(I, J, device=gpu, dtype=BF)
    if J % 4 == 0:
        ops.gemm(ops.transpose(a.to(gpu), Rp), ops.transpose(b.to(gpu), Rp), None, alt, M=I)
        e_alt = rel(alt, ref)
    else:
        e_alt = e
    print(f"backward gemm_tn {R_}x{I}x{J}: rel_l2 {e:.3e} (transposes + gemm: {e_alt:.3e})")
    assert e <= 3e-3 and e <= 1.2 * e_alt + 1e-4, (e, e_alt)
    # strided operands (column slices of wider buffers)
    wide_a, wide_b = torch.zeros(R_, I + 16, dtype=BF), torch.zeros(R_, J + 24, dtype=BF)
    wide_a[:, 8:8 + I], wide_b[:, 16:16 + J] = a, b
    out2 = gemm_tn(wide_a.to(gpu)[:, 8:8 + I], wide_b.to(gpu)[:, 16:16 + J])
    assert torch.equal(out2, out)


def test_gemm_pwg2_two_tile_ring_kernel_matches_default(gpu):
    """Round 4: gemm_pwg2_kernel (UG_GEMM_PWG=4) - the one-wave-per-SIMD kernel on the vendor loop's economy (two-K-tile ring, unrolled; buffer-form
    DMAs; whole-line full-tile epilogue; the next tile's first DMA pieces ahead of the C stores). Same MFMA shape and K order per accumulator as the
    8-phase kernel -> bit-identical under bias / GELU / gated and scaled residual epilogues, with one and with several tiles per workgroup, a
    row-mapped A, K = 256 (the ring's minimum) .. 3072; the ported q/k RMSNorm + RoPE epilogue sums a head's squares inside one wave instead of
    across four, so a few q / k elements may differ by the last bf16 bit of a rotation operand (<= 1e-5 of them, <= 2^-4 absolute at these magnitudes)."""
    import subprocess
    code = r"""
import os, sys
os.environ["UG_ENV_DYNAMIC"] = "1"
sys.path.insert(0, %r)
import torch
from unigen_amd import ops, lib as L
from unigen_amd.ops import RowMap, QkRope
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev).to(torch.bfloat16)
bad = 0
for (M, N, K, epi, mapped) in [(4096, 4096, 256, L.EPI_BIAS, False), (4096, 4096, 384, L.EPI_BIAS_GELU, False), (8192, 4096, 3072, L.EPI_RES_GATE, False),
                               (4096, 8192, 512, L.EPI_RES_SCALE, True), (16384, 3072, 1024, L.EPI_BIAS_GELU, False), (8192, 6144, 640, "qkrope", False)]:
    rows = M // 2 if mapped else 0
    a = rn(M + (256 if mapped else 0), K)                 # mapped: two batches of M / 2 rows at a stride of M / 2 + 256 rows
    w, b = rn(N, K) * 0.1, rn(N)
    r, gate = rn(M, N), rn(M // 1024, N)
    outs = []
    kw = dict(M=M)
    if epi == "qkrope":
        rpb = M // 2
        cs = (torch.rand(rpb, 64, 2, generator=g, device=dev) * 2 - 1).contiguous()
        kw.update(qk_rope=QkRope(rn(128) + 1, rn(128) + 1, cs, rpb, 0, 4096, 1e-6, 128), gelu_from_n=5120)
    else:
        kw.update(epilogue=epi)
    if mapped:
        kw.update(a_map=RowMap(rows, rows + 256))
    if epi in (L.EPI_RES_GATE, L.EPI_RES_SCALE):
        kw.update(residual=r, alpha=0.7)
    if epi == L.EPI_RES_GATE:
        kw.update(gate=gate, gate_ld=N, rows_per_sample=1024)
    for mode in ("0", "4"):
        os.environ["UG_GEMM_PWG"] = mode
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(2):
            ops.gemm(a, w, b, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    diff = (outs[0] != outs[1])
    frac = float(diff.float().mean())
    ok = torch.isfinite(outs[1].float()).all() and (frac == 0.0 if epi != "qkrope" else (frac <= 1e-5 and not diff[:, 4096:].any() and
          float((outs[0].float() - outs[1].float()).abs().max()) <= 2.0 ** -4))      # one ulp of a rotation's operands (values of a few units), not of the rotated result
    print(M, N, K, epi, mapped, "differing elements", frac)
    if not ok:
        bad += 1
        print("MISMATCH", M, N, K, epi, mapped, float((outs[1].float() - outs[0].float()).abs().max()))
sys.exit(1 if bad else 0)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
