"""Instruction census of a kernel's hot loop from the shipped code object (VALU / SALU / DMA per 128 MFMAs, before and after a change).

    python tools/loop_census.py [--lib unigen_amd/libunigen_hip.so] [--match gemm256_kernel] [--out profiles/r05_loop_census.json]

Disassembles the gfx950 code object embedded in the library (clang-offload-bundler + llvm-objdump from /opt/rocm/lib/llvm/bin - no GPU needed),
finds in every kernel whose demangled name contains --match the innermost backward-branch loop that holds the most MFMAs (the K loop) and counts
instruction classes between its head and its backward branch:
  mfma      v_mfma_*
  valu      every other v_* (address arithmetic, moves, selects, accumulator copies)
  salu      s_* except waitcnt / barrier / nop / branches (scalar address arithmetic, wait selection, loop control)
  dma       LDS-DMA loads: global_load_lds_* (64-bit per-lane address, "global form") or buffer_load_* ... lds ("buffer form": SGPR resource + lane offset + SGPR offset)
  ds_read   LDS fragment reads, branch = scalar branches, wait = s_waitcnt, barrier = s_barrier
and scales them to "per 128 MFMAs" (one K-tile of the 256 x 256 x 64 kernel per wave).
Round 6: `--match "flash_attn_kernel<"` gives the attention forward's tile loop (two key tiles per trip: 64 MFMAs at head width 128, 32 at 64). The
count is STATIC: it includes the blocks a trip only enters on the ragged last tile (v_cmp / v_cndmask masking), when a row's reference point moves
(the v_pk_mul_f32 rescale of the O accumulators) and for one wave group only (the LDS-DMA address arithmetic); `op_histogram` lists every opcode.
The per-score softmax work - what every trip executes - is v_fma_f32 + v_exp_f32 + v_add_f32 per score, v_max3_f32 and v_cvt_pk_bf16_f32 per two.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(lib: str) -> str:
    """Every gfx950 code object embedded in `lib` (a shared library holds one offload bundle per linked object, back to back in .hip_fatbin)."""
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    bundler, objdump, objcopy = (os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-objdump", "llvm-objcopy"))
    out = []
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fb}", lib], check=True)
        raw = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(magic), raw)]
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(raw)])):
            part, co = os.path.join(td, f"b{n}"), os.path.join(td, f"b{n}.co")
            with open(part, "wb") as f:
                f.write(raw[a:b])
            r = subprocess.run([bundler, "--list", "--type=o", f"--input={part}"], capture_output=True, text=True, check=True)
            tg = [t for t in r.stdout.split() if "gfx950" in t]
            if not tg:
                continue
            subprocess.run([bundler, "--unbundle", "--type=o", f"--input={part}", f"--targets={tg[0]}", f"--output={co}"], check=True)
            out.append(subprocess.run([objdump, "-d", "--no-show-raw-insn", "-C", co], capture_output=True, text=True, check=True).stdout)
    return "\n".join(out)


def kernels(asm: str):
    """-> {name: [(addr, text)]}"""
    out, cur = {}, None
    for ln in asm.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            cur = m.group(2)
            out[cur] = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", ln)
        if m and cur is not None:
            tm = re.search(r"<.*\+0x([0-9a-fA-F]+)>\s*$", m.group(3))          # branch target, printed behind the encoding as <kernel+0xOFF>
            out[cur].append((int(m.group(2), 16), m.group(1).strip() + (f" <+0x{tm.group(1)}>" if tm else "")))
    return out


def classify(ins: str) -> str:
    op = ins.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and re.search(r"\blds\b", ins)):
        return "dma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_other"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop") or op.startswith("s_sleep") or op.startswith("s_setprio"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def hot_loop(body):
    """The backward-branch span with the highest MFMA density: the steady-state K loop."""
    addr_idx = {a: i for i, (a, _) in enumerate(body)}
    best = None
    for i, (a, ins) in enumerate(body):
        op = ins.split()[0]
        if not (op.startswith("s_cbranch") or op.startswith("s_branch")):
            continue
        tgt = None
        tm = re.search(r"<\+0x([0-9a-fA-F]+)>", ins)
        if tm:
            tgt = body[0][0] + int(tm.group(1), 16)
        if tgt is None or tgt not in addr_idx or tgt > a:
            continue
        j = addr_idx[tgt]
        span = body[j:i + 1]
        n_mfma = sum(1 for _, x in span if classify(x) == "mfma")
        if n_mfma == 0:
            continue
        # the K loop is the span with the highest MFMA density (an enclosing tile loop adds its epilogue, a peeled generic copy its branches)
        dens = n_mfma / float(i - j + 1)
        if best is None or dens > best[0]:
            best = (dens, 0, j, i)
    if best is None:
        return None
    return body[best[2]:best[3] + 1]


def census(lib: str, match: str):
    asm = disassemble(lib)
    res = {}
    for name, body in kernels(asm).items():
        if match not in name or not body:
            continue
        loop = hot_loop(body)
        if loop is None:
            continue
        c = {}
        for _, ins in loop:
            k = classify(ins)
            c[k] = c.get(k, 0) + 1
        forms = dict(global_form=sum(1 for _, x in loop if x.startswith("global_load_lds")),
                     buffer_form=sum(1 for _, x in loop if x.startswith("buffer_load") and re.search(r"\blds\b", x)))
        n = c.get("mfma", 0)
        per128 = {k: round(v * 128.0 / n, 1) for k, v in c.items()} if n else {}
        hist = {}
        for _, ins in loop:
            hist[ins.split()[0]] = hist.get(ins.split()[0], 0) + 1
        res[name] = dict(loop_instructions=len(loop), counts=c, dma_forms=forms, per_128_mfma=per128,
                         valu_ops=_top(loop, "valu"), salu_ops=_top(loop, "salu"),
                         op_histogram=dict(sorted(hist.items(), key=lambda kv: -kv[1])))
    return res


def _top(loop, cls):
    d = {}
    for _, ins in loop:
        if classify(ins) == cls:
            op = ins.split()[0]
            d[op] = d.get(op, 0) + 1
    return dict(sorted(d.items(), key=lambda kv: -kv[1])[:12])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "unigen_amd", "libunigen_hip.so"))
    ap.add_argument("--match", default="gemm256_kernel")
    ap.add_argument("--out", default=None)
    ap.add_argument("--label", default=None)
    a = ap.parse_args()
    res = census(a.lib, a.match)
    doc = dict(lib=os.path.relpath(a.lib, ROOT), match=a.match, label=a.label, kernels=res)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
    for name, r in res.items():
        print(name[:110])
        print("   loop", r["loop_instructions"], "instr; per 128 MFMA:", r["per_128_mfma"], r["dma_forms"])


if __name__ == "__main__":
    main()
