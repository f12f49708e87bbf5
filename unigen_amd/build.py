"""Build libunigen_hip.so (gfx950) in-tree with hipcc. No torch C++ extension: the library is a plain C ABI."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libunigen_hip.so")
SOURCES = ["core.hip", "gemm.hip", "gemm_pwg.hip", "attention.hip", "elementwise.hip", "moe.hip", "verify_f32.hip", "probe.hip", "vae.hip", "backward.hip", "gemm_tn.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# attention: scores are finite or -inf, never NaN; without IEEE mode hipcc drops the NaN-quieting v_max x,x it adds per fmaxf operand
EXTRA = {"attention.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}
# UG_EXTRA_HIPCC_FLAGS: extra flags for a diagnostic build (e.g. -DUG_DIAG_STAMPS); such a build must be made with force=True both ways
FLAGS = os.environ.get("UG_EXTRA_HIPCC_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source to an object and link the shared library. Returns the library path."""
    hdrs = [os.path.join(CSRC, "ug_common.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(HERE, "..", "include", "unigen_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC, *FLAGS, *EXTRA.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
