"""Build libunigen_hip.so (gfx950) in-tree with hipcc. No torch C++ extension: the library is a plain C ABI.

    python -m unigen_amd.build [--force]     the PRODUCT library unigen_amd/libunigen_hip.so: one kernel per dispatch decision, tuning constants fixed
    python -m unigen_amd.build --probe       tools/probe/libunigen_hip_probe.so: the same sources with -DUG_PROBE_BUILD, i.e. every measured-and-dropped
                                             kernel variant compiled in and the UG_* tuning switches read from the environment again (A/B tools only;
                                             select it with UG_LIB_PATH). Objects go to tools/probe/obj/, never next to the product's.

A clean clone has no binaries (*.o / *.so are git-ignored): `unigen_amd.lib.load()` calls build() when the library is missing, and
`__graft_entry__.build()` forces a full recompile (about 30 s) so that "does it build" is really exercised.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libunigen_hip.so")
PROBE_DIR = os.path.join(ROOT, "tools", "probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libunigen_hip_probe.so")
SOURCES = ["core.hip", "gemm.hip", "attention.hip", "elementwise.hip", "moe.hip", "verify_f32.hip", "probe.hip", "vae.hip", "backward.hip"]
PROBE_ONLY_SOURCES = ["gemm_pwg.hip", "gemm_tn.hip"]      # tools/probe/csrc/: kernels that lost their A/B, not in the product library
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


def build(force: bool = False, verbose: bool = False, probe: bool = False) -> str:
    """Compile every .hip source to an object and link the shared library. Returns the library path.

    Safe to call from several processes at once (every rank of a multi-GPU job calls lib.load() on a clean clone): the whole build runs under an
    exclusive flock on <objdir>/.build.lock, staleness is re-checked after the lock is taken (the second process finds fresh outputs and returns),
    and every output is written under a temporary name and os.replace()d into place, so a reader never sees a half-written .o or .so."""
    import fcntl
    objdir = os.path.join(PROBE_DIR, "obj") if probe else CSRC
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, probe, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, probe: bool, objdir: str) -> str:
    hdrs = [os.path.join(CSRC, "ug_common.h"), os.path.join(CSRC, "gemm_epilogue.h"), os.path.join(ROOT, "include", "unigen_hip.h")]
    sources = SOURCES + (PROBE_ONLY_SOURCES if probe else [])
    lib = PROBE_LIB if probe else LIB
    flags = (["-DUG_PROBE_BUILD", "-I", CSRC] if probe else []) + FLAGS
    if probe:
        hdrs.append(os.path.join(PROBE_DIR, "unigen_hip_probe.h"))
    tmp_tag = f".tmp{os.getpid()}"
    objs, jobs = [], []
    for src in sources:
        s = os.path.join(PROBE_DIR, "csrc", src) if src in PROBE_ONLY_SOURCES else os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append(([HIPCC, *flags, *EXTRA.get(src, []), "-c", s, "-o", o + tmp_tag], o))

    def run(job):
        cmd, final = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            try:
                os.unlink(final + tmp_tag)
            except OSError:
                pass
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        os.replace(final + tmp_tag, final)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run(([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + tmp_tag, *objs], lib))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probe="--probe" in sys.argv))
