import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from unigen_amd import lib

    lib.load()  # fail loudly if the HIP library is missing: there is no fallback
    return torch.device("cuda:0")


# ----------------------------------------------------------------------------------------------------------------------------------
# The full-size oracle comparisons (tests/test_fullsize_gpu.py: one forward / one training step of the WHOLE model against the CPU oracle in fp32)
# spend 260 of their 370 s inside the oracle on the host cores while the GPU idles, and the rest of the `-m gpu` suite keeps the GPU busy while the
# host idles (686 s of the driver's 900 s). So the children - tests/fullsize_f32_parity.py, tests/fullsize_train_parity.py, one
# at a time, in a fixed order - start when collection ends and run BESIDE the suite; their tests are moved to the end and only wait for the
# child's record. Same scripts, same arguments, same assertions; nothing is cached or skipped.
# ----------------------------------------------------------------------------------------------------------------------------------
FULLSIZE_JOBS = {          # test id fragment -> child command (relative to the repository root)
    "test_full_model_forward_parity[flux64]": ["tests/fullsize_f32_parity.py", "flux", "64", "--no-ref16"],
    "test_full_depth_gradient_parity": ["tests/fullsize_train_parity.py"],
    "test_full_model_forward_parity[multi]": ["tests/fullsize_f32_parity.py", "multi", "--no-ref16"],
    "test_full_model_forward_parity[sd3]": ["tests/fullsize_f32_parity.py", "sd3"],
}
# A GPU box grants one GPU's share of the host: 16 cores. Two 16-thread pools on them (the child's oracle and the suite's own small oracle evaluations)
# oversubscribe - measured: the chain took 640 s beside the suite against 370 s alone, the suite 642 s; with the split 477 s (profiles/r06k_gputest_durations.log) - so the cores are SPLIT while the chain runs: 13 threads for
# the child's fp32 oracle (its matmuls scale), 3 for the suite's process (its oracle work is small cases).
CHILD_THREADS, SUITE_THREADS = 13, 3
_jobs = {}                 # key -> dict(event, result)
_state = dict(proc=None, stop=False, thread=None)


def _job_key(nodeid: str):
    for k in FULLSIZE_JOBS:
        if nodeid.endswith(k):
            return k
    return None


def pytest_collection_modifyitems(config, items):
    tail = [it for it in items if _job_key(it.nodeid)]
    if tail:
        order = list(FULLSIZE_JOBS)
        tail.sort(key=lambda it: order.index(_job_key(it.nodeid)))
        items[:] = [it for it in items if not _job_key(it.nodeid)] + tail


def _chain(keys):
    import time
    for k in keys:
        if _state["stop"]:
            break
        t0 = time.perf_counter()
        try:
            p = subprocess.Popen([sys.executable, *[os.path.join(ROOT, FULLSIZE_JOBS[k][0]), *FULLSIZE_JOBS[k][1:]]], cwd=ROOT, stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, text=True, env=dict(os.environ, UG_ORACLE_THREADS=str(CHILD_THREADS), OMP_NUM_THREADS=str(CHILD_THREADS)))
            _state["proc"] = p
            out, err = p.communicate(timeout=1100)
            _jobs[k]["result"] = dict(returncode=p.returncode, stdout=out, stderr=err, seconds=time.perf_counter() - t0)
        except Exception as e:           # a timeout or a spawn failure is the test's failure, reported where the test waits
            if _state["proc"] is not None and _state["proc"].poll() is None:
                _state["proc"].kill()
            _jobs[k]["result"] = dict(returncode=-1, stdout="", stderr=f"{type(e).__name__}: {e}", seconds=time.perf_counter() - t0)
        finally:
            _state["proc"] = None
            _jobs[k]["event"].set()


def pytest_collection_finish(session):
    if os.environ.get("UG_FULLSIZE_INLINE") == "1" or session.config.option.collectonly:
        return
    keys = [k for k in FULLSIZE_JOBS if any(_job_key(it.nodeid) == k for it in session.items)]
    if not keys:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    for k in keys:
        _jobs[k] = dict(event=threading.Event(), result=None)
    _state["thread"] = threading.Thread(target=_chain, args=(keys,), daemon=True)
    _state["thread"].start()
    _state["threads_before"] = torch.get_num_threads()
    torch.set_num_threads(SUITE_THREADS)


def pytest_sessionfinish(session, exitstatus):
    _state["stop"] = True
    p = _state["proc"]
    if p is not None and p.poll() is None:      # -x ended the session early: the child we started (exactly that PID) does not outlive it
        p.kill()


@pytest.fixture
def fullsize_child(request):
    """-> run(argv): the record of this test's full-size child. Started in the background at collection time when the session allows it (see above),
    otherwise (UG_FULLSIZE_INLINE=1, or the test selected on its own before the chain could know) run here, inline."""
    key = _job_key(request.node.nodeid)

    def run():
        if key in _jobs:
            _jobs[key]["event"].wait()
            r = _jobs[key]["result"]
            if all(j["event"].is_set() for j in _jobs.values()) and _state.get("threads_before"):
                import torch
                torch.set_num_threads(_state["threads_before"])          # the chain is done: the suite has the cores to itself again
            print(f"[fullsize child {key}: {r['seconds']:.0f} s beside the suite]")
            return r
        import time
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, os.path.join(ROOT, FULLSIZE_JOBS[key][0]), *FULLSIZE_JOBS[key][1:]], capture_output=True, text=True, timeout=1100, cwd=ROOT)
        return dict(returncode=p.returncode, stdout=p.stdout, stderr=p.stderr, seconds=time.perf_counter() - t0)
    return run
