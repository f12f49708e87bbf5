"""RCCL must initialise in this torch / ROCm build BEFORE a multi-GPU node ever sees bench.py (reference: infer.py:173 shards samples by rank,
train.py:520,664,709 barriers). gpurun boxes have one GPU, so the proof is a 1-rank `nccl` process group with `device_id`, formed by a FRESH
child process (this pytest process has touched the GPU and must never be exec'ed over or forked into a communicator), which then runs the
collectives bench.py uses - `DU.barrier`, `DU.max_over_ranks`, `DU.sum_over_ranks`, `DU.all_gather_floats` - on GPU tensors and exits 0."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from unigen_amd import dist_utils as DU
assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
local_rank = int(os.environ["LOCAL_RANK"])
assert local_rank < torch.cuda.device_count()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
t0 = time.perf_counter()
rank, world = DU.init_distributed(dev, force=True)          # nccl (= RCCL) with device_id, as bench.py does at N > 1
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and (rank, world) == (0, 1)
DU.barrier(dev, world)
assert DU.max_over_ranks(1.25, dev, world) == 1.25 and DU.sum_over_ranks(7.0, dev, world) == 7.0
rec = DU.all_gather_floats([8, 0.5, local_rank, 2000.0], dev, world)
assert rec == [[8.0, 0.5, float(local_rank), 2000.0]], rec
x = torch.arange(1024, device=dev, dtype=torch.float32)
dist.all_reduce(x)                                          # a payload collective on the RCCL communicator itself
torch.cuda.synchronize()
assert float(x.sum()) == 1023 * 1024 / 2
DU.barrier(dev, world)
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.get_device_name(local_rank), f"{time.perf_counter() - t0:.2f}s", flush=True)
"""


def test_rccl_process_group_forms_in_a_fresh_child(gpu, tmp_path):
    script = tmp_path / "rccl_child.py"
    script.write_text(_CHILD % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("UG_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL_OK" in r.stdout, r.stdout
    print(r.stdout.strip().splitlines()[-1])
