"""Batch-parallel multi-GPU plumbing: one process per GPU, independent samples per rank, no data-path collective.
The reference shards its DataLoader by rank and has zero collectives in the denoise loop (infer.py:173); here the only
collectives are the barriers around a timed region, one MAX all-reduce of the elapsed time and one all_gather of the per-rank
(images, seconds, device, probe) record for the report - RCCL on GPUs, gloo in CPU tests."""
from __future__ import annotations

import os
from typing import Tuple

import torch


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r of W takes samples [r*G//W, (r+1)*G//W) - contiguous, disjoint, covering, sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (rank * global_batch) // world, ((rank + 1) * global_batch) // world


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank generator seed: reference default 12443 (infer.py:61) + rank."""
    return base_seed + rank


def init_distributed(device: torch.device, force: bool = False):
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world). Backend nccl (= RCCL) on GPUs, gloo on CPU.
    `force`: form the process group even at world size 1 (tests/test_dist_gpu.py proves RCCL initialisation on a one-GPU box that way)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 or force:
        import torch.distributed as dist
        if not dist.is_initialized():
            addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = os.environ.get("MASTER_PORT")
            if port is None:
                if world > 1:
                    raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: launch with torchrun or `python bench.py --gpus N` (both set it)")
                import socket                      # a forced 1-rank group: any free port (a fixed default collides with other jobs on the host)
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    port = s.getsockname()[1]
            init = f"tcp://{addr}:{port}"          # rank / world handed over explicitly: nothing is written into os.environ (children would inherit it)
            backend = os.environ.get("UG_DIST_BACKEND", "nccl" if device.type == "cuda" else "gloo")   # nccl = RCCL over xGMI
            if backend == "nccl":
                dist.init_process_group("nccl", init_method=init, rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, init_method=init, rank=rank, world_size=world)      # rehearsals: several ranks sharing one GPU cannot form an RCCL communicator
    return rank, world


def describe(world: int) -> dict:
    """What the bench line records about the communicator its collectives ran on: torch.distributed's own answer, not the launcher's environment."""
    if not _group_up(world):
        return dict(initialized=False, backend=None, world_size=1, note="single process: no process group, barriers are device synchronisations")
    import torch.distributed as dist
    return dict(initialized=True, backend=str(dist.get_backend()), world_size=int(dist.get_world_size()), rank=int(dist.get_rank()),
                note="backend nccl = RCCL on ROCm")


def _group_up(world: int) -> bool:
    if world > 1:
        return True
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()          # a forced 1-rank group (init_distributed(force=True))


def barrier(device: torch.device, world: int) -> None:
    if _group_up(world):
        import torch.distributed as dist
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _reduce_device(device: torch.device) -> torch.device:
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(value: float, device: torch.device, world: int) -> float:
    if not _group_up(world):
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=_reduce_device(device), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device, world: int) -> float:
    if not _group_up(world):
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=_reduce_device(device), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_gather_floats(values, device: torch.device, world: int):
    """Every rank contributes the same number of floats; returns [world][len(values)] on every rank (the per-rank
    (n_images, seconds, ...) record behind the one JSON line, so that a straggler rank is visible in it)."""
    vals = [float(v) for v in values]
    if not _group_up(world):
        return [vals]
    import torch.distributed as dist
    t = torch.tensor(vals, device=_reduce_device(device), dtype=torch.float64)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]
