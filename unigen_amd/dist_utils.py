"""Batch-parallel multi-GPU plumbing: one process per GPU, independent samples per rank, no data-path collective.
The reference shards its DataLoader by rank and has zero collectives in the denoise loop (infer.py:173); here the only
collectives are the barriers around a timed region and one MAX all-reduce of the elapsed time (RCCL on GPUs, gloo in CPU tests)."""
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


def init_distributed(device: torch.device):
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world). Backend nccl (= RCCL) on GPUs, gloo on CPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            backend = os.environ.get("UG_DIST_BACKEND", "nccl" if device.type == "cuda" else "gloo")   # nccl = RCCL over xGMI
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)      # rehearsals: several ranks sharing one GPU cannot form an RCCL communicator
    return rank, world


def barrier(device: torch.device, world: int) -> None:
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _reduce_device(device: torch.device) -> torch.device:
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def max_over_ranks(value: float, device: torch.device, world: int) -> float:
    if world == 1:
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=_reduce_device(device), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device, world: int) -> float:
    if world == 1:
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=_reduce_device(device), dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
