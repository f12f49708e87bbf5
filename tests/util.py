"""Shared helpers for parity tests."""
import json
import os

import torch

_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_metrics.jsonl")


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    g, r = got.double().flatten().cpu(), ref.double().flatten().cpu()
    return float((g - r).norm() / r.norm().clamp_min(1e-30))


def max_rel(got: torch.Tensor, ref: torch.Tensor) -> float:
    """max |got - ref| / max |ref|"""
    g, r = got.double().flatten().cpu(), ref.double().flatten().cpu()
    return float((g - r).abs().max() / r.abs().max().clamp_min(1e-30))


def report(name: str, got: torch.Tensor, ref: torch.Tensor, **extra) -> dict:
    m = dict(name=name, rel_l2=rel_l2(got, ref), max_rel=max_rel(got, ref),
             mismatch_frac=float((got.cpu() != ref.cpu().to(got.dtype)).float().mean()), **extra)
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(json.dumps(m) + "\n")
    except OSError:
        pass
    print("PARITY", json.dumps(m))
    return m


def bf(t: torch.Tensor) -> torch.Tensor:
    """round to bf16 and back to fp32 (the oracle works on bf16-representable values)"""
    return t.to(torch.bfloat16).float()


def check_routing(dev_idx: torch.Tensor, routing: dict, max_flip_frac: float = 0.02, tie_gap: float = 5e-2) -> int:
    """Top-1 expert indices from the device against the oracle's: they must agree except on tokens whose two largest gate
    probabilities are within `tie_gap` of each other in the oracle (a discrete argmax over logits computed from activations that
    differ in the last bf16 bits cannot be expected to agree on exact near-ties). Returns the number of flipped tokens."""
    di = dev_idx.cpu().long()
    oi = routing["idx"].long()
    flips = torch.nonzero(di != oi).flatten()
    if flips.numel():
        top2 = torch.topk(routing["gates"].float(), 2, dim=1)[0]
        gap = (top2[:, 0] - top2[:, 1])[flips]
        assert float(gap.max()) <= tie_gap, f"expert assignment differs on a clear-cut token (gap {float(gap.max()):.3f})"
        assert flips.numel() <= max(1, int(max_flip_frac * di.numel())), f"{flips.numel()} routing flips"
    return int(flips.numel())
