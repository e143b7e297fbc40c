"""Sharding of independent work units (time windows / flow hypotheses) over the GPUs of one node.

The contrast-maximisation path has no exchange step: every time window (reference: the per-frame loop of
bos_event.py:144-220) and every flow hypothesis (the optuna trial loop of
src/solver/generative_max_likelihood.py:229-236) is evaluated independently.  So the multi-GPU scheme is
one process per GPU, units partitioned statically, results gathered on the host -- and NO collective in the
data path.  ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used only for the
rendezvous, the barrier around timed regions and the final object gather.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def world() -> tuple:
    """(rank, world_size) from torch.distributed if initialised, else from the launcher's environment."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_units(n_units: int, world_size: int, rank: int, mode: str = "round_robin") -> List[int]:
    """Indices of the units owned by ``rank``.

    round_robin: unit i -> rank i mod world_size          (windows: neighbouring windows have similar cost)
    block:       contiguous blocks of ceil(n / world)     (hypothesis grids: 512 / 8 = 64 per GPU)
    """
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    if mode == "round_robin":
        return list(range(rank, n_units, world_size))
    if mode == "block":
        per = (n_units + world_size - 1) // world_size
        return list(range(min(n_units, rank * per), min(n_units, (rank + 1) * per)))
    raise ValueError(f"unknown sharding mode {mode!r}")


def gather_results(local: Dict[int, Any]) -> Dict[int, Any]:
    """Union of every rank's {unit index: result} on every rank (host-side object gather; results are
    scalars / small flow grids)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local)
    parts: List[Optional[Dict[int, Any]]] = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    out: Dict[int, Any] = {}
    for p in parts:
        for k, v in (p or {}).items():
            if k in out:
                raise RuntimeError(f"unit {k} was evaluated by two ranks")
            out[k] = v
    return out


def run_sharded(units: Sequence[Any], fn: Callable[[int, Any], Any], mode: str = "round_robin") -> List[Any]:
    """Evaluate ``fn(index, unit)`` for the units this rank owns and return the full, ordered result list on
    every rank.  ``fn`` runs on this rank's GPU; tensors it returns should be moved to the host first."""
    rank, size = world()
    mine = {i: fn(i, units[i]) for i in shard_units(len(units), size, rank, mode)}
    merged = gather_results(mine)
    missing = [i for i in range(len(units)) if i not in merged]
    if missing:
        raise RuntimeError(f"units {missing[:8]}... were not evaluated by any rank")
    return [merged[i] for i in range(len(units))]


def local_device() -> torch.device:
    """cuda:LOCAL_RANK (one process per GPU)."""
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
