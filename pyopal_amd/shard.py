"""Shard driver helpers for one-process-per-GPU searches.

Targets are independent units (the reference already splits them into
``[start, end)`` chunks per thread, ``src/pyopal/_align.py:150-170``), so each
rank searches its own contiguous slice and the only exchange is one gather of
the int32 scores to the destination rank (RCCL over xGMI when the process
group is ``nccl``; ``gloo`` in the CPU tests).
"""

from __future__ import annotations

import typing

import numpy as np


def balanced_bounds(offsets: np.ndarray, world: int) -> typing.List[int]:
    """Contiguous slices with (nearly) equal residue counts: bounds[r]..bounds[r+1]
    is rank r's slice. Balancing by residues, not by target count, equalises the
    DP cells each GPU has to fill."""
    n = len(offsets) - 1
    total = int(offsets[-1])
    bounds = [0]
    for r in range(1, world):
        goal = total * r // world
        k = int(np.searchsorted(offsets, goal, side="left"))
        bounds.append(min(max(k, bounds[-1]), n))
    bounds.append(n)
    return bounds


def gather_scores(local, bounds: typing.Sequence[int], dst: int = 0):
    """Gather per-rank score tensors (ragged: rank r holds bounds[r+1]-bounds[r]
    entries) on `dst`; returns the concatenated tensor there, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rank = dist.get_rank()
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    width = max(max(sizes), 1)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: sizes[rank]] = local
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, gathered, dst=dst)
    if rank != dst:
        return None
    return torch.cat([g[:s] for g, s in zip(gathered, sizes)])
