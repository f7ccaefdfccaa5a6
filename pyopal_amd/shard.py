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


class OverlappedGather:
    """The gather of one search's scores, issued asynchronously so that it runs beside the next
    search (`bench.py`): `slots` result buffers are used in turn; `acquire()` hands out the next
    one after waiting for the gather that last read it, `submit()` starts the gather of the
    buffer just filled, `drain()` waits for everything in flight. With the ``nccl`` backend a
    wait orders the current stream behind the collective's stream; with ``gloo`` (CPU tests,
    rehearsals) tensors go through host copies.
    """

    def __init__(self, buffers, dst: int = 0, on_device: bool = True, force: bool = False):
        import torch
        import torch.distributed as dist

        self._dist = dist
        self.buffers = list(buffers)
        self.dst = dst
        self.on_device = on_device
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.pending = [None] * len(self.buffers)
        self.received = [None] * len(self.buffers)
        # (force: gather even in a group of one rank - the rehearsal of the N > 1 path on one GPU)
        self.active = self.world > 1 or (force and dist.is_initialized())
        if self.active and self.rank == dst:
            like = self.buffers[0] if on_device else self.buffers[0].cpu()
            self.received = [[torch.empty_like(like) for _ in range(self.world)] for _ in self.buffers]
        self._next = 0
        self.last = 0

    def acquire(self):
        """-> (index, buffer) of the next result buffer, free to be overwritten."""
        b = self._next
        self._next = (b + 1) % len(self.buffers)
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self.last = b
        return b, self.buffers[b]

    def submit(self, b: int) -> None:
        if not self.active:
            return
        tensor = self.buffers[b] if self.on_device else self.buffers[b].cpu()
        self.pending[b] = self._dist.gather(tensor, self.received[b], dst=self.dst, async_op=True)

    def drain(self) -> None:
        for b, work in enumerate(self.pending):
            if work is not None:
                work.wait()
                self.pending[b] = None
