"""`pyopal.align` re-stated for the MI355X path (``src/pyopal/_align.py:28-172``).

Same generator contract (``threads``, ``pool``, ``ordered``, contiguous chunks of
``len(database) // threads`` targets). The chunks are searched concurrently
through the C ABI (GIL released). With several GPUs visible the database is cut
into one contiguous shard per GPU with (nearly) equal residue counts
(`shard.balanced_bounds`: equal DP cells per GPU), every GPU
uploads and keeps only its shard, and a chunk is searched on the GPU(s) whose shard
it lies in - split at the shard boundary when it straddles one. Chunks are
independent, so no collective is involved; target indices are absolute.
"""

from __future__ import annotations

import contextlib
import multiprocessing.pool
import typing

import numpy as np

from . import _capi
from .lib import Aligner, BaseDatabase, Database, ScoreResult, resolve_scoring_matrix
from .shard import balanced_bounds


def align(query, database, scoring_matrix=None, *, gap_open: int = 3, gap_extend: int = 1,
          mode: str = "score", overflow: str = "buckets", algorithm: str = "sw",
          threads: int = 0, pool: typing.Optional[multiprocessing.pool.ThreadPool] = None,
          ordered: bool = False) -> typing.Iterator[ScoreResult]:
    """Align the query to every database sequence, yielding one result per target.

    ``threads=0`` means one chunk per visible GPU (the reference uses one per
    CPU core, ``src/pyopal/_align.py:117-118``); any other value is honoured as
    the number of chunks, as in the reference.
    """
    matrix = resolve_scoring_matrix(scoring_matrix, "got")
    targets = database if isinstance(database, BaseDatabase) else Database(database, matrix.alphabet)
    size = len(targets)
    aligner = Aligner(matrix, gap_open=gap_open, gap_extend=gap_extend)
    options = dict(mode=mode, overflow=overflow, algorithm=algorithm)

    devices = max(1, _capi.lib().miopalDeviceCount())
    chunks = min(threads or devices, size or 1)
    if size == 0 or (chunks == 1 and devices == 1):
        # one search on the calling thread (an empty database yields nothing, however many
        # GPUs there are: src/pyopal/_align.py:129-141)
        yield from aligner.align(query, targets, **options)
        return

    # one contiguous shard per GPU, balanced by residues. The bounds describe the database as it
    # is now: a search that finds it mutated since (its own read lock is taken later) falls back
    # to the whole-database mirror of its device instead of a shard that no longer exists.
    version = None
    if devices > 1:
        with targets.lock.read:
            version = getattr(targets, "_version", None)
            size = targets._get_size()
            lengths = np.fromiter(targets._get_lengths(), dtype=np.int64)[:size]
        offsets = np.zeros(size + 1, dtype=np.int64)
        np.cumsum(lengths, out=offsets[1:])
        bounds = balanced_bounds(offsets, devices)
    else:
        bounds = [0, size]

    jobs = []
    if not threads and devices > 1:
        # threads=0 on several GPUs: the chunks ARE the shards (balanced by residues; chunks of size // devices
        # targets would straddle the shard boundaries of a database with skewed lengths, each in two pieces searched
        # one after the other by one thread)
        for d in range(devices):
            if bounds[d] < bounds[d + 1]:
                jobs.append([(bounds[d], bounds[d + 1], d, (bounds[d], bounds[d + 1]))])
    else:
        # contiguous chunks of size // chunks targets (the remainder makes one more, short chunk,
        # as in the reference); a chunk is cut where it crosses a shard boundary
        step = max(1, size // chunks)
        for begin in range(0, size, step):
            stop = min(begin + step, size)
            pieces = []
            for d in range(devices):
                lo, hi = max(begin, bounds[d]), min(stop, bounds[d + 1])
                if lo < hi:
                    pieces.append((lo, hi, d, (bounds[d], bounds[d + 1]) if devices > 1 else None))
            jobs.append(pieces)

    def search(pieces):
        hits = []
        for begin, stop, device, part in pieces:
            hits.extend(aligner.align(query, targets, start=begin, end=stop, device=device, shard=part,
                                      shard_version=version, **options))
        return hits

    with contextlib.ExitStack() as stack:
        workers = pool if pool is not None else stack.enter_context(multiprocessing.pool.ThreadPool(max(1, len(jobs))))
        results = workers.imap(search, jobs) if ordered else workers.imap_unordered(search, jobs)
        for hits in results:
            yield from hits
