"""`pyopal.align` re-stated for the MI355X path (``src/pyopal/_align.py:28-172``).

Same generator contract (``threads``, ``pool``, ``ordered``, contiguous chunks of
``len(database) // threads`` targets). The chunks are searched concurrently
through the C ABI (GIL released) and dealt round-robin over the visible GPUs,
each GPU holding its own mirror of the database: chunks are independent, so no
collective is involved.
"""

from __future__ import annotations

import contextlib
import multiprocessing.pool
import typing

from . import _capi
from .lib import Aligner, BaseDatabase, Database, ScoreResult, resolve_scoring_matrix


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
    if chunks == 1:
        # one search on the calling thread
        yield from aligner.align(query, targets, **options)
        return

    # contiguous chunks of size // chunks targets (the remainder makes one more, short chunk,
    # as in the reference), chunk k on GPU k mod devices
    step = size // chunks
    jobs = [(begin, begin + step, k % devices) for k, begin in enumerate(range(0, size, step))]

    def search(job):
        begin, stop, device = job
        return aligner.align(query, targets, start=begin, end=stop, device=device, **options)

    with contextlib.ExitStack() as stack:
        workers = pool if pool is not None else stack.enter_context(multiprocessing.pool.ThreadPool(chunks))
        results = workers.imap(search, jobs) if ordered else workers.imap_unordered(search, jobs)
        for hits in results:
            yield from hits
