"""`pyopal.align` re-stated for the MI355X path (``src/pyopal/_align.py:28-172``).

Same generator contract (``threads``, ``pool``, ``ordered``, contiguous chunks of
``len(database) // threads`` targets). The chunks are searched concurrently
through the C ABI (GIL released) and dealt round-robin over the visible GPUs,
each GPU holding its own mirror of the database: chunks are independent, so no
collective is involved.
"""

from __future__ import annotations

import contextlib
import functools
import multiprocessing.pool
import typing

from . import _capi
from .lib import Aligner, BaseDatabase, Database, ScoreResult
from .matrices import ScoringMatrix


def align(query, database, scoring_matrix=None, *, gap_open: int = 3, gap_extend: int = 1,
          mode: str = "score", overflow: str = "buckets", algorithm: str = "sw",
          threads: int = 0, pool: typing.Optional[multiprocessing.pool.ThreadPool] = None,
          ordered: bool = False) -> typing.Iterator[ScoreResult]:
    """Align the query to every database sequence, yielding one result per target.

    ``threads=0`` means one chunk per visible GPU (the reference uses one per
    CPU core, ``src/pyopal/_align.py:117-118``); any other value is honoured as
    the number of chunks, as in the reference.
    """
    if scoring_matrix is None:
        scoring_matrix = Aligner._DEFAULT_SCORING_MATRIX
    elif isinstance(scoring_matrix, str):
        scoring_matrix = ScoringMatrix.from_name(scoring_matrix)
    elif not isinstance(scoring_matrix, ScoringMatrix):
        ty = type(scoring_matrix).__name__
        raise TypeError(f"expected str or ScoringMatrix, got {ty}")
    if not isinstance(database, BaseDatabase):
        database = Database(database, scoring_matrix.alphabet)

    devices = max(1, _capi.lib().miopalDeviceCount())
    if threads == 0:
        threads = devices
    if threads > len(database):
        threads = len(database) or 1

    aligner = Aligner(scoring_matrix, gap_open=gap_open, gap_extend=gap_extend)
    if threads == 1:
        yield from aligner.align(query, database, mode=mode, overflow=overflow, algorithm=algorithm)
        return

    pool_context: typing.ContextManager
    if pool is None:
        pool_context = multiprocessing.pool.ThreadPool(threads)
    else:
        pool_context = contextlib.nullcontext(pool)
    chunk_length = len(database) // threads
    starts = range(0, len(database), chunk_length)
    with pool_context as workers:
        search = functools.partial(aligner.align, query, database, mode=mode, overflow=overflow,
                                   algorithm=algorithm)

        def run(item):
            k, x = item
            return search(start=x, end=x + chunk_length, device=k % devices)

        mapper = workers.imap if ordered else workers.imap_unordered
        for hits in mapper(run, enumerate(starts)):
            yield from hits
