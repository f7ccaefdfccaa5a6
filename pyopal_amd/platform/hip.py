"""The HIP platform plugin: `searchHIP` fills the slot that
``searchSSE2/SSE4/AVX2/NEON`` fill in the reference
(``src/pyopal/platform/pyx.in:16-108``): marshal one query and a database
slice into the C call, map return codes to exceptions, return a list of result
objects. The C call is ``miopalSearch`` on the database's device mirror instead
of ``opalSearchDatabase`` on N host pointers."""

from __future__ import annotations

import typing

import numpy as np

from .. import _capi, _results
from ..lib import (UINT32_MAX, BaseDatabase, EndResult, FullResult, ScoreResult, _int_matrix_array)


def searchHIP(encoded: bytes, database: BaseDatabase, mode: int, overflow: int, algorithm: int,
              gap_open: int, gap_extend: int, int_matrix, start: int = 0, end: int = UINT32_MAX,
              device: int = 0, shard: typing.Optional[typing.Tuple[int, int]] = None) -> typing.List[ScoreResult]:
    if encoded is None or database is None:
        raise TypeError("encoded and database must not be None")
    if mode == _capi.SEARCH["score"]:
        result_type = ScoreResult
    elif mode == _capi.SEARCH["end"]:
        result_type = EndResult
    elif mode == _capi.SEARCH["full"]:
        result_type = FullResult
    else:
        raise ValueError("invalid search mode")
    if overflow not in (0, 1):
        raise ValueError("invalid overflow mode")

    size = database._get_size()
    if end > size:
        end = size
    n = 0 if end == 0 else end - start
    if n < 0:
        raise IndexError("database slice end is lower than start")
    if n == 0:
        return []  # the C call is skipped for an empty slice (pyx.in:75)

    if _capi.lib().miopalDeviceCount() < 1:
        raise RuntimeError("no supported SIMD backend available")
    matrix = _int_matrix_array(int_matrix)
    matrix_size = int(np.sqrt(matrix.shape[0]))
    # (shard: a mirror that holds the targets [lo, hi) only; indices of the results stay absolute)
    mirror = database._device_mirror(device, shard)
    base = 0 if shard is None else shard[0]
    query = np.frombuffer(encoded, dtype=np.uint8)

    mode_name = ("score", "end", "full")[mode]
    algo_name = ("nw", "hw", "ov", "sw")[algorithm] if 0 <= algorithm <= 3 else None
    if algo_name is None:
        _capi.raise_for(_capi.OPAL_ERR_INVALID_MODE)
    if matrix_size != mirror.alphabet_length:
        raise ValueError("database and score matrix have different alphabets")
    out = mirror.search(query, matrix, gap_open, gap_extend, mode_name, algo_name, start - base, end - base)

    scores = np.ascontiguousarray(out["score"], dtype=np.int32)
    if result_type is ScoreResult:
        return _results.score_results(start, scores)
    if result_type is EndResult:
        return _results.end_results(start, scores, out["end_q"], out["end_t"])
    # query and target lengths are recorded so that the coverage can be computed later
    # (pyx.in:95-99)
    return _results.full_results(start, scores, out["end_q"], out["end_t"], out["start_q"], out["start_t"],
                                 len(encoded), database._get_lengths(), out["aln_flat"], out["aln_off"])
