"""Randomised parity: hypothesis draws small databases, queries, matrices, gap
penalties, modes and search types; the HIP path must agree with the CPU checker
bit for bit on every one of them."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1
    return _capi


@st.composite
def cases(draw):
    A = draw(st.sampled_from([2, 4, 20, 24, 32]))
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    n = draw(st.integers(1, 40))
    # mostly short targets, sometimes long ones and strip-boundary query lengths
    max_len = draw(st.sampled_from([6, 40, 200, 700]))
    lengths = rng.integers(0, max_len + 1, size=n)
    qlen = draw(st.sampled_from([1, 2, 7, 8, 9, 31, 63, 64, 65, 127, 128, 129, 200]))
    low, high = draw(st.sampled_from([(-1, 1), (-4, 11), (-12, 5), (0, 3), (-30, 40)]))
    matrix = rng.integers(low, high + 1, size=(A, A)).astype(np.int32)
    if draw(st.booleans()):
        matrix = np.minimum(matrix, matrix.T)  # symmetric half of the time
    gap_open = draw(st.sampled_from([0, 1, 3, 11, 40]))
    gap_ext = draw(st.sampled_from([0, 1, 2, 7]))
    # few distinct residues => many ties between equal-scoring cells
    span = draw(st.sampled_from([A, min(A, 2)]))
    seqs = [rng.integers(0, span, size=int(L)).astype(np.uint8) for L in lengths]
    if draw(st.booleans()) and qlen <= max_len:
        seqs[0] = None  # replaced by a mutated copy of the query below
    query = rng.integers(0, span, size=qlen).astype(np.uint8)
    if seqs[0] is None:
        copy = query.copy()
        copy[rng.integers(0, qlen, size=max(1, qlen // 8))] = rng.integers(0, span)
        seqs[0] = copy
    mode = draw(st.sampled_from(["score", "end", "full"]))
    algo = draw(st.sampled_from(["nw", "hw", "ov", "sw"]))
    return A, matrix, query, seqs, gap_open, gap_ext, mode, algo


@settings(max_examples=1500, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture,
                                                                   HealthCheck.too_slow, HealthCheck.data_too_large])
@given(case=cases())
def test_random_cases_match_the_checker(capi, case):
    A, matrix, query, seqs, go, ge, mode, algo = case
    res, off = _oracle.flatten(seqs)
    ref = _oracle.search(query, res, off, matrix.ravel(), go, ge, mode, algo)
    db = capi.DeviceDatabase(res, off, A)
    try:
        # lane-per-target kernels (the tier's default), then the production routing of small
        # searches (wavefront-per-pair kernels)
        runs = [db.search(query, matrix.ravel(), go, ge, mode, algo)]
        with capi.tuning(NO_SMALL_SEARCH=None):
            runs.append(db.search(query, matrix.ravel(), go, ge, mode, algo))
    finally:
        db.close()
    for gpu in runs:
        np.testing.assert_array_equal(gpu["score"], ref["score"])
        if mode != "score":
            np.testing.assert_array_equal(gpu["end_q"], ref["end_q"])
            np.testing.assert_array_equal(gpu["end_t"], ref["end_t"])
        if mode == "full":
            np.testing.assert_array_equal(gpu["start_q"], ref["start_q"])
            np.testing.assert_array_equal(gpu["start_t"], ref["start_t"])
            for a, b in zip(gpu["aln"], ref["aln"]):
                assert a.tolist() == b.tolist()


@st.composite
def long_cases(draw):
    """Fewer, larger cases: targets around the packed / long-target boundary (8192),
    several strips, scores that leave the half-float and int16 ranges."""
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    A = draw(st.sampled_from([4, 24]))
    qlen = draw(st.sampled_from([40, 64, 65, 130, 257, 520]))
    kind = draw(st.sampled_from(["random", "homolog", "boundary"]))
    query = rng.integers(0, A, size=qlen).astype(np.uint8)
    if kind == "random":
        lengths = rng.integers(1, 3000, size=draw(st.integers(1, 6)))
        seqs = [rng.integers(0, A, size=int(L)).astype(np.uint8) for L in lengths]
    elif kind == "homolog":
        # many copies of the query with few changes: high scores, long diagonal runs
        reps = draw(st.integers(1, 12))
        base = np.tile(query, reps)
        seqs = []
        for _ in range(draw(st.integers(1, 4))):
            s = base.copy()
            s[rng.integers(0, len(s), size=max(1, len(s) // 50))] = rng.integers(0, A)
            seqs.append(s)
        seqs.append(rng.integers(0, A, size=int(rng.integers(1, 500))).astype(np.uint8))
    else:
        lengths = [8190, 8192, 8193, int(rng.integers(1, 200))]
        seqs = [rng.integers(0, A, size=L).astype(np.uint8) for L in lengths[:draw(st.integers(1, 4))]]
    match = draw(st.sampled_from([2, 5, 11]))
    matrix = rng.integers(-6, 3, size=(A, A)).astype(np.int32)
    matrix[np.arange(A), np.arange(A)] = match
    gap_open = draw(st.sampled_from([1, 3, 11]))
    gap_ext = draw(st.sampled_from([0, 1, 2]))
    mode = draw(st.sampled_from(["score", "end", "full"]))
    algo = draw(st.sampled_from(["nw", "hw", "ov", "sw"]))
    return A, matrix, query, seqs, gap_open, gap_ext, mode, algo


@settings(max_examples=120, deadline=None, derandomize=True,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow,
                                 HealthCheck.data_too_large])
@given(case=long_cases())
def test_random_long_cases_match_the_checker(capi, case):
    test_random_cases_match_the_checker.hypothesis.inner_test(capi, case)
