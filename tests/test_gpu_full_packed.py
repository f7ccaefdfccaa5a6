"""`full` searches through the kernels that carry two pairs per lane on 16-bit halves (round 5,
pyopal_amd/csrc/perpair_packed.hip: the start-cell scan of Smith-Waterman prefixes and the direction pass of every
mode; src/pyopal/opal.pxd:17-19 OPAL_SEARCH_ALIGNMENT, what the reference does with the result:
src/pyopal/platform/pyx.in:95-99, src/pyopal/lib.pyx:999-1037) against the CPU checker.

tests/test_gpu_full_profile.py already holds these kernels to the checker and to the 32-bit kernels on its
cases (the default routing takes them). Here: what is theirs alone -
  * the bias that makes the profile an unsigned byte (BLOSUM50 under gap 3/1: score + open + ext < 0),
  * scoring schemes that do NOT fit (open < ext, steps beyond a byte, values beyond the half floats) and must
    leave for the 32-bit kernels with the same answers,
  * one direction launch over several batches' sorted lists, with the outliers at the head of every batch
    left to the wavefront-per-pair kernel,
  * halves of a lane that belong to pairs of very different shape (a long window beside an empty one),
  * ties: related sequences under cheap gaps, where the flags' tie-breaks (diagonal > INS > DEL, close before
    extend) and the first-maximum rule of the scan decide the result.
"""
import numpy as np
import pytest

import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix
from test_gpu_parity import compare

pytestmark = pytest.mark.gpu

B50 = np.array(ScoringMatrix.from_name("BLOSUM50").int_array(), dtype=np.int32)
B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
KEYS = ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat")
PACKED_TRACE, PACKED_SCAN, ONE_LAUNCH = 64, 128, 256


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


@pytest.fixture()
def lane_per_pair(tuning):
    tuning.setenv("MIOPAL_NO_SMALL_SEARCH", "1")
    tuning.setenv("MIOPAL_NO_HYBRID_TRACE", "1")
    tuning.setenv("MIOPAL_FORCE_LANE_PER_PAIR", "1")


def search(capi, q, res, off, matrix, go, ge, algo="sw"):
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, matrix, go, ge, "full", algo)
        return got, capi.DeviceDatabase.last_full_routing()
    finally:
        db.close()


def related(rng, q, n, edits=12, flank=30):
    seqs = []
    for _ in range(n):
        t = q.copy()
        for _ in range(rng.integers(0, edits)):
            k = rng.integers(0, len(t))
            op = rng.integers(0, 3)
            if op == 0:
                t[k] = rng.integers(0, 20)
            elif op == 1 and len(t) > 2:
                t = np.delete(t, k)
            else:
                t = np.insert(t, k, rng.integers(0, 20))
        f = _data.random_protein(rng, int(rng.integers(0, flank)))
        seqs.append(np.concatenate([f, t, f[::-1]]).astype(np.uint8))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), off


@pytest.mark.parametrize("algo", ["sw", "nw", "hw", "ov"])
@pytest.mark.parametrize("matrix,go,ge", [(B50, 3, 1), (B50, 4, 0), (B62, 3, 1), (B62, 11, 1), (B62, 2, 2), (B62, 0, 0)],
                         ids=["b50-3-1-biased", "b50-4-0", "b62-3-1", "b62-11-1", "b62-2-2", "b62-0-0"])
def test_every_mode_against_the_checker(capi, lane_per_pair, algo, matrix, go, ge):
    rng = np.random.default_rng(7 + go)
    res, off = _data.random_db(rng, rng.integers(1, 350, size=700))
    for qlen in (1, 7, 53, 64, 65, 129, 260):
        q = _data.random_protein(rng, qlen)
        got, routing = search(capi, q, res, off, matrix, go, ge, algo)
        assert routing & PACKED_TRACE, (routing, qlen)
        assert bool(routing & PACKED_SCAN) == (algo != "nw"), (routing, qlen)   # (NW has no scan)
        ref = _oracle.search(q, res, off, matrix, go, ge, "full", algo)
        compare(got, ref, "full", f"{algo} Q={qlen} gaps {go}/{ge}")


def test_ties_on_related_sequences(capi, lane_per_pair):
    rng = np.random.default_rng(11)
    for qlen in (40, 64, 100, 300):
        q = _data.random_protein(rng, qlen)
        res, off = related(rng, q, 400)
        for matrix in (B62, B50):
            for go, ge in ((3, 1), (1, 1), (11, 1), (6, 2)):
                got, routing = search(capi, q, res, off, matrix, go, ge)
                assert routing & (PACKED_TRACE | PACKED_SCAN) == PACKED_TRACE | PACKED_SCAN, routing
                ref = _oracle.search(q, res, off, matrix, go, ge, "full", "sw")
                compare(got, ref, "full", f"related Q={qlen} gaps {go}/{ge}")


def test_low_complexity_sequences(capi, lane_per_pair):
    # runs of one residue: every cell of a run ties with its neighbours (the first maximum of the scan in column-major
    # order, the close-before-extend rule of the flags)
    rng = np.random.default_rng(13)
    seqs = []
    for _ in range(500):
        parts = [np.full(int(rng.integers(1, 25)), rng.integers(0, 4), dtype=np.uint8) for _ in range(int(rng.integers(1, 8)))]
        seqs.append(np.concatenate(parts))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    res = np.concatenate(seqs)
    for qlen in (30, 90):
        q = np.concatenate([np.full(qlen // 3, 0), np.full(qlen // 3, 1), np.full(qlen - 2 * (qlen // 3), 0)]).astype(np.uint8)
        for algo in ("sw", "ov"):
            for go, ge in ((3, 1), (1, 1), (0, 0)):
                got, routing = search(capi, q, res, off, B62, go, ge, algo)
                assert routing & PACKED_TRACE, routing
                ref = _oracle.search(q, res, off, B62, go, ge, "full", algo)
                compare(got, ref, "full", f"runs {algo} Q={qlen} gaps {go}/{ge}")


def test_schemes_that_do_not_fit_take_the_wide_kernels(capi, lane_per_pair):
    rng = np.random.default_rng(17)
    res, off = _data.random_db(rng, rng.integers(1, 250, size=400))
    q = _data.random_protein(rng, 70)
    # open < ext: the borders are no constants on the kernels' scales - neither pass
    got, routing = search(capi, q, res, off, B62, 1, 3)
    assert routing & (PACKED_TRACE | PACKED_SCAN) == 0, routing
    compare(got, _oracle.search(q, res, off, B62, 1, 3, "full", "sw"), "full", "open < ext")
    # score + open beyond 31: eight times it is no byte - the direction pass alone
    for matrix, go, ge in ((B62 * 2, 12, 2), (B62, 100, 1), (B62 * 9, 20, 3)):
        got, routing = search(capi, q, res, off, matrix, go, ge)
        assert routing & (PACKED_TRACE | PACKED_SCAN) == PACKED_TRACE, (routing, go, ge)
        compare(got, _oracle.search(q, res, off, matrix, go, ge, "full", "sw"), "full", f"direction pass alone, gaps {go}/{ge}")


def test_windows_beyond_the_half_floats_take_the_wide_kernels(capi, lane_per_pair):
    # global alignments of 2100 x 2100: rows x best score + the scale's (rows + columns) x ext leave the normal half
    # floats - the 32-bit direction pass (Smith-Waterman's scan too: eight times the query's best)
    rng = np.random.default_rng(31)
    q = _data.random_protein(rng, 2100)
    res, off = related(rng, q, 12, edits=40, flank=5)
    for algo in ("nw", "sw"):
        got, routing = search(capi, q, res, off, B62, 3, 1, algo)
        assert routing & (PACKED_TRACE | PACKED_SCAN) == 0, (routing, algo)
        compare(got, _oracle.search(q, res, off, B62, 3, 1, "full", algo), "full", f"long windows {algo}")


def test_long_windows_beside_empty_ones(capi, lane_per_pair):
    # halves of one lane: a window of several strips beside a pair without an alignment (no residue of the target
    # scores above zero against the query), beside a one-residue target
    rng = np.random.default_rng(19)
    q = _data.random_protein(rng, 200)
    long = [np.concatenate([q, q])[: int(n)] for n in rng.integers(150, 400, size=80)]
    stop = [np.full(int(n), 23, dtype=np.uint8) for n in rng.integers(1, 50, size=80)]   # '*': negative against everything
    tiny = [q[k:k + 1] for k in rng.integers(0, 200, size=80)]
    seqs = [s.astype(np.uint8) for trio in zip(long, stop, tiny) for s in trio]
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    res = np.concatenate(seqs)
    for algo in ("sw", "hw", "ov", "nw"):
        got, routing = search(capi, q, res, off, B62, 3, 1, algo)
        assert routing & PACKED_TRACE, routing
        compare(got, _oracle.search(q, res, off, B62, 3, 1, "full", algo), "full", f"mixed shapes {algo}")


@pytest.mark.parametrize("qlen", [53, 150])
def test_one_launch_over_several_batches(capi, tuning, qlen):
    # enough pairs for four traceback batches: the direction pass of two batches per launch, the outliers at the head of
    # each batch's sorted list on the wavefront-per-pair kernel; against the 32-bit kernels (every array) and the checker
    # (a sample)
    rng = np.random.default_rng(23)
    n = 270_000
    lengths = np.clip(rng.lognormal(mean=3.6, sigma=0.5, size=n), 5, 400).astype(np.int64)
    lengths[rng.choice(n, size=300, replace=False)] = rng.integers(600, 900, size=300)   # outliers: more than twice the 90th percentile
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    # (the outliers get something to align: copies of the query inside them)
    for k in np.flatnonzero(lengths >= 600)[::3]:
        at = off[k] + rng.integers(0, lengths[k] - qlen)
        res[at:at + qlen] = q
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, B62, 3, 1, "full", "sw")
        routing = capi.DeviceDatabase.last_full_routing()
        assert routing & (PACKED_TRACE | PACKED_SCAN | ONE_LAUNCH) == PACKED_TRACE | PACKED_SCAN | ONE_LAUNCH, routing
        tuning.setenv("MIOPAL_ONE_LAUNCH_GROUP", "4")
        one = db.search(q, B62, 3, 1, "full", "sw")
        tuning.delenv("MIOPAL_ONE_LAUNCH_GROUP")
        tuning.setenv("MIOPAL_NO_ONE_LAUNCH", "1")
        each = db.search(q, B62, 3, 1, "full", "sw")
        assert capi.DeviceDatabase.last_full_routing() & (PACKED_TRACE | ONE_LAUNCH) == PACKED_TRACE
        tuning.setenv("MIOPAL_NO_PACKED_TRACE", "1")
        tuning.setenv("MIOPAL_NO_PACKED_SCAN", "1")
        wide = db.search(q, B62, 3, 1, "full", "sw")
        assert capi.DeviceDatabase.last_full_routing() & (PACKED_TRACE | PACKED_SCAN | ONE_LAUNCH) == 0
    finally:
        db.close()
    for key in KEYS:
        np.testing.assert_array_equal(got[key], wide[key], err_msg=f"{key}: groups of two batches")
        np.testing.assert_array_equal(one[key], wide[key], err_msg=f"{key}: one launch")
        np.testing.assert_array_equal(each[key], wide[key], err_msg=f"{key}: a launch per batch")
    pick = np.sort(np.concatenate([rng.choice(n, size=300, replace=False), np.flatnonzero(lengths >= 600)[:40]]))
    sub_res = np.concatenate([res[off[k]:off[k + 1]] for k in pick])
    sub_off = np.concatenate([[0], np.cumsum(lengths[pick])]).astype(np.int64)
    ref = _oracle.search(q, sub_res, sub_off, B62, 3, 1, "full", "sw")
    np.testing.assert_array_equal(got["score"][pick], ref["score"])
    np.testing.assert_array_equal(got["start_q"][pick], ref["start_q"])
    np.testing.assert_array_equal(got["start_t"][pick], ref["start_t"])
    for x, k in enumerate(pick):
        assert got["aln"][k].tolist() == ref["aln"][x].tolist(), f"alignment of target {k}"


def test_values_beyond_the_half_floats_leave_the_form(capi, lane_per_pair):
    # a query against copies of itself: scores in the thousands times eight do not fit the scan's patterns; the
    # direction pass still does (its values are not scaled)
    rng = np.random.default_rng(29)
    q = _data.random_protein(rng, 600)
    res, off = related(rng, q, 60, edits=6, flank=10)
    got, routing = search(capi, q, res, off, B62, 11, 1)
    assert routing & PACKED_SCAN == 0, routing
    compare(got, _oracle.search(q, res, off, B62, 11, 1, "full", "sw"), "full", "near-identical copies")


def test_orders_and_overlaps_of_the_pipeline_do_not_change_results(capi, tuning):
    """Round 5's scheduling choices - the scan's pairs longest prefixes first (an index sort: intraseq.hip
    launchSortIndicesByKey), the windows' rows as a minor key of the direction jobs' sort, the next group's job lists
    built on the side stream, the host's shares on helper threads - are orders and overlaps, never answers: each
    switched off alone gives the arrays of the default, and a sample of them equals the checker's."""
    rng = np.random.default_rng(77)
    n = 270_000   # (four traceback batches; the scan's order needs four pairs per lane of the chip: 65 536)
    q = _data.encode(_data.README_QUERY)
    lengths = np.clip(rng.lognormal(mean=3.6, sigma=0.5, size=n), 5, 400).astype(np.int64)
    res, off = _data.random_db(rng, lengths)
    tuning.setenv("MIOPAL_NO_SMALL_SEARCH", "1")
    db = capi.DeviceDatabase(res, off, 24)
    try:
        base = db.search(q, B62, 3, 1, "full", "sw")
        routing = capi.DeviceDatabase.last_full_routing()
        assert routing & PACKED_TRACE and routing & PACKED_SCAN and routing & ONE_LAUNCH, routing
        base = {k: np.array(base[k], copy=True) for k in KEYS}
        for switch in ("MIOPAL_NO_SCAN_ORDER", "MIOPAL_NO_SORT_BY_ROWS", "MIOPAL_NO_JOBS_AHEAD", "MIOPAL_NO_ASYNC_SHARES"):
            tuning.setenv(switch, "1")
            got = db.search(q, B62, 3, 1, "full", "sw")
            tuning.delenv(switch)
            for k in KEYS:
                assert np.array_equal(got[k], base[k]), (switch, k)
    finally:
        db.close()
    sample = np.sort(rng.choice(n, size=300, replace=False))
    seqs = [res[off[t]:off[t + 1]] for t in sample]
    sub_off = np.zeros(len(seqs) + 1, dtype=np.int64)
    sub_off[1:] = np.cumsum([len(x) for x in seqs])
    want = _oracle.search(q, np.concatenate(seqs), sub_off, B62, 3, 1, "full", "sw")
    for rank, t in enumerate(sample):
        assert base["score"][t] == want["score"][rank]
        assert (base["end_q"][t], base["end_t"][t]) == (want["end_q"][rank], want["end_t"][rank])
        assert (base["start_q"][t], base["start_t"][t]) == (want["start_q"][rank], want["start_t"][rank])
        a0, a1 = base["aln_off"][t], base["aln_off"][t + 1]
        assert np.array_equal(base["aln_flat"][a0:a1], want["aln"][rank]), t


@pytest.mark.parametrize("algo,qlen", [("hw", 53), ("hw", 150), ("ov", 53), ("ov", 64), ("ov", 9), ("ov", 150), ("ov", 200)])
def test_hw_and_ov_start_cells_on_the_packed_scan(capi, lane_per_pair, tuning, algo, qlen):
    """OV: the answer in the pair's OWN last row - the reversed prefix ends where the forward pass
    ended, in the last row or the last column - or anywhere in its last column: a select tree over the rows per column.
    HW: the reversed-prefix scan answers only in the query's last row (perpair_packed.hip, scanLastRow), cells above it
    may exceed the optimum (targets of tryptophans against a query without one: every real cell is worse than the query
    in one gap; the case where THAT is the optimum at a real end cell - a border cell of the reversed problem, which no
    scan computes - is the property tier's). Against the checker, and against the 32-bit scan array by array."""
    rng = np.random.default_rng(5 + qlen)
    q = _data.random_protein(rng, qlen)
    w = int(_data.encode("W")[0])
    q[q == w] = int(_data.encode("A")[0])
    lengths = rng.integers(1, 320, size=900)
    res, off = _data.random_db(rng, lengths)
    for k in range(0, 900, 9):   # a ninth of the targets: nothing to align with
        res[off[k]:off[k + 1]] = w
    for k in range(4, 900, 9):   # another ninth: the query itself inside, edited
        if lengths[k] > qlen + 8:
            at = off[k] + rng.integers(0, lengths[k] - qlen - 4)
            res[at:at + qlen] = q
            res[at + qlen // 2] = w
    # (OV: targets that end inside the query, so that the forward pass ends in the last COLUMN - a copy of the query's
    # first half at the target's end)
    for k in range(7, 900, 9):
        half = qlen // 2
        if lengths[k] > half + 4 and half > 0:
            res[off[k + 1] - half:off[k + 1]] = q[:half]
    got, routing = search(capi, q, res, off, B62, 3, 1, algo)
    assert routing & PACKED_SCAN and routing & PACKED_TRACE, routing
    ref = _oracle.search(q, res, off, B62, 3, 1, "full", algo)
    compare(got, ref, "full", f"{algo} Q={qlen}")
    if algo == "ov":
        ends = np.asarray(got["end_q"])
        assert np.count_nonzero((ends >= 0) & (ends < qlen - 1)) > 20   # (pairs whose own last row is not the query's)
    tuning.setenv("MIOPAL_NO_PACKED_HW_SCAN", "1")
    wide, routing = search(capi, q, res, off, B62, 3, 1, algo)
    assert routing & PACKED_SCAN == 0, routing
    for key in KEYS:
        np.testing.assert_array_equal(got[key], wide[key], err_msg=key)
