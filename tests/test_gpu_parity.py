"""Parity of the HIP path (through the C ABI) against the CPU checker.

Bit-exact: scores, end/start locations and alignment operations are integers.
"""
import json
import os

import numpy as np
import pytest

import _cpu_baseline
import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
B50 = np.array(ScoringMatrix.from_name("BLOSUM50").int_array(), dtype=np.int32)
B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
ALGOS = ["nw", "hw", "ov", "sw"]


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


def compare(gpu, ref, mode, tag=""):
    np.testing.assert_array_equal(gpu["score"], ref["score"], err_msg=f"score {tag}")
    if mode in ("end", "full"):
        np.testing.assert_array_equal(gpu["end_q"], ref["end_q"], err_msg=f"end_q {tag}")
        np.testing.assert_array_equal(gpu["end_t"], ref["end_t"], err_msg=f"end_t {tag}")
    if mode == "full":
        np.testing.assert_array_equal(gpu["start_q"], ref["start_q"], err_msg=f"start_q {tag}")
        np.testing.assert_array_equal(gpu["start_t"], ref["start_t"], err_msg=f"start_t {tag}")
        for k, (a, b) in enumerate(zip(gpu["aln"], ref["aln"])):
            assert a.tolist() == b.tolist(), f"alignment {k} {tag}"


def run_both(capi, query, res, off, matrix, go, ge, mode, algo, **kw):
    db = capi.DeviceDatabase(res, off, 24)
    try:
        gpu = db.search(query, matrix, go, ge, mode, algo, **kw)
    finally:
        db.close()
    start = kw.get("start", 0)
    end = kw.get("end", len(off) - 1)
    end = min(end if end is not None else len(off) - 1, len(off) - 1)
    sub_off = off[start:end + 1] - off[start]
    sub_res = res[off[start]:off[end]]
    ref = _oracle.search(query, sub_res, sub_off, matrix, go, ge, mode, algo)
    return gpu, ref


# ---- the reference's own known-answer vectors through the C ABI -------------
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    VECTORS = json.load(f)["vectors"]


@pytest.mark.parametrize("vec", VECTORS, ids=[v["id"] for v in VECTORS])
@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_reference_vectors_small_search_routing(capi, vec, mode, small_search_routing):
    # the reference's vectors are tiny databases: in production they take the
    # wavefront-per-pair kernels (host.hip, kSmallSearch)
    test_reference_vectors(capi, vec, mode)


def test_results_written_into_the_arrays_of_an_earlier_search(capi):
    """`reuse=`: a caller that searches again with the per-target arrays of an earlier result (pyopal_amd/_capi.py)."""
    rng = np.random.default_rng(61)
    res, off = _data.random_db(rng, rng.integers(1, 300, size=3000))
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for mode in ("score", "end", "full"):
            q1, q2 = _data.random_protein(rng, 40), _data.random_protein(rng, 70)
            first = db.search(q1, B62, 3, 1, mode, "sw")
            kept = {k: v for k, v in first.items() if isinstance(v, np.ndarray) and k != "aln_flat"}
            second = db.search(q2, B62, 3, 1, mode, "sw", reuse=first)
            for k, v in kept.items():
                assert second[k] is v, k            # the same arrays, written again
            ref = _oracle.search(q2, res, off, B62, 3, 1, mode, "sw")
            compare(second, ref, mode, f"reuse {mode}")
            # arrays of another shape or type are not taken
            other = db.search(q2, B62, 3, 1, mode, "sw", 0, 100, reuse=second)
            assert other["score"] is not second["score"] and len(other["score"]) == 100
        # the operations buffer (miopalSearchFlatInto): written in place when it is large enough ...
        q3 = _data.random_protein(rng, 70)
        a = db.search(q3, B62, 3, 1, "full", "sw")
        a["aln_flat"][:] = 7   # (whatever the earlier result held is gone)
        b = db.search(q3, B62, 3, 1, "full", "sw", reuse=a)
        assert b["_ops_owner"] is a["_ops_owner"]
        compare(b, _oracle.search(q3, res, off, B62, 3, 1, "full", "sw"), "full", "operations in place")
        part = db.search(q3, B62, 3, 1, "full", "sw", 10, 60, reuse=b)   # (fewer targets: the same buffer, a shorter view of it)
        assert part["_ops_owner"] is b["_ops_owner"]
        compare(part, _oracle.search(q3, res[off[10]:off[60]], off[10:61] - off[10], B62, 3, 1, "full", "sw"), "full", "a slice in place")
        # ... and left alone, with its owner, when it is too small
        small = db.search(q3, B62, 3, 1, "full", "sw", 0, 50)
        before = small["aln_flat"].copy()
        big = db.search(_data.random_protein(rng, 300), B62, 3, 1, "full", "sw", reuse=small)
        assert big["_ops_owner"] is not small["_ops_owner"]
        assert np.array_equal(small["aln_flat"], before)
        again = db.search(q3, B62, 3, 1, "full", "sw", reuse=big)
        assert again["_ops_owner"] is big["_ops_owner"]
        compare(again, _oracle.search(q3, res, off, B62, 3, 1, "full", "sw"), "full", "after a larger search")
        empty = db.search(q3, B62, 3, 1, "full", "sw", 5, 5, reuse=again)
        assert len(empty["aln_flat"]) == 0 and len(empty["score"]) == 0
    finally:
        db.close()


@pytest.mark.parametrize("qlen", [65, 150, 300, 1000])
def test_small_searches_of_longer_queries(capi, small_search_routing, qlen):
    """Production routing of small searches (host_search.inc, kSmallSteps): a few targets against a query of
    several strips stay on the wavefront-per-pair kernel (shorter start-up than the packed kernels' views and
    tables) while an estimate of its time stays below one of theirs; more targets, or long ones, take the packed
    kernels."""
    rng = np.random.default_rng(300 + qlen)
    q = _data.random_protein(rng, qlen)
    for n, lengths_hi in ((1, 400), (7, 400), (300, 300)):
        lengths = rng.integers(1, lengths_hi, size=n)
        res, off = _data.random_db(rng, lengths)
        strips = -(-qlen // 64)
        steps = strips * (int(off[-1]) + 63 * n)
        # (the host's estimate, host_search.inc)
        per_pair = 0.075 + max(steps / 7e6, (int(lengths.max()) + 64 * strips) * 0.00025)
        expect_small = per_pair < 0.26 + 0.0006 * qlen
        for algo in ALGOS:
            for mode in ("score", "end", "full"):
                gpu, ref = run_both(capi, q, res, off, B62, 3, 1, mode, algo)
                compare(gpu, ref, mode, f"{algo}/{mode} Q={qlen} n={n}")
                if mode == "score":
                    routed = capi.DeviceDatabase.last_routing()
                    assert (routed[0] == n) == expect_small, (routed, steps, per_pair)
        assert expect_small or n == 300, "the handful of targets is what the routing is for"
    # beyond the bound: the packed kernels
    res, off = _data.random_db(rng, np.full(20_000, 300))
    gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "score", "sw")
    compare(gpu, ref, "score", f"20000 x 300 Q={qlen}")
    assert capi.DeviceDatabase.last_routing()[0] == 0


@pytest.mark.parametrize("vec", VECTORS, ids=[v["id"] for v in VECTORS])
@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_reference_vectors(capi, vec, mode):
    m = np.array(ScoringMatrix.from_name(vec["matrix"]).int_array(), dtype=np.int32)
    q = _oracle.encode(vec["query"])
    res, off = _oracle.flatten([_oracle.encode(t) for t in vec["targets"]])
    gpu, ref = run_both(capi, q, res, off, m, vec["gap_open"], vec["gap_extend"], mode, vec["algorithm"])
    compare(gpu, ref, mode, vec["id"])
    for k, want in enumerate(vec["score"]):
        if want is not None:
            assert int(gpu["score"][k]) == want
    if mode == "full" and "cigar" in vec:
        assert gpu["aln"][0].tolist() == [2, 0, 0, 0, 3, 0, 2, 0]


def test_readme_example(capi):
    # BASELINE.json configs[0]: README query vs its 4 targets, BLOSUM50, SW score
    q = _oracle.encode(_data.README_QUERY)
    res, off = _oracle.flatten([_oracle.encode(t) for t in _data.README_TARGETS])
    for mode in ("score", "end", "full"):
        gpu, ref = run_both(capi, q, res, off, B50, 3, 1, mode, "sw")
        compare(gpu, ref, mode, "readme")


# ---- seeded random databases --------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_random_small(capi, algo, mode):
    rng = np.random.default_rng(11)
    lengths = rng.integers(1, 400, size=300)
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 53)
    gpu, ref = run_both(capi, q, res, off, B62, 3, 1, mode, algo)
    compare(gpu, ref, mode, f"{algo}/{mode}")


@pytest.mark.parametrize("algo", ALGOS)
def test_related_sequences_with_ties(capi, algo):
    # noisy copies of the query give long alignments, many equal-scoring cells
    rng = np.random.default_rng(5)
    q = _data.random_protein(rng, 120)
    seqs = [_data.mutate(rng, q, rate) for rate in np.linspace(0.0, 0.6, 60)]
    seqs += [np.concatenate([_data.random_protein(rng, 30), s, _data.random_protein(rng, 17)]) for s in seqs[:20]]
    res, off = _oracle.flatten(seqs)
    for go, ge in ((3, 1), (11, 1), (2, 2), (0, 0), (5, 0)):
        gpu, ref = run_both(capi, q, res, off, B62, go, ge, "full", algo)
        compare(gpu, ref, "full", f"{algo} gaps {go}/{ge}")


@pytest.mark.parametrize("switch", ["MIOPAL_NO_PAIR_TABLE", "MIOPAL_NO_DIAG_SHIFT"])
def test_alternative_kernel_variants(capi, tuning, switch):
    # the variants the default dispatch does not pick on small inputs: v_perm profile
    # fetch for one-strip SW, unshifted signed lanes for NW / HW / OV
    tuning.setenv(switch, "1")
    rng = np.random.default_rng(17)
    res, off = _data.random_db(rng, rng.integers(1, 500, size=700))
    for qlen in (53, 150, 600):
        q = _data.random_protein(rng, qlen)
        for algo in ALGOS:
            for mode in ("score", "end"):
                gpu, ref = run_both(capi, q, res, off, B62, 3, 1, mode, algo)
                compare(gpu, ref, mode, f"{switch} {algo}/{mode} Q={qlen}")


@pytest.mark.parametrize("switch", ["MIOPAL_HOST_TRACEBACK", "MIOPAL_NO_PERPAIR", "MIOPAL_NO_SIDE_STREAM",
                                    "MIOPAL_NO_HYBRID_TRACE"])
def test_alternative_full_mode_paths(capi, tuning, switch):
    # fallbacks of `full`: traceback batches built on the host, wavefront-per-pair kernels for
    # one-strip queries, long targets recomputed after (not beside) the packed kernel
    tuning.setenv(switch, "1")
    tuning.setenv("MIOPAL_NO_SEGMENTS", "1")
    rng = np.random.default_rng(23)
    lengths = rng.integers(1, 400, size=5000)
    lengths[:40] = rng.integers(2000, 9000, size=40)
    res, off = _data.random_db(rng, lengths)
    for qlen in (53, 130):
        q = _data.random_protein(rng, qlen)
        for algo in ALGOS:
            gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "full", algo)
            compare(gpu, ref, "full", f"{switch} {algo} Q={qlen}")


def test_full_in_batches_with_and_without_the_side_stream_copies(capi, tuning):
    """A `full` search of enough targets for several batches (host_full.inc: the per-target arrays leave on the
    side stream batch by batch and are copied out and prefix-summed between batches) against the same search
    with everything sent behind the last batch (`MIOPAL_NO_SIDE_COPIES`), and the head of it against the checker."""
    rng = np.random.default_rng(77)
    n = 4 * 65536 + 1000
    res, off = _data.random_db(rng, rng.integers(20, 90, size=n))
    q = _data.random_protein(rng, 40)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for algo in ("sw", "hw"):
            a = db.search(q, B62, 3, 1, "full", algo)
            tuning.setenv("MIOPAL_NO_SIDE_COPIES", "1")
            b = db.search(q, B62, 3, 1, "full", algo)
            tuning.delenv("MIOPAL_NO_SIDE_COPIES")
            # (the host's share of a batch behind the next batch's gather, the last share without the crew: round 4's first form)
            tuning.setenv("MIOPAL_NO_EARLY_HOST_SHARE", "1")
            tuning.setenv("MIOPAL_NO_UNPACK_CREW", "1")
            d = db.search(q, B62, 3, 1, "full", algo)
            tuning.delenv("MIOPAL_NO_EARLY_HOST_SHARE")
            tuning.delenv("MIOPAL_NO_UNPACK_CREW")
            for key in ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat"):
                np.testing.assert_array_equal(d[key], b[key], err_msg=f"{algo} {key} (late host share)")
            c = db.search(q, B62, 3, 1, "full", algo, reuse=a)   # (and into the arrays and the operations buffer of `a`)
            for key in ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat"):
                np.testing.assert_array_equal(c[key], b[key], err_msg=f"{algo} {key}")
            head = 3000
            ref = _oracle.search(q, res[:off[head]], off[:head + 1], B62, 3, 1, "full", algo)
            part = {k: (v[:head] if k != "aln" else v[:head]) for k, v in c.items() if k in ("score", "end_q", "end_t", "start_q", "start_t", "aln")}
            compare(part, ref, "full", f"batches {algo}")
    finally:
        db.close()


@pytest.mark.parametrize("qlen", [1, 7, 8, 9, 53, 63, 64, 65, 100, 128, 129, 200, 333])
def test_query_lengths_sw_score(capi, qlen):
    # strip boundaries of the inter-sequence kernel (8-row blocks, 64-row strips)
    rng = np.random.default_rng(qlen)
    lengths = rng.integers(1, 300, size=400)
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "score", "sw")
    compare(gpu, ref, "score", f"Q={qlen}")


def test_edge_cases(capi):
    rng = np.random.default_rng(2)
    seqs = [np.zeros(0, dtype=np.uint8), _data.random_protein(rng, 1), _data.random_protein(rng, 2),
            np.zeros(0, dtype=np.uint8), _data.random_protein(rng, 5),
            _oracle.encode("WWWWWWWW"), _oracle.encode("X*BZX*BZ")]
    res, off = _oracle.flatten(seqs)
    for q in (_oracle.encode("W"), _oracle.encode("ACDWWWWWWY"), _data.random_protein(rng, 70)):
        for algo in ALGOS:
            for mode in ("score", "end", "full"):
                gpu, ref = run_both(capi, q, res, off, B62, 3, 1, mode, algo)
                compare(gpu, ref, mode, f"edge {algo}/{mode}/Q={len(q)}")


def test_slices(capi):
    rng = np.random.default_rng(9)
    res, off = _data.random_db(rng, rng.integers(5, 200, size=257))
    q = _data.random_protein(rng, 40)
    for start, end in ((0, 257), (0, 1), (5, 5), (10, 139), (128, 257), (200, 10_000)):
        for mode in ("score", "full"):
            gpu, ref = run_both(capi, q, res, off, B62, 3, 1, mode, "sw", start=start, end=end)
            compare(gpu, ref, mode, f"slice {start}:{end}")


def test_sw_int16_saturation(capi):
    # identical long sequences: SW score = sum of diagonal scores >> 32767,
    # forcing the 16 -> 32 bit recompute for some targets only
    rng = np.random.default_rng(4)
    q = _data.random_protein(rng, 7000)
    seqs = [q.copy(), _data.mutate(rng, q, 0.05), _data.random_protein(rng, 6000), q[:5000].copy(),
            _data.random_protein(rng, 300), q[1000:6000].copy()]
    seqs += [_data.random_protein(rng, int(n)) for n in rng.integers(10, 500, size=200)]
    res, off = _oracle.flatten(seqs)
    gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "score", "sw")
    assert ref["score"].max() > 32767
    compare(gpu, ref, "score", "saturation")


@pytest.mark.parametrize("first_rung", ["shifted", "half"])
def test_sw_lane_width_ladder(capi, tuning, first_rung):
    # half-float lanes are exact below 2048, int16 lanes below 32767, then int32:
    # thousands of close homologues push most targets past the first rung (so the
    # whole view is redone with int16 lanes), a few past the second. The column-shifted first rung
    # (round 2) holds the 480-residue query's hits itself and only hands the 6500-residue ones on.
    if first_rung == "half":
        tuning.setenv("MIOPAL_NO_SW_SHIFT", "1")
        tuning.setenv("MIOPAL_NO_PAIR_STRIPS", "1")
    rng = np.random.default_rng(14)
    q = _data.random_protein(rng, 6500)
    short = q[:480].copy()
    seqs = [_data.mutate(rng, short, 0.08) for _ in range(2300)]
    seqs += [_data.random_protein(rng, int(n)) for n in rng.integers(20, 600, size=700)]
    seqs += [q.copy(), _data.mutate(rng, q, 0.02)]
    res, off = _oracle.flatten(seqs)
    for query in (short, q):
        gpu, ref = run_both(capi, query, res, off, B62, 3, 1, "score", "sw")
        assert (ref["score"] >= 2048).sum() > 2048
        compare(gpu, ref, "score", f"ladder Q={len(query)}")
    assert ref["score"].max() > 32767


@pytest.mark.parametrize("algo", ALGOS)
def test_long_targets_overflow_ladder(capi, algo):
    # shape of src/pyopal/tests/test_aligner.py:24-37 (lengths 1000..35000,
    # query = the 1000-aa one); the reference asserts no values, the oracle does.
    # Sub-sampled to keep the scalar checker to a few seconds.
    rng = np.random.default_rng(0)
    lengths = [1000, 2000, 9000, 17000, 35000, 45000]
    seqs = [_data.random_protein(rng, n) for n in lengths]
    res, off = _oracle.flatten(seqs)
    gpu, ref = run_both(capi, seqs[0], res, off, B50, 3, 1, "score", algo)
    compare(gpu, ref, "score", algo)
    if algo == "nw":
        assert ref["score"].min() < -32768  # really leaves the 16-bit range


@pytest.mark.parametrize("A", [4, 20, 32])
def test_other_alphabets(capi, A):
    # alphabets from 4 letters to the 32-letter maximum (src/pyopal/lib.pxd:28-32),
    # asymmetric matrix on purpose: rows are indexed by the QUERY residue
    rng = np.random.default_rng(A)
    matrix = rng.integers(-6, 8, size=(A, A)).astype(np.int32)
    matrix[np.arange(A), np.arange(A)] = rng.integers(3, 12, size=A)
    seqs = [rng.integers(0, A, size=int(n)).astype(np.uint8) for n in rng.integers(1, 260, size=300)]
    res, off = _oracle.flatten(seqs)
    for qlen in (37, 150):
        q = rng.integers(0, A, size=qlen).astype(np.uint8)
        for algo in ALGOS:
            db = capi.DeviceDatabase(res, off, A)
            try:
                for mode in ("score", "full"):
                    gpu = db.search(q, matrix.ravel(), 5, 2, mode, algo)
                    ref = _oracle.search(q, res, off, matrix.ravel(), 5, 2, mode, algo)
                    compare(gpu, ref, mode, f"A={A} {algo} {mode} Q={qlen}")
            finally:
                db.close()


def test_opal_search_database_entry(capi):
    """The literal opal.h entry point (host pointers in, result structs out)."""
    import ctypes
    rng = np.random.default_rng(3)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 120, size=37)]
    q = _data.random_protein(rng, 31)
    res, off = _oracle.flatten(seqs)
    for mode, algo in (("score", "sw"), ("end", "hw"), ("full", "ov"), ("full", "nw")):
        ref = _oracle.search(q, res, off, B62, 3, 1, mode, algo)
        n = len(seqs)
        results = (capi.OpalSearchResult * n)()
        rptr = (ctypes.POINTER(capi.OpalSearchResult) * n)()
        for k in range(n):
            capi.lib().opalInitSearchResult(ctypes.byref(results[k]))
            rptr[k] = ctypes.pointer(results[k])
        seq_ptrs = (ctypes.c_void_p * n)(*[s.ctypes.data for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        rc = capi.lib().opalSearchDatabase(
            q.ctypes.data, len(q), ctypes.cast(seq_ptrs, ctypes.c_void_p), n, lens.ctypes.data, 3, 1,
            B62.ctypes.data, 24, ctypes.cast(rptr, ctypes.c_void_p), capi.SEARCH[mode], capi.MODE[algo], 1)
        assert rc == 0, capi.last_error()
        for k in range(n):
            r = results[k]
            assert r.scoreSet and r.score == ref["score"][k]
            if mode != "score":
                assert (r.endLocationQuery, r.endLocationTarget) == (ref["end_q"][k], ref["end_t"][k])
            if mode == "full":
                assert (r.startLocationQuery, r.startLocationTarget) == (ref["start_q"][k], ref["start_t"][k])
                assert [r.alignment[i] for i in range(r.alignmentLength)] == ref["aln"][k].tolist()


def test_device_scores_entry(capi):
    """Results left in HBM (miopalSearchDeviceScores), torch only as the allocator."""
    import torch
    rng = np.random.default_rng(21)
    res, off = _data.random_db(rng, rng.integers(1, 350, size=1000))
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    out = torch.full((1000,), -7, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for algo in ALGOS:
        db.search_device_scores(q, B62, out.data_ptr(), stream, 3, 1, algo)
        torch.cuda.synchronize()
        ref = _oracle.search(q, res, off, B62, 3, 1, "score", algo)
        np.testing.assert_array_equal(out.cpu().numpy(), ref["score"], err_msg=algo)
    db.close()


def test_errors(capi):
    rng = np.random.default_rng(1)
    res, off = _data.random_db(rng, [10, 20])
    db = capi.DeviceDatabase(res, off, 24)
    q = _data.random_protein(rng, 5)
    with pytest.raises(RuntimeError, match="code=3"):
        capi.raise_for(capi.lib().miopalSearch(db.handle, q.ctypes.data, 5, 3, 1, B62.ctypes.data, 24, 0, 9,
                                               0, 2, q.ctypes.data, None, None, None, None, None, None))
    with pytest.raises(RuntimeError, match="alphabet"):
        db.alphabet_length = 20
        db.search(q, B62[:400], 3, 1, "score", "sw")
    db.alphabet_length = 24
    big = (B62.astype(np.int64) * 0 + 2 ** 28).astype(np.int32)
    with pytest.raises(OverflowError):
        db.search(q, big, 3, 1, "score", "nw")
    db.close()


@pytest.mark.parametrize("qlen", [53, 24, 64])
def test_long_groups_beside_the_packed_kernel(capi, qlen, tuning):
    """Variable-length database without segmented views: the longest groups of a one-strip
    Smith-Waterman search are computed by the wavefront-per-pair kernel on a side stream, the
    rest by the lane-per-target kernel - both must agree with the checker, for scores, end
    locations and full alignments. (What NW / HW / OV searches of such a database do.)"""
    tuning.setenv("MIOPAL_NO_SEGMENTS", "1")
    rng = np.random.default_rng(100 + qlen)
    lengths = np.clip(rng.lognormal(5.3, 0.5, size=60_000), 10, 1500).astype(np.int64)
    lengths[rng.integers(0, len(lengths), size=700)] = rng.integers(1500, 3500, size=700)   # long groups
    lengths[rng.integers(0, len(lengths), size=6)] = rng.integers(7000, 8100, size=6)       # extreme
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    # plant a few strong hits inside long targets so that locations far from the ends are checked
    order = np.argsort(lengths)
    for k in order[-40:]:
        at = off[k] + lengths[k] // 2
        copy = _data.mutate(rng, q, 0.1)
        m = min(len(copy), int(lengths[k] // 2))
        res[at:at + m] = copy[:m]
    db = capi.DeviceDatabase(res, off, 24)
    try:
        score = db.search(q, B62, 3, 1, "score", "sw")
        routed = capi.DeviceDatabase.last_routing()
        assert routed[0] >= 128 and routed[2] > 0, f"expected long groups on the int32 kernel, got {routed}"
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search_sw(q, B62, 3, 1, 8)
        cpu.close()
        np.testing.assert_array_equal(score["score"], want)
        end = db.search(q, B62, 3, 1, "end", "sw")
        routed = capi.DeviceDatabase.last_routing()
        assert routed[0] >= 128 and routed[2] > 0, f"expected long groups on the int32 kernel, got {routed}"
        np.testing.assert_array_equal(end["score"], want)
        # the checker on the longest 150 targets and on a random sample
        sample = np.unique(np.concatenate([order[-150:], rng.integers(0, len(lengths), size=300)]))
        sub = [res[off[k]:off[k + 1]] for k in sample]
        sres, soff = _oracle.flatten(sub)
        ref = _oracle.search(q, sres, soff, B62, 3, 1, "full", "sw")
        for key in ("score", "end_q", "end_t"):
            np.testing.assert_array_equal(end[key][sample], ref[key], err_msg=key)
        if qlen == 53:
            full = db.search(q, B62, 3, 1, "full", "sw")
            for key in ("score", "end_q", "end_t", "start_q", "start_t"):
                np.testing.assert_array_equal(full[key][sample], ref[key], err_msg=key)
            for x, k in enumerate(sample):
                assert full["aln"][int(k)].tolist() == ref["aln"][x].tolist(), f"alignment of target {k}"
    finally:
        db.close()


@pytest.mark.parametrize("qlen,gaps,matrix", [(53, (3, 1), "B62"), (20, (3, 1), "B62"), (64, (11, 1), "B50"),
                                                (53, (1, 2), "B62"), (40, (5, 2), "B62"), (53, (3, 0), "B62"),
                                                (120, (3, 1), "B62")])
def test_segmented_views_every_score(capi, qlen, gaps, matrix):
    """Smith-Waterman scores of long targets as the maximum over overlapping windows
    (host.hip, `overlap`): every score equals the CPU baseline's, for several window sizes
    (the reach Q + Q max(S) / min(open, ext) varies), with targets beyond the packed kernel's
    usual length limit, and for a gap model that rules segmentation out (ext = 0)."""
    rng = np.random.default_rng(7 * qlen + gaps[0])
    lengths = np.clip(rng.lognormal(5.3, 0.6, size=30_000), 5, 2500).astype(np.int64)
    lengths[rng.integers(0, len(lengths), size=300)] = rng.integers(2500, 8100, size=300)
    lengths[rng.integers(0, len(lengths), size=4)] = [9000, 12_345, 8193, 20_000]
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    order = np.argsort(lengths)
    for k in order[-60:]:     # strong hits at random places of the longest targets, some across window borders
        copy = _data.mutate(rng, q, 0.15)
        at = int(rng.integers(0, max(1, lengths[k] - len(copy))))
        m = min(len(copy), int(lengths[k]) - at)
        res[off[k] + at:off[k] + at + m] = copy[:m]
    mat = B62 if matrix == "B62" else B50
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, mat, gaps[0], gaps[1], "score", "sw")["score"]
        routed = capi.DeviceDatabase.last_routing()
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search_sw(q, mat, gaps[0], gaps[1], 8)
        cpu.close()
        np.testing.assert_array_equal(got, want)
        if gaps[1] > 0 and qlen <= 64:
            assert routed[0] < 64, f"long targets should stay in the packed kernel, got {routed}"
        part = db.search(q, mat, gaps[0], gaps[1], "score", "sw", 1000, 20_000)["score"]
        np.testing.assert_array_equal(part, want[1000:20_000])
        # end locations and alignments: the windows' first maxima merged by (score, column, row)
        end = db.search(q, mat, gaps[0], gaps[1], "end", "sw")
        np.testing.assert_array_equal(end["score"], want)
        if gaps[1] > 0 and qlen <= 64:
            assert capi.DeviceDatabase.last_routing()[0] < 64
        sample = np.unique(np.concatenate([order[-120:], rng.integers(0, len(lengths), size=200)]))
        sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
        ref = _oracle.search(q, sres, soff, mat, gaps[0], gaps[1], "full" if qlen == 53 else "end", "sw")
        for key in ("score", "end_q", "end_t"):
            np.testing.assert_array_equal(end[key][sample], ref[key], err_msg=key)
        if qlen == 53:
            full = db.search(q, mat, gaps[0], gaps[1], "full", "sw")
            for key in ("score", "end_q", "end_t", "start_q", "start_t"):
                np.testing.assert_array_equal(full[key][sample], ref[key], err_msg=key)
            for x, k in enumerate(sample):
                assert full["aln"][int(k)].tolist() == ref["aln"][x].tolist(), f"alignment of target {k}"
    finally:
        db.close()


@pytest.mark.parametrize("rich", [False, True])
@pytest.mark.parametrize("gaps", [(3, 1), (1, 1), (2, 3)])
def test_segmented_views_alignments_across_long_gaps(capi, rich, gaps):
    """The windows of a long target overlap by the farthest a positive local alignment can reach:
    Q + (sum over the query of each residue's best score) / min(open, ext) columns (host_search.inc,
    `pairsBest`; it was Q max(S) before). Alignments that really use that reach: the two halves of the
    query, exact copies, with up to hundreds of unrelated residues between them, anywhere in targets
    of thousands of residues - for a query of ordinary composition and for one of tryptophans and
    cysteines (sum of best scores close to Q max(S)). Every score, end location and - on a sample - start
    location against the checker."""
    rng = np.random.default_rng(11 + gaps[0] + 10 * rich)
    q = _data.random_protein(rng, 48)
    if rich:
        q = rng.choice(_data.encode("WCWWHC"), size=48).astype(np.uint8)
    half = len(q) // 2
    seqs = []
    for gap_columns in list(range(0, 120, 3)) + list(range(120, 640, 13)):
        length = int(rng.integers(1500, 6000))
        t = _data.random_protein(rng, length)
        insert = np.concatenate([q[:half], _data.random_protein(rng, gap_columns), q[half:]])
        at = int(rng.integers(0, length - len(insert)))
        t[at:at + len(insert)] = insert
        seqs.append(t)
    # short targets around them, so that the long ones are the tail of a packed view
    seqs += [_data.random_protein(rng, int(n)) for n in rng.integers(20, 400, size=3000)]
    res, off = _oracle.flatten(seqs)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        end = db.search(q, B62, gaps[0], gaps[1], "end", "sw")
        assert capi.DeviceDatabase.last_routing()[0] < 64, "long targets should stay in the packed kernel"
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search_sw(q, B62, gaps[0], gaps[1], 8)
        cpu.close()
        np.testing.assert_array_equal(end["score"], want)
        sample = np.arange(0, 81)
        sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
        ref = _oracle.search(q, sres, soff, B62, gaps[0], gaps[1], "full", "sw")
        full = db.search(q, B62, gaps[0], gaps[1], "full", "sw", 0, 81)
        for key in ("score", "end_q", "end_t", "start_q", "start_t"):
            np.testing.assert_array_equal(full[key], ref[key], err_msg=key)
        np.testing.assert_array_equal(end["end_t"][:81], ref["end_t"])
        # some of these alignments do span the gap: longer than the query
        spans = ref["end_t"] - ref["start_t"] + 1
        assert (spans > len(q) + 20).sum() >= 5, spans
    finally:
        db.close()


@pytest.mark.parametrize("qlen,gaps,matrix", [(53, (3, 1), "B62"), (20, (3, 1), "B62"), (64, (11, 1), "B50"),
                                                (53, (1, 2), "B62"), (30, (5, 2), "B62"), (53, (3, 0), "B62"),
                                                (100, (11, 1), "B62")])
def test_segmented_views_hw(capi, qlen, gaps, matrix):
    """HW (whole query, free ends in the target) on long targets cut into overlapping windows: every
    score against the AVX2 checker, end locations and alignments of the longest targets against the
    scalar one. Scores are negative for most targets (the merge starts from minus infinity, the keys
    carry a bias); hits are planted across window borders; ext = 0 rules windows out."""
    rng = np.random.default_rng(11 * qlen + gaps[0])
    lengths = np.clip(rng.lognormal(5.3, 0.6, size=20_000), 1, 2500).astype(np.int64)
    lengths[rng.integers(0, len(lengths), size=300)] = rng.integers(2500, 8100, size=300)
    lengths[rng.integers(0, len(lengths), size=4)] = [9000, 12_345, 8193, 20_000]
    lengths[:3] = [0, 1, 2]
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    order = np.argsort(lengths)
    for k in order[-60:]:
        copy = _data.mutate(rng, q, 0.1)
        at = int(rng.integers(0, max(1, lengths[k] - len(copy))))
        m = min(len(copy), int(lengths[k]) - at)
        res[off[k] + at:off[k] + at + m] = copy[:m]
    mat = B62 if matrix == "B62" else B50
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, mat, gaps[0], gaps[1], "score", "hw")["score"]
        routed = capi.DeviceDatabase.last_routing()
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search(q, mat, gaps[0], gaps[1], "hw", 8)
        cpu.close()
        np.testing.assert_array_equal(got, want)
        if gaps[1] > 0 and gaps[0] > 0 and qlen <= 64:
            assert routed[0] < 64, f"long targets should stay in the packed kernel, got {routed}"
        part = db.search(q, mat, gaps[0], gaps[1], "score", "hw", 1000, 15_000)["score"]
        np.testing.assert_array_equal(part, want[1000:15_000])
        end = db.search(q, mat, gaps[0], gaps[1], "end", "hw")
        np.testing.assert_array_equal(end["score"], want)
        sample = np.unique(np.concatenate([order[-120:], rng.integers(0, len(lengths), size=200), [0, 1, 2]]))
        sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
        mode = "full" if qlen == 53 else "end"
        ref = _oracle.search(q, sres, soff, mat, gaps[0], gaps[1], mode, "hw")
        for key in ("score", "end_q", "end_t"):
            np.testing.assert_array_equal(end[key][sample], ref[key], err_msg=key)
        if mode == "full":
            full = db.search(q, mat, gaps[0], gaps[1], "full", "hw")
            for key in ("score", "end_q", "end_t", "start_q", "start_t"):
                np.testing.assert_array_equal(full[key][sample], ref[key], err_msg=key)
            for x, k in enumerate(sample):
                assert full["aln"][int(k)].tolist() == ref["aln"][x].tolist(), f"alignment of target {k}"
    finally:
        db.close()


@pytest.mark.parametrize("config", ["", "4,4", "4,2", "4,1", "8,8", "8,4", "3,1", "6,2", "5,1", "16,8"])
def test_strip_configurations(capi, tuning, config):
    """Queries of more than 64 rows: every split into strips and wavefronts that the dispatch
    may pick (host.hip, cost model; MIOPAL_STRIPS forces one) gives the checker's scores and end
    locations - including requests that would leave the last strip without a query row, which
    must be refused rather than run."""
    if config:
        tuning.setenv("MIOPAL_STRIPS", config)
    rng = np.random.default_rng(171)
    res, off = _data.random_db(rng, rng.integers(1, 500, size=700))
    for qlen in (65, 100, 150, 200, 333):
        q = _data.random_protein(rng, qlen)
        for algo in ALGOS:
            gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "end", algo)
            compare(gpu, ref, "end", f"strips {config or 'default'} {algo} Q={qlen}")


@pytest.mark.parametrize("config", ["4,4", "4,2", "4,1", "8,4", "8,2", "6,2", "5,1", "16,8", "12,4"])
def test_unit_mode_of_multi_round_score_searches(capi, tuning, config):
    """Scores of queries whose strips take several rounds: with MIOPAL_UNITS=1 a workgroup takes
    (group, round) units from a counter instead of owning a group; boundary rows and the partial
    answers of a wavefront travel through HBM between rounds, possibly between workgroups. Every
    algorithm (the all-cells maximum of SW and the last-row / last-column answers of HW / OV are
    carried across rounds), ragged groups, more units than workgroups can be resident."""
    tuning.setenv("MIOPAL_STRIPS", config)
    tuning.setenv("MIOPAL_UNITS", "1")
    rng = np.random.default_rng(173)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 400, size=1500)]
    for qlen in (150, 333, 700):
        q = _data.random_protein(rng, qlen)
        extra = [np.concatenate([_data.random_protein(rng, 40), _data.mutate(rng, q, 0.1)]) for _ in range(6)]
        res, off = _oracle.flatten(seqs + extra)
        for algo in ALGOS:
            gpu, ref = run_both(capi, q, res, off, B62, 3, 1, "score", algo)
            compare(gpu, ref, "score", f"units {config} {algo} Q={qlen}")


@pytest.mark.parametrize("first_rung", ["shifted", "half"])
def test_segmented_view_with_lanes_leaving_the_half_float_range(capi, tuning, first_rung):
    """A window whose lane saturates the half-float rung (scores >= 2048) is flagged like any other
    lane, and its target recomputed whole by the next rungs - over the merged window maxima. The
    column-shifted first rung of round 2 (ArithSwU16) holds these scores itself: nothing is redone."""
    if first_rung == "half":
        tuning.setenv("MIOPAL_NO_SW_SHIFT", "1")
        tuning.setenv("MIOPAL_NO_PAIR_STRIPS", "1")
    rng = np.random.default_rng(5)
    lengths = np.clip(rng.lognormal(5.3, 0.5, size=20_000), 10, 1500).astype(np.int64)
    lengths[:30] = rng.integers(4000, 7000, size=30)
    res, off = _data.random_db(rng, lengths)
    w = _data.NCBI.index("W")
    q = np.concatenate([np.full(230, w, dtype=np.uint8), _data.random_protein(rng, 70)])
    for k in range(0, 30, 3):          # the query itself inside long targets: scores around 2800
        at = int(rng.integers(100, lengths[k] - 400))
        res[off[k] + at:off[k] + at + len(q)] = q
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, B62, 5, 2, "score", "sw")["score"]
        routed = capi.DeviceDatabase.last_routing()
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search_sw(q, B62, 5, 2, 8)
        cpu.close()
        assert want.max() >= 2048 and (routed[3] >= 1) == (first_rung == "half"), (int(want.max()), routed)
        np.testing.assert_array_equal(got, want)
        end = db.search(q, B62, 5, 2, "end", "sw")
        np.testing.assert_array_equal(end["score"], want)
        sample = np.arange(0, 60)
        sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
        ref = _oracle.search(q, sres, soff, B62, 5, 2, "end", "sw")
        for key in ("score", "end_q", "end_t"):
            np.testing.assert_array_equal(end[key][sample], ref[key], err_msg=key)
    finally:
        db.close()


def test_outlier_windows_in_the_direction_pass(capi):
    """A few long alignments among many short ones: the head of the sorted job list goes to the
    wavefront-per-pair kernel, the rest to the lane-per-pair kernel (host.hip, headWaves) - the
    alignments of both parts must equal the checker's."""
    rng = np.random.default_rng(99)
    n = 30_000
    res, off = _data.random_db(rng, np.full(n, 300))
    q = _data.random_protein(rng, 200)
    homologs = rng.choice(n, size=150, replace=False)
    for k in homologs:                      # noisy copies of the query: alignments of ~200 columns
        copy = _data.mutate(rng, q, 0.2)[:290]
        at = int(rng.integers(0, 300 - len(copy) + 1))
        res[off[k] + at:off[k] + at + len(copy)] = copy
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(q, B62, 11, 1, "full", "sw")
    finally:
        db.close()
    sample = np.unique(np.concatenate([homologs, rng.integers(0, n, size=400)]))
    sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
    ref = _oracle.search(q, sres, soff, B62, 11, 1, "full", "sw")
    spans = ref["end_t"] - ref["start_t"] + 1
    planted = np.isin(sample, homologs)
    assert np.median(spans[planted]) > 2 * np.percentile(spans[~planted], 90), "the test needs outliers"
    for key in ("score", "end_q", "end_t", "start_q", "start_t"):
        np.testing.assert_array_equal(got[key][sample], ref[key], err_msg=key)
    for x, k in enumerate(sample):
        assert got["aln"][int(k)].tolist() == ref["aln"][x].tolist(), f"alignment of target {k}"


@pytest.mark.parametrize("qlen", [65, 128, 129, 300])
def test_long_pairs_one_wavefront_per_strip(capi, qlen, tuning):
    """The wavefront-per-pair int32 kernel with the strips of a pair side by side (intraseq_strips_kernel):
    targets too long for a lane each (> 8192 residues), all modes, scores and end locations, lengths
    that are no multiple of the 64-column blocks, repeats (ties between strips) - against the checker and
    against the strip-after-strip kernel it replaces. Shorter targets beside them stay on the packed kernel."""
    rng = np.random.default_rng(500 + qlen)
    query = np.concatenate([_data.random_protein(rng, 40)] * 8)[:qlen]   # repeats: equal scores in different strips
    long_ones = [_data.random_protein(rng, int(n)) for n in (8193, 8200, 8255, 8256, 8257, 9001)]
    long_ones.append(np.concatenate([_data.random_protein(rng, 4000), np.tile(query[:40], 6), _data.random_protein(rng, 4100)]))
    long_ones.append(np.concatenate([_data.random_protein(rng, 8100), _data.mutate(rng, query, 0.1), _data.random_protein(rng, 77)]))
    seqs = long_ones + [_data.random_protein(rng, int(n)) for n in (1, 64, 700)] + [np.zeros(0, dtype=np.uint8)]
    res, off = _oracle.flatten(seqs)
    for algo in ALGOS:
        for mode in ("score", "end"):
            gpu, ref = run_both(capi, query, res, off, B62, 11, 1, mode, algo)
            compare(gpu, ref, mode, f"strip units {algo}/{mode}/Q={qlen}")
            if algo in ("nw", "ov"):   # (SW and HW searches see long targets through windows on the packed kernel)
                assert capi.DeviceDatabase.last_routing()[0] >= len(long_ones)   # the long pairs went to the int32 kernel
            tuning.setenv("MIOPAL_NO_PAIR_STRIP_UNITS", "1")
            old, _ = run_both(capi, query, res, off, B62, 11, 1, mode, algo)
            tuning.delenv("MIOPAL_NO_PAIR_STRIP_UNITS")
            compare(old, ref, mode, f"strip after strip {algo}/{mode}/Q={qlen}")


def test_scores_written_by_the_kernel_into_database_order(capi):
    # headline fast path (one strip, Smith-Waterman scores, nothing can leave its range): the kernel
    # scatters into database order itself - into the pinned staging buffer of miopalSearch, into a
    # caller's re-used array, into a caller's PINNED array (no copy at all) - for whole databases and
    # slices, ragged groups, and with the switches that restore the scatter kernel
    import os
    import torch
    rng = np.random.default_rng(404)
    query = _oracle.encode(_data.README_QUERY)
    res, off = _data.random_db(rng, rng.integers(1, 400, size=9000))
    want = _oracle.search_parallel(query, res, off, B62, 3, 1, "score", "sw")["score"]
    want_end = _oracle.search_parallel(query, res, off, B62, 3, 1, "end", "sw")
    db = capi.DeviceDatabase(res, off, 24)
    try:
        pinned = torch.empty(9000, dtype=torch.int32).pin_memory().numpy()
        plain = np.empty(9000, dtype=np.int32)
        for switch in (None, "MIOPAL_NO_HOST_SCATTER", "MIOPAL_NO_DIRECT_SCATTER", "MIOPAL_NO_CALLER_PINNED"):
            with capi.tuning(**({switch: "1"} if switch else {})):
                for lo, hi in ((0, 9000), (1, 9000), (4000, 4001), (137, 8999)):
                    got = db.search(query, B62, 3, 1, "score", "sw", lo, hi)["score"]
                    np.testing.assert_array_equal(got, want[lo:hi], err_msg=f"{switch} [{lo},{hi})")
                    # (end locations leave the kernel the same way: three arrays)
                    ends = db.search(query, B62, 3, 1, "end", "sw", lo, hi)
                    for key in ("score", "end_q", "end_t"):
                        np.testing.assert_array_equal(ends[key], want_end[key][lo:hi], err_msg=f"{switch} [{lo},{hi}) end {key}")
                    for buf in (plain, pinned):
                        buf[:] = -7
                        db.search(query, B62, 3, 1, "score", "sw", lo, hi, score_out=buf[: hi - lo])
                        np.testing.assert_array_equal(buf[: hi - lo], want[lo:hi], err_msg=f"{switch} [{lo},{hi}) into a buffer")
                        assert (buf[hi - lo:] == -7).all()
        # a result array that starts INSIDE a pinned allocation: the kernel must write at that address
        big = torch.empty(3 * 9000, dtype=torch.int32).pin_memory().numpy()
        big[:] = -7
        db.search(query, B62, 3, 1, "score", "sw", score_out=big[9000:18000])
        np.testing.assert_array_equal(big[9000:18000], want)
        assert (big[:9000] == -7).all() and (big[18000:] == -7).all()
        with pytest.raises(ValueError):
            db.search(query, B62, 3, 1, "score", "sw", score_out=np.empty(5, dtype=np.int32))
    finally:
        db.close()


@pytest.mark.parametrize("algo", ["sw", "nw", "hw", "ov"])
def test_refused_pair_table_launch_with_host_visible_outputs(capi, algo, tuning):
    # A one-strip search hands the pair-table kernel host-visible result arrays (the library's pinned staging
    # buffer, or the caller's own pinned array). When the runtime refuses that launch (its 150 KB of dynamic
    # LDS, say) the score pass starts over on the general kernel, which writes the DEVICE arrays: the result
    # must come from there, not from the host-visible buffer nothing wrote (round-3 advisor finding).
    import torch
    rng = np.random.default_rng(405)
    query = _oracle.encode(_data.README_QUERY)
    res, off = _data.random_db(rng, rng.integers(1, 300, size=6000))
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for mode in ("score", "end"):
            ref = _oracle.search_parallel(query, res, off, B62, 3, 1, mode, algo)
            plain = db.search(query, B62, 3, 1, mode, algo)
            compare(plain, ref, mode, f"{algo}/{mode}")
            routed = capi.DeviceDatabase.last_routing()[1]
            assert (routed & 15) in (4, 5) and not (routed & 16), routed      # the pair-table kernels ran
            tuning.setenv("MIOPAL_TEST_REFUSE_PAIR_LAUNCH", "1")
            for pinned in (False, True):
                buf = (torch.empty(6000, dtype=torch.int32).pin_memory().numpy() if pinned
                       else np.empty(6000, dtype=np.int32))
                buf[:] = -7
                got = db.search(query, B62, 3, 1, mode, algo, score_out=buf)
                routed = capi.DeviceDatabase.last_routing()[1]
                assert (routed & 15) == 1 or (routed & 16), "the general kernel took over"
                compare(got, ref, mode, f"{algo}/{mode} refused, pinned={pinned}")
                np.testing.assert_array_equal(buf, ref["score"])
            tuning.delenv("MIOPAL_TEST_REFUSE_PAIR_LAUNCH")
    finally:
        db.close()
