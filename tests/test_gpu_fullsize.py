"""Full-size checks at BASELINE.json's configurations (sizes the scalar checker
cannot finish): every score against the AVX2 CPU baseline for Smith-Waterman,
and size-independent properties for the other modes."""
import numpy as np
import pytest

import _cpu_baseline
import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

pytestmark = pytest.mark.gpu
B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1
    return _capi


def test_cfg2_every_score(capi):
    # configs[1]: 53-aa query vs 1M x 300, BLOSUM62, gap 3/1, SW score
    rng = np.random.default_rng(1)
    res, off = _data.random_db(rng, np.full(1_000_000, 300))
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    gpu = db.search(q, B62, 3, 1, "score", "sw")["score"]
    cpu = _cpu_baseline.CpuDatabase(res, off)
    want = cpu.search_sw(q, B62, 3, 1, 16)
    cpu.close()
    np.testing.assert_array_equal(gpu, want)
    # idempotence and slice consistency on the resident database
    again = db.search(q, B62, 3, 1, "score", "sw")["score"]
    np.testing.assert_array_equal(gpu, again)
    part = db.search(q, B62, 3, 1, "score", "sw", 123_457, 654_321)["score"]
    np.testing.assert_array_equal(part, gpu[123_457:654_321])
    # end mode agrees with score mode and the checker on a sample
    end = db.search(q, B62, 3, 1, "end", "sw")
    np.testing.assert_array_equal(end["score"], gpu)
    ref = _oracle.search(q, res[:off[2000]], off[:2001], B62, 3, 1, "end", "sw")
    for key in ("score", "end_q", "end_t"):
        np.testing.assert_array_equal(end[key][:2000], ref[key])
    db.close()


def test_cfg3_full_alignments_sample_and_invariants(capi):
    # configs[2]: SW full on the cfg2 database (a 200k slice keeps host memory modest)
    rng = np.random.default_rng(1)
    n = 200_000
    res, off = _data.random_db(rng, np.full(n, 300))
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    out = db.search(q, B62, 3, 1, "full", "sw")
    ref = _oracle.search(q, res[:off[3000]], off[:3001], B62, 3, 1, "full", "sw")
    for key in ("score", "end_q", "end_t", "start_q", "start_t"):
        np.testing.assert_array_equal(out[key][:3000], ref[key])
    assert all(a.tolist() == b.tolist() for a, b in zip(out["aln"][:3000], ref["aln"]))
    # every alignment re-scores to its reported score and spans [start, end]
    flat, aoff = out["aln_flat"], out["aln_off"]
    S = B62.reshape(24, 24)
    for k in rng.integers(0, n, size=2000):
        ops = flat[aoff[k]:aoff[k + 1]]
        i, j, score, gap = out["start_q"][k], off[k] + out["start_t"][k], 0, None
        for op in ops:
            if op in (0, 3):
                score += S[q[i], res[j]]
                assert (q[i] == res[j]) == (op == 0)
                i += 1; j += 1; gap = None
            else:
                score -= 1 if gap == op else 3
                gap = op
                if op == 1: i += 1
                else: j += 1
        assert score == out["score"][k]
        assert i - 1 == out["end_q"][k] and j - 1 - off[k] == out["end_t"][k]
    db.close()


def test_cfg4_mode_ordering_and_sample(capi):
    # configs[3]: 2000-aa query vs 2000-aa targets (20k of the 100k keep the run short),
    # plus the long tail that forces 32-bit lanes. NW <= HW <= OV <= SW holds for any
    # pair because each mode frees more of the borders than the one before.
    rng = np.random.default_rng(2)
    lengths = np.concatenate([np.full(20_000, 2000), np.arange(1000, 52000, 10000)])
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 2000)
    db = capi.DeviceDatabase(res, off, 24)
    scores = {a: db.search(q, B62, 3, 1, "score", a)["score"] for a in ("nw", "hw", "ov", "sw")}
    assert (scores["nw"] <= scores["hw"]).all() and (scores["hw"] <= scores["ov"]).all()
    assert (scores["ov"] <= scores["sw"]).all() and (scores["sw"] >= 0).all()
    assert scores["nw"].min() < -32768 < 32767  # the tail really leaves 16 bits
    sample = [0, 1, 19_999, 20_000, 20_003, 20_005]
    sub = [res[off[k]:off[k + 1]] for k in sample]
    sres, soff = _oracle.flatten(sub)
    for algo in ("nw", "hw", "ov", "sw"):
        ref = _oracle.search(q, sres, soff, B62, 3, 1, "score", algo)["score"]
        np.testing.assert_array_equal(scores[algo][sample], ref)
    db.close()


def test_cfg5_whole_database_on_one_gpu(capi):
    # configs[4] unsharded: 53-aa query vs 10M x 400 (4e9 residues: offsets beyond 2^31), SW score.
    # Every score against the AVX2 CPU baseline on the two ends and the middle of the database
    # (the part of it where 32-bit residue offsets would wrap), the checker on a sample, and
    # slice consistency across the 2^31 boundary.
    n, length = 10_000_000, 400
    rng = np.random.default_rng(3)
    res = _data.AA20_CODES[rng.integers(0, 20, size=n * length, dtype=np.uint8)]
    off = np.arange(n + 1, dtype=np.int64) * length
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        gpu = db.search(q, B62, 3, 1, "score", "sw")["score"]
        assert gpu.shape == (n,)
        for lo in (0, 5_368_000, n - 200_000):      # 5_368_709 * 400 = 2^31
            hi = lo + 200_000
            cpu = _cpu_baseline.CpuDatabase(res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo])
            want = cpu.search_sw(q, B62, 3, 1, 16)
            cpu.close()
            np.testing.assert_array_equal(gpu[lo:hi], want, err_msg=f"targets {lo}..{hi}")
        for lo in (0, 5_368_700, n - 500):
            hi = lo + 500
            ref = _oracle.search(q, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], B62, 3, 1, "end", "sw")
            np.testing.assert_array_equal(gpu[lo:hi], ref["score"])
            part = db.search(q, B62, 3, 1, "end", "sw", lo, hi)
            for key in ("score", "end_q", "end_t"):
                np.testing.assert_array_equal(part[key], ref[key], err_msg=f"{key} {lo}..{hi}")
        part = db.search(q, B62, 3, 1, "score", "sw", 5_000_000, 6_000_000)["score"]
        np.testing.assert_array_equal(part, gpu[5_000_000:6_000_000])
    finally:
        db.close()
