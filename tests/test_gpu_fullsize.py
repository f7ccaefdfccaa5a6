"""Full-size checks at BASELINE.json's configurations, at BASELINE size: every score against the
AVX2 CPU checker (all four modes), every end location and every alignment of the 1M-target
database against the scalar checker run on the host cores, frozen whole-database checksums."""
import numpy as np
import pytest

import _cpu_baseline
import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

import os

pytestmark = pytest.mark.gpu
B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
THREADS = max(1, min(16, os.cpu_count() or 1))  # host threads of the CPU checkers


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
    want = cpu.search_sw(q, B62, 3, 1, THREADS)
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


def test_cfg2_every_end_location(capi):
    # configs[1] with end locations: all 1M (score, end_q, end_t) against the scalar checker
    rng = np.random.default_rng(1)
    res, off = _data.random_db(rng, np.full(1_000_000, 300))
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        end = db.search(q, B62, 3, 1, "end", "sw")
    finally:
        db.close()
    ref = _oracle.search_parallel(q, res, off, B62, 3, 1, "end", "sw", THREADS)
    for key in ("score", "end_q", "end_t"):
        np.testing.assert_array_equal(end[key], ref[key], err_msg=key)


# sums over the whole cfg3 result, from the scalar checker (tests/golden/make_cfg3_checksum.py)
CFG3_SCORE_SUM = 57_737_236
CFG3_ALIGNMENT_BYTES = 66_834_735
CFG3_OPS_CRC32 = 0x6BCA4F59


def test_cfg3_every_alignment(capi):
    # configs[2]: SW full on the whole cfg2 database, 1M x 300. Every score, end, start and
    # alignment operation against the scalar checker (run here on the host cores), and the
    # frozen whole-database checksums.
    import zlib
    rng = np.random.default_rng(1)
    n = 1_000_000
    res, off = _data.random_db(rng, np.full(n, 300))
    q = _oracle.encode(_data.README_QUERY)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        out = db.search(q, B62, 3, 1, "full", "sw")
    finally:
        db.close()
    flat, aoff = out["aln_flat"], out["aln_off"]
    assert int(out["score"].sum()) == CFG3_SCORE_SUM
    assert int(aoff[-1]) == CFG3_ALIGNMENT_BYTES == len(flat)
    assert zlib.crc32(np.ascontiguousarray(flat).tobytes()) == CFG3_OPS_CRC32
    ref = _oracle.search_parallel(q, res, off, B62, 3, 1, "full", "sw", THREADS)
    for key in ("score", "end_q", "end_t", "start_q", "start_t"):
        np.testing.assert_array_equal(out[key], ref[key], err_msg=key)
    np.testing.assert_array_equal(aoff, ref["aln_off"])
    np.testing.assert_array_equal(flat, ref["aln_flat"])
    # every sampled alignment re-scores to its reported score and spans [start, end]
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


def test_cfg4_every_score_all_modes(capi):
    # configs[3]: 2000-aa query vs 100k x 2000 plus the 35 long targets of the reference's overflow
    # test (1000 ... 35000 residues, src/pyopal/tests/test_aligner.py:31-34), which really leave
    # 16 bits. NW, HW, OV and SW: every score against the AVX2 CPU checker (itself held to the
    # scalar checker by tests/test_cpu_baseline.py), 500 targets per mode (the whole tail among
    # them) against the scalar checker, end locations of 100, and NW <= HW <= OV <= SW.
    rng = np.random.default_rng(2)
    n_main = 100_000
    lengths = np.concatenate([np.full(n_main, 2000), np.arange(1000, 35001, 1000)])
    n = len(lengths)
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 2000)
    tail = np.arange(n_main, n)
    sample = np.concatenate([rng.choice(n_main, size=500 - len(tail), replace=False), tail])
    sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in sample])
    ends = np.concatenate([sample[:90], tail[:10]])
    eres, eoff = _oracle.flatten([res[off[k]:off[k + 1]] for k in ends])
    cpu = _cpu_baseline.CpuDatabase(res, off)
    db = capi.DeviceDatabase(res, off, 24)
    scores = {}
    try:
        for algo in ("nw", "hw", "ov", "sw"):
            scores[algo] = db.search(q, B62, 3, 1, "score", algo)["score"]
            if algo != "sw":
                # the tail really leaves 16 bits: cells of a 2000 x L matrix reach -(2000 + L) and
                # below, the targets of 30000 residues and more were computed by the int32 kernel
                assert capi.DeviceDatabase.last_routing()[0] >= 6, capi.DeviceDatabase.last_routing()
            want = cpu.search(q, B62, 3, 1, algo, THREADS)
            np.testing.assert_array_equal(scores[algo], want, err_msg=f"{algo}: every score")
            ref = _oracle.search_parallel(q, sres, soff, B62, 3, 1, "score", algo, THREADS, chunk=4)["score"]
            np.testing.assert_array_equal(scores[algo][sample], ref, err_msg=f"{algo}: scalar checker")
            end = db.search(q, B62, 3, 1, "end", algo)
            np.testing.assert_array_equal(end["score"], scores[algo], err_msg=f"{algo}: end vs score mode")
            ref = _oracle.search_parallel(q, eres, eoff, B62, 3, 1, "end", algo, THREADS, chunk=4)
            for key in ("score", "end_q", "end_t"):
                np.testing.assert_array_equal(end[key][ends], ref[key], err_msg=f"{algo} {key}")
    finally:
        db.close()
        cpu.close()
    assert (scores["nw"] <= scores["hw"]).all() and (scores["hw"] <= scores["ov"]).all()
    assert (scores["ov"] <= scores["sw"]).all() and (scores["sw"] >= 0).all()


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
            want = cpu.search_sw(q, B62, 3, 1, THREADS)
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


def test_long_pairs_beside_the_packed_kernel_repeatedly(capi, tuning):
    # The int32 kernel's (pair, strip) units run on the side stream BESIDE the packed launch, their rows
    # crossing XCDs behind progress counters. A counter once overtook its rows under exactly this load (one
    # wrong score in twenty searches of configs[3] with its tail): every mode, twelve searches each, against
    # the strip-after-strip kernel's answer - which test_cfg4_every_score_all_modes holds to the CPU checkers.
    rng = np.random.default_rng(12)
    lengths = np.concatenate([np.full(20_000, 2000), np.arange(1000, 35001, 1000)])
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 2000)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for algo in ("nw", "hw", "ov", "sw"):
            tuning.setenv("MIOPAL_NO_PAIR_STRIP_UNITS", "1")
            want = db.search(q, B62, 3, 1, "score", algo)["score"]
            tuning.delenv("MIOPAL_NO_PAIR_STRIP_UNITS")
            for run in range(12):
                got = db.search(q, B62, 3, 1, "score", algo)["score"]
                np.testing.assert_array_equal(got, want, err_msg=f"{algo}, search {run}")
    finally:
        db.close()
