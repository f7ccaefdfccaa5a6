"""`full` searches through the query-profile form of the lane-per-pair kernels (perpair.hip,
perpair_profile_kernel: start-cell scan and direction pass; src/pyopal/opal.pxd:17-19
OPAL_SEARCH_ALIGNMENT, semantics src/pyopal/lib.pyx:999-1037) against the CPU checker, and against
the kernels they replace (MIOPAL_NO_PERPAIR_PROFILE=1).

What the form adds over the other kernels and what could go wrong with it: signed bytes of
score + open in LDS (matrices / gaps that do not fit leave it), target residues read four at a time
at addresses clamped into the database (first and last targets, prefixes of fewer than four
residues), bit planes of 32 rows (query windows of 8 ... 64 rows per strip, several strips), the
known optimum ending a lane's scan.
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


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


@pytest.fixture()
def lane_per_pair(tuning):
    # (searches of a few hundred targets take the wavefront-per-pair kernels unless told otherwise)
    tuning.setenv("MIOPAL_NO_SMALL_SEARCH", "1")
    tuning.setenv("MIOPAL_NO_HYBRID_TRACE", "1")
    tuning.setenv("MIOPAL_FORCE_LANE_PER_PAIR", "1")   # (the cost estimates prefer a wavefront per pair on few pairs)


KEYS = ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat")


def packed_applies(matrix, go, ge):
    # perpair_packed.hip (two pairs per lane, round 5): gap models with open >= ext whose scores leave room in a byte
    return go >= ge >= 0


def both_forms(capi, tuning, q, res, off, matrix, go, ge, algo="sw", expect_profile=True):
    """The default routing (since round 5: two pairs per lane on 16-bit halves where the scoring scheme allows), the
    32-bit query-profile kernels (MIOPAL_NO_PACKED_*), which must give the very same arrays, and the kernels before
    both (MIOPAL_NO_PERPAIR_PROFILE): returns (default, before)."""
    db = capi.DeviceDatabase(res, off, 24)
    try:
        new = db.search(q, matrix, go, ge, "full", algo)
        packed_routing = capi.DeviceDatabase.last_full_routing()
        tuning.setenv("MIOPAL_NO_PACKED_TRACE", "1")
        tuning.setenv("MIOPAL_NO_PACKED_SCAN", "1")
        wide = db.search(q, matrix, go, ge, "full", algo)
        routing = capi.DeviceDatabase.last_full_routing()
        assert routing & (64 | 128) == 0, routing
        tuning.setenv("MIOPAL_NO_PERPAIR_PROFILE", "1")
        old = db.search(q, matrix, go, ge, "full", algo)
        assert capi.DeviceDatabase.last_full_routing() & (10 | 64 | 128) == 0
        tuning.delenv("MIOPAL_NO_PERPAIR_PROFILE")
        tuning.delenv("MIOPAL_NO_PACKED_TRACE")
        tuning.delenv("MIOPAL_NO_PACKED_SCAN")
    finally:
        db.close()
    for key in KEYS:
        np.testing.assert_array_equal(new[key], wide[key], err_msg=f"{key}: packed halves against the 32-bit profile kernels")
    if expect_profile and off[-1] >= 4:
        # directions always; start cells of every mode that has a scan (NW starts at the origin); queries of one
        # strip by the persistent wavefronts that refill their lanes
        assert routing & 12 == 12, routing
        assert (routing & 3 == 3) == (algo != "nw"), routing
        assert bool(routing & 32) == (algo != "nw" and len(q) <= 64), routing
        if packed_applies(matrix, go, ge):
            # the direction pass of every mode, the scan of Smith-Waterman, HW and OV prefixes
            # (the scan when eight times its values fit the half floats as well: not gap 5/5 against 400 columns)
            assert packed_routing & 64, packed_routing
            assert algo != "nw" or not packed_routing & 128, packed_routing   # (NW has no scan)
            assert packed_routing & 128 or algo != "sw" or ge > 1, packed_routing
        else:
            assert packed_routing & (64 | 128) == 0, packed_routing
    return new, old


@pytest.mark.parametrize("algo", ["sw", "hw", "ov"])
@pytest.mark.parametrize("qlen", [1, 8, 33, 53, 64])
def test_one_strip_scan_without_refill(capi, lane_per_pair, tuning, qlen, algo):
    # the start-cell scan of a one-strip query on the kernel that longer queries use (a wavefront lasts as long as
    # its longest lane), against the persistent one and the checker
    rng = np.random.default_rng(500 + qlen)
    res, off = _data.random_db(rng, rng.integers(1, 400, size=5000))
    q = _data.random_protein(rng, qlen)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        tuning.setenv("MIOPAL_NO_PACKED_SCAN", "1")   # (the 32-bit kernels: Smith-Waterman scans take two pairs per lane otherwise)
        refill = db.search(q, B62, 3, 1, "full", algo)
        assert capi.DeviceDatabase.last_full_routing() & (32 | 128) == 32
        tuning.setenv("MIOPAL_NO_SCAN_REFILL", "1")
        plain = db.search(q, B62, 3, 1, "full", algo)
        assert capi.DeviceDatabase.last_full_routing() & (35 | 128) == 3
    finally:
        db.close()
    ref = _oracle.search(q, res, off, B62, 3, 1, "full", algo)
    compare(refill, ref, "full", f"refill {algo} Q={qlen}")
    compare(plain, ref, "full", f"no refill {algo} Q={qlen}")


@pytest.mark.parametrize("qlen", [1, 5, 8, 9, 31, 32, 33, 53, 63, 64, 65, 96, 97, 128, 150, 300])
@pytest.mark.parametrize("gaps", [(3, 1), (11, 1), (1, 1), (5, 5)])
def test_against_the_checker(capi, lane_per_pair, tuning, qlen, gaps):
    rng = np.random.default_rng(1000 * qlen + gaps[0])
    lengths = rng.integers(1, 400, size=600)
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, qlen)
    new, old = both_forms(capi, tuning, q, res, off, B62, *gaps)
    ref = _oracle.search(q, res, off, B62, gaps[0], gaps[1], "full", "sw")
    compare(new, ref, "full", f"profile form Q={qlen} gaps {gaps}")
    compare(old, ref, "full", f"form before Q={qlen} gaps {gaps}")


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
def test_other_modes_take_the_profile_form_too(capi, lane_per_pair, tuning, algo):
    # HW / OV: start cells in the regions "last row" / "last row or column" (the lane's last query row picked out of
    # the 64 registers every column; the whole last column for OV); NW: no scan; the direction pass is shared
    rng = np.random.default_rng(77)
    res, off = _data.random_db(rng, rng.integers(1, 300, size=500))
    for qlen in (1, 20, 63, 64, 65, 130, 200):
        q = _data.random_protein(rng, qlen)
        new, old = both_forms(capi, tuning, q, res, off, B62, 3, 1, algo)
        ref = _oracle.search(q, res, off, B62, 3, 1, "full", algo)
        compare(new, ref, "full", f"{algo} Q={qlen}")
        compare(old, ref, "full", f"{algo} Q={qlen} (form before)")


def test_related_sequences(capi, lane_per_pair, tuning):
    # long alignments with ties: mutated copies of the query, cheap and dear gaps, two matrices
    rng = np.random.default_rng(5)
    for qlen in (60, 200):
        q = _data.random_protein(rng, qlen)
        seqs = []
        for _ in range(300):
            t = q.copy()
            for _ in range(rng.integers(0, 12)):
                k = rng.integers(0, len(t))
                op = rng.integers(0, 3)
                if op == 0:
                    t[k] = rng.integers(0, 20)
                elif op == 1 and len(t) > 2:
                    t = np.delete(t, k)
                else:
                    t = np.insert(t, k, rng.integers(0, 20))
            flank = _data.random_protein(rng, int(rng.integers(0, 30)))
            seqs.append(np.concatenate([flank, t, flank[::-1]]).astype(np.uint8))
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        res = np.concatenate(seqs)
        for matrix in (B62, B50):
            for go, ge in ((3, 1), (11, 1), (2, 2)):
                new, old = both_forms(capi, tuning, q, res, off, matrix, go, ge)
                ref = _oracle.search(q, res, off, matrix, go, ge, "full", "sw")
                compare(new, ref, "full", f"related Q={qlen} gaps {go}/{ge}")
                compare(old, ref, "full", f"related Q={qlen} gaps {go}/{ge} (form before)")


def test_short_targets_at_both_ends_of_the_database(capi, lane_per_pair, tuning):
    # the four-residue loads are clamped into the database: the first targets (reversed prefixes reach below
    # their first residue) and the last ones (forward windows reach beyond the last residue) are the ones
    # where the clamp moves the load
    rng = np.random.default_rng(9)
    q = _data.random_protein(rng, 40)
    for first, last in ((1, 1), (2, 3), (3, 2), (1, 5), (4, 4), (5, 1)):
        lengths = np.concatenate([[first, 1, 2, 3], rng.integers(1, 60, size=200), [3, 2, 1, last]])
        res, off = _data.random_db(rng, lengths)
        # (a residue that matches, so that the tiny targets do have alignments)
        res[:first] = q[:first]
        res[off[-2]:] = q[-last:]
        new, old = both_forms(capi, tuning, q, res, off, B62, 3, 1)
        ref = _oracle.search(q, res, off, B62, 3, 1, "full", "sw")
        compare(new, ref, "full", f"ends {first}/{last}")
        compare(old, ref, "full", f"ends {first}/{last} (form before)")


def test_a_database_of_fewer_than_four_residues(capi, lane_per_pair, tuning):
    q = _data.random_protein(np.random.default_rng(3), 30)
    for lengths in ([1], [2], [1, 2], [3]):
        res = q[:sum(lengths)].copy()
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        new, old = both_forms(capi, tuning, q, res, off, B62, 3, 1)
        ref = _oracle.search(q, res, off, B62, 3, 1, "full", "sw")
        compare(new, ref, "full", f"tiny {lengths}")


def test_scores_beyond_the_byte_leave_the_form(capi, lane_per_pair, tuning):
    # score + open does not fit a signed byte: the search takes the kernels of before (same answers)
    rng = np.random.default_rng(21)
    res, off = _data.random_db(rng, rng.integers(1, 200, size=300))
    q = _data.random_protein(rng, 50)
    big = B62 * 9          # 99 on the diagonal
    for matrix, go, ge in ((B62, 120, 1), (big, 40, 3), (B62, 127, 127)):
        db = capi.DeviceDatabase(res, off, 24)
        try:
            got = db.search(q, matrix, go, ge, "full", "sw")
            assert capi.DeviceDatabase.last_full_routing() & 15 == 5   # a lane per pair, not the profile form
        finally:
            db.close()
        ref = _oracle.search(q, res, off, matrix, go, ge, "full", "sw")
        compare(got, ref, "full", f"gaps {go}/{ge}")


@pytest.mark.parametrize("algo", ["hw", "ov"])
def test_many_pairs_of_the_other_modes(capi, tuning, algo):
    # the regions "last row" / "last row or column" at a size where the host picks one lane per pair by itself:
    # every alignment against the kernels of before, a sample against the checker
    rng = np.random.default_rng(41)
    lengths = np.clip(rng.lognormal(mean=5.0, sigma=0.5, size=120_000), 10, 1500).astype(np.int64)
    res, off = _data.random_db(rng, lengths)
    for qlen in (53, 150):
        q = _data.random_protein(rng, qlen)
        new, old = both_forms(capi, tuning, q, res, off, B62, 3, 1, algo)
        for key in ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat"):
            np.testing.assert_array_equal(new[key], old[key], err_msg=f"{key} {algo} Q={qlen}")
        pick = np.sort(rng.choice(len(lengths), size=200, replace=False))
        sub_res = np.concatenate([res[off[k]:off[k + 1]] for k in pick])
        sub_off = np.concatenate([[0], np.cumsum(lengths[pick])]).astype(np.int64)
        ref = _oracle.search(q, sub_res, sub_off, B62, 3, 1, "full", algo)
        for x, k in enumerate(pick):
            assert new["score"][k] == ref["score"][x]
            assert new["aln"][k].tolist() == ref["aln"][x].tolist(), f"alignment of target {k} {algo} Q={qlen}"


def test_many_pairs_in_batches(capi, tuning):
    # enough pairs for several direction batches and for the copy of one batch's operations beside the
    # next batch; every alignment against the form before
    rng = np.random.default_rng(31)
    lengths = np.clip(rng.lognormal(mean=5.0, sigma=0.5, size=300_000), 10, 2000).astype(np.int64)
    res, off = _data.random_db(rng, lengths)
    for qlen in (53, 150):
        q = _data.random_protein(rng, qlen)
        new, old = both_forms(capi, tuning, q, res, off, B62, 3, 1)   # (asserts the profile form ran)
        for key in ("score", "end_q", "end_t", "start_q", "start_t", "aln_off", "aln_flat"):
            np.testing.assert_array_equal(new[key], old[key], err_msg=f"{key} Q={qlen}")
        # a sample against the checker
        pick = rng.choice(len(lengths), size=300, replace=False)
        pick.sort()
        sub_res = np.concatenate([res[off[k]:off[k + 1]] for k in pick])
        sub_off = np.concatenate([[0], np.cumsum(lengths[pick])]).astype(np.int64)
        ref = _oracle.search(q, sub_res, sub_off, B62, 3, 1, "full", "sw")
        for x, k in enumerate(pick):
            assert new["score"][k] == ref["score"][x]
            assert new["aln"][k].tolist() == ref["aln"][x].tolist(), f"alignment of target {k} Q={qlen}"
