"""The biased-integer lanes of the pair-table kernel (interseq_impl.h: scores, and scores with
end locations) against the CPU checker: every strip height, the edges of the exact range (lanes
that must be flagged and redone), rebasing of the column shift, gap models on both sides of
open == ext, ragged groups. Bit-exact, through the C ABI."""
import os

import numpy as np
import pytest

import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

pytestmark = pytest.mark.gpu

B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
PAIR_BIASED = 4  # miopalLastRouting counts[1]: 2 + kPairSwBiased


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


def check(capi, query, res, off, matrix, go, ge, modes=("score", "end"), expect_kernel=PAIR_BIASED, tag=""):
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for mode in modes:
            got = db.search(query, matrix, go, ge, mode, "sw")
            kernel = capi.DeviceDatabase.last_routing()[1] & 31
            want = _oracle.search(query, res, off, matrix, go, ge, mode, "sw")
            for key in want:
                if key == "aln":
                    for k, (a, b) in enumerate(zip(got[key], want[key])):
                        assert a.tolist() == b.tolist(), f"{tag} {mode} alignment {k}"
                else:
                    np.testing.assert_array_equal(got[key], want[key], err_msg=f"{tag} {mode} {key}")
            if expect_kernel is not None:
                assert kernel == expect_kernel, f"{tag} {mode}: lane-per-target pass ran kernel {kernel}"
    finally:
        db.close()


@pytest.mark.parametrize("qlen", list(range(1, 65)))
def test_every_strip_height(capi, qlen):
    rng = np.random.default_rng(1000 + qlen)
    query = _data.random_protein(rng, qlen)
    lengths = rng.integers(1, 120, size=300)
    res, off = _data.random_db(rng, lengths)
    # a few related targets: scores well above the random background, ends away from the borders
    pieces, lens = [], []
    for k in range(20):
        t = np.concatenate([_data.random_protein(rng, int(rng.integers(0, 30))), _data.mutate(rng, query, 0.1),
                            _data.random_protein(rng, int(rng.integers(0, 30)))])
        pieces.append(t)
        lens.append(len(t))
    res = np.concatenate([res] + pieces)
    off = np.concatenate([off, off[-1] + np.cumsum(lens)])
    # the pair table of the 25-symbol alphabet fits the CU's LDS up to 60 rows (15 slots of 16 bytes
    # per row); taller strips take the general kernel
    rows = max(2, (qlen + 1) // 2 * 2)
    fits = 25 * 25 * (((rows + 3) // 4) | 1) * 16 <= 158 * 1024
    check(capi, query, res, off, B62, 3, 1, expect_kernel=PAIR_BIASED if fits else 1, tag=f"Q={qlen}")


@pytest.mark.parametrize("go,ge", [(3, 1), (11, 1), (1, 1), (2, 5), (0, 3), (5, 0), (14, 12), (40, 12)])
def test_gap_models(capi, go, ge):
    rng = np.random.default_rng(go * 100 + ge)
    query = _data.random_protein(rng, 53)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 400, size=400)]
    seqs += [_data.mutate(rng, query, 0.2) for _ in range(40)]
    res, off = _oracle.flatten(seqs)
    # (a gap model outside the guard band of the flavour must fall back, not fail: no kernel check)
    check(capi, query, res, off, B62, go, ge, expect_kernel=None, tag=f"gap {go}/{ge}")


def test_column_shift_is_rebased(capi):
    # ext = 12 with end locations (6 row bits): 768 pattern units per column, a rebase every
    # chunk (and steps up of 23 << 6: the lowered limit); scores-only at ext = 400: a rebase every
    # other chunk. Targets long enough for dozens.
    rng = np.random.default_rng(77)
    query = _data.random_protein(rng, 60)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(200, 1500, size=260)]
    seqs += [np.concatenate([_data.random_protein(rng, 700), _data.mutate(rng, query, 0.05)]) for _ in range(10)]
    res, off = _oracle.flatten(seqs)
    check(capi, query, res, off, B62, 14, 12, tag="ext 12")
    check(capi, query, res, off, B62, 500, 400, modes=("score",), tag="ext 400")


def test_long_targets_many_rebases_at_unit_extension(capi):
    rng = np.random.default_rng(78)
    query = _data.encode(_data.README_QUERY)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(4000, 8000, size=130)]
    seqs[5] = np.concatenate([seqs[5][:6000], _data.mutate(rng, query, 0.05), seqs[5][6000:6100]])
    res, off = _oracle.flatten(seqs)
    # (Smith-Waterman searches of long targets use segmented views: windows, still the same kernel)
    check(capi, query, res, off, B62, 3, 1, tag="long targets")


def scaled_identity(A, match, mismatch):
    m = np.full((A, A), mismatch, dtype=np.int32)
    np.fill_diagonal(m, match)
    return m.ravel()


@pytest.mark.parametrize("match,qlen", [(15, 60), (15, 30), (15, 16), (11, 53)])
def test_end_location_range_is_left_and_lanes_are_redone(capi, match, qlen):
    # exact copies of the query score match * qlen: 900 at 60 rows (limit 384 at 6 row bits),
    # 450 at 30 rows (limit 768), 240 at 16 rows (limit 1536), 583 at 53 rows
    rng = np.random.default_rng(match * 100 + qlen)
    m = scaled_identity(24, match, -4)
    query = _data.random_protein(rng, qlen)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 200, size=300)]
    for frac in (1.0, 0.9, 0.7, 0.5, 0.41, 0.4, 0.39, 0.3):   # scores on both sides of the limits
        k = max(1, int(qlen * frac))
        seqs.append(np.concatenate([_data.random_protein(rng, 17), query[:k], _data.random_protein(rng, 9)]))
        seqs.append(np.concatenate([query[qlen - k:], _data.random_protein(rng, 23)]))
    res, off = _oracle.flatten(seqs)
    check(capi, query, res, off, m, 5, 2, tag=f"match {match} Q={qlen}")


def test_score_range_is_left_and_lanes_are_redone(capi):
    # match 500: copies of k query residues score 500 k, the flavour is exact below 25600 (k = 51)
    rng = np.random.default_rng(5)
    m = scaled_identity(24, 500, -300)
    query = _data.random_protein(rng, 60)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 200, size=300)]
    for k in (60, 56, 53, 52, 51, 50, 49, 40, 10):
        seqs.append(np.concatenate([_data.random_protein(rng, 11), query[:k], _data.random_protein(rng, 5)]))
    res, off = _oracle.flatten(seqs)
    check(capi, query, res, off, m, 700, 100, modes=("score",), tag="match 500")
    db = capi.DeviceDatabase(res, off, 24)
    try:
        db.search(query, m, 700, 100, "score", "sw")
        assert capi.DeviceDatabase.last_routing()[3] >= 4   # the copies of 52+ residues were redone
    finally:
        db.close()


def test_many_lanes_leave_the_range(capi):
    # more flagged lanes than the direct recompute takes: the whole view runs the next rung
    rng = np.random.default_rng(6)
    m = scaled_identity(24, 15, -4)
    query = _data.random_protein(rng, 60)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 100, size=200)]
    seqs += [np.concatenate([_data.random_protein(rng, int(rng.integers(0, 9))), query]) for _ in range(2300)]
    res, off = _oracle.flatten(seqs)
    check(capi, query, res, off, m, 5, 2, modes=("end",), expect_kernel=None, tag="many flagged")


def test_large_steps_lower_the_limit(capi):
    # match + ext = 3000 > 0x0400: a finite half could jump over the NaN patterns; the limit is
    # lowered by the excess so that the cell it would jump from is flagged
    rng = np.random.default_rng(18)
    m = scaled_identity(24, 2900, -900)
    query = _data.random_protein(rng, 40)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 100, size=200)]
    seqs += [np.concatenate([query[:k], _data.random_protein(rng, 3)]) for k in (40, 12, 11, 10, 9, 8, 7, 6, 5)]
    res, off = _oracle.flatten(seqs)
    check(capi, query, res, off, m, 1000, 100, modes=("score",), tag="match 2900")


def test_steps_up_beyond_the_guard_band_take_another_flavour(capi):
    # match + ext > 0x1000: the flavour must not run
    rng = np.random.default_rng(8)
    m = scaled_identity(24, 4100, -900)
    query = _data.random_protein(rng, 40)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 100, size=200)]
    seqs += [np.concatenate([query[:k], _data.random_protein(rng, 3)]) for k in (40, 33, 32, 31, 5)]
    res, off = _oracle.flatten(seqs)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for mode in ("score", "end"):
            got = db.search(query, m, 1200, 100, mode, "sw")
            assert (capi.DeviceDatabase.last_routing()[1] & 15) != PAIR_BIASED
            want = _oracle.search(query, res, off, m, 1200, 100, mode, "sw")
            for key in want:
                np.testing.assert_array_equal(got[key], want[key], err_msg=f"{mode} {key}")
    finally:
        db.close()


def test_switch_restores_the_half_float_rung(capi, tuning):
    rng = np.random.default_rng(9)
    query = _data.encode(_data.README_QUERY)
    res, off = _data.random_db(rng, rng.integers(20, 300, size=500))
    tuning.setenv("MIOPAL_NO_BIASED", "1")
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(query, B62, 3, 1, "score", "sw")
        assert capi.DeviceDatabase.last_routing()[1] == 3   # pair table, half floats
        want = _oracle.search(query, res, off, B62, 3, 1, "score", "sw")
        np.testing.assert_array_equal(got["score"], want["score"])
    finally:
        db.close()


PAIR_GLOBAL = 5  # miopalLastRouting counts[1]: 2 + kPairGlobalBiased


def check_algo(capi, algo, query, res, off, matrix, go, ge, modes=("score", "end", "full"), expect_kernel=PAIR_GLOBAL, tag=""):
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for mode in modes:
            got = db.search(query, matrix, go, ge, mode, algo)
            kernel = capi.DeviceDatabase.last_routing()[1] & 31
            want = _oracle.search(query, res, off, matrix, go, ge, mode, algo)
            for key in want:
                if key == "aln":
                    for k, (a, b) in enumerate(zip(got[key], want[key])):
                        assert a.tolist() == b.tolist(), f"{tag} {algo} {mode} alignment {k}"
                else:
                    np.testing.assert_array_equal(got[key], want[key], err_msg=f"{tag} {algo} {mode} {key}")
            if expect_kernel is not None:
                assert kernel == expect_kernel, f"{tag} {algo} {mode}: lane-per-target pass ran kernel {kernel}"
    finally:
        db.close()


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
@pytest.mark.parametrize("qlen", [1, 2, 3, 7, 8, 16, 17, 31, 32, 33, 52, 53, 54, 59, 60])
def test_global_modes_every_kind_of_strip(capi, algo, qlen):
    # one-strip NW / HW / OV on the pair-table kernel: odd and even query lengths (the last query row
    # is row R - 1 or R - 2), ragged groups, empty and one-residue targets, related targets
    rng = np.random.default_rng(2000 + qlen)
    query = _data.random_protein(rng, qlen)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(0, 150, size=300)]
    seqs += [np.concatenate([_data.random_protein(rng, int(rng.integers(0, 30))), _data.mutate(rng, query, 0.1),
                             _data.random_protein(rng, int(rng.integers(0, 30)))]) for _ in range(20)]
    seqs += [_data.random_protein(rng, 1), np.zeros(0, np.uint8), query.copy()]
    res, off = _oracle.flatten(seqs)
    check_algo(capi, algo, query, res, off, B62, 3, 1, tag=f"Q={qlen}")


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
@pytest.mark.parametrize("go,ge", [(11, 1), (1, 1), (5, 0), (0, 0), (14, 12), (2, 5), (40, 12)])
def test_global_modes_gap_models(capi, algo, go, ge):
    rng = np.random.default_rng(go * 100 + ge + 7)
    query = _data.random_protein(rng, 47)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 300, size=300)]
    seqs += [_data.mutate(rng, query, 0.2) for _ in range(30)]
    res, off = _oracle.flatten(seqs)
    # (NW with open < ext, and gap costs beyond the static bounds, take the general kernel: no kernel check)
    check_algo(capi, algo, query, res, off, B62, go, ge, modes=("score", "end"), expect_kernel=None, tag=f"gap {go}/{ge}")


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
def test_global_modes_no_target_leaves_the_lanes_for_its_length(capi, algo):
    # gap 70/60: (Q + L) ext passes 32000 at L = 480, the general kernel's int16 lanes would hand the
    # longer targets to the int32 kernel; here a pattern carries zero + j ext and the true value is
    # read as a 32-bit number, so everything stays in the lanes. HW / OV rebase their shift every
    # 68 columns at this ext.
    rng = np.random.default_rng(91)
    query = _data.random_protein(rng, 53)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(380, 512, size=300)]
    seqs[3] = np.concatenate([seqs[3][:200], _data.mutate(rng, query, 0.05), seqs[3][200:330]])
    res, off = _oracle.flatten(seqs)
    check_algo(capi, algo, query, res, off, B62, 70, 60, modes=("score", "end"), tag="gap 70/60")
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(query, B62, 70, 60, "score", algo)["score"]
        assert capi.DeviceDatabase.last_routing()[0] == 0   # nothing on the int32 kernel
        if algo == "nw":
            assert got.min() < -20000
    finally:
        db.close()


@pytest.mark.parametrize("switch", [None, "MIOPAL_NO_BIASED", "MIOPAL_NO_PAIR_TABLE"])
def test_scores_beyond_the_half_float_range(capi, tuning, switch):
    # match 1024: a copy of the 64-residue query scores 65536, beyond the largest finite half float
    # (65504), where a half-float lane would turn inf and, next to -inf padding, NaN - which converts to
    # 0 and would slip through the "best >= limit" flag. Every rung must hand such lanes on; the answer
    # comes from the int32 kernel whatever the first rung was.
    if switch:
        tuning.setenv(switch, "1")
    rng = np.random.default_rng(64)
    m = scaled_identity(24, 1024, -1024)
    query = _data.random_protein(rng, 64)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 200, size=300)]
    for k in (64, 63, 48, 33, 32, 31, 26, 25, 24, 2):   # 65536 .. 2048: every rung's limit is crossed
        seqs.append(np.concatenate([_data.random_protein(rng, 7), query[:k], _data.random_protein(rng, 4)]))
    seqs.append(np.concatenate([query, query]))
    res, off = _oracle.flatten(seqs)
    for go, ge in ((2000, 1000), (30, 10)):
        check(capi, query, res, off, m, go, ge, expect_kernel=None, tag=f"match 1024 gap {go}/{ge} {switch}")


# --- Smith-Waterman scores of several strips: column-shifted unsigned lanes of the general kernel ---
GENERAL_SHIFTED = 1 + 32 * 6   # miopalLastRouting counts[1]: general kernel, kSwShifted
GENERAL_HALF = 1 + 32 * 0
GENERAL_INT16 = 1 + 32 * 1


PAIR_STRIPS = 2 + 4            # pair table, one strip after the other (interseq_pair_strips_kernel)


@pytest.fixture(params=["strips", "general"])
def multi_strip(request, tuning):
    """Both first rungs of a multi-strip Smith-Waterman score search: the pair-table kernel strip by
    strip (default when the scores fit its guard band) and the general kernel's column-shifted lanes."""
    if request.param == "general":
        tuning.setenv("MIOPAL_NO_PAIR_STRIPS", "1")
        return GENERAL_SHIFTED
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")   # (also for searches of few units, where the host prefers the general kernel)
    return PAIR_STRIPS


def shifted_check(capi, query, res, off, matrix, go, ge, expect=GENERAL_SHIFTED, tag=""):
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(query, matrix, go, ge, "score", "sw")["score"]
        routed = capi.DeviceDatabase.last_routing()
    finally:
        db.close()
    want = _oracle.search(query, res, off, matrix, go, ge, "score", "sw")["score"]
    np.testing.assert_array_equal(got, want, err_msg=tag)
    if expect is not None:
        assert routed[1] == expect, f"{tag}: lane-per-target pass ran {routed[1]}"
    return routed


@pytest.mark.parametrize("qlen", [61, 64, 65, 100, 127, 128, 129, 192, 193, 333, 700])
def test_shifted_lanes_every_kind_of_strip_count(capi, qlen, multi_strip):
    rng = np.random.default_rng(4000 + qlen)
    query = _data.random_protein(rng, qlen)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 500, size=300)]
    seqs += [np.concatenate([_data.random_protein(rng, int(rng.integers(0, 40))), _data.mutate(rng, query, 0.15),
                             _data.random_protein(rng, int(rng.integers(0, 40)))]) for _ in range(20)]
    seqs += [query[: qlen // 2], query[qlen // 3:], np.zeros(0, dtype=np.uint8)]
    res, off = _oracle.flatten(seqs)
    shifted_check(capi, query, res, off, B62, 11, 1, expect=multi_strip, tag=f"Q={qlen}")


@pytest.mark.parametrize("go,ge", [(3, 1), (1, 1), (2, 5), (0, 3), (5, 0), (0, 0), (14, 12), (40, 12), (700, 30)])
def test_shifted_lanes_gap_models(capi, go, ge, multi_strip):
    rng = np.random.default_rng(go * 100 + ge + 7)
    query = _data.random_protein(rng, 150)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 400, size=300)]
    seqs += [_data.mutate(rng, query, 0.2) for _ in range(30)]
    res, off = _oracle.flatten(seqs)
    # (whatever rung the gap model gets, the scores are the checker's)
    shifted_check(capi, query, res, off, B62, go, ge, expect=None, tag=f"gap {go}/{ge}")


def test_shifted_lanes_leave_their_range_and_are_redone(capi, multi_strip):
    # match 100, ext 2, targets of at most ~420 residues: exact below 0x7C00 - 0x1000 - 2 (420 + 8) or so;
    # copies of k query residues score 100 k on both sides of that (k = 267)
    rng = np.random.default_rng(31)
    m = scaled_identity(24, 100, -40)
    query = _data.random_protein(rng, 300)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 400, size=300)]
    ks = (300, 280, 270, 268, 267, 266, 265, 264, 263, 262, 260, 250, 100, 3)
    for k in ks:
        seqs.append(np.concatenate([_data.random_protein(rng, 30), query[:k], _data.random_protein(rng, 20)]))
        seqs.append(np.concatenate([query[300 - k:], _data.random_protein(rng, 50)]))
    res, off = _oracle.flatten(seqs)
    routed = shifted_check(capi, query, res, off, m, 5, 2, expect=multi_strip, tag="match 100")
    # (the strips kernel rebases its column shift: exact below 25600 whatever the targets' lengths)
    limit = 25600 if multi_strip == PAIR_STRIPS else 0x7C00 - 0x1000 - 2 * (max(len(s) for s in seqs) + 64)
    assert 2 * sum(1 for k in ks if 100 * k >= limit) <= routed[3] <= 2 * sum(1 for k in ks if 100 * k >= limit - 2000)


def test_shifted_lanes_large_steps_lower_the_limit(capi, multi_strip):
    # match + ext = 3000: a finite pattern could step over 0x7C00..0x7FFF; the limit gives the excess away
    rng = np.random.default_rng(32)
    m = scaled_identity(24, 2900, -900)
    query = _data.random_protein(rng, 130)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 200, size=300)]
    seqs += [np.concatenate([_data.random_protein(rng, 7), query[:k], _data.random_protein(rng, 3)])
             for k in (130, 40, 12, 11, 10, 9, 8, 7, 6, 5, 4)]
    res, off = _oracle.flatten(seqs)
    shifted_check(capi, query, res, off, m, 1000, 100, expect=multi_strip, tag="match 2900")


def test_shifted_lanes_negative_scores_are_biased(capi, multi_strip):
    # mismatch -900 at ext 1: profile entries are s + ext + 899 >= 0, the column term takes the 899 back
    rng = np.random.default_rng(33)
    m = scaled_identity(24, 60, -900)
    query = _data.random_protein(rng, 200)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 300, size=300)]
    seqs += [_data.mutate(rng, query, f) for f in (0.0, 0.01, 0.02, 0.05, 0.1)]
    res, off = _oracle.flatten(seqs)
    shifted_check(capi, query, res, off, m, 20, 1, expect=multi_strip, tag="mismatch -900")
    # beyond the room below zero: another rung, same scores
    m = scaled_identity(24, 60, -3000)
    routed = shifted_check(capi, query, res, off, m, 20, 1, expect=None, tag="mismatch -3000")
    assert routed[1] in (GENERAL_HALF, GENERAL_INT16)


def test_shifted_lanes_are_not_used_when_the_columns_eat_the_range(capi, multi_strip):
    # ext 200 x a few hundred columns leaves nothing above zero for the general kernel's shifted lanes
    # (the half-float / int16 rungs run); the strips kernel rebases its shift and takes it
    rng = np.random.default_rng(34)
    query = _data.random_protein(rng, 100)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(300, 400, size=330)]
    seqs[3] = np.concatenate([seqs[3][:200], _data.mutate(rng, query, 0.05), seqs[3][200:250]])
    res, off = _oracle.flatten(seqs)
    routed = shifted_check(capi, query, res, off, B62, 210, 200, expect=None, tag="ext 200")
    assert routed[1] == PAIR_STRIPS if multi_strip == PAIR_STRIPS else routed[1] in (GENERAL_HALF, GENERAL_INT16)
    # at ext 1 the same targets fit, and so do much longer ones (Smith-Waterman searches see long
    # targets through windows of a few query lengths)
    shifted_check(capi, query, res, off, B62, 11, 1, expect=multi_strip, tag="ext 1")
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1500, 2000, size=130)]
    seqs[3] = np.concatenate([seqs[3][:900], _data.mutate(rng, query, 0.05), seqs[3][900:1000]])
    res, off = _oracle.flatten(seqs)
    shifted_check(capi, query, res, off, B62, 14, 12, expect=multi_strip, tag="ext 12, long targets")


def test_switch_restores_the_half_float_rung_of_the_general_kernel(capi, tuning):
    rng = np.random.default_rng(35)
    query = _data.random_protein(rng, 150)
    res, off = _data.random_db(rng, rng.integers(20, 300, size=500))
    tuning.setenv("MIOPAL_NO_SW_SHIFT", "1")
    tuning.setenv("MIOPAL_NO_PAIR_STRIPS", "1")
    shifted_check(capi, query, res, off, B62, 3, 1, expect=GENERAL_HALF, tag="switch")


# --- the strips kernel's own corners ---
def test_strips_many_rebases_and_long_ragged_groups(capi, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    # ext 400: the column shift is rebased every other chunk, in every strip at the same chunks (the rows
    # handed from strip to strip must mean the same on both sides); log-normal lengths: groups of very
    # different lengths in one batch, wavefronts that wait at the unit's barrier
    rng = np.random.default_rng(61)
    query = _data.random_protein(rng, 170)
    lengths = np.clip(rng.lognormal(5.5, 0.8, size=3000).astype(int), 1, 4000)
    seqs = [_data.random_protein(rng, int(n)) for n in lengths]
    seqs += [np.concatenate([_data.random_protein(rng, 600), _data.mutate(rng, query, 0.1), _data.random_protein(rng, 40)])
             for _ in range(10)]
    res, off = _oracle.flatten(seqs)
    shifted_check(capi, query, res, off, B62, 500, 400, expect=PAIR_STRIPS, tag="ext 400")
    shifted_check(capi, query, res, off, B62, 11, 1, expect=PAIR_STRIPS, tag="ext 1")


def test_strips_overflow_crosses_strip_boundaries(capi, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    # match 500 on a 200-residue query: copies of 52 residues and more leave the exact range (25600), the
    # pattern that turns inf / NaN in one strip travels down the boundary rows; every such lane is redone
    rng = np.random.default_rng(62)
    m = scaled_identity(24, 500, -300)
    query = _data.random_protein(rng, 200)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 300, size=400)]
    for k in (200, 120, 60, 53, 52, 51, 50, 49, 30):
        for at in (0, 37, 100):   # the copy starts in different strips
            seqs.append(np.concatenate([_data.random_protein(rng, 11), query[at:at + k], _data.random_protein(rng, 5)]))
    res, off = _oracle.flatten(seqs)
    routed = shifted_check(capi, query, res, off, m, 700, 100, expect=PAIR_STRIPS, tag="match 500")
    assert routed[3] >= 9


@pytest.mark.parametrize("A,qlen", [(4, 130), (12, 333), (32, 64), (32, 100), (32, 50)])
def test_strips_other_alphabets(capi, A, qlen, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    # the pair table of a 33-symbol alphabet holds 36 rows: 64 rows are two strips of 32, 100 rows three
    # of 34; 50 rows would be two strips of 26, below the kernel's 32, and stay on the general kernel
    rng = np.random.default_rng(A * 1000 + qlen)
    matrix = rng.integers(-6, 8, size=(A, A)).astype(np.int32)
    matrix[np.arange(A), np.arange(A)] = rng.integers(3, 12, size=A)
    seqs = [rng.integers(0, A, size=int(n)).astype(np.uint8) for n in rng.integers(1, 400, size=500)]
    q = rng.integers(0, A, size=qlen).astype(np.uint8)
    seqs += [np.concatenate([seqs[k][:50], q, seqs[k][:30]]) for k in range(5)]
    res, off = _oracle.flatten(seqs)
    db = capi.DeviceDatabase(res, off, A)
    try:
        got = db.search(q, matrix.ravel(), 5, 2, "score", "sw")["score"]
        routed = capi.DeviceDatabase.last_routing()
    finally:
        db.close()
    want = _oracle.search(q, res, off, matrix.ravel(), 5, 2, "score", "sw")["score"]
    np.testing.assert_array_equal(got, want)
    assert (routed[1] == PAIR_STRIPS) == ((A, qlen) != (32, 50))


def test_strips_few_groups_long_query(capi, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    # 40 strips over 3 batches: the units of a batch run side by side in different workgroups, each
    # wavefront two chunks behind the one above it
    rng = np.random.default_rng(63)
    query = _data.random_protein(rng, 2000)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1500, 2500, size=4000)]
    seqs[7] = np.concatenate([seqs[7][:300], _data.mutate(rng, query, 0.3), seqs[7][300:400]])
    res, off = _oracle.flatten(seqs)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(query, B62, 11, 1, "score", "sw")["score"]
        assert capi.DeviceDatabase.last_routing()[1] == PAIR_STRIPS
    finally:
        db.close()
    import _cpu_baseline
    cpu = _cpu_baseline.CpuDatabase(res, off)
    want = cpu.search_sw(query, B62, 11, 1, 8)
    cpu.close()
    np.testing.assert_array_equal(got, want)


def test_strips_routing_by_size(capi):
    # few (group, strip) units, or a longest group that is long against the whole launch: the general
    # kernel; many units of groups that are short against the launch: the strips kernel
    rng = np.random.default_rng(64)
    res, off = _data.random_db(rng, np.full(100_000, 100))
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for qlen, lo, hi, want in ((100, 0, 100_000, 1), (100, 0, 3000, 1), (600, 0, 100_000, PAIR_STRIPS), (300, 0, 3000, 1),
                                   (900, 0, 3000, PAIR_STRIPS)):   # (16 strips or more: the strips kernel whatever the size)
            q = _data.random_protein(rng, qlen)
            got = db.search(q, B62, 11, 1, "score", "sw", lo, hi)["score"]
            assert (capi.DeviceDatabase.last_routing()[1] & 31) == want, (qlen, hi)
            ref = _oracle.search(q, res[:off[200]], off[:201], B62, 11, 1, "score", "sw")["score"]
            np.testing.assert_array_equal(got[:200], ref)
    finally:
        db.close()


# --- the strips kernel with end locations (values scaled by 2^bits, the row inside the strip in the low bits) ---
def strips_end_check(capi, query, res, off, matrix, go, ge, expect=PAIR_STRIPS, modes=("end",), tag=""):
    db = capi.DeviceDatabase(res, off, 24)
    routed = None
    try:
        for mode in modes:
            got = db.search(query, matrix, go, ge, mode, "sw")
            if mode == "end":
                routed = capi.DeviceDatabase.last_routing()
            want = _oracle.search(query, res, off, matrix, go, ge, mode, "sw")
            for key in want:
                if key == "aln":
                    for k, (a, b) in enumerate(zip(got[key], want[key])):
                        assert a.tolist() == b.tolist(), f"{tag} {mode} alignment {k}"
                else:
                    np.testing.assert_array_equal(got[key], want[key], err_msg=f"{tag} {mode} {key}")
    finally:
        db.close()
    if expect is not None and routed is not None:
        assert (routed[1] & 31) == expect, f"{tag}: lane-per-target pass ran {routed[1]}"
    return routed


@pytest.mark.parametrize("qlen", [61, 64, 65, 96, 97, 129, 193, 333, 700])
def test_strips_end_locations_every_kind_of_strip_count(capi, qlen, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    rng = np.random.default_rng(7000 + qlen)
    query = _data.random_protein(rng, qlen)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 500, size=300)]
    # related targets whose best cells fall into every strip: pieces of the query from different rows
    for k in range(24):
        a = int(rng.integers(0, max(1, qlen - 20)))
        piece = _data.mutate(rng, query[a:a + int(rng.integers(12, 40))], 0.1)
        seqs.append(np.concatenate([_data.random_protein(rng, int(rng.integers(0, 40))), piece,
                                    _data.random_protein(rng, int(rng.integers(0, 40)))]))
    seqs += [query[: qlen // 2], query[qlen // 3:], np.zeros(0, dtype=np.uint8), query[-5:]]
    res, off = _oracle.flatten(seqs)
    strips_end_check(capi, query, res, off, B62, 11, 1, modes=("end", "full"), tag=f"Q={qlen}")


def test_strips_end_locations_ties_across_strips(capi, tuning):
    # the query repeats one block in every strip: a target holding the block once has equally good
    # alignments ending in the same column at rows of different strips (the smallest row wins), a target
    # holding it twice has them in different columns too (the smallest column wins)
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    rng = np.random.default_rng(71)
    block = _data.random_protein(rng, 30)
    query = np.concatenate([block, _data.random_protein(rng, 14)] * 4)[:170]
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(20, 300, size=200)]
    for k in range(30):
        pre, mid, post = (_data.random_protein(rng, int(rng.integers(0, 50))) for _ in range(3))
        seqs.append(np.concatenate([pre, block, post]))
        seqs.append(np.concatenate([pre, block, mid, block, post]))
        seqs.append(np.concatenate([pre, block[:17], post]))
    res, off = _oracle.flatten(seqs)
    m = scaled_identity(24, 5, -4)
    strips_end_check(capi, query, res, off, m, 6, 2, modes=("end", "full"), tag="repeats")
    strips_end_check(capi, query, res, off, B62, 11, 1, tag="repeats, BLOSUM62")


@pytest.mark.parametrize("go,ge", [(3, 1), (11, 1), (1, 1), (2, 5), (5, 0), (14, 12)])
def test_strips_end_locations_gap_models(capi, go, ge, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    rng = np.random.default_rng(go * 100 + ge + 72)
    query = _data.random_protein(rng, 150)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 400, size=300)]
    seqs += [_data.mutate(rng, query[a:a + 40], 0.2) for a in range(0, 110, 10)]
    res, off = _oracle.flatten(seqs)
    strips_end_check(capi, query, res, off, B62, go, ge, expect=None, tag=f"gap {go}/{ge}")


def test_strips_end_locations_range_is_left_and_lanes_are_redone(capi, tuning):
    # strips of 44 rows: 6 row bits, exact below 384; copies of k query residues score 11 k under this
    # matrix: both sides of the limit, in one strip and across several
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    rng = np.random.default_rng(73)
    m = scaled_identity(24, 11, -4)
    query = _data.random_protein(rng, 130)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 300, size=300)]
    for k in (130, 80, 40, 36, 35, 34, 33, 30, 10):
        for at in (0, 25, 50):
            seqs.append(np.concatenate([_data.random_protein(rng, 17), query[at:at + k], _data.random_protein(rng, 9)]))
    res, off = _oracle.flatten(seqs)
    routed = strips_end_check(capi, query, res, off, m, 5, 2, modes=("end", "full"), tag="match 11")
    assert routed[3] >= 12


def test_strips_end_locations_probe_declines_scores_in_the_thousands(capi):
    # long query, long targets, cheap gaps: random pairs score far beyond 384. The twelve longest groups
    # go through the scores-only kernel first; since round 3 the search then takes TWO sweeps of the strips
    # kernel (scores, then the first cell that holds each score) instead of the general kernel's row scans
    # (MIOPAL_NO_TWO_PASS_ENDS restores them); under the usual gap costs the same search stays on the row
    # keys. (Sizes at which the host picks the strips kernel by itself; the general kernel's own answer is
    # the reference, the checker on a sample.)
    rng = np.random.default_rng(74)
    query = _data.random_protein(rng, 600)
    res, off = _data.random_db(rng, np.full(90_000, 600))
    db = capi.DeviceDatabase(res, off, 24)
    try:
        with capi.tuning(NO_TWO_PASS_ENDS="1"):
            db.search(query, B62, 3, 1, "end", "sw")
            assert (capi.DeviceDatabase.last_routing()[1] & 31) == 1
        for go, ge, want in ((3, 1, PAIR_STRIPS), (11, 1, PAIR_STRIPS)):
            got = db.search(query, B62, go, ge, "end", "sw")
            assert (capi.DeviceDatabase.last_routing()[1] & 31) == want, (go, ge)
            ref = _oracle.search(query, res[:off[100]], off[:101], B62, go, ge, "end", "sw")
            for key in ("score", "end_q", "end_t"):
                np.testing.assert_array_equal(got[key][:100], ref[key], err_msg=f"{go}/{ge} {key}")
            with capi.tuning(NO_PAIR_STRIPS="1"):
                general = db.search(query, B62, go, ge, "end", "sw")
            for key in ("score", "end_q", "end_t"):
                np.testing.assert_array_equal(got[key], general[key], err_msg=f"{go}/{ge} {key}: every target")
    finally:
        db.close()


@pytest.mark.parametrize("qlen", [65, 97, 130, 193, 333, 700])
def test_strips_end_locations_in_two_sweeps(capi, qlen, tuning):
    # the second form of multi-strip end locations (scores beyond the row keys' range): a scores-only sweep,
    # then a sweep that looks for each target's known score - first column, then first row, over all strips;
    # targets without a positive cell, ties across strips and columns, lanes beyond 25600 (redone), `full`
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    tuning.setenv("MIOPAL_TWO_PASS_ENDS", "1")
    rng = np.random.default_rng(9000 + qlen)
    query = _data.random_protein(rng, qlen)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1, 500, size=300)]
    for k in range(24):
        a = int(rng.integers(0, max(1, qlen - 20)))
        piece = _data.mutate(rng, query[a:a + int(rng.integers(12, 60))], 0.1)
        seqs.append(np.concatenate([_data.random_protein(rng, int(rng.integers(0, 40))), piece,
                                    _data.random_protein(rng, int(rng.integers(0, 40)))]))
    seqs += [query[: qlen // 2], query[qlen // 3:], np.zeros(0, dtype=np.uint8), query[-5:], query.copy()]
    res, off = _oracle.flatten(seqs)
    strips_end_check(capi, query, res, off, B62, 3, 1, modes=("end", "full"), tag=f"two sweeps Q={qlen}")
    # ties: a block repeated in the query, once and twice in the targets (smallest column, then smallest row)
    block = _data.random_protein(rng, 30)
    rep = np.concatenate([block, _data.random_protein(rng, 14)] * 8)[:qlen]
    seqs2 = [_data.random_protein(rng, int(n)) for n in rng.integers(20, 300, size=100)]
    for k in range(20):
        pre, mid, post = (_data.random_protein(rng, int(rng.integers(0, 50))) for _ in range(3))
        seqs2 += [np.concatenate([pre, block, post]), np.concatenate([pre, block, mid, block, post])]
    res2, off2 = _oracle.flatten(seqs2)
    strips_end_check(capi, rep, res2, off2, scaled_identity(24, 5, -4), 6, 2, tag="two sweeps, repeats")
    # scores beyond the lanes' own range (25600): those lanes are redone by the int32 kernel
    m = scaled_identity(24, 500, -300)
    seqs3 = [_data.random_protein(rng, int(n)) for n in rng.integers(10, 300, size=200)]
    seqs3 += [np.concatenate([_data.random_protein(rng, 11), query[at:at + k], _data.random_protein(rng, 5)])
              for k in (60, 52, 51, 50, 30) for at in (0, 37) if at + k <= qlen]
    res3, off3 = _oracle.flatten(seqs3)
    strips_end_check(capi, query, res3, off3, m, 700, 100, tag="two sweeps, match 500")
