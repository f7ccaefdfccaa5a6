"""NW / HW / OV of several strips on the pair table (interseq_pair_global_strips_kernel,
interseq_impl.h) against the CPU checker: every kind of strip count, last query rows that are and
are not a strip's last row, the top / left borders of every mode across strip boundaries, rebased
column shifts, gap models on both sides of open == ext, end locations merged over the strips
(OV's last column against its last row), other alphabets, routing, and the time-out escapes of the
strip hand-over (fault injection). Bit-exact, through the C ABI."""
import ctypes

import numpy as np
import pytest

import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

pytestmark = pytest.mark.gpu

B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
GLOBAL_STRIPS = 2 + 5   # miopalLastRouting counts[1]: 2 + kPairGlobalStrips
PAIR_STRIPS = 2 + 4
ALGOS = ["nw", "hw", "ov"]


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


@pytest.fixture
def forced(tuning):
    # (also for searches of few units, where the host prefers the general kernel)
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")


def check(capi, algo, query, res, off, matrix, go, ge, modes=("score", "end"), expect=GLOBAL_STRIPS, tag="", alphabet=24):
    db = capi.DeviceDatabase(res, off, alphabet)
    routed = None
    try:
        for mode in modes:
            got = db.search(query, matrix, go, ge, mode, algo)
            routed = capi.DeviceDatabase.last_routing()
            want = _oracle.search(query, res, off, matrix, go, ge, mode, algo)
            for key in want:
                if key == "aln":
                    for k, (a, b) in enumerate(zip(got[key], want[key])):
                        assert a.tolist() == b.tolist(), f"{tag} {algo} {mode} alignment {k}"
                else:
                    np.testing.assert_array_equal(got[key], want[key], err_msg=f"{tag} {algo} {mode} {key}")
            if expect is not None:
                assert (routed[1] & 31) == expect, f"{tag} {algo} {mode}: lane-per-target pass ran kernel {routed[1]}"
    finally:
        db.close()
    return routed


def mixed_targets(rng, query, n=300, longest=500):
    seqs = [_data.random_protein(rng, int(k)) for k in rng.integers(1, longest, size=n)]
    seqs += [np.concatenate([_data.random_protein(rng, int(rng.integers(0, 40))), _data.mutate(rng, query, 0.15),
                             _data.random_protein(rng, int(rng.integers(0, 40)))]) for _ in range(20)]
    seqs += [query[: len(query) // 2], query[len(query) // 3:], query.copy(), np.zeros(0, dtype=np.uint8),
             _data.random_protein(rng, 1)]
    return _oracle.flatten(seqs)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("qlen", [65, 66, 96, 97, 100, 101, 104, 129, 150, 193, 300, 333, 700])
def test_every_kind_of_strip_count(capi, algo, qlen, forced):
    # two strips and many; queries whose last row is the last strip's last row (100 = 2 x 50, 150, 300)
    # and queries that leave padding rows below it (101, 333, 700); empty, one-residue and related targets
    rng = np.random.default_rng(7000 + qlen)
    query = _data.random_protein(rng, qlen)
    res, off = mixed_targets(rng, query)
    check(capi, algo, query, res, off, B62, 3, 1, tag=f"Q={qlen}")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("go,ge", [(11, 1), (1, 1), (5, 0), (0, 0), (14, 12), (2, 5), (40, 12), (120, 100)])
def test_gap_models(capi, algo, go, ge, forced):
    # (NW with open < ext and gap costs beyond the static bounds take the general kernel: no kernel
    # check; ext 100: HW / OV rebase their shift every tenth chunk, in every strip at the same chunks)
    rng = np.random.default_rng(go * 100 + ge + 11)
    query = _data.random_protein(rng, 147)
    res, off = mixed_targets(rng, query, longest=400)
    check(capi, algo, query, res, off, B62, go, ge, expect=None, tag=f"gap {go}/{ge}")


@pytest.mark.parametrize("algo", ALGOS)
def test_rebased_shift_crosses_strip_boundaries(capi, algo, forced):
    # ext 40: the column shift of HW / OV is rebased every 25th chunk while rows travel from strip to
    # strip; long ragged groups (log-normal lengths) in one batch
    rng = np.random.default_rng(71)
    query = _data.random_protein(rng, 120)
    lengths = np.clip(rng.lognormal(5.5, 0.8, size=2000).astype(int), 1, 3000)
    seqs = [_data.random_protein(rng, int(n)) for n in lengths]
    seqs += [np.concatenate([_data.random_protein(rng, 600), _data.mutate(rng, query, 0.1), _data.random_protein(rng, 40)])
             for _ in range(10)]
    res, off = _oracle.flatten(seqs)
    check(capi, algo, query, res, off, B62, 45, 40, tag="ext 40")


@pytest.mark.parametrize("algo", ALGOS)
def test_end_location_ties_across_strips(capi, algo, forced):
    # a two-letter alphabet with small scores: many equal candidates - OV's last column against its
    # last row, the first of equal rows in different strips, the first of equal columns
    rng = np.random.default_rng(72)
    A = 2
    matrix = np.array([[1, -1], [-1, 1]], dtype=np.int32).ravel()
    query = rng.integers(0, A, size=140).astype(np.uint8)
    seqs = [rng.integers(0, A, size=int(n)).astype(np.uint8) for n in rng.integers(1, 200, size=400)]
    seqs += [query.copy(), query[:70], query[70:], np.zeros(50, np.uint8), np.ones(90, np.uint8)]
    res, off = _oracle.flatten(seqs)
    check(capi, algo, query, res, off, matrix, 1, 1, tag="ties", alphabet=A)
    check(capi, algo, query, res, off, matrix, 0, 0, expect=None, tag="ties, free gaps", alphabet=A)


@pytest.mark.parametrize("A,qlen", [(4, 130), (12, 333), (32, 64), (32, 100)])
def test_other_alphabets(capi, A, qlen, forced):
    # the pair table of a 33-symbol alphabet holds 36 rows: 64 rows are two strips of 32, 100 rows three of 34
    rng = np.random.default_rng(A * 1000 + qlen + 1)
    matrix = rng.integers(-6, 8, size=(A, A)).astype(np.int32)
    matrix[np.arange(A), np.arange(A)] = rng.integers(3, 12, size=A)
    seqs = [rng.integers(0, A, size=int(n)).astype(np.uint8) for n in rng.integers(1, 400, size=500)]
    q = rng.integers(0, A, size=qlen).astype(np.uint8)
    seqs += [np.concatenate([seqs[k][:50], q, seqs[k][:30]]) for k in range(5)]
    res, off = _oracle.flatten(seqs)
    for algo in ALGOS:
        check(capi, algo, q, res, off, matrix.ravel(), 5, 2, tag=f"A={A}", alphabet=A)


@pytest.mark.parametrize("algo", ALGOS)
def test_long_targets_stay_in_the_lanes(capi, algo, forced):
    # targets of thousands of residues against a 300-residue query: NW scores far below -32768 are read
    # as 32-bit values at each lane's own last column, nothing is redone for its length
    rng = np.random.default_rng(73)
    query = _data.random_protein(rng, 300)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(3000, 8000, size=200)]
    seqs += [_data.random_protein(rng, int(n)) for n in rng.integers(100, 600, size=300)]
    res, off = _oracle.flatten(seqs)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        got = db.search(query, B62, 11, 5, "end", algo)
        routed = capi.DeviceDatabase.last_routing()
    finally:
        db.close()
    import _cpu_baseline
    cpu = _cpu_baseline.CpuDatabase(res, off)
    want = cpu.search(query, B62, 11, 5, algo, 8)
    cpu.close()
    np.testing.assert_array_equal(got["score"], want)
    assert (routed[1] & 31) == GLOBAL_STRIPS and routed[3] == 0
    if algo == "nw":
        assert got["score"].min() < -33000
    sample = np.concatenate([np.arange(0, 8), np.arange(200, 230)])
    for k in sample:
        ref = _oracle.search(query, res[off[k]:off[k + 1]], np.array([0, off[k + 1] - off[k]]), B62, 11, 5, "end", algo)
        assert (got["score"][k], got["end_q"][k], got["end_t"][k]) == (ref["score"][0], ref["end_q"][0], ref["end_t"][0]), k


def test_few_groups_long_query(capi, forced):
    # 40 strips over 3 batches: the units of a batch run side by side in different workgroups, each
    # wavefront two chunks behind the one above it (BASELINE configs[3]'s shape, 4000 targets)
    rng = np.random.default_rng(74)
    query = _data.random_protein(rng, 2000)
    seqs = [_data.random_protein(rng, int(n)) for n in rng.integers(1500, 2500, size=4000)]
    seqs[7] = np.concatenate([seqs[7][:300], _data.mutate(rng, query, 0.3), seqs[7][300:400]])
    res, off = _oracle.flatten(seqs)
    import _cpu_baseline
    cpu = _cpu_baseline.CpuDatabase(res, off)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for algo in ALGOS:
            got = db.search(query, B62, 3, 1, "score", algo)["score"]
            assert capi.DeviceDatabase.last_routing()[1] == GLOBAL_STRIPS
            np.testing.assert_array_equal(got, cpu.search(query, B62, 3, 1, algo, 8), err_msg=algo)
    finally:
        db.close()
        cpu.close()


def test_routing(capi):
    # few (group, strip) units: the general kernel; many, or 16 strips and more: the strips kernel; a query beyond the static
    # range of the patterns (Q (max S + ext) above 0x7C00): the general kernel; the switch
    rng = np.random.default_rng(75)
    res, off = _data.random_db(rng, np.full(100_000, 100))
    db = capi.DeviceDatabase(res, off, 24)
    try:
        for qlen, lo, hi, want in ((100, 0, 100_000, 1), (600, 0, 100_000, GLOBAL_STRIPS), (300, 0, 3000, 1),
                                   (900, 0, 3000, GLOBAL_STRIPS), (2600, 0, 100_000, 1)):
            q = _data.random_protein(rng, qlen)
            for algo in ALGOS:
                got = db.search(q, B62, 11, 1, "score", algo, lo, hi)["score"]
                assert (capi.DeviceDatabase.last_routing()[1] & 31) == want, (qlen, hi, algo)
                ref = _oracle.search(q, res[:off[60]], off[:61], B62, 11, 1, "score", algo)["score"]
                np.testing.assert_array_equal(got[:60], ref)
    finally:
        db.close()


def test_switch_restores_the_general_kernel(capi, tuning):
    tuning.setenv("MIOPAL_PAIR_STRIPS", "1")
    tuning.setenv("MIOPAL_NO_GLOBAL_STRIPS", "1")
    rng = np.random.default_rng(76)
    query = _data.random_protein(rng, 150)
    res, off = mixed_targets(rng, query)
    for algo in ALGOS:
        check(capi, algo, query, res, off, B62, 3, 1, expect=1, tag="switch")


def _inject(capi, kind, unit, spin_cap):
    fn = capi.lib().miopalTestInjectFault
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    fn.restype = None
    fn(kind, unit, spin_cap)


@pytest.mark.parametrize("algo", ["sw", "nw", "ov"])
def test_a_unit_that_never_publishes_is_survived(capi, algo, forced):
    # Time-out escape of the packed strips kernels, run once: one (batch, strip) unit publishes nothing
    # (test hook, an argument of the next search - not an environment switch), the unit below it gives
    # up after its spin cap, flags its lanes and poisons its own counter, the units further down see
    # the poison at once, and the int32 kernel recomputes every flagged target: the scores are right.
    rng = np.random.default_rng(77)
    query = _data.random_protein(rng, 260)
    res, off = _data.random_db(rng, rng.integers(50, 400, size=6000))
    want = _oracle.search_parallel(query, res, off, B62, 11, 1, "score", algo)["score"]
    db = capi.DeviceDatabase(res, off, 24)
    try:
        clean = db.search(query, B62, 11, 1, "score", algo)["score"]
        routed = capi.DeviceDatabase.last_routing()
        assert (routed[1] & 31) == (PAIR_STRIPS if algo == "sw" else GLOBAL_STRIPS) and routed[3] == 0
        np.testing.assert_array_equal(clean, want)
        _inject(capi, 1, 1, 1 << 12)     # unit 1 = strip 0 of batch 1 (strip-major unit order)
        hurt = db.search(query, B62, 11, 1, "score", algo)["score"]
        routed = capi.DeviceDatabase.last_routing()
        np.testing.assert_array_equal(hurt, want)
        # the lanes of batch 1 were flagged and redone (1 to 12 groups of 128 targets)
        assert 128 <= routed[3] <= 12 * 128, routed
        again = db.search(query, B62, 11, 1, "score", algo)["score"]   # the hook was for one search
        assert capi.DeviceDatabase.last_routing()[3] == 0
        np.testing.assert_array_equal(again, want)
    finally:
        db.close()


def test_a_pair_strip_unit_that_never_publishes_fails_the_search(capi):
    # Time-out escape of the int32 kernel's (pair, strip) units, run once: the unit below the silent one
    # gives up, the search returns MIOPAL_ERR_INTERNAL (102), and the handle stays usable.
    rng = np.random.default_rng(78)
    query = _data.random_protein(rng, 300)
    lengths = np.concatenate([np.full(3000, 200), [20000, 24000]])
    res, off = _data.random_db(rng, lengths)
    db = capi.DeviceDatabase(res, off, 24)
    try:
        want = db.search(query, B62, 11, 1, "score", "nw")["score"]
        assert capi.DeviceDatabase.last_routing()[0] >= 2     # the two long targets took the int32 kernel
        ref = _oracle.search(query, res[off[3000]:], off[3000:] - off[3000], B62, 11, 1, "score", "nw")["score"]
        np.testing.assert_array_equal(want[3000:], ref)
        _inject(capi, 2, 1, 1 << 12)      # strip 1 of the first long pair
        with pytest.raises(RuntimeError) as err:
            db.search(query, B62, 11, 1, "score", "nw")
        assert "102" in str(err.value) or "gave up" in str(err.value)
        again = db.search(query, B62, 11, 1, "score", "nw")["score"]
        np.testing.assert_array_equal(again, want)
    finally:
        db.close()
