"""`Aligner.align` / `pyopal.align` on the GPU: the reference's own assertions
(src/pyopal/tests/test_aligner.py:39-131, test_align.py:9-37, doctests) run
against ``pyopal_amd`` unchanged in meaning."""
import multiprocessing.pool
import random
import threading

import numpy as np
import pytest

import _data
import _oracle
import pyopal_amd as pyopal

pytestmark = pytest.mark.gpu


def test_nw_test1():
    # src/pyopal/tests/test_aligner.py:42-79
    aligner, db = pyopal.Aligner(), pyopal.Database(["AACCGCTG"])
    for kw in ({}, {"mode": "score"}):
        r = aligner.align("ACCTCG", db, algorithm="nw", **kw)
        assert len(r) == 1 and type(r[0]) is pyopal.ScoreResult and r[0].score == 44
    r = aligner.align("ACCTCG", db, algorithm="nw", mode="end")
    assert type(r[0]) is pyopal.EndResult and (r[0].score, r[0].query_end, r[0].target_end) == (44, 5, 7)
    r = aligner.align("ACCTCG", db, algorithm="nw", mode="full")[0]
    assert type(r) is pyopal.FullResult and r.alignment is not None
    assert (r.score, r.query_start, r.query_end, r.target_start, r.target_end) == (44, 0, 5, 0, 7)
    assert r.coverage("query") == 1 and r.coverage("target") == 7 / 8
    assert r.cigar() == "1D5M1D1M"                      # doctest src/pyopal/lib.pyx:1006-1010
    assert (r.query_length, r.target_length, r.target_index) == (6, 8, 0)


def test_sw_test1():
    # src/pyopal/tests/test_aligner.py:93-131
    aligner, db = pyopal.Aligner(), pyopal.Database(["AACCGCTG"])
    assert aligner.align("ACCTCG", db, algorithm="sw")[0].score == 47
    r = aligner.align("ACCTCG", db, algorithm="sw", mode="end")[0]
    assert (r.score, r.query_end, r.target_end) == (47, 5, 7)
    r = aligner.align("ACCTCG", db, algorithm="sw", mode="full")[0]
    assert (r.score, r.query_start, r.query_end, r.target_start, r.target_end) == (47, 0, 5, 1, 7)
    assert r.coverage("query") == pytest.approx(1) and r.coverage("target") == pytest.approx(7 / 8)


@pytest.mark.parametrize("algorithm", ["nw", "hw", "ov", "sw"])
def test_overflow_does_not_raise(algorithm):
    # src/pyopal/tests/test_aligner.py:24-37 (35 proteins of 1000..35000 aa); the
    # reference checks only that no exception escapes; scores are checked against
    # the oracle for a sub-sample in test_gpu_parity.py
    rnd = random.Random(0)
    proteins = ["".join(rnd.choices(_data.AA20, k=k)) for k in range(1000, 36000, 1000)]
    results = pyopal.Aligner().align(proteins[0], pyopal.Database(proteins), mode="score", algorithm=algorithm)
    assert len(results) == 35 and [r.target_index for r in results] == list(range(35))


@pytest.mark.parametrize("threads", [1, 2, 3])
def test_align_threads(threads):
    # src/pyopal/tests/test_align.py:9-37
    target = ["AACCGCTG", "AACCGCTA", "AACCGCTC", "AACCGCTT"]
    results = list(pyopal.align("ACCTCG", target, threads=threads, mode="full", algorithm="nw", ordered=True))
    assert [r.target_index for r in results] == [0, 1, 2, 3]
    r = results[0]
    assert (r.target_start, r.target_end, r.query_start, r.query_end, r.score) == (0, 7, 0, 5, 44)
    assert [x.score for x in results] == [44, 36, 39, 34]   # oracle values for targets 1-3


def test_align_doctest_and_pool():
    # src/pyopal/_align.py:106-111
    targets = ["AACCGCTG", "ATGCGCT", "TTATTACG"]
    assert [r.score for r in pyopal.align("ACCTG", targets, gap_open=2, ordered=True)] == [41, 31, 23]
    with multiprocessing.pool.ThreadPool(2) as pool:
        unordered = list(pyopal.align("ACCTG", targets, gap_open=2, threads=2, pool=pool))
    assert sorted((r.target_index, r.score) for r in unordered) == [(0, 41), (1, 31), (2, 23)]


def test_slices_and_mirror_invalidation():
    rng = np.random.default_rng(6)
    seqs = ["".join(_data.AA20[i] for i in rng.integers(0, 20, size=int(n))) for n in rng.integers(5, 80, size=50)]
    db = pyopal.Database(seqs)
    aligner = pyopal.Aligner("BLOSUM62")
    full = [r.score for r in aligner.align(seqs[3], db)]
    part = aligner.align(seqs[3], db, start=10, end=20)
    assert [r.target_index for r in part] == list(range(10, 20))
    assert [r.score for r in part] == full[10:20]
    assert [r.score for r in aligner.align(seqs[3], db, start=45, end=10_000)] == full[45:]
    # mutators invalidate the device mirror (they hold the write lock)
    db.append(seqs[3])
    db.reverse()
    again = [r.score for r in aligner.align(seqs[3], db)]
    assert again == [full[3]] + full[::-1]
    assert again[0] == max(again)
    sub = db.extract([0, 5, 7])
    assert [r.score for r in aligner.align(seqs[3], sub)] == [again[0], again[5], again[7]]


def test_scores_array_fast_path():
    rng = np.random.default_rng(4)
    seqs = ["".join(_data.AA20[i] for i in rng.integers(0, 20, size=int(n))) for n in rng.integers(1, 90, size=64)]
    db, aligner = pyopal.Database(seqs), pyopal.Aligner()
    for algo in ("sw", "nw", "hw", "ov"):
        want = [r.score for r in aligner.align(seqs[0], db, algorithm=algo)]
        assert aligner.scores(seqs[0], db, algorithm=algo).tolist() == want
    assert aligner.scores(seqs[0], db, start=10, end=20).tolist() == [r.score for r in aligner.align(seqs[0], db)][10:20]


def test_concurrent_queries_share_one_database():
    # README.md:116-144: several threads query one Database through one Aligner
    rng = np.random.default_rng(12)
    res, off = _data.random_db(rng, rng.integers(10, 200, size=400))
    seqs = [_oracle.NCBI and "".join(_oracle.NCBI[c] for c in res[off[k]:off[k + 1]]) for k in range(400)]
    db, aligner = pyopal.Database(seqs), pyopal.Aligner("BLOSUM62")
    queries = seqs[:8]
    want = [[r.score for r in aligner.align(q, db)] for q in queries]
    got = [None] * len(queries)

    def work(i):
        got[i] = [r.score for r in aligner.align(queries[i], db, mode="end")]
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(queries))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert got == want
    m = np.array(aligner.scoring_matrix.int_array(), dtype=np.int32)
    ref = _oracle.search(_oracle.encode(queries[0]), res, off, m, 3, 1, "score", "sw")
    assert want[0] == ref["score"].tolist()


@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_align_arrays_extension_matches_align(mode):
    # array results, equal to the list of result objects
    rng = np.random.default_rng(77)
    seqs = ["".join(rng.choice(list(_data.AA20), size=int(n))) for n in rng.integers(1, 120, size=300)]
    db = pyopal.Database(seqs)
    aligner = pyopal.Aligner(gap_open=3, gap_extend=1)
    for algo in ("nw", "hw", "ov", "sw"):
        want = aligner.align(seqs[5], db, mode=mode, algorithm=algo, start=7, end=290)
        got = aligner.align_arrays(seqs[5], db, mode=mode, algorithm=algo, start=7, end=290)
        assert len(got) == len(want) == 283
        assert got.score.tolist() == [r.score for r in want]
        assert list(got) == want
        assert got[-1] == want[-1]
        if mode == "full":
            assert got.alignment(3) == want[3].alignment
            assert got.target_length.tolist() == [len(s) for s in seqs[7:290]]
    empty = aligner.align_arrays(seqs[5], db, mode=mode, start=10, end=10)
    assert len(empty) == 0 and list(empty) == []
    with pytest.raises(ValueError):
        aligner.align_arrays(seqs[5], db, mode="nope")


def _oracle_for(seqs, picks, query, aligner, mode, algo):
    """The CPU checker's answer for the targets `picks` of `seqs` (test infrastructure: oracle/)."""
    enc = [_oracle.encode(seqs[k]) for k in picks]
    res, off = _oracle.flatten(enc)
    m = np.array(aligner.scoring_matrix.int_array(), dtype=np.int32)
    return _oracle.search(_oracle.encode(query), res, off, m, aligner.gap_open, aligner.gap_extend, mode, algo)


def _assert_arrays_equal_oracle(got, ref, mode, label):
    np.testing.assert_array_equal(got.score, ref["score"], err_msg=f"{label} score")
    if mode != "score":
        np.testing.assert_array_equal(got.query_end, ref["end_q"], err_msg=f"{label} end_q")
        np.testing.assert_array_equal(got.target_end, ref["end_t"], err_msg=f"{label} end_t")
    if mode == "full":
        np.testing.assert_array_equal(got.query_start, ref["start_q"], err_msg=f"{label} start_q")
        np.testing.assert_array_equal(got.target_start, ref["start_t"], err_msg=f"{label} start_t")
        assert len(got.operation_offsets) == len(ref["aln"]) + 1
        for k, ops in enumerate(ref["aln"]):
            lo, hi = got.operation_offsets[k], got.operation_offsets[k + 1]
            assert got.operations[lo:hi].tolist() == ops.tolist(), f"{label} operations of target {k}"


@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_align_arrays_and_scores_equal_the_checker(mode):
    # the bulk results against the CPU checker itself, not only against
    # Aligner.align - every algorithm, a slice, a query of one strip and one of several
    rng = np.random.default_rng(78)
    seqs = ["".join(rng.choice(list(_data.AA20), size=int(n))) for n in rng.integers(1, 160, size=257)]
    db = pyopal.Database(seqs)
    aligner = pyopal.Aligner("BLOSUM62", gap_open=3, gap_extend=1)
    long_query = "".join(rng.choice(list(_data.AA20), size=131))
    for query in (seqs[5], long_query):
        for algo in ("nw", "hw", "ov", "sw"):
            for lo, hi in ((0, 257), (7, 250)):
                ref = _oracle_for(seqs, range(lo, hi), query, aligner, mode, algo)
                got = aligner.align_arrays(query, db, mode=mode, algorithm=algo, start=lo, end=hi)
                _assert_arrays_equal_oracle(got, ref, mode, f"{algo} Q={len(query)} [{lo},{hi})")
                if mode == "score":
                    np.testing.assert_array_equal(aligner.scores(query, db, algorithm=algo, start=lo, end=hi), ref["score"])


@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_subsets_equal_the_checker(mode):
    # mask / extract / slices (mirrors gathered on the device from the parent's,
    # miopalDbCreateSubset) and a subset of a subset, against the CPU checker on the same sequences
    rng = np.random.default_rng(79)
    seqs = ["".join(rng.choice(list(_data.AA20), size=int(n))) for n in rng.integers(1, 200, size=180)]
    db = pyopal.Database(seqs)
    aligner = pyopal.Aligner("BLOSUM62")
    query = seqs[11]
    aligner.align(query, db)            # the parent is resident: the subsets below gather from it
    keep = rng.random(180) < 0.4
    picks_mask = [k for k in range(180) if keep[k]]
    picks_extract = [int(k) for k in rng.permutation(180)[:50]] + [3, 3]      # any order, repeats
    subsets = {"mask": (db.mask(keep.tolist()), picks_mask), "extract": (db.extract(picks_extract), picks_extract),
               "slice": (db[20:150:3], list(range(20, 150, 3)))}
    inner = subsets["extract"][0].extract([1, 0, 17])
    subsets["extract of extract"] = (inner, [picks_extract[1], picks_extract[0], picks_extract[17]])
    for name, (sub, picks) in subsets.items():
        assert len(sub) == len(picks), name
        for algo in ("nw", "hw", "ov", "sw"):
            ref = _oracle_for(seqs, picks, query, aligner, mode, algo)
            got = aligner.align_arrays(query, sub, mode=mode, algorithm=algo)
            _assert_arrays_equal_oracle(got, ref, mode, f"{name} {algo}")
            objs = aligner.align(query, sub, mode=mode, algorithm=algo)
            assert [r.score for r in objs] == ref["score"].tolist(), f"{name} {algo} objects"


def test_subsets_gather_their_mirror_from_the_parent_on_the_device(monkeypatch):
    # Database.mask / Database.extract / slices (src/pyopal/lib.pyx:694-778 share the parent's buffers):
    # while the parent is resident and unchanged, the subset's mirror is gathered from the parent's on
    # the device (miopalDbCreateSubset) - nothing is uploaded again; after a mutation of either side the
    # ordinary upload takes over. Scores, indices and full alignments equal the parent's.
    from pyopal_amd import _capi
    rng = np.random.default_rng(8)
    seqs = ["".join(_data.AA20[i] for i in rng.integers(0, 20, size=int(n))) for n in rng.integers(1, 120, size=300)]
    db = pyopal.Database(seqs)
    aligner = pyopal.Aligner("BLOSUM62")
    query = seqs[17]
    whole = aligner.align(query, db, mode="full", algorithm="sw")
    uploads, gathers = [], []
    real_init, real_subset = _capi.DeviceDatabase.__init__, _capi.DeviceDatabase.subset
    monkeypatch.setattr(_capi.DeviceDatabase, "__init__",
                        lambda self, *a, **k: (uploads.append(1), real_init(self, *a, **k))[1])
    monkeypatch.setattr(_capi.DeviceDatabase, "subset",
                        lambda self, idx: (gathers.append(len(idx)), real_subset(self, idx))[1])
    picks = [299, 0, 17, 17, 150, 3]
    mask = [bool(k % 3 == 0) for k in range(300)]
    for sub, ids in ((db.extract(picks), picks), (db.mask(mask), [k for k in range(300) if mask[k]]),
                     (db[40:200:7], list(range(40, 200, 7))), (db.extract([]), [])):
        got = aligner.align(query, sub, mode="full", algorithm="sw")
        assert [r.target_index for r in got] == list(range(len(ids)))
        assert [(r.score, r.query_end, r.target_end, r.query_start, r.target_start, r.alignment) for r in got] == \
               [(whole[k].score, whole[k].query_end, whole[k].target_end, whole[k].query_start, whole[k].target_start,
                 whole[k].alignment) for k in ids]
    assert uploads == [] and gathers == [6, 100, 23]     # (an empty subset is never searched on the device)
    # a subset of a subset chains; an edited subset, or a subset of an edited parent, uploads
    sub = db.extract(picks)
    subsub = sub.extract([1, 2])
    assert [r.score for r in aligner.align(query, subsub)] == [whole[0].score, whole[17].score]
    assert uploads == [] and gathers[-1] == 2      # gathered from the root's mirror: `sub` itself was never resident
    sub2 = db.extract(picks)
    sub2.append(query)
    assert [r.score for r in aligner.align(query, sub2)] == [whole[k].score for k in picks] + [whole[17].score]
    assert len(uploads) == 1
    sub3 = db.extract([1, 2])
    db.append(query)
    assert [r.score for r in aligner.align(query, sub3)] == [whole[1].score, whole[2].score]
    assert len(uploads) == 2


def test_align_over_several_real_devices_equals_one_device():
    # pyopal_amd.align() in ONE process with several GPUs: one residue-balanced shard of the database per
    # device, chunks cut at the shard boundaries, absolute target indices, database order with
    # ordered=True, full alignments - against the answer of device 0 alone. (The per-device launch
    # attributes of the pair-table kernels - 150 KB of dynamic LDS, hipFuncSetAttribute per device - are
    # exercised by the 53-residue query.) On a box with one GPU: logical devices.
    from pyopal_amd import _capi
    logical = _capi.lib().miopalDeviceCount() < 2
    if logical:
        # one GPU: three device ordinals on it (include/miopal.h, miopalTestSetLogicalDevices) - a handle, a
        # mirror keyed by (device, shard), a stream set and hipSetDevice per call for each, as with real devices
        _capi.raise_for(_capi.lib().miopalTestSetLogicalDevices(3))
    try:
        _several_devices_equal_one(_capi)
    finally:
        if logical:
            _capi.lib().miopalTestSetLogicalDevices(0)


def _several_devices_equal_one(_capi):
    assert _capi.lib().miopalDeviceCount() >= 2
    rng = np.random.default_rng(12)
    lengths = np.concatenate([rng.integers(200, 600, size=3000), rng.integers(20, 120, size=9000)])   # skewed
    seqs = ["".join(_data.AA20[i] for i in rng.integers(0, 20, size=int(n))) for n in lengths]
    db = pyopal.Database(seqs)
    query = "".join(_data.AA20[i] for i in rng.integers(0, 20, size=53))
    long_query = "".join(_data.AA20[i] for i in rng.integers(0, 20, size=200))
    aligner = pyopal.Aligner("BLOSUM62")
    for q, mode, algo in ((query, "score", "sw"), (query, "full", "sw"), (long_query, "end", "nw"), (long_query, "score", "sw")):
        want = aligner.align(q, db, mode=mode, algorithm=algo)      # device 0, whole database
        for threads in (0, 3):
            got = list(pyopal.align(q, db, "BLOSUM62", mode=mode, algorithm=algo, threads=threads, ordered=True))
            assert [r.target_index for r in got] == list(range(len(seqs))), (mode, threads)
            assert [r.score for r in got] == [r.score for r in want], (mode, algo, threads)
            if mode != "score":
                assert [(r.query_end, r.target_end) for r in got] == [(r.query_end, r.target_end) for r in want]
            if mode == "full":
                assert [(r.query_start, r.target_start, r.alignment) for r in got] == \
                       [(r.query_start, r.target_start, r.alignment) for r in want]
    # every device holds a shard, none the whole database
    keys = sorted(k for k in db._mirrors if k[1] is not None)
    assert len(keys) >= 2 and all(hi - lo < len(seqs) for _, (lo, hi) in keys)
