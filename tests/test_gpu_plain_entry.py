"""opalSearchDatabase as the reference binds it (src/pyopal/opal.pxd:38-52, pyx.in:76-91): N host
pointers and lengths in, N result structs out, on every call. The library gathers the sequences
piece by piece on a few threads while earlier pieces cross PCIe, checks the residues on the way,
and refills the handle of the previous call. Checked here: the gather (ragged and empty sequences,
sequences across piece boundaries), the residue check wherever the bad byte sits, refilling with
larger / smaller / other-alphabet databases, concurrent callers, the cache controls."""
import ctypes
import threading

import numpy as np
import pytest

import _cpu_baseline
import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

pytestmark = pytest.mark.gpu
# OpalSearchResult (include/opal.h) as a numpy record: six ints, the alignment pointer, its length
RESULT = np.dtype({"names": ["scoreSet", "score", "endLocationTarget", "endLocationQuery", "startLocationTarget",
                             "startLocationQuery", "alignment", "alignmentLength"],
                   "formats": ["<i4"] * 6 + ["<u8", "<i4"], "offsets": [0, 4, 8, 12, 16, 20, 24, 32], "itemsize": 40})
MIOPAL_ERR_BAD_ARGUMENT = 101
B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)


@pytest.fixture(scope="module")
def capi():
    from pyopal_amd import _capi
    assert _capi.lib().miopalDeviceCount() >= 1, "no gfx950 device visible"
    return _capi


def plain_search(capi, query, res, off, matrix, alphabet, go=3, ge=1, mode="score", algo="sw", expect_rc=0):
    """One opalSearchDatabase call over pointers into `res`; returns the result-struct array."""
    n = len(off) - 1
    assert ctypes.sizeof(capi.OpalSearchResult) == RESULT.itemsize
    results = np.zeros(n, dtype=RESULT)
    step = RESULT.itemsize
    rptrs = (results.ctypes.data + np.arange(n, dtype=np.uint64) * step).astype(np.uint64)
    ptrs = (res.ctypes.data + np.asarray(off[:-1], dtype=np.uint64)).astype(np.uint64)
    lens = np.diff(off).astype(np.int32)
    query = np.ascontiguousarray(query, dtype=np.uint8)
    matrix = np.ascontiguousarray(matrix, dtype=np.int32)
    rc = capi.lib().opalSearchDatabase(query.ctypes.data, len(query), ptrs.ctypes.data, n, lens.ctypes.data, go, ge,
                                       matrix.ctypes.data, alphabet, rptrs.ctypes.data, capi.SEARCH[mode],
                                       capi.MODE[algo], 1)
    assert rc == expect_rc, (rc, capi.last_error())
    return results


def test_gather_of_ragged_sequences_across_piece_boundaries(capi):
    # 60 MB of residues in sequences of 0 .. 3000 (many shorter than one 16-byte block, some empty):
    # eight bounce pieces of 8 MB, most boundaries inside a sequence. Every score against the AVX2 CPU
    # checker, a sample against the scalar one.
    rng = np.random.default_rng(91)
    lengths = np.concatenate([rng.integers(0, 40, size=20_000), rng.integers(0, 3000, size=40_000),
                              np.zeros(50, dtype=np.int64)])
    rng.shuffle(lengths)
    res, off = _data.random_db(rng, lengths)
    assert off[-1] > 7 * (8 << 20)
    q = _oracle.encode(_data.README_QUERY)
    out = plain_search(capi, q, res, off, B62, 24)
    assert out["scoreSet"].all()
    cpu = _cpu_baseline.CpuDatabase(res, off)
    want = cpu.search_sw(q, B62, 3, 1, 8)
    cpu.close()
    np.testing.assert_array_equal(out["score"], want)
    pick = rng.choice(len(lengths), size=300, replace=False)
    sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in pick])
    ref = _oracle.search(q, sres, soff, B62, 3, 1, "end", "sw")
    end = plain_search(capi, q, sres, soff, B62, 24, mode="end")
    np.testing.assert_array_equal(out["score"][pick], ref["score"])
    np.testing.assert_array_equal(end["endLocationQuery"], ref["end_q"])
    np.testing.assert_array_equal(end["endLocationTarget"], ref["end_t"])


@pytest.mark.parametrize("where", ["first byte", "last byte", "short sequence", "block tail", "piece edge", "second piece"])
def test_residue_check_wherever_the_bad_byte_sits(capi, where):
    rng = np.random.default_rng(92)
    lengths = np.concatenate([[5, 300, 17, 1, 33], np.full(40_000, 300)])
    res, off = _data.random_db(rng, lengths)
    res = res.copy()
    at = {"first byte": 0, "last byte": len(res) - 1, "short sequence": off[3], "block tail": off[2] - 1,
          "piece edge": (8 << 20) - 1, "second piece": (8 << 20) + 12345}[where]
    res[at] = 24
    q = _oracle.encode(_data.README_QUERY)
    plain_search(capi, q, res, off, B62, 24, expect_rc=MIOPAL_ERR_BAD_ARGUMENT)
    assert "residue 24 out of range" in capi.last_error()
    # the same bytes are fine in an alphabet that has the symbol; and the library is usable afterwards
    res[at] = 3
    out = plain_search(capi, q, res, off, B62, 24)
    ref = _oracle.search(q, res[:off[200]], off[:201], B62, 3, 1, "score", "sw")
    np.testing.assert_array_equal(out["score"][:200], ref["score"])


def test_handle_is_refilled_with_other_databases(capi):
    # a chain of calls with databases of very different sizes, alphabets and length profiles: each is
    # answered from the sequences of THAT call (the previous call's packed views must not survive)
    rng = np.random.default_rng(93)
    capi.lib().miopalReleaseCaches()
    for step, (n, lo, hi, A) in enumerate([(3000, 1, 400, 24), (50_000, 200, 400, 24), (40, 1, 90, 24), (5000, 1, 300, 4),
                                           (5000, 1, 300, 32), (3000, 1, 400, 24), (1, 1, 2, 24), (200, 5000, 9000, 24)]):
        lengths = rng.integers(lo, hi, size=n)
        if A == 24:
            res, off = _data.random_db(rng, lengths)
            m = B62
            q = _data.random_protein(rng, int(rng.integers(10, 120)))
        else:
            res = rng.integers(0, A, size=int(lengths.sum())).astype(np.uint8)
            off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
            m = rng.integers(-5, 7, size=(A, A)).astype(np.int32)
            m[np.arange(A), np.arange(A)] = 6
            m = m.ravel()
            q = rng.integers(0, A, size=40).astype(np.uint8)
        for mode, algo in (("score", "sw"), ("end", "nw")):
            out = plain_search(capi, q, res, off, m, A, mode=mode, algo=algo)
            k = min(n, 150)
            ref = _oracle.search(q, res[:off[k]], off[:k + 1], m, 3, 1, mode, algo)
            np.testing.assert_array_equal(out["score"][:k], ref["score"], err_msg=f"step {step} {mode} {algo}")
            if mode == "end":
                np.testing.assert_array_equal(out["endLocationTarget"][:k], ref["end_t"], err_msg=f"step {step}")
            if n > 1000 and A == 24 and algo == "sw":
                cpu = _cpu_baseline.CpuDatabase(res, off)
                want = cpu.search_sw(q, m, 3, 1, 8)
                cpu.close()
                np.testing.assert_array_equal(out["score"], want, err_msg=f"step {step}: every score")


def test_full_alignments_through_a_refilled_handle(capi):
    rng = np.random.default_rng(94)
    q = _data.random_protein(rng, 45)
    for n in (60, 25):
        seqs = [_data.random_protein(rng, int(x)) for x in rng.integers(1, 150, size=n)]
        res, off = _oracle.flatten(seqs)
        out = plain_search(capi, q, res, off, B62, 24, mode="full", algo="ov")
        ref = _oracle.search(q, res, off, B62, 3, 1, "full", "ov")
        libc = ctypes.CDLL(None)
        libc.free.argtypes = [ctypes.c_void_p]
        for k in range(n):
            assert out["score"][k] == ref["score"][k]
            assert (out["startLocationQuery"][k], out["startLocationTarget"][k]) == (ref["start_q"][k], ref["start_t"][k])
            length = int(out["alignmentLength"][k])
            ops = (ctypes.c_ubyte * length).from_address(int(out["alignment"][k])) if length else []
            assert list(ops) == ref["aln"][k].tolist()
            if length:
                libc.free(int(out["alignment"][k]))


def test_concurrent_callers_each_get_their_own_answer(capi):
    # the reference's thread pool calls the entry point from several threads, each with its own chunk
    rng = np.random.default_rng(95)
    q = _oracle.encode(_data.README_QUERY)
    chunks = []
    for t in range(6):
        res, off = _data.random_db(rng, rng.integers(1, 500, size=int(rng.integers(2000, 30_000))))
        chunks.append((res, off))
    got = [None] * len(chunks)
    errors = []

    def work(t):
        try:
            for _ in range(3):
                got[t] = plain_search(capi, q, chunks[t][0], chunks[t][1], B62, 24)["score"].copy()
        except BaseException as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(chunks))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t, (res, off) in enumerate(chunks):
        cpu = _cpu_baseline.CpuDatabase(res, off)
        want = cpu.search_sw(q, B62, 3, 1, 4)
        cpu.close()
        np.testing.assert_array_equal(got[t], want, err_msg=f"thread {t}")


def test_cache_controls(capi, tuning):
    import torch
    rng = np.random.default_rng(96)
    q = _oracle.encode(_data.README_QUERY)
    res, off = _data.random_db(rng, np.full(200_000, 300))   # 60 MB of residues, as much again packed
    ref = _oracle.search(q, res[:off[100]], off[:101], B62, 3, 1, "score", "sw")["score"]
    capi.lib().miopalReleaseCaches()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    np.testing.assert_array_equal(plain_search(capi, q, res, off, B62, 24)["score"][:100], ref)
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held >= 100 << 20, "the handle of the call is kept for the next one"
    np.testing.assert_array_equal(plain_search(capi, q, res, off, B62, 24)["score"][:100], ref)
    assert free0 - torch.cuda.mem_get_info()[0] <= held + (16 << 20), "a second call re-uses it"
    capi.lib().miopalReleaseCaches()
    assert free0 - torch.cuda.mem_get_info()[0] <= 16 << 20, "miopalReleaseCaches gives the memory back"
    tuning.setenv("MIOPAL_SPARE_HANDLE_MB", "0")
    np.testing.assert_array_equal(plain_search(capi, q, res, off, B62, 24)["score"][:100], ref)
    assert free0 - torch.cuda.mem_get_info()[0] <= 16 << 20, "nothing is kept with MIOPAL_SPARE_HANDLE_MB=0"


def test_search_under_the_upload(capi, tuning):
    # Round 4 (opt-in, MIOPAL_SEARCH_UNDER_UPLOAD=1: no faster on this pool, host.hip): a large database of short
    # targets searched in segments while its later segments still cross PCIe. Every score against the AVX2 CPU checker
    # and against the same call with the search behind the upload (the default); end locations of a sample against
    # the scalar checker; a bad residue in the LAST segment still fails the call.
    rng = np.random.default_rng(95)
    n = 300_000
    lengths = rng.integers(300, 385, size=n)              # ragged, all within the prefetched views' bound
    lengths[rng.integers(0, n, size=500)] = 0             # ... and a few empty ones
    res, off = _data.random_db(rng, lengths)
    assert off[-1] >= (96 << 20)
    q = _oracle.encode(_data.README_QUERY)
    behind = plain_search(capi, q, res, off, B62, 24)
    tuning.setenv("MIOPAL_SEARCH_UNDER_UPLOAD", "1")
    under = plain_search(capi, q, res, off, B62, 24)
    assert under["scoreSet"].all()
    cpu = _cpu_baseline.CpuDatabase(res, off)
    want = cpu.search_sw(q, B62, 3, 1, 8)
    cpu.close()
    np.testing.assert_array_equal(under["score"], want)
    np.testing.assert_array_equal(behind["score"], want)
    # end locations: the whole database again, a sample of every segment against the scalar checker
    ends = plain_search(capi, q, res, off, B62, 24, mode="end", algo="hw")
    pick = np.sort(rng.choice(n, size=400, replace=False))
    sres, soff = _oracle.flatten([res[off[k]:off[k + 1]] for k in pick])
    ref = _oracle.search(q, sres, soff, B62, 3, 1, "end", "hw")
    np.testing.assert_array_equal(ends["score"][pick], ref["score"])
    np.testing.assert_array_equal(ends["endLocationQuery"][pick], ref["end_q"])
    np.testing.assert_array_equal(ends["endLocationTarget"][pick], ref["end_t"])
    # a residue outside the alphabet in the last segment: the call fails, whatever the earlier segments returned
    bad = res.copy()
    bad[off[-1] - 5] = 24
    plain_search(capi, q, bad, off, B62, 24, expect_rc=MIOPAL_ERR_BAD_ARGUMENT)
    assert "out of range" in capi.last_error()
    # ... and the call after it is clean
    again = plain_search(capi, q, res, off, B62, 24)
    np.testing.assert_array_equal(again["score"], want)
    capi.lib().miopalReleaseCaches()
