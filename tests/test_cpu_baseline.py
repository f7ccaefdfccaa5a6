"""The AVX2 CPU baseline (bench.py's cpu_baseline leg) agrees with the oracle."""
import numpy as np
import pytest

import _cpu_baseline
import _data
import _oracle
from pyopal_amd.matrices import ScoringMatrix

B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
B50 = np.array(ScoringMatrix.from_name("BLOSUM50").int_array(), dtype=np.int32)


@pytest.mark.parametrize("threads", [1, 2])
def test_random_db(threads):
    rng = np.random.default_rng(7)
    res, off = _data.random_db(rng, rng.integers(1, 400, size=700))
    q = _oracle.encode(_data.README_QUERY)
    db = _cpu_baseline.CpuDatabase(res, off)
    for m, go, ge in ((B62, 3, 1), (B50, 11, 1), (B62, 0, 0)):
        got = db.search_sw(q, m, go, ge, threads)
        want = _oracle.search(q, res, off, m, go, ge, "score", "sw")["score"]
        np.testing.assert_array_equal(got, want)
    db.close()


def test_overflow_ladder():
    rng = np.random.default_rng(8)
    q = _data.random_protein(rng, 900)
    seqs = [q.copy(), _data.mutate(rng, q, 0.1), q[100:300].copy(), _data.random_protein(rng, 700)]
    seqs += [_data.random_protein(rng, int(n)) for n in rng.integers(0, 300, size=100)]
    big = _data.random_protein(rng, 8000)
    seqs += [big]
    res, off = _oracle.flatten(seqs)
    db = _cpu_baseline.CpuDatabase(res, off)
    for query in (q, big):
        got = db.search_sw(query, B62, 3, 1, 2)
        want = _oracle.search(query, res, off, B62, 3, 1, "score", "sw")["score"]
        np.testing.assert_array_equal(got, want)
    assert want.max() > 32767
    db.close()


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
def test_global_modes_match_the_oracle(algo):
    rng = np.random.default_rng(11)
    lengths = np.concatenate([rng.integers(1, 400, size=300), [0, 1, 2, 399, 400]])
    res, off = _data.random_db(rng, lengths)
    db = _cpu_baseline.CpuDatabase(res, off)
    for qlen in (1, 53, 200):
        q = _data.random_protein(rng, qlen)
        for m, go, ge in ((B62, 3, 1), (B50, 11, 1), (B62, 2, 5), (B62, 0, 0)):
            got = db.search(q, m, go, ge, algo, 2)
            want = _oracle.search(q, res, off, m, go, ge, "score", algo)["score"]
            np.testing.assert_array_equal(got, want, err_msg=f"{algo} Q={qlen} gap {go}/{ge}")
    db.close()


@pytest.mark.parametrize("algo", ["nw", "hw", "ov"])
def test_global_modes_leave_16_bits(algo):
    # the reference's overflow test shape (src/pyopal/tests/test_aligner.py:24-37): long targets
    rng = np.random.default_rng(12)
    q = _data.random_protein(rng, 300)
    lengths = [300, 1000, 9000, 20000, 35000, 299, 31000]
    res, off = _data.random_db(rng, lengths)
    db = _cpu_baseline.CpuDatabase(res, off)
    got = db.search(q, B62, 3, 1, algo, 2)
    want = _oracle.search(q, res, off, B62, 3, 1, "score", algo)["score"]
    np.testing.assert_array_equal(got, want)
    if algo == "nw":
        assert want.min() < -32768
    db.close()
