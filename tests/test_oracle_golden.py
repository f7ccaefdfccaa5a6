"""Pins the CPU checker (oracle/) on every known-answer vector the reference's
tests hold for the opalSearchDatabase path (tests/golden/reference_vectors.json)."""
import json
import os

import numpy as np
import pytest

import _oracle
from pyopal_amd.matrices import ScoringMatrix, _self_check

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
    VECTORS = json.load(f)["vectors"]


def cigar(ops):
    # same run-length encoding as FullResult.cigar (src/pyopal/lib.pyx:1017-1037)
    out, i = [], 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] % 3 == ops[i] % 3:
            j += 1
        out.append(f"{j - i}{'MID'[ops[i] % 3]}")
        i = j
    return "".join(out)


def coverage(res, k, qlen, tlen, reference):
    # restates FullResult.coverage (src/pyopal/lib.pyx:1095-1119), including
    # the trimming of edge operations that are gaps in the reference sequence
    ops = list(res["aln"][k])
    if reference == "query":
        reflen, length, op = qlen, res["end_q"][k] + 1 - res["start_q"][k], 1
    else:
        reflen, length, op = tlen, res["end_t"][k] + 1 - res["start_t"][k], 2
    for o in ops:
        if o != op:
            break
        length -= 1
    for o in reversed(ops):
        if o != op:
            break
        length -= 1
    return 0.0 if length < 0 else length / reflen


def test_matrix_self_check():
    _self_check()


@pytest.mark.parametrize("vec", VECTORS, ids=[v["id"] for v in VECTORS])
@pytest.mark.parametrize("mode", ["score", "end", "full"])
def test_reference_vectors(vec, mode):
    m = ScoringMatrix.from_name(vec["matrix"]).int_array()
    q = _oracle.encode(vec["query"])
    res, off = _oracle.flatten([_oracle.encode(t) for t in vec["targets"]])
    out = _oracle.search(q, res, off, m, vec["gap_open"], vec["gap_extend"], mode, vec["algorithm"])

    def check(key, field):
        if field in vec:
            for k, want in enumerate(vec[field]):
                if want is not None:
                    assert int(out[key][k]) == want, (vec["id"], field, k)

    check("score", "score")
    if mode in ("end", "full"):
        check("end_q", "query_end")
        check("end_t", "target_end")
    if mode == "full":
        check("start_q", "query_start")
        check("start_t", "target_start")
        for k, want in enumerate(vec.get("cigar", [])):
            assert cigar(list(out["aln"][k])) == want
        for k, want in enumerate(vec.get("coverage_query", [])):
            assert coverage(out, k, len(q), len(vec["targets"][k]), "query") == pytest.approx(want)
        for k, want in enumerate(vec.get("coverage_target", [])):
            assert coverage(out, k, len(q), len(vec["targets"][k]), "target") == pytest.approx(want)


def test_g1_ops_exact():
    v = VECTORS[0]
    m = ScoringMatrix.from_name("BLOSUM50").int_array()
    res, off = _oracle.flatten([_oracle.encode(v["targets"][0])])
    out = _oracle.search(_oracle.encode(v["query"]), res, off, m, 3, 1, "full", "nw")
    assert list(out["aln"][0]) == [2, 0, 0, 0, 3, 0, 2, 0]
