"""Generate the seeded fixtures of SURVEY.md section 8c (ii) and (iii).

    python tests/golden/make_oracle_vectors.py

The expected values come from this repo's CPU oracle (oracle/opal_oracle.c), NOT from the
reference (whose DP core is absent from the reference tree): they freeze the
oracle's behaviour so that a change of the oracle, or of the HIP path, is seen against data
that does not move. The reference's own known answers live in reference_vectors.json.

oracle_random_pairs.json   200 seeded protein pairs (lengths 1..400, a third of them noisy
                           copies so that ties and long gaps occur) x NW/HW/OV/SW, full
                           results: score, ends, starts, run-length encoded operations.
oracle_promotion.json      the lane-width ladder cases modelled on the reference's
                           tests/test_aligner.py:28-37 (target lengths 1000..35000 against a
                           query of the same composition): scores that cross the int8 and
                           int16 ranges in both directions, all four algorithms, plus ends.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _data  # noqa: E402
import _oracle  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pyopal_amd.matrices import ScoringMatrix  # noqa: E402

ALGOS = ["nw", "hw", "ov", "sw"]


def decode(codes):
    return "".join(_oracle.NCBI[c] for c in codes)


def rle(ops):
    """uint8 ops -> e.g. '3M1I2X' over the reference's letters (lib.pyx:991)."""
    out = []
    i = 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append(f"{j - i}{'MDIX'[ops[i]]}")
        i = j
    return "".join(out)


def random_pairs():
    rng = np.random.default_rng(20240607)
    m62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
    m50 = np.array(ScoringMatrix.from_name("BLOSUM50").int_array(), dtype=np.int32)
    cases = []
    for k in range(200):
        qlen = int(np.exp(rng.uniform(0, np.log(400))))
        query = _data.random_protein(rng, qlen)
        if k % 3 == 0:
            target = _data.mutate(rng, query, rate=0.25)
            if len(target) == 0:
                target = _data.random_protein(rng, 1)
        else:
            target = _data.random_protein(rng, int(np.exp(rng.uniform(0, np.log(400)))))
        gap_open, gap_extend = [(3, 1), (11, 1), (5, 2), (1, 1), (2, 4)][k % 5]
        name, matrix = ("BLOSUM62", m62) if k % 2 == 0 else ("BLOSUM50", m50)
        res, off = _oracle.flatten([target])
        expected = {}
        for algo in ALGOS:
            r = _oracle.search(query, res, off, matrix, gap_open, gap_extend, "full", algo)
            expected[algo] = {
                "score": int(r["score"][0]),
                "end": [int(r["end_q"][0]), int(r["end_t"][0])],
                "start": [int(r["start_q"][0]), int(r["start_t"][0])],
                "ops": rle(r["aln"][0]),
            }
        cases.append({"query": decode(query), "target": decode(target), "matrix": name,
                      "gap_open": gap_open, "gap_extend": gap_extend, "expected": expected})
    return cases


def _checksum(res):
    return int(np.sum(res.astype(np.int64) * (np.arange(len(res)) % 251 + 1)))


def build_block(name):
    """Seeded inputs of one block of oracle_promotion.json -> (query, [targets])."""
    if name == "ladder":
        # target lengths of the reference's overflow test; the query is short enough for the
        # NW scores to fall below -32768 on the longest targets
        rng = np.random.default_rng(31)
        query = _data.random_protein(rng, 300)
        targets = []
        for idx, n in enumerate(range(1000, 36000, 1000)):
            if idx % 4 == 0:   # carries a noisy copy of the query: scores far above +127
                t = np.concatenate([_data.random_protein(rng, n // 2), _data.mutate(rng, query, 0.1),
                                    _data.random_protein(rng, n)])[:n]
            else:
                t = _data.random_protein(rng, n)
            targets.append(t)
        return query, targets
    if name == "int16_positive":
        # scores above +32767 need thousands of aligned residues
        rng = np.random.default_rng(32)
        query = _data.random_protein(rng, 8000)
        targets = [_data.mutate(rng, query, 0.05), _data.random_protein(rng, 3000),
                   np.concatenate([_data.random_protein(rng, 9000), _data.mutate(rng, query, 0.1),
                                   _data.random_protein(rng, 5000)])]
        return query, targets
    raise KeyError(name)


def promotion():
    m62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
    blocks = {}
    for name in ("ladder", "int16_positive"):
        # the sequences are regenerated from the seed by the tests (build_block): only the
        # expected numbers and a checksum of the inputs are stored
        query, targets = build_block(name)
        res, off = _oracle.flatten(targets)
        expected = {}
        for algo in ALGOS:
            r = _oracle.search(query, res, off, m62, 3, 1, "end", algo)
            expected[algo] = {"score": r["score"].tolist(), "end_q": r["end_q"].tolist(),
                              "end_t": r["end_t"].tolist()}
        blocks[name] = {"lengths": [len(t) for t in targets], "query_length": len(query),
                        "residue_checksum": _checksum(res), "expected": expected}
    return {"_comment": "generated by make_oracle_vectors.py from the CPU oracle (end mode)",
            "matrix": "BLOSUM62", "gap_open": 3, "gap_extend": 1, "blocks": blocks}


def promotion_inputs(name, block):
    """Rebuild the inputs of one block and check them against the fixture's checksum."""
    query, targets = build_block(name)
    res, off = _oracle.flatten(targets)
    assert _checksum(res) == block["residue_checksum"], "seeded inputs differ from the fixture's"
    return query, res, off


if __name__ == "__main__":
    with open(os.path.join(HERE, "oracle_random_pairs.json"), "w") as f:
        json.dump({"_comment": "generated by make_oracle_vectors.py from the CPU oracle; "
                               "ops are run-length encoded over MDIX (alignment letters)",
                   "cases": random_pairs()}, f, indent=0)
    with open(os.path.join(HERE, "oracle_promotion.json"), "w") as f:
        json.dump(promotion(), f)
    print("written")
