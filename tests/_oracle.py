"""ctypes binding to the CPU checker in oracle/ (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

SEARCH = {"score": 0, "end": 1, "full": 2}
MODE = {"nw": 0, "hw": 1, "ov": 2, "sw": 3}
from _data import NCBI  # noqa: E402


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "libopal_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.oracleSearchFlat.restype = ctypes.c_int
    return _LIB


def encode(seq, alphabet=NCBI):
    table = {c: i for i, c in enumerate(alphabet)}
    return np.array([table[c] for c in seq], dtype=np.uint8)


def flatten(seqs):
    """list of uint8 arrays -> (residues, offsets[int64])"""
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    res = np.concatenate(seqs) if len(seqs) and off[-1] > 0 else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(res, dtype=np.uint8), off


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def search(query, residues, offsets, matrix, gap_open=3, gap_extend=1, mode="score", algorithm="sw"):
    """Run the oracle. Returns dict of numpy arrays (score, end_t, end_q,
    start_t, start_q, aln (list of uint8 arrays))."""
    n = len(offsets) - 1
    A = int(round(len(matrix) ** 0.5))
    S = np.ascontiguousarray(matrix, dtype=np.int32)
    q = np.ascontiguousarray(query, dtype=np.uint8)
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    st = SEARCH[mode]
    out = {"score": np.zeros(n, dtype=np.int32)}
    et = eq = s_t = s_q = aln = aoff = None
    cap = 0
    if st >= 1:
        et = np.full(n, -1, dtype=np.int32)
        eq = np.full(n, -1, dtype=np.int32)
    if st == 2:
        s_t = np.full(n, -1, dtype=np.int32)
        s_q = np.full(n, -1, dtype=np.int32)
        cap = int(offsets[-1]) + n * len(q) + 1
        aln = np.zeros(cap, dtype=np.uint8)
        aoff = np.zeros(n + 1, dtype=np.int64)
    rc = lib().oracleSearchFlat(
        _p(q, ctypes.c_ubyte), ctypes.c_int(len(q)), _p(residues, ctypes.c_ubyte),
        _p(offsets, ctypes.c_int64), ctypes.c_int(n), ctypes.c_int(gap_open),
        ctypes.c_int(gap_extend), _p(S, ctypes.c_int), ctypes.c_int(A), ctypes.c_int(st),
        ctypes.c_int(MODE[algorithm]), _p(out["score"], ctypes.c_int), _p(et, ctypes.c_int),
        _p(eq, ctypes.c_int), _p(s_t, ctypes.c_int), _p(s_q, ctypes.c_int),
        _p(aln, ctypes.c_ubyte), ctypes.c_int64(cap), _p(aoff, ctypes.c_int64))
    if rc != 0:
        raise RuntimeError(f"oracle failed with code {rc}")
    if st >= 1:
        out.update(end_t=et, end_q=eq)
    if st == 2:
        out.update(start_t=s_t, start_q=s_q,
                   aln=[aln[aoff[k]:aoff[k + 1]].copy() for k in range(n)])
    return out


def search_parallel(query, residues, offsets, matrix, gap_open=3, gap_extend=1, mode="score", algorithm="sw",
                    threads=8, chunk=4096):
    """search() over chunks of targets on a thread pool (the C call releases the GIL and the
    checker keeps no global state): the scalar checker on a million targets in seconds.
    Same keys as search(), plus aln_flat / aln_off (alignments concatenated) in "full" mode."""
    from concurrent.futures import ThreadPoolExecutor
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    bounds = list(range(0, n, chunk)) + [n]

    def run(k):
        lo, hi = bounds[k], bounds[k + 1]
        return search(query, residues[offsets[lo]:offsets[hi]], offsets[lo:hi + 1] - offsets[lo], matrix,
                      gap_open, gap_extend, mode, algorithm)

    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(run, range(len(bounds) - 1)))
    out = {}
    for key in parts[0] if parts else ():
        if key == "aln":
            continue
        out[key] = np.concatenate([p[key] for p in parts])
    if parts and "aln" in parts[0]:
        lens = np.concatenate([[len(a) for a in p["aln"]] for p in parts]).astype(np.int64)
        out["aln_off"] = np.concatenate([[0], np.cumsum(lens)])
        out["aln_flat"] = np.concatenate([np.concatenate(p["aln"]) if len(p["aln"]) else np.zeros(0, np.uint8)
                                          for p in parts]).astype(np.uint8)
    return out
