"""Scratch: the strips kernels forced (MIOPAL_PAIR_STRIPS=1) against the host's own choice, on the
searches where the routing table shows the general kernel: wall ms with host results."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)

def lengths_of(dist, n, rng):
    if dist == "uniform300":
        return np.full(n, 300)
    if dist == "lognormal":
        return np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
    return np.where(rng.random(n) < 0.1, 3000, 100)

def best_of(db, q, mode, algo):
    db.search(q, m, 3, 1, mode, algo)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); db.search(q, m, 3, 1, mode, algo); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, _capi.DeviceDatabase.last_routing()

cases = [("uniform300", 20_000), ("uniform300", 50_000), ("uniform300", 100_000), ("uniform300", 200_000),
         ("lognormal", 100_000), ("lognormal", 500_000), ("lognormal", 2_000_000), ("bimodal100_3000", 500_000)]
for dist, n in cases:
    rng = np.random.default_rng(1000 + n)
    res, off = _data.random_db(rng, lengths_of(dist, n, rng))
    db = _capi.DeviceDatabase(res, off, 24)
    for qlen in (65, 150, 300, 1000):
        q = _data.random_protein(np.random.default_rng(qlen), qlen)
        for algo in ("sw", "nw"):
            _capi.set_tuning("MIOPAL_PAIR_STRIPS", None)
            a, ra = best_of(db, q, "score", algo)
            _capi.set_tuning("MIOPAL_PAIR_STRIPS", "1")
            b, rb = best_of(db, q, "score", algo)
            _capi.set_tuning("MIOPAL_PAIR_STRIPS", None)
            print(f"{dist:16s} N={n:8d} Q={qlen:5d} {algo}: default {a:8.3f} ms (code {ra[1]}, side {ra[0]}) | strips forced {b:8.3f} ms (code {rb[1]}, side {rb[0]}) {'<-- strips better' if b < 0.95 * a else ''}", flush=True)
    db.close()
