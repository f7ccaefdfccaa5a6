"""Scratch: SW score on log-normal target lengths (lane-packing / balance check)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY)
rng = np.random.default_rng(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
lengths = np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
for _ in range(3): db.search(q, m, 3, 1, "score", "sw")
t0 = time.perf_counter()
for _ in range(5):
    out = db.search(q, m, 3, 1, "score", "sw")
dt = (time.perf_counter() - t0) / 5
print(f"{dt*1e3:.3f} ms -> {53*lengths.sum()/dt/1e9:.1f} GCUPS; lengths>780: {(lengths>780).sum()}, >975: {(lengths>975).sum()}")
