"""Scratch: configs[4] unsharded on one GPU (10M x 400, SW score) - steady-state GCUPS."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY)
n, length = 10_000_000, 400
rng = np.random.default_rng(3)
t = time.perf_counter()
res = _data.AA20_CODES[rng.integers(0, 20, size=n * length, dtype=np.uint8)]
off = np.arange(n + 1, dtype=np.int64) * length
print(f"generated in {time.perf_counter() - t:.1f} s", file=sys.stderr)
t = time.perf_counter(); db = _capi.DeviceDatabase(res, off, 24); print(f"upload {time.perf_counter() - t:.2f} s", file=sys.stderr)
db.set_profiling(True)
for rep in range(5):
    t = time.perf_counter(); out = db.search(q, m, 3, 1, "score", "sw"); dt = time.perf_counter() - t
    nk, ms = db.last_kernel_time()
    print(f"score #{rep}: {dt*1e3:.1f} ms  {len(q)*n*length/dt/1e9:.0f} GCUPS  (kernel {ms:.2f} ms, {len(q)*n*length/ms/1e6:.0f} GCUPS)", file=sys.stderr)
