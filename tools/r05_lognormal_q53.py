"""Round 5: log-normal target lengths (500k, clipped at 8000) at Q = 53, scores of every algorithm: ms, TCUPS, routing.
With TRACE=1 in the environment the caller wraps it in rocprofv3 (tools/r05_lognormal_trace.sh)."""
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
n = 500_000
lengths = np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
cells = 53.0 * float(lengths.sum())
for algo in (sys.argv[1:] or ["sw", "nw", "hw", "ov"]):
    for _ in range(3): db.search(q, m, 3, 1, "score", algo)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); db.search(q, m, 3, 1, "score", algo); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print(f"{algo}: {dt*1e3:.3f} ms -> {cells/dt/1e12:.2f} TCUPS, routing {_capi.DeviceDatabase.last_routing()}", flush=True)
