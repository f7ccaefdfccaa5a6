"""Scratch timing: multi-strip Smith-Waterman scores, pair-table strips kernel against the general
kernel (MIOPAL_NO_PAIR_STRIPS=1), same process, same database. Usage: quick_strips.py [targets] [length]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
QS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [61, 100, 150, 300, 600]
MODE = os.environ.get("QS_MODE", "score")   # score | end
rng = np.random.default_rng(5)
lengths = np.full(N, L) if L > 0 else np.clip(rng.lognormal(5.5, 0.6, size=N).astype(int), 1, 6000)   # L = 0: log-normal
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)


def timed(q, reps=5):
    out = db.search(q, m, 3, 1, MODE, "sw")
    routed = _capi.DeviceDatabase.last_routing()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); db.search(q, m, 3, 1, MODE, "sw"); ts.append(time.perf_counter() - t)
    db.set_profiling(True)
    for _ in range(3):
        db.search(q, m, 3, 1, MODE, "sw")
    n, ms = db.last_kernel_time()
    db.set_profiling(False)
    return out, float(np.median(ts)), routed, ms / max(n, 1)


for Q in QS:
    q = _data.random_protein(rng, Q)
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIPS", None)
    a, ta, ra, ka = timed(q)
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIPS", "1")
    b, tb, rb, kb = timed(q)
    same = all(bool(np.array_equal(a[k], b[k])) for k in a)
    cells = float(Q) * float(off[-1])
    print(f"Q={Q:5d} {N}x{L}: strips {ta*1e3:8.2f} ms, kernel {ka:8.3f} ms {cells/ka/1e9:6.2f} TCUPS (code {ra[1]}, redone {ra[3]}) | "
          f"general {tb*1e3:8.2f} ms, kernel {kb:8.3f} ms {cells/kb/1e9:6.2f} TCUPS (code {rb[1]}) | equal {same}", flush=True)
    assert same
