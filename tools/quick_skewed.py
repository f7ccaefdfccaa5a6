"""Scratch: skewed-length databases (log-normal, the bench's extras.lognormal_lengths database; bimodal) at one query
length: wall time, TCUPS and routing per algorithm, over the multiples of the balanced share beyond which leading
groups leave the packed launch (MIOPAL_SKIP_SHARES). usage: quick_skewed.py [Q] [N] [dist]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 53
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
dist = sys.argv[3] if len(sys.argv) > 3 else "lognormal"
rng = np.random.default_rng(7)
if dist == "lognormal":
    lengths = np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
else:
    lengths = np.where(rng.random(n) < 0.1, 3000, 100)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
q = _data.README_QUERY if Q == 53 else None
q = _data.encode(q) if q else _data.random_protein(rng, Q)
cells = float(len(q)) * float(off[-1])
shares = [None] + [s for s in os.environ.get("QS_SHARES", "1.0,1.5,2.0,3.5,5.0,100").split(",") if s]
want = {}
for algo in os.environ.get("QS_ALGOS", "sw,nw,hw,ov").split(","):
    for sh in shares:
        _capi.set_tuning("SKIP_SHARES", sh)
        r = db.search(q, m, 3, 1, "score", algo)
        if algo not in want:
            want[algo] = r["score"].copy()
        assert np.array_equal(r["score"], want[algo]), (algo, sh)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); db.search(q, m, 3, 1, "score", algo); ts.append(time.perf_counter() - t)
        dt = min(ts)
        print(f"{dist} N={n} Q={len(q)} {algo} shares={sh or 'default':>7}: {dt*1e3:7.3f} ms {cells/dt/1e12:6.2f} TCUPS routing {_capi.DeviceDatabase.last_routing()}", flush=True)
_capi.set_tuning("SKIP_SHARES", None)
db.close()
