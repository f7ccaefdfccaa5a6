import os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
def lengths_of(dist, n, rng):
    if dist == "lognormal":
        return np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
    return np.where(rng.random(n) < 0.1, 3000, 100)
for dist, n in (("bimodal", 500_000), ("bimodal", 2_000_000), ("lognormal", 500_000), ("lognormal", 2_000_000)):
    rng = np.random.default_rng(1000 + n)
    res, off = _data.random_db(rng, lengths_of(dist, n, rng))
    db = _capi.DeviceDatabase(res, off, 24)
    for qlen in (20, 53, 150):
        q = _data.random_protein(np.random.default_rng(qlen), qlen)
        for algo, mode in (("sw", "score"), ("sw", "end"), ("hw", "score")):
            out = []
            for env in ("MIOPAL_SHORT_STRIDE", None):
                _capi.set_tuning("MIOPAL_SHORT_STRIDE", None)
                if env: _capi.set_tuning(env, "1")
                r = db.search(q, m, 3, 1, mode, algo); ts = []
                for _ in range(3):
                    t = time.perf_counter(); r = db.search(q, m, 3, 1, mode, algo); ts.append(time.perf_counter() - t)
                out.append((min(ts) * 1e3, r, _capi.DeviceDatabase.last_routing()))
            _capi.set_tuning("MIOPAL_SHORT_STRIDE", None)
            same = all(np.array_equal(out[0][1][k], out[1][1][k]) for k in out[0][1])
            print(f"{dist:10s} N={n:8d} Q={qlen:4d} {algo} {mode:5s}: short stride {out[0][0]:8.3f} ms {out[0][2]} | adaptive {out[1][0]:8.3f} ms {out[1][2]} | equal {same}", flush=True)
            assert same
    db.close()
