import os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(7)
n = 500_000
lengths = np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
q = _data.random_protein(rng, 150)
db.set_profiling(True)
for mode in ("score", "end"):
    for _ in range(3): db.search(q, m, 3, 1, mode, "nw")
    t = time.perf_counter()
    for _ in range(5): db.search(q, m, 3, 1, mode, "nw")
    nk, kms = db.last_kernel_time()
    print(mode, f"{(time.perf_counter()-t)/5*1e3:.2f} ms", f"kernel {kms/max(nk,1):.2f} ms x{nk}", _capi.DeviceDatabase.last_routing(), os.environ.get("MIOPAL_UNITS"), os.environ.get("MIOPAL_NO_UNSIGNED_DIAG"))
