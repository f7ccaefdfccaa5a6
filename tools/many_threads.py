"""Scratch: many host threads, each with its own slice, full mode - parked workspace memory."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(5)
T = 32
n = 320_000
lengths = np.clip(rng.lognormal(5.3, 0.7, size=n), 5, 6000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
q = _data.random_protein(rng, 60)
ref = db.search(q, m, 3, 1, "score", "nw")["score"].copy()
free0 = torch.cuda.mem_get_info()[0]
errors = []
def worker(t):
    lo, hi = t * (n // T), (t + 1) * (n // T)
    for it in range(3):
        out = db.search(q, m, 3, 1, "full", "nw", lo, hi)
        if not np.array_equal(out["score"], ref[lo:hi]): errors.append(t)
t0 = time.time()
threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[t.start() for t in threads]; [t.join() for t in threads]
held = (free0 - torch.cuda.mem_get_info()[0]) / 2**30
print(f"{T} threads x 3 global full searches of their slice: {time.time()-t0:.1f} s, errors={len(errors)}, device memory held {held:.1f} GiB", file=sys.stderr)
assert not errors
