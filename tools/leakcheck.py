"""Scratch: create / search / destroy databases in a loop and watch device and host memory."""
import os, sys, resource
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(3)
q = _data.random_protein(rng, 53)
q2 = _data.random_protein(rng, 150)
free0 = None
for it in range(40):
    lengths = np.clip(rng.lognormal(5.3, 0.6, size=50_000), 5, 6000).astype(np.int64)
    res, off = _data.random_db(rng, lengths)
    db = _capi.DeviceDatabase(res, off, 24)
    for mode in ("score", "end", "full"):
        db.search(q, m, 3, 1, mode, "sw")
        db.search(q2, m, 3, 1, mode, "hw", 100, 20_000)
    db.close()
    free = torch.cuda.mem_get_info()[0]
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
    if it == 4: free0 = free
    if it % 5 == 4:
        print(f"iteration {it}: device memory in use vs iteration 4: {((free0 or free) - free)/2**20:.0f} MiB, max RSS {rss:.0f} MiB", file=sys.stderr)
