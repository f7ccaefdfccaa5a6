"""Scratch timing of the SW score path (not the driver's bench.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 53
algo = sys.argv[4] if len(sys.argv) > 4 else "sw"
rng = np.random.default_rng(1)
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY) if Q == 53 else _data.random_protein(rng, Q)
res, off = _data.random_db(rng, np.full(N, L))
t0 = time.time(); db = _capi.DeviceDatabase(res, off, 24); t1 = time.time()
out = torch.zeros(N, dtype=torch.int32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
db.search_device_scores(q, m, out.data_ptr(), stream, 3, 1, algo); torch.cuda.synchronize(); t2 = time.time()
print(f"create {t1-t0:.3f}s first search (incl. view build) {t2-t1:.3f}s")
db.set_profiling(True)
for it in range(int(os.environ.get("QB_ROUNDS", "6"))):
    torch.cuda.synchronize(); t = time.time()
    K = 10
    for _ in range(K):
        db.search_device_scores(q, m, out.data_ptr(), stream, 3, 1, algo)
    torch.cuda.synchronize(); dt = (time.time() - t) / K
    n, ms = db.last_kernel_time()
    cells = Q * N * L
    print(f"wall {dt*1e3:.3f} ms/search -> {cells/dt/1e9:.1f} GCUPS; kernel {ms/max(n,1):.3f} ms x{n} -> {cells/(ms/max(n,1)*1e-3)/1e9:.1f} GCUPS")
print("checksum", int(out.sum().item()))
