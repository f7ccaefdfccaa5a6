"""Scratch: stability soak -- many searches of every kind on one handle, watching
device memory and results."""
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
lengths = np.clip(rng.lognormal(5.3, 0.7, size=60_000), 5, 12_000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
queries = [_data.random_protein(rng, n) for n in (53, 64, 65, 200, 700)]
ref = {}
for qi, q in enumerate(queries):
    for algo in ("sw", "nw", "hw", "ov"):
        ref[(qi, algo)] = db.search(q, m, 3, 1, "score", algo)["score"].copy()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
errors = []
def worker(tid, iters):
    r = np.random.default_rng(tid)
    for it in range(iters):
        qi = int(r.integers(0, len(queries))); algo = ("sw", "nw", "hw", "ov")[int(r.integers(0, 4))]
        mode = ("score", "score", "end", "full")[int(r.integers(0, 4))]
        lo = int(r.integers(0, 1000)) if r.random() < 0.2 else 0
        out = db.search(queries[qi], m, 3, 1, mode, algo, lo, None)
        if not np.array_equal(out["score"], ref[(qi, algo)][lo:]):
            errors.append((tid, it, qi, algo, mode))
for rnd in range(3):
    threads = [threading.Thread(target=worker, args=(t + 10 * rnd, 150)) for t in range(4)]
    [t.start() for t in threads]; [t.join() for t in threads]
    print(f"round {rnd}: device memory in use beyond warm-up {(free0 - torch.cuda.mem_get_info()[0])/2**20:.0f} MiB")
free1 = torch.cuda.mem_get_info()[0]
print(f"1800 mixed searches on 4 threads in {time.time()-t0:.1f} s; errors={len(errors)}; "
      f"device memory held after run: {(free0-free1)/2**20:.0f} MiB more than after warm-up; "
      f"mirror bytes {db.device_bytes()/2**20:.0f} MiB")
assert not errors, errors[:5]
released = db.release_workspaces()
free2 = torch.cuda.mem_get_info()[0]
print(f"miopalDbReleaseWorkspaces: {released/2**20:.0f} MiB of idle workspaces released; device memory held now: "
      f"{(free0-free2)/2**20:.0f} MiB more than after warm-up")
assert db.search(queries[0], m, 3, 1, "full", "sw")["score"].tolist() == ref[(0, "sw")].tolist()

