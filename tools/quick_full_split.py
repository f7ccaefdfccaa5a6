"""Scratch: where a `full` search's wall time goes at the Python level - the call itself, and the release of the
previous result - with the operations crossing PCIe packed (two bits each) or as bytes. Same process, same database."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
Q, N, L = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (53, 1_000_000, 300)))
rng = np.random.default_rng(1)
res, off = _data.random_db(rng, np.full(N, L))
q = _data.random_protein(rng, Q)
db = _capi.DeviceDatabase(res, off, 24)
ref = None
for label, switches in (("packed", {}), ("bytes", {"NO_PACKED_OPS": "1"}), ("packed", {}), ("bytes", {"NO_PACKED_OPS": "1"})):
    with _capi.tuning(**switches):
        r = db.search(q, m, 3, 1, "full", "sw")
        if ref is None:
            ref = (r["aln_flat"].copy(), r["aln_off"].copy())
        assert np.array_equal(r["aln_flat"], ref[0]) and np.array_equal(r["aln_off"], ref[1])
        calls, frees, reused = [], [], []
        for _ in range(8):
            t0 = time.perf_counter(); del r; t1 = time.perf_counter()
            r = db.search(q, m, 3, 1, "full", "sw"); t2 = time.perf_counter()
            frees.append(t1 - t0); calls.append(t2 - t1)
        for _ in range(8):
            t1 = time.perf_counter(); r2 = db.search(q, m, 3, 1, "full", "sw", reuse=r); t2 = time.perf_counter()
            reused.append(t2 - t1); r = r2; del r2
        print(f"{label:7s} Q={Q} {N}x{L}: call {np.median(calls)*1e3:7.2f} ms (min {min(calls)*1e3:.2f}), release of the previous result "
              f"{np.median(frees)*1e3:6.2f} ms, call with reuse= {np.median(reused)*1e3:7.2f} ms", flush=True)
db.close()
