"""Scratch: wall time per search at the C ABI (host results) for every algorithm and search
type, for a few query lengths, on a uniform and a log-normal database."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
kinds = {"uniform300": np.full(N, 300),
         "lognormal": np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=N), 20, 8000).astype(np.int64)}
for kind, lengths in kinds.items():
    res, off = _data.random_db(rng, lengths)
    db = _capi.DeviceDatabase(res, off, 24)
    total = int(lengths.sum())
    for Q in (53, 150, 300):
        q = _data.random_protein(rng, Q)
        for algo in ("sw", "nw", "hw", "ov"):
            row = []
            for mode in ("score", "end", "full"):
                if mode == "full" and (Q > 64 or algo != "sw") and N > 200_000:
                    row.append("   -   "); continue
                for _ in range(2): out = db.search(q, m, 3, 1, mode, algo)
                t = time.perf_counter()
                for _ in range(3): out = db.search(q, m, 3, 1, mode, algo)
                dt = (time.perf_counter() - t) / 3
                row.append(f"{dt*1e3:7.2f}")
                del out
            print(f"{kind:10s} Q={Q:3d} {algo}: score/end/full ms = {' '.join(row)}   (score {Q*total/float(row[0])/1e6:.0f} GCUPS)", file=sys.stderr)
    db.close()
