"""Round 5: how many of the cells a direction wavefront sweeps belong to its pairs' [start..end] rectangles?

usage: r05_sweep_efficiency.py [N] [Q] [open] [ext]
A wavefront of perpair_packed_trace_kernel holds 128 pairs of the sorted job list and sweeps, for all of them, the
rows of its tallest window (in groups of eight) times the columns of its longest. Prints useful / swept cells for the
list sorted by window length (the build), and by (length, rows / 8) and (rows / 8, length).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 53
GO = int(sys.argv[3]) if len(sys.argv) > 3 else 3
GE = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(N, 300))
q = _data.encode(_data.README_QUERY) if Q == 53 else _data.random_protein(np.random.default_rng(4), Q)
db = _capi.DeviceDatabase(res, off, 24)
r = db.search(q, m, GO, GE, "full", "sw")
cols = (r["end_t"].astype(np.int64) - r["start_t"] + 1)
rows = (r["end_q"].astype(np.int64) - r["start_q"] + 1)
live = (r["end_t"] >= 0) & (r["end_q"] >= 0)
cols[~live] = 0; rows[~live] = 0
useful = float((cols * rows).sum())
print(f"Q={Q} N={N} {GO}/{GE}: mean window {cols.mean():.1f} columns x {rows.mean():.1f} rows, useful cells {useful:.3e}")
scan_rows = r["end_q"].astype(np.int64) + 1
scan_rows[~live] = 0
print(f"start-cell scan: cells between the end cell and the start cell's column, all rows of the prefix: "
      f"{float((cols * scan_rows).sum()):.3e} (mean {scan_rows.mean():.1f} rows); whole prefixes "
      f"{float(((r['end_t'].astype(np.int64) + 1) * scan_rows)[live].sum()):.3e}")
print("rows / 8 histogram:", np.bincount((rows + 7) // 8)[:40])


def swept(order_key, batches=4, per_wave=128):
    total = 0.0
    nb = -(-N // batches)
    for b in range(batches):
        sl = slice(b * nb, min(N, (b + 1) * nb))
        c, rw = cols[sl], rows[sl]
        o = np.argsort(-order_key(c, rw), kind="stable")
        c, rw = c[o], rw[o]
        pad = (-len(c)) % per_wave
        c = np.concatenate([c, np.zeros(pad, np.int64)]).reshape(-1, per_wave)
        rw = np.concatenate([rw, np.zeros(pad, np.int64)]).reshape(-1, per_wave)
        total += float((per_wave * ((rw.max(1) + 7) // 8 * 8) * ((c.max(1) + 3) // 4 * 4)).sum())
    return total


for name, key in (("by length (the build)", lambda c, rw: c),
                  ("by (length, rows / 8)", lambda c, rw: c * 64 + (rw + 7) // 8),
                  ("by (length / 4, rows / 8)", lambda c, rw: (c >> 2) * 64 + (rw + 7) // 8),
                  ("by (length / 8, rows / 8)", lambda c, rw: (c >> 3) * 64 + (rw + 7) // 8),
                  ("by (length / 16, rows / 8)", lambda c, rw: (c >> 4) * 64 + (rw + 7) // 8),
                  ("by (length / 32, rows / 8)", lambda c, rw: (c >> 5) * 64 + (rw + 7) // 8),
                  ("by (rows / 8, length)", lambda c, rw: ((rw + 7) // 8) * 100000 + c),
                  ("by swept area of the pair itself", lambda c, rw: ((rw + 7) // 8) * c)):
    s = swept(key)
    print(f"  {name:36s} swept {s:.3e}  useful / swept {useful / s:.3f}")
