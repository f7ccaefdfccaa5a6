"""Round 5: `full` searches with the packed (two pairs per lane) later passes against the 32-bit ones.

usage: quick_full_ab.py [N] [Q] [open] [ext] [algo]
Prints, per configuration of the switches, the median of five searches and whether the results of the two forms
are the same arrays (scores, ends, starts, operations).
"""
import os, sys, time
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
ALGO = sys.argv[5] if len(sys.argv) > 5 else "sw"
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(N, 300))
q = _data.encode(_data.README_QUERY) if Q == 53 else _data.random_protein(np.random.default_rng(4), Q)
db = _capi.DeviceDatabase(res, off, 24)


def run(label, **switches):
    with _capi.tuning(**switches):
        r = None
        for _ in range(2):
            r = db.search(q, m, GO, GE, "full", ALGO, reuse=r)
        ts = []
        for _ in range(int(os.environ.get("REPS", "9"))):
            t0 = time.perf_counter(); r = db.search(q, m, GO, GE, "full", ALGO, reuse=r); ts.append(time.perf_counter() - t0)
        routing = _capi.DeviceDatabase.last_full_routing()
    print(f"{label:28s} Q={Q} N={N} {GO}/{GE} {ALGO}: median {np.median(ts)*1e3:8.2f} ms  min {min(ts)*1e3:8.2f}  routing {routing}", flush=True)
    return {k: np.array(v, copy=True) for k, v in r.items() if isinstance(v, np.ndarray)}


only = os.environ.get("ONLY")
if only == "packed":
    run("packed"); sys.exit(0)
if only == "old":
    run("32-bit scan + directions", NO_PACKED_TRACE="1", NO_PACKED_SCAN="1"); sys.exit(0)
new = run("packed")
old = run("32-bit directions", NO_PACKED_TRACE="1")
old2 = run("32-bit scan + directions", NO_PACKED_TRACE="1", NO_PACKED_SCAN="1")
same = all(np.array_equal(new[k], old2[k]) for k in old2)
print("same results:", same, {k: bool(np.array_equal(new[k], old2[k])) for k in old2}, flush=True)
if os.environ.get("PHASES"):
    with _capi.tuning(PHASE_TIMING="1"):
        db.search(q, m, GO, GE, "full", ALGO)
sys.exit(0 if same else 1)
