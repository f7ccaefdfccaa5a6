"""Round 5: one switch on and off by turns in ONE process (box-to-box and run-to-run differences of `full` are larger
than most effects). usage: r05_switch_ab.py SWITCH[=VALUE] [N] [Q] [open] [ext] [algo]   (SWITCH without the MIOPAL_ prefix)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

SW, _, VALUE = sys.argv[1].partition("=")
VALUE = VALUE or "1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 53
GO = int(sys.argv[4]) if len(sys.argv) > 4 else 3
GE = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ALGO = sys.argv[6] if len(sys.argv) > 6 else "sw"
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
res, off = _data.random_db(np.random.default_rng(1), np.full(N, 300))
q = _data.encode(_data.README_QUERY) if Q == 53 else _data.random_protein(np.random.default_rng(4), Q)
db = _capi.DeviceDatabase(res, off, 24)
r = None
for _ in range(3):
    r = db.search(q, m, GO, GE, "full", ALGO, reuse=r)
meds = {"default": [], SW: []}
for turn in range(int(os.environ.get("TURNS", "6"))):
    for label, sw in (("default", {}), (SW, {SW: VALUE})):
        with _capi.tuning(**sw):
            r = db.search(q, m, GO, GE, "full", ALGO, reuse=r)
            ts = []
            for _ in range(int(os.environ.get("REPS", "9"))):
                t0 = time.perf_counter(); r = db.search(q, m, GO, GE, "full", ALGO, reuse=r); ts.append(time.perf_counter() - t0)
        meds[label].append(np.median(ts) * 1e3)
for k, v in meds.items():
    print(f"Q={Q} N={N} {GO}/{GE} {ALGO} {k:24s} medians by turn: " + " ".join(f"{x:.2f}" for x in v) + f"   median {np.median(v):.2f} ms", flush=True)
