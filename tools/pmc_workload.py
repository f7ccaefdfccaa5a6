"""One named workload, a few searches of it, nothing else: the command the PMC passes of
tools/collect_pmc.sh profile (the product path only; no CPU checker).

usage: pmc_workload.py WORKLOAD ALGO MODE [REPS]
  WORKLOAD  cfg4        2000-aa query vs 100k x 2000 (seed 2)
            cfg4tail    the same plus the reference's 35 long targets (1000 ... 35000 residues)
            qQ_NxL      Q-residue query vs N x L, e.g. q150_1000000x300, q300_200000x300 (seed 1)
  ALGO      sw | nw | hw | ov
  MODE      score | end | full
Prints wall ms per search, TCUPS and the routing of the last search to stderr.
"""
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data  # noqa: E402
from pyopal_amd import _capi  # noqa: E402
from pyopal_amd.matrices import ScoringMatrix  # noqa: E402

workload, algo, mode = sys.argv[1], sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
gap_open = int(os.environ.get("PW_OPEN", "3"))
gap_ext = int(os.environ.get("PW_EXT", "1"))
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
if workload.startswith("cfg4"):
    rng = np.random.default_rng(2)
    n_main = int(os.environ.get("PW_N", "100000"))
    lengths = np.full(n_main, 2000)
    if workload == "cfg4tail":
        lengths = np.concatenate([lengths, np.arange(1000, 35001, 1000)])
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 2000)
else:
    mt = re.fullmatch(r"q(\d+)_(\d+)x(\d+)", workload)
    if not mt:
        raise SystemExit(__doc__)
    Q, N, L = (int(x) for x in mt.groups())
    rng = np.random.default_rng(1)
    res, off = _data.random_db(rng, np.full(N, L))
    q = _data.random_protein(rng, Q)
db = _capi.DeviceDatabase(res, off, 24)
cells = float(len(q)) * float(off[-1])
db.search(q, m, gap_open, gap_ext, mode, algo)
t = time.perf_counter()
for _ in range(reps):
    db.search(q, m, gap_open, gap_ext, mode, algo)
dt = (time.perf_counter() - t) / reps
print(f"{workload} {algo} {mode}: {dt * 1e3:.2f} ms/search, {cells / dt / 1e12:.2f} TCUPS, cells={cells:.0f}, routing "
      f"{_capi.DeviceDatabase.last_routing()}", file=sys.stderr, flush=True)
db.close()
