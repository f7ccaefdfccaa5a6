"""cfg4 as BASELINE.json has it: 2000-aa query vs 100k x 2000 PLUS the 35 long targets of the reference's
overflow test (1000 ... 35000 residues): wall ms per score search, every mode."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(2)
n_main = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
lengths = np.concatenate([np.full(n_main, 2000), np.arange(1000, 35001, 1000)])
res, off = _data.random_db(rng, lengths)
q = _data.random_protein(rng, 2000)
db = _capi.DeviceDatabase(res, off, 24)
cells = 2000.0 * float(off[-1])
for algo in ("sw", "nw", "hw", "ov"):
    for mode in ("score", "end"):
        db.search(q, m, 3, 1, mode, algo)
        t = time.perf_counter()
        for _ in range(2): db.search(q, m, 3, 1, mode, algo)
        dt = (time.perf_counter() - t) / 2
        print(f"{algo} {mode}: {dt*1e3:8.1f} ms -> {cells/dt/1e9:7.0f} GCUPS  routing {_capi.DeviceDatabase.last_routing()}", flush=True)
