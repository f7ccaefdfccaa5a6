import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
import pyopal_amd as pyopal
N = 1_000_000
rng = np.random.default_rng(1)
letters = np.frombuffer(_data.AA20.encode(), dtype=np.uint8)
flat = letters[rng.integers(0, 20, size=N * 300)].tobytes().decode("ascii")
db = pyopal.Database([flat[k * 300:(k + 1) * 300] for k in range(N)])
aligner = pyopal.Aligner("BLOSUM62", gap_open=3, gap_extend=1)
aligner.align(_data.README_QUERY, db, end=10)
for rep in range(4):
    t = time.perf_counter(); a = aligner.align_arrays(_data.README_QUERY, db, mode="full"); t2 = time.perf_counter() - t
    print(f"=== align_arrays full {t2*1e3:.1f} ms", file=sys.stderr)
    del a
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
a = aligner.align_arrays(_data.README_QUERY, db, mode="full")
pr.disable()
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(12)
