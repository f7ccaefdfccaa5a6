"""Scratch: API-level wall time on cfg2's database - list of result objects (`Aligner.align`,
the reference's contract) against the array extension (`Aligner.align_arrays`). The array
calls are timed first: releasing a million result objects leaves the allocator in a state
that slows the next large allocation down."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
import pyopal_amd as pyopal
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
letters = np.frombuffer(_data.AA20.encode(), dtype=np.uint8)
flat = letters[rng.integers(0, 20, size=N * 300)].tobytes().decode("ascii")
t = time.perf_counter()
db = pyopal.Database([flat[k * 300:(k + 1) * 300] for k in range(N)])
print(f"Database({N} x 300) built in {time.perf_counter() - t:.2f} s", file=sys.stderr)
aligner = pyopal.Aligner("BLOSUM62", gap_open=3, gap_extend=1)
t = time.perf_counter(); aligner.align(_data.README_QUERY, db, end=10); print(f"first call (mirror upload) {time.perf_counter() - t:.2f} s", file=sys.stderr)
for mode in ("score", "end", "full"):
    for rep in range(4):
        t = time.perf_counter(); a = aligner.align_arrays(_data.README_QUERY, db, mode=mode); t2 = time.perf_counter() - t
        del a
    print(f"{mode}: align_arrays {t2*1e3:.1f} ms", file=sys.stderr)
for mode in ("score", "end", "full"):
    for rep in range(3):
        t = time.perf_counter(); r = aligner.align(_data.README_QUERY, db, mode=mode); t1 = time.perf_counter() - t
        del r
    print(f"{mode}: align -> list of result objects {t1*1e3:.1f} ms", file=sys.stderr)
