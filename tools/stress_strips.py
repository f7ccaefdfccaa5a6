"""Repeats multi-strip Smith-Waterman score searches on the pair-table strips kernel and compares every
run with the general kernel's answer: the rows handed from strip to strip cross XCDs on relaxed
agent-scope atomics ordered by hand (interseq_impl.h), a lost ordering would show as a wrong score in
some run. Usage: stress_strips.py [iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 100
_capi.set_tuning("MIOPAL_PAIR_STRIPS", "1")   # also where the host would prefer the general kernel (few units)
rng = np.random.default_rng(17)
cases = [
    ("100k x 2000, Q=2000", np.full(100_000, 2000), 2000, max(1, ITER // 4), "score"),
    ("1M x 300, Q=300", np.full(1_000_000, 300), 300, ITER, "score"),
    ("1M x 300, Q=300, end locations", np.full(1_000_000, 300), 300, ITER, "end"),
    ("log-normal 500k, Q=150", np.clip(rng.lognormal(5.5, 0.6, size=500_000).astype(int), 1, 6000), 150, ITER, "score"),
    ("log-normal 500k, Q=150, end locations", np.clip(rng.lognormal(5.5, 0.6, size=500_000).astype(int), 1, 6000), 150, ITER, "end"),
    ("30k x 300, Q=640 (few batches, 13 strips)", np.full(30_000, 300), 640, ITER, "score"),
    # round 3: NW / HW / OV of several strips (interseq_pair_global_strips_kernel) against the general kernel
    ("100k x 2000, Q=2000, nw", np.full(100_000, 2000), 2000, max(1, ITER // 4), "score", "nw"),
    ("100k x 2000, Q=2000, ov end locations", np.full(100_000, 2000), 2000, max(1, ITER // 4), "end", "ov"),
    ("1M x 300, Q=300, hw", np.full(1_000_000, 300), 300, ITER, "score", "hw"),
    ("1M x 300, Q=150, nw end locations", np.full(1_000_000, 300), 150, ITER, "end", "nw"),
    ("log-normal 500k, Q=150, ov", np.clip(rng.lognormal(5.5, 0.6, size=500_000).astype(int), 1, 6000), 150, ITER, "score", "ov"),
    ("30k x 300, Q=640, hw (few batches, 13 strips)", np.full(30_000, 300), 640, ITER, "score", "hw"),
]
for case in cases:
    name, lengths, Q, iters, mode = case[:5]
    algo = case[5] if len(case) > 5 else "sw"
    off_switch = "MIOPAL_NO_PAIR_STRIPS" if algo == "sw" else "MIOPAL_NO_GLOBAL_STRIPS"
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, Q)
    db = _capi.DeviceDatabase(res, off, 24)
    _capi.set_tuning(off_switch, "1")
    want = db.search(q, m, 3, 1, mode, algo)
    assert (_capi.DeviceDatabase.last_routing()[1] & 15) == 1
    _capi.set_tuning(off_switch, None)
    bad = 0
    t0 = time.perf_counter()
    for k in range(iters):
        got = db.search(q, m, 3, 1, mode, algo)
        assert _capi.DeviceDatabase.last_routing()[1] == (6 if algo == "sw" else 7)
        if not all(np.array_equal(got[key], want[key]) for key in want):
            bad += 1
            print(f"  {name}: run {k}: {sum(int((got[key] != want[key]).sum()) for key in want)} values differ", flush=True)
    print(f"{name}: {iters} runs, {bad} with differences, {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per search", flush=True)
    db.close()
    assert bad == 0

# ---- the int32 kernel with a pair's strips side by side (intraseq_strips_kernel): the reference's 35 long
# targets against a 2000-residue query, every mode, against the strip-after-strip kernel's answer
_capi.set_tuning("MIOPAL_PAIR_STRIPS", None)
res, off = _data.random_db(rng, np.arange(1000, 35001, 1000))
q = _data.random_protein(rng, 2000)
db = _capi.DeviceDatabase(res, off, 24)
for algo in ("nw", "hw", "ov", "sw"):
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIP_UNITS", "1")
    want = db.search(q, m, 3, 1, "end", algo)
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIP_UNITS", None)
    bad = 0
    t0 = time.perf_counter()
    for k in range(ITER):
        got = db.search(q, m, 3, 1, "end", algo)
        if not all(np.array_equal(got[key], want[key]) for key in want):
            bad += 1
            print(f"  tail {algo}: run {k} differs", flush=True)
    print(f"35 tail targets x Q=2000, {algo} end: {ITER} runs, {bad} with differences, {(time.perf_counter() - t0) / ITER * 1e3:.2f} ms per search", flush=True)
    assert bad == 0
db.close()

# ---- the same strip units BESIDE the packed kernel (side stream), where a progress counter once overtook its
# rows: BASELINE configs[3] with its tail, every mode, against the strip-after-strip kernel's answer
lengths = np.concatenate([np.full(100_000, 2000), np.arange(1000, 35001, 1000)])
res, off = _data.random_db(rng, lengths)
q = _data.random_protein(rng, 2000)
db = _capi.DeviceDatabase(res, off, 24)
for algo in ("nw", "hw", "ov", "sw"):
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIP_UNITS", "1")
    want = db.search(q, m, 3, 1, "score", algo)["score"]
    _capi.set_tuning("MIOPAL_NO_PAIR_STRIP_UNITS", None)
    bad = 0
    for k in range(ITER):
        got = db.search(q, m, 3, 1, "score", algo)["score"]
        d = np.nonzero(got != want)[0]
        if len(d):
            bad += 1
            print(f"  cfg4 with tail, {algo}: run {k}: targets {d[:5]} (lengths {lengths[d[:5]]}) got {got[d[:5]]} want {want[d[:5]]}", flush=True)
    print(f"cfg4 with its tail, {algo} score: {ITER} runs, {bad} with differences", flush=True)
    assert bad == 0
db.close()
