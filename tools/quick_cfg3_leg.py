"""Scratch: the cfg3 leg of bench.py (README query, full, 1M x 300) call by call: results held (as bench.py does), results
released between calls (outside the timed call), with and without reuse= of the per-target arrays."""
import os, resource, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data
if os.environ.get("WITH_TORCH"):
    import torch
    _keep = torch.zeros(1000000, dtype=torch.int32, device="cuda:0"); _pin = torch.empty(1000000, dtype=torch.int32).pin_memory()
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
def where():
    import ctypes, glob
    cpu = ctypes.CDLL(None).sched_getcpu()
    node = [os.path.basename(p) for p in glob.glob(f"/sys/devices/system/cpu/cpu{cpu}/node*")]
    return f"cpu {cpu} {node}"
rng = np.random.default_rng(1)
res, off = _data.random_db(rng, np.full(int(os.environ.get("N", 1_000_000)), 300))
q = _data.encode(_data.README_QUERY)
if os.environ.get("RANDOMQ"):
    q = _data.random_protein(np.random.default_rng(int(os.environ["RANDOMQ"])), 53)
db = _capi.DeviceDatabase(res, off, 24)
print(f"database created on {where()}", flush=True)
SUSTAIN = os.environ.get("SUSTAIN", "")   # (what bench.py runs before its cfg3 leg: "s" six seconds of the headline search, "d" another database, "q" a longer query)
if "s" in SUSTAIN:
    t_end = time.perf_counter() + 6.0
    while time.perf_counter() < t_end:
        db.search(q, m, 3, 1, "score", "sw")
if "S" in SUSTAIN:
    for _ in range(300):
        db.search(q, m, 3, 1, "score", "sw")
if "d" in SUSTAIN or "k" in SUSTAIN or "c" in SUSTAIN:
    lengths = np.concatenate([np.full(100_000, 2000), np.arange(1000, 35001, 1000)])
    r2, o2 = _data.random_db(np.random.default_rng(2), lengths)
    cdb = _capi.DeviceDatabase(r2, o2, 24)
    ql = _data.random_protein(np.random.default_rng(3), 2000)
    if "c" not in SUSTAIN:
        for algo in ("nw", "sw"):
            cdb.search(ql, m, 3, 1, "score", algo)
    if "k" not in SUSTAIN:
        cdb.close()
if "r" in SUSTAIN:
    db.release_workspaces()
if "q" in SUSTAIN:
    qq = _data.random_protein(np.random.default_rng(4), 300)
    for mode in ("score", "end"):
        for _ in range(3):
            db.search(qq, m, 3, 1, mode, "sw")
if "F" in SUSTAIN:   # (bench.py's Q = 300 `full` leg)
    qq = _data.random_protein(np.random.default_rng(4), 300)
    rr = None
    for _ in range(6):
        rr = db.search(qq, m, 3, 1, "full", "sw", reuse=rr)
    del rr
    for _ in range(7):
        db.search(q, m, 3, 1, "end", "sw")
print(f"before the full searches: {where()}, affinity {len(os.sched_getaffinity(0))} cpus", flush=True)
for _ in range(2):
    db.search(q, m, 3, 1, "full", "sw")
if os.environ.get("PHASES"):   # (the library's phase timers of three calls, nothing else)
    r = db.search(q, m, 3, 1, "full", "sw")
    with _capi.tuning(PHASE_TIMING="1"):
        for _ in range(3):
            r = db.search(q, m, 3, 1, "full", "sw", reuse=r)
    db.close()
    sys.exit(0)
if os.environ.get("AB_SWITCH"):   # (a tuning switch on and off by turns, same process: medians of 8 calls each)
    name = os.environ["AB_SWITCH"]
    ab_mode = os.environ.get("AB_MODE", "full")
    if ab_mode != "full":   # (score / end searches into re-used, pageable arrays)
        out = np.empty(len(off) - 1, dtype=np.int32)
        for turn in range(6):
            on = turn % 2 == 1
            _capi.set_tuning(name, "1" if on else None)
            r = db.search(q, m, 3, 1, ab_mode, "sw", score_out=out)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); r = db.search(q, m, 3, 1, ab_mode, "sw", score_out=out, reuse=r); ts.append(time.perf_counter() - t0)
            print(f"{ab_mode}: {name} {'set  ' if on else 'unset'}: median {np.median(ts)*1e3:.3f} ms  min {min(ts)*1e3:.3f}", flush=True)
        _capi.set_tuning(name, None)
        db.close()
        sys.exit(0)
    r = db.search(q, m, 3, 1, "full", "sw")
    for turn in range(6):
        on = turn % 2 == 1
        _capi.set_tuning(name, "1" if on else None)
        r = db.search(q, m, 3, 1, "full", "sw", reuse=r)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); r = db.search(q, m, 3, 1, "full", "sw", reuse=r); ts.append(time.perf_counter() - t0)
        print(f"{name} {'set  ' if on else 'unset'}: median {np.median(ts)*1e3:.2f} ms  min {min(ts)*1e3:.2f}", flush=True)
    _capi.set_tuning(name, None)
    db.close()
    sys.exit(0)
for label, hold, reuse in (("held, reuse", True, True), ("released between calls, reuse", False, True), ("released between calls, fresh arrays", False, False),
                           ("held, reuse", True, True)):
    r = db.search(q, m, 3, 1, "full", "sw")
    held, ts = [], []
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(8):
        t0 = time.perf_counter()
        r2 = db.search(q, m, 3, 1, "full", "sw", reuse=r if reuse else None)
        ts.append(time.perf_counter() - t0)
        if hold:
            held.append(r2)
        r = r2
        del r2
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    print(f"    now on {where()}", flush=True)
    print(f"{label:40s}: " + " ".join(f"{t*1e3:6.2f}" for t in ts) + f"  | median {np.median(ts)*1e3:.2f} ms, ops {int(r['aln_off'][-1])}"
          f" | per call: {(ru1.ru_minflt - ru0.ru_minflt) / 8:.0f} minor faults, {(ru1.ru_stime - ru0.ru_stime) / 8 * 1e3:.2f} ms system, "
          f"{(ru1.ru_utime - ru0.ru_utime) / 8 * 1e3:.2f} ms user", flush=True)
    del held
db.close()
