"""Routing regression table: TCUPS and the routing of the score pass over
    targets N  x  query length Q  x  length distribution  x  algorithm  x  search type,
and every cell whose throughput is below 0.6 x the better of its neighbours along N or Q (same
distribution, algorithm and search type) flagged as a cliff. Times the product path only (wall time
of miopalSearch with host results, best of 3 after a warm-up; no CPU checker).

usage: routing_table.py OUT.txt [quick]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data  # noqa: E402
from pyopal_amd import _capi  # noqa: E402
from pyopal_amd.matrices import ScoringMatrix  # noqa: E402

out_path = sys.argv[1]
quick = len(sys.argv) > 2
NS = [4, 1000, 20_000, 100_000, 500_000, 2_000_000]
QS = [20, 53, 64, 65, 150, 300, 1000, 2000]
if quick:
    NS, QS = [1000, 100_000], [53, 300]
DISTS = ("uniform300", "lognormal", "bimodal100_3000")
ALGOS = ("sw", "nw", "hw", "ov")
MODES = ("score", "end")
# (a part of the table: RT_NS=20000,100000 RT_QS=64,150 RT_DISTS=lognormal RT_ALGOS=sw,hw RT_MODES=score)
if os.environ.get("RT_NS"):
    NS = [int(x) for x in os.environ["RT_NS"].split(",")]
if os.environ.get("RT_QS"):
    QS = [int(x) for x in os.environ["RT_QS"].split(",")]
if os.environ.get("RT_DISTS"):
    DISTS = tuple(os.environ["RT_DISTS"].split(","))
if os.environ.get("RT_ALGOS"):
    ALGOS = tuple(os.environ["RT_ALGOS"].split(","))
if os.environ.get("RT_MODES"):
    MODES = tuple(os.environ["RT_MODES"].split(","))
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)


def lengths_of(dist, n, rng):
    if dist == "uniform300":
        return np.full(n, 300)
    if dist == "lognormal":
        return np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
    return np.where(rng.random(n) < 0.1, 3000, 100)   # a tenth of the targets thirty times as long as the rest


rows = {}   # (dist, algo, mode, N, Q) -> (tcups, ms, routing)
log = open(out_path, "w")


def say(line):
    print(line, flush=True)
    log.write(line + "\n")
    log.flush()


say("# dist algo mode N Q ms TCUPS routing(int32 targets, kernel code, groups, redone)")
for dist in DISTS:
    for n in NS:
        rng = np.random.default_rng(1000 + n)
        lengths = lengths_of(dist, n, rng)
        res, off = _data.random_db(rng, lengths)
        db = _capi.DeviceDatabase(res, off, 24)
        total = float(off[-1])
        for qlen in QS:
            q = _data.random_protein(np.random.default_rng(qlen), qlen)
            cells = qlen * total
            for algo in ALGOS:
                for mode in MODES:
                    # (a cell that would take more than ~2 s per search says enough after one)
                    t0 = time.perf_counter()
                    db.search(q, m, 3, 1, mode, algo)
                    first = time.perf_counter() - t0
                    best = first
                    for _ in range(0 if first > 2.0 else 3):
                        t0 = time.perf_counter()
                        db.search(q, m, 3, 1, mode, algo)
                        best = min(best, time.perf_counter() - t0)
                    routing = _capi.DeviceDatabase.last_routing()
                    rows[(dist, algo, mode, n, qlen)] = (cells / best / 1e12, best * 1e3, routing)
                    say(f"{dist:16s} {algo} {mode:5s} {n:8d} {qlen:5d} {best * 1e3:10.3f} {cells / best / 1e12:8.3f} {routing}")
        db.close()
        del res, off

say("")
say("# cliffs: cells below 0.6 x the better of their neighbours along N or Q (small searches are latency-bound: only")
say("# cells of at least 100k targets, or whose smaller neighbour is faster, count)")
flagged = 0
for (dist, algo, mode, n, qlen), (tc, ms, routing) in sorted(rows.items()):
    ni, qi = NS.index(n), QS.index(qlen)
    neighbours = []
    for dn, dq in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        a, b = ni + dn, qi + dq
        if 0 <= a < len(NS) and 0 <= b < len(QS):
            other = rows.get((dist, algo, mode, NS[a], QS[b]))
            if other:
                # throughput grows with the size of the search: a larger neighbour being faster is no cliff
                if (dn > 0 or dq > 0) and n < 100_000:
                    continue
                neighbours.append((other[0], NS[a], QS[b]))
    if not neighbours:
        continue
    top = max(neighbours)
    if tc < 0.6 * top[0]:
        flagged += 1
        say(f"CLIFF {dist:16s} {algo} {mode:5s} N={n:8d} Q={qlen:5d}: {tc:7.3f} TCUPS ({ms:.3f} ms, routing {routing}) against "
            f"{top[0]:7.3f} at N={top[1]} Q={top[2]}")
say(f"# {flagged} cliffs among {len(rows)} cells")
