"""Round 5: the persistent start-cell scan (perpair_packed_scan_kernel) hands its pairs out in database order; every
half of a lane takes the next pair when >= 24 halves of its wavefront idle. How long is a wavefront's life in columns,
against the balanced share - and what would an order by score (known after the end pass; the window's length is not)
or by window length (the ideal) give?      usage: r05_scan_schedule_sim.py [N] [Q] [open] [ext]"""
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
live = (r["end_t"] >= 0) & (r["end_q"] >= 0)
cols = np.where(live, r["end_t"].astype(np.int64) - r["start_t"] + 1, 0)
score = r["scores"].astype(np.int64) if "scores" in r else r["score"].astype(np.int64)


def simulate(lengths, waves=3072, halves=128, refill_at=24):
    """columns until the last wavefront is done; mean busy share of the halves over the wavefronts' lives"""
    n = len(lengths)
    left = np.zeros((waves, halves), np.int64)      # columns the half still has to sweep
    nxt = 0
    t = 0
    busy_cols = 0
    alive = np.ones(waves, bool)
    life = np.zeros(waves, np.int64)
    while alive.any():
        idle = (left <= 0)
        nidle = idle.sum(1)
        want = alive & (nidle > 0) & ((nidle >= refill_at) | (t == 0)) & (nxt < n)
        for w in np.nonzero(want)[0]:                # the counter: in order of the wavefronts
            if nxt >= n: break
            k = min(int(nidle[w]), n - nxt)
            slots = np.nonzero(idle[w])[0][:k]
            left[w, slots] = lengths[nxt:nxt + k]
            nxt += k
        done = alive & (nxt >= n) & ((left <= 0).all(1))
        life[done] = t
        alive &= ~done
        step = 4
        busy_cols += int(np.minimum(np.maximum(left[alive], 0), step).sum())
        left[alive] -= step
        t += step
    return t, life.mean(), busy_cols / (life.sum() * halves)


total = cols.sum()
print(f"Q={Q} N={N} {GO}/{GE}: {total:.3e} columns, balanced share {total / (3072 * 128):.1f} columns per half")
for name, order in (("database order (the build)", np.arange(N)),
                    ("by score, highest first", np.argsort(-score, kind="stable")),
                    ("by the end cell's column, longest prefixes first", np.argsort(-r["end_t"].astype(np.int64), kind="stable")),
                    ("by the end cell's column in steps of 32, longest first", np.argsort(-(r["end_t"].astype(np.int64) >> 5), kind="stable")),
                    ("by the end cell's row, last rows first", np.argsort(-r["end_q"].astype(np.int64), kind="stable")),
                    ("by min(end row, end column)", np.argsort(-np.minimum(r["end_q"], r["end_t"]).astype(np.int64), kind="stable")),
                    ("by window length, longest first (not known beforehand)", np.argsort(-cols, kind="stable"))):
    for waves in (3072,):
        end, mean_life, busy = simulate(cols[order], waves=waves)
        print(f"  {name:56s} {waves} wavefronts: last one done at column {end}, mean life {mean_life:.0f}, halves busy {busy:.2f}")
