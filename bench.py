#!/usr/bin/env python3
"""Headline benchmark: GCUPS of Smith-Waterman score-only database search,
53-aa query vs 1M x 300-aa synthetic proteins, BLOSUM62, gap 3/1
(BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one search of the query against the device-resident database
shard of every rank (weak scaling: each rank owns its own 1M x 300 shard; the
only exchange is the gather of the int32 scores to rank 0 over RCCL).
`value` is that search with the scores left in HBM (where the gather reads
them); `value_host_results` is the same search with the scores delivered into a
host buffer, the form the reference's boundary returns (N = 1).
`extras.cfg5_strong` is BASELINE.json configs[4]: ONE 10M x 400 database cut into
residue-balanced contiguous shards over the ranks (strong scaling), one gather.
Rank 0 prints one JSON line.
"""
import ctypes
import hashlib
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md, HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--targets", type=int, default=1_000_000, help="targets per GPU")
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the 10M x 400 strong-scaling leg")
    ap.add_argument("--cfg5-targets", type=int, default=10_000_000)
    ap.add_argument("--cfg5-steps", type=int, default=5)
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import _data
    from pyopal_amd import _capi
    from pyopal_amd.matrices import ScoringMatrix

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the search path has no CPU fallback)")
    # Rehearsal switches for boxes with fewer GPUs than ranks (never set by the driver):
    # MIOPAL_BENCH_BACKEND=gloo gathers through host copies, MIOPAL_BENCH_SHARE_DEVICE=1
    # puts every rank on cuda:0.
    backend = os.environ.get("MIOPAL_BENCH_BACKEND", "nccl")
    if os.environ.get("MIOPAL_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # leave the collective's kernels a few CUs beside the persistent search kernel, so that
        # the gather of one step really runs during the next step's search
        os.environ.setdefault("MIOPAL_RESERVE_CUS", "8")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- synthetic shard (seed 1, uniform over 20 amino acids)
    N, L = args.targets, args.length
    matrix = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
    query = _data.encode(_data.README_QUERY)
    Q = len(query)
    rng = np.random.default_rng(1 + rank)
    residues, offsets = _data.random_db(rng, np.full(N, L))
    t0 = time.time()
    db = _capi.DeviceDatabase(residues, offsets, 24, device=local_rank)
    # two result buffers: the gather of step k runs beside the search of step k + 1
    outs = [torch.zeros(N, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(2)]
    out = outs[0]
    stream = torch.cuda.current_stream().cuda_stream
    db.search_device_scores(query, matrix, out.data_ptr(), stream, 3, 1, "sw")  # builds the packed view
    torch.cuda.synchronize()
    build_s = time.time() - t0
    on_device = backend == "nccl"
    from pyopal_amd.shard import OverlappedGather
    pipe = OverlappedGather(outs, dst=0, on_device=on_device)

    def step():
        b, buf = pipe.acquire()  # waits (stream-ordered for RCCL) for the gather that read it last
        db.search_device_scores(query, matrix, buf.data_ptr(), stream, 3, 1, "sw")
        # the one exchange of the path: per-shard scores to rank 0 (RCCL over xGMI). Issued
        # asynchronously: it waits for the search on the current stream, then runs on the
        # collective's own stream while the next step's search starts.
        pipe.submit(b)

    def fence():
        pipe.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    db.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    last = pipe.last   # buffer of the last step
    out = outs[last]
    n_launch, kernel_ms = db.last_kernel_time()
    db.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            # the gathered vector of this rank's own shard is what the search wrote
            assert torch.equal(pipe.received[last][0].cpu(), out.cpu())

    # (correctness gate: in the cpu_baseline leg below - every score against the AVX2 port,
    # a sample against the scalar checker; the other legs never touch the code under oracle/)
    got = out.cpu().numpy()
    checksum = int(got.astype(np.int64).sum())

    cfg5 = None
    if not args.no_cfg5:
        cfg5 = cfg5_strong(args, query, matrix, rank, world, local_rank, on_device, stream)

    if rank == 0:
        cells_per_step = float(Q) * N * L * world
        ms_per_step = elapsed / args.steps * 1e3
        gcups = cells_per_step / (elapsed / args.steps) / 1e9
        # algorithmic bytes of one launch: every residue once,
        # 8 B of offset/length metadata and 4 B of score per target
        alg_bytes = float(N) * L + 12.0 * N + Q + 4 * 24 * 24
        k_ms = kernel_ms / max(n_launch, 1)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pmc = pmc_summary() if (N, L) == (1_000_000, 300) else None
        host_form = host_results(db, query, matrix, Q, N, L) if world == 1 else None
        plain_form = plain_entry(query, residues, offsets, matrix, Q, N, L, got) if world == 1 else None
        line = {
            "metric": "GCUPS (billion DP cells/s) SW score-only, 53aa query vs 1Mx300aa DB",
            "value": round(gcups, 1),
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # packed unsigned 16-bit integer lanes (compared through the half-float max, exact below
            # 25600; lanes that reach it are redone at int16 / int32)
            "dtype": "u16",
            "data": "synthetic",
            # the same search with the 4 MB of scores delivered to a host buffer (miopalSearch: + D2H
            # + sync), the form the reference's boundary returns; `value` leaves them in HBM
            "value_host_results": host_form["gcups"] if host_form else None,
            "ms_per_step_host_results": host_form["ms"] if host_form else None,
            # opalSearchDatabase exactly as the reference binds it: N host pointers in, the database
            # uploaded and packed on every call, N result structs out (PCIe-inclusive, never `value`)
            "value_pcie_inclusive": plain_form["gcups"] if plain_form else None,
            "ms_per_step_pcie_inclusive": plain_form["ms"] if plain_form else None,
            "config": {
                "workload": f"sw_score q{Q} (README.md:86) vs {N}x{L} uniform-random proteins per GPU, "
                            "BLOSUM62, gap_open 3, gap_extend 1; value: scores left in HBM "
                            "(value_host_results: scores on host)",
                "targets_per_gpu": N, "target_length": L, "query_length": Q,
                "sharding": f"{world} independent shards, RCCL gather of int32 scores to rank 0"
                            if world > 1 else "single shard",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                # HBM bytes per launch from separate rocprofv3 --pmc passes over THIS binary (the
                # summary carries the sha256 of the libmiopal.so it profiled; null when it differs)
                "traffic": pmc["hbm_traffic_bytes_per_launch"] if pmc else None,
                "traffic_source": pmc["file"] + " (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction of "
                                  "MI355X_MICROARCH.md; library " + pmc["library_sha256"][:16] + ")" if pmc else
                                  "no PMC summary for this build of libmiopal.so (sha256 "
                                  + library_sha256()[:16] + ")",
                "kernel": f"interseq_pair_biased_kernel<{max(2, (Q + 1) // 2 * 2)}, false>",
                "kernel_ms": round(k_ms, 4),
                "kernel_gcups": round(float(Q) * N * L / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                "algorithmic_bytes": alg_bytes,
                "note": "integer-VALU-bound by construction (6.9 VALU instructions per pair of cells of a "
                        "lane, 0.02 B/cell): see DESIGN.md",
                # secondary ceiling: VALU issue, from SQ_INSTS_VALU of this binary
                "valu_issue": valu_ceiling(Q, k_ms, N, L, pmc),
            },
            "db_build_s": round(build_s, 3),
            "score_checksum": checksum,
        }
        line["extras"] = {}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(query, residues, offsets, matrix, Q, N, L, got)
            line["extras"] = extras(db, query, matrix, Q, N, L)
        if cfg5 is not None:
            line["extras"]["cfg5_strong"] = cfg5
        print(json.dumps(line), flush=True)
    db.close()
    if world > 1:
        dist.destroy_process_group()


def valu_ceiling(Q, kernel_ms, N, L, pmc):
    """Where the dominant kernel stands against the SIMDs' issue rate (informational, next to the
    mandated HBM roofline). instructions / cycles come from the PMC summary of this binary
    (SQ_INSTS_VALU per launch); `full_rate_peak` prices every instruction at the 2 cycles a wave64
    VALU instruction takes on a SIMD-32 (MI355X_MICROARCH.md), which packed (VOP3P) and three-operand
    instructions do not reach: alone they issue at 4.4-4.9 cycles, plain 32-bit adds at 2.6, mixed as
    in this kernel about 3.5 (profiles/r01_valu_issue_rates.txt, profiles/r02_ubench_mix.txt)."""
    simds, clock_hz = 1024, 2.4e9
    achieved = float(Q) * N * L / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out = {"achieved": round(achieved, 1), "unit": "GCUPS", "instructions_per_launch": None,
           "instructions_per_cell_pair": None, "cycles_per_instruction": None, "full_rate_peak": None, "frac": None}
    if pmc and pmc.get("valu_instructions_per_launch") and kernel_ms > 0:
        instr = pmc["valu_instructions_per_launch"]   # wave64 instructions
        cells = float(Q) * N * L
        per_pair = instr * 128.0 / cells              # per lane: one instruction updates one pair of cells
        full = simds * clock_hz / 2.0 / instr * cells / 1e9
        out.update({
            "instructions_per_launch": instr,
            "instructions_per_cell_pair": round(per_pair, 3),
            "cycles_per_instruction": round(kernel_ms * 1e-3 * clock_hz * simds / instr, 3),
            "full_rate_peak": round(full, 1),
            "frac": round(achieved / full, 3),
        })
    return out


def library_sha256():
    from pyopal_amd import _capi
    h = hashlib.sha256()
    with open(_capi.LIB_PATH, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def pmc_summary():
    """The newest committed PMC summary of the headline kernel, if it was taken on this very build
    of libmiopal.so (tools/summarize_pmc.py stores the library's sha256 beside the counters)."""
    import glob
    mine = library_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_headline*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("library_sha256") != mine:
            continue
        c = d.get("counters", {})
        return {
            "file": os.path.relpath(path, ROOT),
            "library_sha256": mine,
            "hbm_traffic_bytes_per_launch": d.get("hbm_traffic_bytes_per_launch"),
            "valu_instructions_per_launch": c.get("SQ_INSTS_VALU", {}).get("mean_per_launch"),
        }
    return None


def host_results(db, query, matrix, Q, N, L):
    """The host-buffer form of the same search: miopalSearch incl. the D2H of the scores and the sync."""
    for _ in range(5):
        db.search(query, matrix, 3, 1, "score", "sw")
    times = []
    for _ in range(20):
        t0 = time.perf_counter()
        db.search(query, matrix, 3, 1, "score", "sw")
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"ms": round(dt * 1e3, 4), "gcups": round(float(Q) * N * L / dt / 1e9, 1)}


def plain_entry(query, residues, offsets, matrix, Q, N, L, expect):
    """opalSearchDatabase (include/opal.h; src/pyopal/opal.pxd:38-52) on the same workload: pointers to
    the N host sequences in, N OpalSearchResult structs out, nothing resident between calls."""
    from pyopal_amd import _capi
    lib = _capi.lib()
    record = np.dtype({"names": ["scoreSet", "score", "rest"], "formats": ["<i4", "<i4", "V32"],
                       "offsets": [0, 4, 8], "itemsize": ctypes.sizeof(_capi.OpalSearchResult)})
    results = np.zeros(N, dtype=record)
    rptrs = (results.ctypes.data + np.arange(N, dtype=np.uint64) * record.itemsize).astype(np.uint64)
    ptrs = (residues.ctypes.data + offsets[:-1].astype(np.uint64)).astype(np.uint64)
    lens = np.diff(offsets).astype(np.int32)
    q = np.ascontiguousarray(query, dtype=np.uint8)
    m = np.ascontiguousarray(matrix, dtype=np.int32)
    times = []
    for k in range(8):
        t0 = time.perf_counter()
        rc = lib.opalSearchDatabase(q.ctypes.data, Q, ptrs.ctypes.data, N, lens.ctypes.data, 3, 1, m.ctypes.data, 24,
                                    rptrs.ctypes.data, 0, 3, 1)   # OPAL_SEARCH_SCORE, OPAL_MODE_SW
        times.append(time.perf_counter() - t0)
        _capi.raise_for(rc)
    assert np.array_equal(results["score"], expect) and results["scoreSet"].all()
    lib.miopalReleaseCaches()
    dt = float(np.median(times[2:]))
    return {"ms": round(dt * 1e3, 3), "gcups": round(float(Q) * N * L / dt / 1e9, 1), "first_call_ms": round(times[0] * 1e3, 1)}


def cfg5_block(block, targets, length):
    """Targets [block * 100000, ...) of the cfg5 database: seed (3, block), so that every rank can
    build its own shard of the SAME database whatever the number of ranks."""
    import _data
    rng = np.random.default_rng([3, block])
    return _data.AA20_CODES[rng.integers(0, 20, size=targets * length, dtype=np.uint8)]


def cfg5_strong(args, query, matrix, rank, world, local_rank, on_device, stream):
    """BASELINE.json configs[4]: one 10M x 400 database (seed 3, generated in blocks of 100k
    targets), cut into contiguous shards of equal residue counts (pyopal_amd.shard.balanced_bounds,
    the reference's [start, end) chunks of src/pyopal/_align.py:150-170 balanced by residues), each
    rank searches its shard, ONE gather of the int32 scores to rank 0. Strong scaling: the total
    work is fixed. Timed like the headline: barrier + sync on both sides, max over ranks."""
    import torch
    import torch.distributed as dist
    from pyopal_amd import _capi
    from pyopal_amd.shard import OverlappedGather, balanced_bounds

    n_total, length, block = args.cfg5_targets, 400, 100_000
    all_offsets = np.arange(n_total + 1, dtype=np.int64) * length
    bounds = balanced_bounds(all_offsets, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    parts = []
    for b in range(lo // block, (max(hi, lo + 1) - 1) // block + 1):
        first = b * block
        count = min(block, n_total - first)
        piece = cfg5_block(b, count, length)
        a, z = max(lo, first) - first, min(hi, first + count) - first
        parts.append(piece[a * length:z * length])
    residues = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    n = hi - lo
    offsets = np.arange(n + 1, dtype=np.int64) * length
    t0 = time.time()
    db = _capi.DeviceDatabase(residues, offsets, 24, device=local_rank)
    width = max(bounds[r + 1] - bounds[r] for r in range(world))   # equal-size gather slots
    outs = [torch.zeros(width, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(2)]
    db.search_device_scores(query, matrix, outs[0].data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize()
    build_s = time.time() - t0
    pipe = OverlappedGather(outs, dst=0, on_device=on_device)

    def fence():
        pipe.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        b, buf = pipe.acquire()
        db.search_device_scores(query, matrix, buf.data_ptr(), stream, 3, 1, "sw")
        pipe.submit(b)

    step()
    db.set_profiling(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.cfg5_steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = db.last_kernel_time()
    k_ms = kernel_ms / max(n_launch, 1)
    checksum = int(outs[pipe.last][:n].sum(dtype=torch.int64).item())
    stats = torch.tensor([elapsed, k_ms, float(checksum)], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if on_device else "cpu")
    if world > 1:
        gathered = [torch.zeros_like(stats) for _ in range(world)] if rank == 0 else None
        dist.gather(stats, gathered, dst=0)
    else:
        gathered = [stats]
    db.close()
    if rank != 0:
        return None
    elapsed = max(float(g[0]) for g in gathered)
    cells = float(len(query)) * n_total * length
    return {
        "workload": f"sw_score q{len(query)} vs ONE {n_total}x{length} database (seed (3, block)), "
                    f"{world} residue-balanced contiguous shard(s), one gather of int32 scores to rank 0",
        "scaling": "strong",
        "steps": args.cfg5_steps,
        "gcups": round(cells / (elapsed / args.cfg5_steps) / 1e9, 1),
        "ms_per_step": round(elapsed / args.cfg5_steps * 1e3, 3),
        "kernel_ms_per_rank": [round(float(g[1]), 3) for g in gathered],
        "targets_per_rank": [bounds[r + 1] - bounds[r] for r in range(world)],
        "score_checksum": int(sum(float(g[2]) for g in gathered)),
        "db_build_s_rank0": round(build_s, 2),
    }


def extras(db, query, matrix, Q, N, L):
    """Secondary measurements asked for by SURVEY.md section 8d; never part of `value`."""
    import _data
    from pyopal_amd import _capi
    out = {}
    # lane-packing efficiency on UniProt-like lengths (log-normal, mean about 300)
    rng = np.random.default_rng(7)
    n = min(N, 500_000)
    lengths = np.clip(rng.lognormal(mean=5.55, sigma=0.6, size=n), 20, 8000).astype(np.int64)
    res, off = _data.random_db(rng, lengths)
    vdb = _capi.DeviceDatabase(res, off, 24, device=db.device)
    for _ in range(3):
        vdb.search(query, matrix, 3, 1, "score", "sw")
    t0 = time.perf_counter()
    for _ in range(5):
        vdb.search(query, matrix, 3, 1, "score", "sw")
    dt = (time.perf_counter() - t0) / 5
    vdb.close()
    out["lognormal_lengths"] = {"targets": int(n), "mean_length": round(float(lengths.mean()), 1),
                                "max_length": int(lengths.max()),
                                "host_results_gcups": round(float(Q) * float(lengths.sum()) / dt / 1e9, 1)}
    del res, off
    # longer queries on the headline database (the pair-table kernel strip by strip): scores and end locations
    rng = np.random.default_rng(11)
    longer = {}
    for qlen in (150, 300):
        q = _data.random_protein(rng, qlen)
        row = {}
        for mode in ("score", "end"):
            for _ in range(2):
                db.search(q, matrix, 3, 1, mode, "sw")
            t0 = time.perf_counter()
            for _ in range(5):
                db.search(q, matrix, 3, 1, mode, "sw")
            dt = (time.perf_counter() - t0) / 5
            row[mode] = {"ms": round(dt * 1e3, 3), "host_results_gcups": round(float(qlen) * N * L / dt / 1e9, 1)}
        longer[f"q{qlen}"] = row
    out["longer_queries_sw"] = longer
    # BASELINE configs[3] as written: 2000-aa query vs 100k x 2000 PLUS the reference's 35 long targets
    # (1000 ... 35000 residues: the ones that really leave 16 bits), every algorithm, scores
    if (N, L) == (1_000_000, 300):
        rng = np.random.default_rng(2)
        lengths = np.concatenate([np.full(100_000, 2000), np.arange(1000, 35001, 1000)])
        res, off = _data.random_db(rng, lengths)
        q = _data.random_protein(rng, 2000)
        cdb = _capi.DeviceDatabase(res, off, 24, device=db.device)
        cells = 2000.0 * float(off[-1])
        cfg4 = {}
        for algo in ("nw", "hw", "ov", "sw"):
            cdb.search(q, matrix, 3, 1, "score", algo)
            t0 = time.perf_counter()
            for _ in range(2):
                cdb.search(q, matrix, 3, 1, "score", algo)
            dt = (time.perf_counter() - t0) / 2
            cfg4[algo] = {"ms": round(dt * 1e3, 1), "host_results_gcups": round(cells / dt / 1e9, 1),
                          "targets_on_the_int32_kernel": int(_capi.DeviceDatabase.last_routing()[0])}
        cdb.close()
        out["cfg4_with_tail"] = cfg4
    return out


def host_cpu():
    """CPU model, physical cores of the host, and the CPUs this process may use."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "model name":
                    model = val
                elif key == "physical id":
                    phys = val
                elif key == "core id":
                    core = val
                elif not key and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:   # CPU quota of the container, if any
            quota, period = f.read().split()
            if quota != "max":
                usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    logical = os.cpu_count() or 1
    physical = len(cores) or logical
    return model, physical, logical, usable


def cpu_baseline(query, residues, offsets, matrix, Q, N, L, gpu_scores):
    """Own AVX2 inter-sequence SW (oracle/cpu_simd_baseline.c) on this host's cores: one thread,
    and every core the process may use (one thread per physical core)."""
    import _cpu_baseline
    model, physical, logical, usable = host_cpu()
    threads = max(1, min(physical, usable, _cpu_baseline.max_threads()))
    n = N  # the whole workload: an all-core pass takes tens of milliseconds, a one-thread pass ~1 s
    cdb = _cpu_baseline.CpuDatabase(residues[:offsets[n]], offsets[:n + 1], 24)

    def timed(nthreads, passes):
        cdb.search_sw(query, matrix, 3, 1, nthreads)  # warm-up
        times = []
        for _ in range(passes):
            t0 = time.perf_counter()
            got = cdb.search_sw(query, matrix, 3, 1, nthreads)
            times.append(time.perf_counter() - t0)
        return float(np.median(times)), got

    med_all, scores = timed(threads, 11)
    med_one, _ = timed(1, 5)
    cdb.close()
    if not np.array_equal(scores, gpu_scores[:n]):
        raise SystemExit("CPU baseline and GPU disagree")
    # and a sample against the scalar checker (oracle/opal_oracle.c)
    import _oracle
    sample = min(512, n)
    ref = _oracle.search(query, residues[:offsets[sample]], offsets[:sample + 1], matrix, 3, 1, "score", "sw")
    if not np.array_equal(gpu_scores[:sample], ref["score"]):
        raise SystemExit("GPU scores differ from the CPU checker")
    cells = float(Q) * n * L
    return {
        "value": round(cells / med_all / 1e9, 2),
        "unit": "GCUPS",
        "cores": threads,
        "kind": "port",
        "value_one_thread": round(cells / med_one / 1e9, 2),
        "cpu_model": model,
        "host_physical_cores": physical,
        "host_logical_cpus": logical,
        "cpus_usable_by_this_process": usable,
        "sample": f"all {n} targets of the same database; {threads} threads: median of 11 passes "
                  f"({med_all:.3f} s each), 1 thread: median of 5 ({med_one:.3f} s); own AVX2 8/16/32-bit "
                  f"SWIPE-style code (not Opal: its source is absent), every score equal to the GPU's",
    }


if __name__ == "__main__":
    main()
