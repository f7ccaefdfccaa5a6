#!/usr/bin/env python3
"""Headline benchmark: GCUPS of Smith-Waterman score-only database search,
53-aa query vs 1M x 300-aa synthetic proteins, BLOSUM62, gap 3/1
(BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one search of the query against the device-resident database
shard of every rank (weak scaling: each rank owns its own 1M x 300 shard; the
only exchange is the gather of the int32 scores to rank 0 over RCCL).
Rank 0 prints one JSON line.
"""
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

    if rank == 0:
        cells_per_step = float(Q) * N * L * world
        ms_per_step = elapsed / args.steps * 1e3
        gcups = cells_per_step / (elapsed / args.steps) / 1e9
        # algorithmic bytes of one launch: every residue once,
        # 8 B of offset/length metadata and 4 B of score per target
        alg_bytes = float(N) * L + 12.0 * N + Q + 4 * 24 * 24
        k_ms = kernel_ms / max(n_launch, 1)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
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
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": f"sw_score q{Q} (README.md:86) vs {N}x{L} uniform-random proteins per GPU, "
                            "BLOSUM62, gap_open 3, gap_extend 1",
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
                "traffic": pmc_traffic() if (N, L) == (1_000_000, 300) else None,
                "traffic_source": "profiles/r01d_pmc_interseq_pair_kernel.json (separate rocprofv3 --pmc passes "
                                  "FETCH_SIZE x2 + WRITE_SIZE, same kernel and workload)",
                "kernel": "interseq_pair_kernel<56, ArithSwF16>",
                "kernel_ms": round(k_ms, 4),
                "kernel_gcups": round(float(Q) * N * L / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                "algorithmic_bytes": alg_bytes,
                "note": "integer-VALU-bound by construction (about 4 packed VALU ops per cell, "
                        "0.02 B/cell): see DESIGN.md",
                # secondary ceiling: VALU issue. 418 packed instructions per
                # wavefront-column of 128 x 56 cells (SQ_INSTS_VALU, profiles/r01d_pmc_*), each
                # ~4.42 SIMD-cycles at the instruction mix's measured issue rate
                # (profiles/r01_valu_issue_rates.txt), 1024 SIMDs at the 2.4 GHz boost clock
                "valu_issue": valu_ceiling(Q, k_ms, N, L),
            },
            "db_build_s": round(build_s, 3),
            "score_checksum": checksum,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(query, residues, offsets, matrix, Q, N, L, got)
            line["extras"] = extras(db, query, matrix, Q, N, L)
        print(json.dumps(line), flush=True)
    db.close()
    if world > 1:
        dist.destroy_process_group()


def valu_ceiling(Q, kernel_ms, N, L):
    """Cells/s the dominant kernel could reach if every SIMD issued its measured instruction
    mix back to back: informational, next to the mandated HBM roofline."""
    simds, clock_hz = 1024, 2.4e9
    instr_per_column, cycles_per_instr, targets_per_wave = 418.0, 4.42, 128
    peak = simds * clock_hz / (instr_per_column * cycles_per_instr) * targets_per_wave * Q / 1e9
    achieved = float(Q) * N * L / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return {"achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GCUPS",
            "frac": round(achieved / peak, 3)}


def extras(db, query, matrix, Q, N, L):
    """Secondary measurements asked for by SURVEY.md section 8d; never part of `value`."""
    import _data
    from pyopal_amd import _capi
    out = {}
    # (a) the host-buffer form of the same search: miopalSearch incl. the 4 MB D2H and sync
    for _ in range(3):
        db.search(query, matrix, 3, 1, "score", "sw")
    t0 = time.perf_counter()
    for _ in range(5):
        db.search(query, matrix, 3, 1, "score", "sw")
    dt = (time.perf_counter() - t0) / 5
    out["host_results_ms_per_search"] = round(dt * 1e3, 3)
    out["host_results_gcups"] = round(float(Q) * N * L / dt / 1e9, 1)
    # (b) lane-packing efficiency on UniProt-like lengths (log-normal, mean about 300)
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
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (counters cannot be collected from inside the timed run)."""
    path = os.path.join(ROOT, "profiles", "r01d_pmc_interseq_pair_kernel.json")
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(query, residues, offsets, matrix, Q, N, L, gpu_scores):
    """Own AVX2 inter-sequence SW (oracle/cpu_simd_baseline.c) on this host's cores."""
    import _cpu_baseline
    threads = min(os.cpu_count() or 1, _cpu_baseline.max_threads(), 16)
    # the whole workload: the AVX2 code needs well under a second per pass
    n = N
    cdb = _cpu_baseline.CpuDatabase(residues[:offsets[n]], offsets[:n + 1], 24)
    cdb.search_sw(query, matrix, 3, 1, threads)  # warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        scores = cdb.search_sw(query, matrix, 3, 1, threads)
        times.append(time.perf_counter() - t0)
    cdb.close()
    if not np.array_equal(scores, gpu_scores[:n]):
        raise SystemExit("CPU baseline and GPU disagree")
    # and a sample against the scalar checker (oracle/opal_oracle.c)
    import _oracle
    sample = min(512, n)
    ref = _oracle.search(query, residues[:offsets[sample]], offsets[:sample + 1], matrix, 3, 1, "score", "sw")
    if not np.array_equal(gpu_scores[:sample], ref["score"]):
        raise SystemExit("GPU scores differ from the CPU checker")
    med = sorted(times)[1]
    return {
        "value": round(float(Q) * n * L / med / 1e9, 2),
        "unit": "GCUPS",
        "cores": threads,
        "kind": "port",
        "sample": f"first {n} targets of the same database, median of 3 passes "
                  f"({med:.3f} s each), own AVX2 8/16/32-bit SWIPE-style code (not Opal), "
                  f"all {n} scores equal to the GPU's",
    }


if __name__ == "__main__":
    main()
