#!/usr/bin/env python3
"""Headline benchmark: GCUPS of Smith-Waterman score-only database search,
53-aa query vs 1M x 300-aa synthetic proteins, BLOSUM62, gap 3/1
(BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one search of the query against the device-resident database
shard of every rank (weak scaling: each rank owns its own 1M x 300 shard; the
only exchange is the gather of the int32 scores to rank 0 over RCCL).
At N = 1 `value` is the search as the reference's boundary returns it: a blocking
call (miopalSearch) that leaves the scores in a HOST array of the caller (a pinned
one: the kernel writes database order straight into it; `value_host_results_pageable`
is the same call with an ordinary array, through the library's pinned bounce buffer)
and `value_device_results` the search with the scores left in HBM. At N > 1 (and
with --force-collective) a step is the search with the scores left in HBM, where the
gather reads them, plus that gather.
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
RESERVED_CUS = 8       # ranks of a multi-GPU run keep these out of the persistent search launch (room for RCCL's kernels)


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
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 rehearsal of the N > 1 path: a 1-rank nccl group (launch under torch.distributed.run "
                         "--nproc-per-node 1), RCCL gather of device tensors, all_reduce of the elapsed time, "
                         "MIOPAL_RESERVE_CUS")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` typed bare (no WORLD_SIZE in the environment): start the N ranks as a
    child `torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line
    and the child's exit code. This parent never touches the GPU - nothing is imported that could
    initialise HIP before the child starts, and the child is a child process, not an exec."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in child.stdout:          # rank 0's JSON line to stdout; whatever else the ranks print (gloo's
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr   # connection notes, ...) to stderr
        out.write(line)
        out.flush()
    return child.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
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
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    if collective:
        # leave the collective's kernels a few CUs beside the persistent search kernel, so that
        # the gather of one step really runs during the next step's search (a handle option:
        # include/miopal.h, miopalDbSetOption)
        db.set_option("reserve_cus", RESERVED_CUS)
    # two result buffers: the gather of step k runs beside the search of step k + 1
    outs = [torch.zeros(N, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(2)]
    out = outs[0]
    stream = torch.cuda.current_stream().cuda_stream
    db.search_device_scores(query, matrix, out.data_ptr(), stream, 3, 1, "sw")  # builds the packed view
    torch.cuda.synchronize()
    build_s = time.time() - t0
    on_device = backend == "nccl"
    from pyopal_amd.shard import OverlappedGather
    pipe = OverlappedGather(outs, dst=0, on_device=on_device, force=args.force_collective)
    # N = 1: the caller's result array, pinned (the kernel writes into it; nothing is copied on the host)
    host_out = None
    if not collective:
        host_pinned = torch.empty(N, dtype=torch.int32).pin_memory()
        host_out = host_pinned.numpy()

    def host_step():
        db.search(query, matrix, 3, 1, "score", "sw", score_out=host_out)

    def step():
        if not collective:
            return host_step()
        b, buf = pipe.acquire()  # waits (stream-ordered for RCCL) for the gather that read it last
        db.search_device_scores(query, matrix, buf.data_ptr(), stream, 3, 1, "sw")
        # the one exchange of the path: per-shard scores to rank 0 (RCCL over xGMI). Issued
        # asynchronously: it waits for the search on the current stream, then runs on the
        # collective's own stream while the next step's search starts.
        pipe.submit(b)

    def fence():
        pipe.drain()
        if collective:
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
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            # the gathered vector of this rank's own shard is what the search wrote
            assert torch.equal(pipe.received[last][0].cpu(), out.cpu())

    # (correctness gate: in the cpu_baseline leg below - every score against the AVX2 port,
    # a sample against the scalar checker; the other legs never touch the code under oracle/)
    got = host_out.copy() if host_out is not None else out.cpu().numpy()
    checksum = int(got.astype(np.int64).sum())
    device_form = None
    if not collective:
        # the same search with the scores left in HBM (what `value` was up to round 2)
        device_form = device_results(db, query, matrix, outs[0], stream, Q, N, L)
        assert np.array_equal(outs[0].cpu().numpy(), got)

    cfg5 = None
    if not args.no_cfg5:
        cfg5 = cfg5_strong(args, query, matrix, rank, world, local_rank, on_device, stream, collective)
    if collective:
        # every timed region is over: the group ends here, so that the other ranks are gone (and their
        # cores free) while rank 0 times the CPU baseline on its own shard
        dist.barrier()
        dist.destroy_process_group()

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
        host_form = host_results(db, query, matrix, Q, N, L) if not collective else None
        plain_form = plain_entry(query, residues, offsets, matrix, Q, N, L, got) if not collective else None
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
            # N = 1: `value` is the blocking call with the scores in the caller's pinned host array; the same
            # call with an ordinary (pageable) array goes through the library's pinned bounce buffer and
            # one 4 MB copy on the host; with the scores left in HBM nothing crosses PCIe
            "value_host_results_pageable": host_form["gcups"] if host_form else None,
            "ms_per_step_host_results_pageable": host_form["ms"] if host_form else None,
            "value_device_results": device_form["gcups"] if device_form else None,
            "ms_per_step_device_results": device_form["ms"] if device_form else None,
            "forced_collective": bool(args.force_collective),
            # opalSearchDatabase exactly as the reference binds it: N host pointers in, the database
            # uploaded and packed on every call, N result structs out (PCIe-inclusive, never `value`)
            "value_pcie_inclusive": plain_form["gcups"] if plain_form else None,
            "ms_per_step_pcie_inclusive": plain_form["ms"] if plain_form else None,
            "config": {
                "workload": f"sw_score q{Q} (README.md:86) vs {N}x{L} uniform-random proteins per GPU, "
                            "BLOSUM62, gap_open 3, gap_extend 1; value: "
                            + ("scores left in HBM + RCCL gather to rank 0" if collective else
                               "blocking miopalSearch, scores in the caller's pinned host array "
                               "(value_device_results: scores left in HBM)"),
                "targets_per_gpu": N, "target_length": L, "query_length": Q,
                "sharding": f"{world} independent shards, RCCL gather of int32 scores to rank 0"
                            if collective else "single shard",
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
                "traffic_ratio": round(pmc["hbm_traffic_bytes_per_launch"] / alg_bytes, 3)
                                 if pmc and pmc.get("hbm_traffic_bytes_per_launch") else None,
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
                # LDS counters of the same PMC passes (north_star: "LDS-hit counters reported"): the pair table
                # is read with one ds_read_b128 per four query rows of a lane; two thirds of the LDS array's
                # busy cycles are bank conflicts of the random pair rows
                "lds": pmc["lds"] if pmc else None,
            },
            "db_build_s": round(build_s, 3),
            "score_checksum": checksum,
        }
        line["extras"] = {}
        if not args.no_cpu_baseline:
            # (N > 1: rank 0's shard against rank 0's scores - the same 1M x 300 workload per GPU)
            line["cpu_baseline"] = cpu_baseline(query, residues, offsets, matrix, Q, N, L, got)
            if not collective:
                line["extras"] = extras(db, query, matrix, Q, N, L)
        if cfg5 is not None:
            line["extras"]["cfg5_strong"] = cfg5
        print(json.dumps(compact_line(line)), flush=True)
    db.close()


def compact_line(line, limit=8000):
    """The ONE printed line: the contract's keys, `roofline` with one flat scalar per secondary leg (a record that keeps
    the scalar keys of `roofline` and the last 8 KB of the line keeps every number DESIGN.md quotes), `cpu_baseline`.
    Everything nested - per-leg and per-kernel roofline blocks, wave-cycle fractions, LDS counters, notes - goes to
    bench_details.json beside the working directory's bench.py (and under gpurun_out/ when that exists)."""
    ex = line.pop("extras", {}) or {}
    roof = line["roofline"]
    details = {"roofline": {k: roof.pop(k) for k in ("valu_issue", "lds") if k in roof}, "extras": ex,
               "value": line["value"], "ms_per_step": line["ms_per_step"], "library_sha256": library_sha256()}
    valu = details["roofline"].get("valu_issue") or {}
    flat = {
        "valu_issue_frac": valu.get("frac"), "valu_instructions_per_cell_pair": valu.get("instructions_per_cell_pair"),
        "pageable_gcups": line.get("value_host_results_pageable"), "device_results_gcups": line.get("value_device_results"),
        "pcie_inclusive_ms": line.get("ms_per_step_pcie_inclusive"), "pcie_inclusive_gcups": line.get("value_pcie_inclusive"),
    }

    def get(*path):
        d = ex
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    flat.update({
        "sustained_median_ms": get("sustained", "ms_per_search", "median"), "sustained_p99_ms": get("sustained", "ms_per_search", "p99"),
        "lognormal_gcups": get("lognormal_lengths", "host_results_gcups"),
        "cfg2_end_ms": get("cfg2_end", "ms"), "cfg3_full_ms": get("cfg3_full", "ms"),
        "cfg3_full_operations": get("cfg3_full", "alignment_operations"),
        "q150_score_ms": get("longer_queries_sw", "q150", "score", "ms"), "q150_end_ms": get("longer_queries_sw", "q150", "end", "ms"),
        "q300_score_ms": get("longer_queries_sw", "q300", "score", "ms"), "q300_end_ms": get("longer_queries_sw", "q300", "end", "ms"),
        "q300_full_ms": get("longer_queries_sw", "q300", "full", "ms"),
        "cfg4_nw_ms": get("cfg4_with_tail", "nw", "ms"), "cfg4_hw_ms": get("cfg4_with_tail", "hw", "ms"),
        "cfg4_ov_ms": get("cfg4_with_tail", "ov", "ms"), "cfg4_sw_ms": get("cfg4_with_tail", "sw", "ms"),
        "cfg4_nw_gcups": get("cfg4_with_tail", "nw", "host_results_gcups"),
        "cfg5_gcups": get("cfg5_strong", "gcups"), "cfg5_ms": get("cfg5_strong", "ms_per_step"),
    })
    roof.update({k: v for k, v in flat.items() if v is not None})
    line["details_file"] = "bench_details.json"
    for where in (os.getcwd(), os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(where):
                with open(os.path.join(where, "bench_details.json"), "w") as f:
                    json.dump(details, f, indent=1)
        except OSError:
            pass
    # (never beyond the limit: the longest strings go first)
    for key in ("note", "traffic_source"):
        if len(json.dumps(line)) > limit and key in roof:
            roof[key] = roof[key][:80]
    if len(json.dumps(line)) > limit and isinstance(line.get("cpu_baseline"), dict):
        for key in ("note", "sample", "cpu"):
            if len(json.dumps(line)) > limit and isinstance(line["cpu_baseline"].get(key), str):
                line["cpu_baseline"][key] = line["cpu_baseline"][key][:120]
    return line


def device_results(db, query, matrix, out, stream, Q, N, L):
    """The search with the scores left in HBM (miopalSearchDeviceScores), back to back on one stream."""
    import torch
    for _ in range(5):
        db.search_device_scores(query, matrix, out.data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        db.search_device_scores(query, matrix, out.data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    return {"ms": round(dt * 1e3, 4), "gcups": round(float(Q) * N * L / dt / 1e9, 1)}


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


def pmc_summary(tag="headline"):
    """The newest committed PMC summary of a workload (`headline`: the default bench command; others:
    tools/collect_pmc.sh), if it was taken on this very build of libmiopal.so (tools/summarize_pmc.py
    stores the library's sha256 beside the counters)."""
    import glob
    mine = library_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{tag}.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("library_sha256") != mine:
            continue
        c = d.get("counters", {})
        def mean(name):
            return c.get(name, {}).get("mean_per_launch")
        return {
            "file": os.path.relpath(path, ROOT),
            "library_sha256": mine,
            "kernel": d.get("kernel"),
            "kernel_ms": d.get("kernel_ms_unprofiled"),
            "hbm_traffic_bytes_per_launch": d.get("hbm_traffic_bytes_per_launch"),
            "valu_instructions_per_launch": mean("SQ_INSTS_VALU"),
            # LDS: cycles the LDS arrays were busy, the extra cycles of bank conflicts among them, and the
            # wave-cycles instructions could not issue for the LDS (quad-cycle units, per launch)
            "lds": {"idx_active": mean("SQ_LDS_IDX_ACTIVE"), "bank_conflict": mean("SQ_LDS_BANK_CONFLICT"),
                    "wait_inst_lds": mean("SQ_WAIT_INST_LDS"), "wave_cycles": mean("SQ_WAVE_CYCLES"),
                    "bank_conflict_fraction": d.get("lds_bank_conflict_fraction")},
            "wave_cycle_fractions": d.get("fractions_of_wave_cycles"),
        }
    return None


LANE_KERNELS = {1: "interseq_kernel (general, v_perm profile)", 2: "interseq_pair_kernel<int16>", 3: "interseq_pair_kernel<half>",
                4: "interseq_pair_biased_kernel", 5: "interseq_pair_global_kernel", 6: "interseq_pair_strips_kernel",
                7: "interseq_pair_global_strips_kernel"}


def leg_roofline(kernel_ms, cells, alg_bytes, routing, pmc_tag, boundary_bytes=None, kernel=None):
    """The mandated HBM roofline of one secondary leg (the bound per config):
    algorithmic bytes over the dominant kernel's HIP-event time, and where that kernel stands against
    the SIMDs' issue rate when a PMC summary of this very build exists (null otherwise)."""
    pmc = pmc_summary(pmc_tag) if pmc_tag else None
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    valu = None
    if pmc and pmc.get("valu_instructions_per_launch") and kernel_ms > 0:
        instr = pmc["valu_instructions_per_launch"]
        full = 1024 * 2.4e9 / 2.0 / instr * cells / 1e9
        valu = {"instructions_per_cell_pair": round(instr * 128.0 / cells, 3),
                "cycles_per_instruction": round(kernel_ms * 1e-3 * 2.4e9 * 1024 / instr, 3),
                "full_rate_peak_gcups": round(full, 1), "frac": round(cells / (kernel_ms * 1e-3) / 1e9 / full, 3),
                "wave_cycle_fractions": pmc.get("wave_cycle_fractions"), "source": pmc["file"]}
    return {"bound": "hbm", "kernel": kernel or LANE_KERNELS.get(int(routing[1]) & 15, "wavefront-per-pair kernels"),
            "kernel_ms": round(kernel_ms, 4), "kernel_gcups": round(cells / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None,
            "algorithmic_bytes": alg_bytes, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6),
            # rows handed from strip to strip through HBM (16 B per column, lane and strip boundary: one
            # store, one load): real traffic of multi-strip queries, not algorithmic
            "strip_boundary_bytes": boundary_bytes,
            "traffic": pmc["hbm_traffic_bytes_per_launch"] if pmc else None,
            # counter traffic over algorithmic bytes (1 = nothing read twice; the strips kernels' boundary rows are
            # real traffic that is not algorithmic: see strip_boundary_bytes)
            "traffic_ratio": round(pmc["hbm_traffic_bytes_per_launch"] / alg_bytes, 3)
                             if pmc and pmc.get("hbm_traffic_bytes_per_launch") else None,
            "traffic_source": pmc["file"] if pmc else None,
            "valu_issue": valu}


def full_pipeline_roofline(res, Q, N, L, wall_ms, alg_bytes_whole):
    """BASELINE configs[2] (`full`): the call is a pipeline of kernels, each with a bound of its own. One entry per
    kernel: launches per search, time per launch and HBM traffic from the rocprofv3 summaries of THIS build under
    profiles/ (null when they were taken on another build), algorithmic bytes per launch derived from the result of
    the search itself (windows, cells and operations of the alignments), the ratio of the two."""
    end_t, end_q = res["end_t"].astype(np.int64), res["end_q"].astype(np.int64)
    start_t, start_q = res["start_t"].astype(np.int64), res["start_q"].astype(np.int64)
    live = (end_t >= 0) & (end_q >= 0)
    ops = float(res["aln_off"][-1])
    prefix_residues = float((end_t[live] + 1).sum())                       # the reversed prefixes the start-cell scan reads
    window_cols = float((end_t[live] - start_t[live] + 1).sum())
    window_cells = float(((end_t[live] - start_t[live] + 1) * (end_q[live] - start_q[live] + 1)).sum())
    batches = 4.0
    groups = 2.0   # (round 5: the direction pass of two batches per launch)
    stages = [
        ("end pass (scores + end cells)", "cfg3full_interseq_pair_biased_kernel", 1.0, float(N) * L + 20.0 * N),
        ("start cells: scan of the reversed prefixes, two pairs per lane", "cfg3full_perpair_packed_scan_kernel", 1.0,
         prefix_residues + 28.0 * N),
        ("directions of the [start..end] rectangles (4 bits a cell), two pairs per lane", "cfg3full_perpair_packed_trace_kernel", groups,
         (window_cols + 0.5 * window_cells + 60.0 * N) / groups),
        ("walk: operations from the direction bits", "cfg3full_walk_planes_kernel", batches, (2.5 * ops + 60.0 * N) / batches),
        ("operations compacted into slice order", "cfg3full_gather_ops_kernel", batches, (2.0 * ops + 12.0 * N) / batches),
        ("operations to pinned host memory, two bits each", "cfg3full_copy_out_packed_kernel", batches, 1.25 * ops / batches),
    ]
    kernels = []
    for what, tag, launches, alg in stages:
        pmc = pmc_summary(tag)
        k_ms = pmc.get("kernel_ms") if pmc else None
        traffic = pmc["hbm_traffic_bytes_per_launch"] if pmc else None
        achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms else None
        kernels.append({"stage": what, "kernel": pmc["kernel"] if pmc else tag.split("_", 1)[1],
                        "launches_per_search": launches, "kernel_ms": round(k_ms, 4) if k_ms else None,
                        "algorithmic_bytes": round(alg), "achieved": round(achieved, 1) if achieved else None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                        "traffic": traffic, "traffic_ratio": round(traffic / alg, 2) if traffic else None,
                        "wave_cycle_fractions": pmc.get("wave_cycle_fractions") if pmc else None,
                        "source": pmc["file"] if pmc else None})
    achieved = alg_bytes_whole / (wall_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "pipeline (see kernels)", "kernel_ms": round(wall_ms, 4), "algorithmic_bytes": alg_bytes_whole,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "note": "kernel_ms / achieved of this block: wall time of the whole call (host-visible results); per kernel below. "
                    "The direction and scan passes are VALU-issue-bound (two pairs per lane on 16-bit halves: 8 / 4.5 instructions "
                    "per cell in the row loops), the walk by memory latency (one round trip per two 64-byte lines of direction bits and wavefront)",
            "kernels": kernels}


def strips_of(qlen, sw):
    """Strips of a multi-strip query on the pair-table kernels (host.hip: strips of at most 52 rows; the
    NW / HW / OV kernel prefers heights that divide the query)."""
    if qlen <= 64:
        return 1
    if sw:
        return max(2, -(-qlen // 52))
    best = None
    for rows in range(52, 30, -2):
        ns = -(-qlen // rows)
        if ns < 2 or (ns - 1) * rows >= qlen:
            continue
        cost = ns * (rows + 4) + (rows // 8 if ns * rows - qlen > 1 else 0)
        if best is None or cost < best[0]:
            best = (cost, ns)
    return best[1] if best else 1


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


CFG5_CHECKSUM = 596_667_328   # sum of the 10M scores of BASELINE configs[4] (frozen at N = 1 in round 2)


def cfg5_targets(lo, hi, length, n_total, block=100_000):
    """Residues of the targets [lo, hi) of the cfg5 database."""
    parts = []
    for b in range(lo // block, (max(hi, lo + 1) - 1) // block + 1):
        first = b * block
        count = min(block, n_total - first)
        piece = cfg5_block(b, count, length)
        a, z = max(lo, first) - first, min(hi, first + count) - first
        parts.append(piece[a * length:z * length])
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def cfg5_strong(args, query, matrix, rank, world, local_rank, on_device, stream, collective):
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
    residues = cfg5_targets(lo, hi, length, n_total, block)
    n = hi - lo
    offsets = np.arange(n + 1, dtype=np.int64) * length
    t0 = time.time()
    db = _capi.DeviceDatabase(residues, offsets, 24, device=local_rank)
    if collective:
        db.set_option("reserve_cus", RESERVED_CUS)
    width = max(bounds[r + 1] - bounds[r] for r in range(world))   # equal-size gather slots
    outs = [torch.zeros(width, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(2)]
    db.search_device_scores(query, matrix, outs[0].data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize()
    build_s = time.time() - t0
    pipe = OverlappedGather(outs, dst=0, on_device=on_device, force=args.force_collective)

    def fence():
        pipe.drain()
        if collective:
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
    if collective:
        gathered = [torch.zeros_like(stats) for _ in range(world)] if rank == 0 else None
        dist.gather(stats, gathered, dst=0)
    else:
        gathered = [stats]
    db.close()
    if rank != 0:
        return None
    elapsed = max(float(g[0]) for g in gathered)
    cells = float(len(query)) * n_total * length
    # Self-check of the sharded form (a wrong shard boundary, a gather slot in the wrong place or a rank
    # that searched the wrong block would otherwise go unnoticed at N > 1): the whole-database checksum
    # equals the one frozen at N = 1, and the first and last 1000 GATHERED scores of every shard equal a
    # direct search of those targets (regenerated here; 2000 targets take the wavefront-per-pair kernel,
    # not the kernel that produced the gathered scores).
    total_checksum = int(sum(float(g[2]) for g in gathered))
    if n_total == 10_000_000 and total_checksum != CFG5_CHECKSUM:
        raise SystemExit(f"cfg5: checksum {total_checksum} over {world} shard(s), expected {CFG5_CHECKSUM}")
    received = pipe.received[pipe.last] if collective else [outs[pipe.last]]
    for r in range(world):
        size = bounds[r + 1] - bounds[r]
        k = min(1000, size)
        if k == 0:
            continue
        mine = received[r][:size].cpu().numpy()
        for a, z in ((bounds[r], bounds[r] + k), (bounds[r + 1] - k, bounds[r + 1])):
            res = cfg5_targets(a, z, length, n_total, block)
            sdb = _capi.DeviceDatabase(res, np.arange(z - a + 1, dtype=np.int64) * length, 24, device=local_rank)
            direct = sdb.search(query, matrix, 3, 1, "score", "sw")["score"]
            sdb.close()
            if not np.array_equal(mine[a - bounds[r]:z - bounds[r]], direct):
                raise SystemExit(f"cfg5: gathered scores of shard {r}, targets [{a}, {z}), differ from a direct search")
    return {
        "workload": f"sw_score q{len(query)} vs ONE {n_total}x{length} database (seed (3, block)), "
                    f"{world} residue-balanced contiguous shard(s), one gather of int32 scores to rank 0",
        "scaling": "strong",
        "steps": args.cfg5_steps,
        "gcups": round(cells / (elapsed / args.cfg5_steps) / 1e9, 1),
        "ms_per_step": round(elapsed / args.cfg5_steps * 1e3, 3),
        "kernel_ms_per_rank": [round(float(g[1]), 3) for g in gathered],
        "targets_per_rank": [bounds[r + 1] - bounds[r] for r in range(world)],
        "score_checksum": total_checksum,
        "self_check": "checksum equals the N = 1 value; first and last 1000 gathered scores of every shard equal a direct search",
        "db_build_s_rank0": round(build_s, 2),
    }


def extras(db, query, matrix, Q, N, L):
    """Secondary measurements asked for by SURVEY.md section 8d; never part of `value`."""
    import _data
    from pyopal_amd import _capi
    out = {}
    # the headline search (the call `value` times) back to back for about six seconds: clocks and
    # temperature settled, and long enough for a 5-s utilisation sampler to see the card busy
    import torch
    pinned = torch.empty(N, dtype=torch.int32).pin_memory()
    host_out = pinned.numpy()
    per_call = []
    t_end = time.perf_counter() + 6.0
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        db.search(query, matrix, 3, 1, "score", "sw", score_out=host_out)
        per_call.append(time.perf_counter() - t0)
    per_call = np.array(per_call)
    out["sustained"] = {"seconds": round(float(per_call.sum()), 2), "searches": int(len(per_call)),
                        "gcups": round(float(Q) * N * L * len(per_call) / float(per_call.sum()) / 1e9, 1),
                        "ms_per_search": {"min": round(float(per_call.min()) * 1e3, 4),
                                          "median": round(float(np.median(per_call)) * 1e3, 4),
                                          "p99": round(float(np.quantile(per_call, 0.99)) * 1e3, 4),
                                          "max": round(float(per_call.max()) * 1e3, 4)}}
    del pinned, host_out
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

    def timed_leg(dbx, q, mode, algo, reps):
        for _ in range(2):
            dbx.search(q, matrix, 3, 1, mode, algo)
        dbx.set_profiling(True)
        dbx.last_kernel_time()
        held = []   # (a `full` search hands out tens of MB of operations: the caller's free() of the previous
                    # result - 3 ms per 57 MB on this host - is not part of the next search; and like the headline
                    # search, which writes into one array of the caller's every step, it re-uses its per-target
                    # arrays: 36 MB of pages that are not faulted in again)
        res = dbx.search(q, matrix, 3, 1, mode, algo) if mode == "full" else None
        calls = []
        for _ in range(reps):
            t0 = time.perf_counter()
            res = dbx.search(q, matrix, 3, 1, mode, algo, reuse=res if mode == "full" else None)
            calls.append(time.perf_counter() - t0)
            if mode == "full":
                held.append(res)
        dt = float(np.median(calls))   # (median of the calls: these legs are a handful of calls each)
        del held
        n_launch, k_total = dbx.last_kernel_time()
        dbx.set_profiling(False)
        return dt, (k_total / n_launch if n_launch else 0.0), _capi.DeviceDatabase.last_routing(), res

    for qlen in (150, 300):
        q = _data.random_protein(rng, qlen)
        row = {}
        for mode in ("score", "end"):
            dt, k_ms, routing, _ = timed_leg(db, q, mode, "sw", 5)
            cells = float(qlen) * N * L
            row[mode] = {"ms": round(dt * 1e3, 3), "host_results_gcups": round(cells / dt / 1e9, 1)}
            alg = float(N) * L + (12.0 if mode == "score" else 20.0) * N + qlen + 4 * 24 * 24
            bnd = 16.0 * 64 * (-(-L // 4) * 4) * -(-N // 128) * (strips_of(qlen, True) - 1) if (routing[1] & 15) == 6 else None
            row[mode]["roofline"] = leg_roofline(k_ms, cells, alg, routing, f"q{qlen}_sw" if mode == "score" else f"q{qlen}_sw_end", bnd)
        if qlen == 300:
            q300 = q
        longer[f"q{qlen}"] = row
    out["longer_queries_sw"] = longer
    # the headline search with end locations (row keys in the low bits of every value: no row scan)
    dt, k_ms, routing, _ = timed_leg(db, query, "end", "sw", 5)
    cells = float(Q) * N * L
    out["cfg2_end"] = {"ms": round(dt * 1e3, 3), "host_results_gcups": round(cells / dt / 1e9, 1),
                       "roofline": leg_roofline(k_ms, cells, float(N) * L + 20.0 * N + Q + 4 * 24 * 24, routing, "cfg2_end",
                                                kernel=f"interseq_pair_biased_kernel<{max(2, (Q + 1) // 2 * 2)}, true>")}
    # BASELINE configs[2]: Smith-Waterman with full alignments on the headline database
    if (N, L) == (1_000_000, 300):
        dt, k_ms, routing, res = timed_leg(db, query, "full", "sw", 5)
        ops = int(res["aln_off"][-1])
        cells = float(Q) * N * L
        alg = float(N) * L + 12.0 * N + 16.0 * N + ops + Q + 4 * 24 * 24
        out["cfg3_full"] = {"ms": round(dt * 1e3, 3), "host_results_gcups": round(cells / dt / 1e9, 1),
                            "end_pass_kernel_ms": round(k_ms, 4), "alignment_operations": ops,
                            "roofline": full_pipeline_roofline(res, Q, N, L, dt * 1e3, alg)}
        del res
    # (behind cfg3: a search that hands out 360 MB of operations and takes 4 GB of direction bits)
    if (N, L) == (1_000_000, 300):
        # full alignments with a query of five strips of rows (the earlier result lent back, as in cfg3_full)
        dt, k_ms, routing, res = timed_leg(db, q300, "full", "sw", 3)
        stages = []
        for what, tag, launches in (("end pass", "q300full_interseq_pair_strips_kernel", 1),
                                    ("start cells: scan of the reversed prefixes, two pairs per lane", "q300full_perpair_packed_scan_strips_kernel", 1),
                                    ("directions (4 bits a cell), two pairs per lane", "q300full_perpair_packed_trace_kernel", 4),
                                    ("walk", "q300full_walk_planes_kernel", 4),
                                    ("operations to pinned host memory, two bits each", "q300full_copy_out_packed_kernel", 4)):
            pmc = pmc_summary(tag)
            stages.append({"stage": what, "launches_per_search": launches,
                           "kernel_ms": round(pmc["kernel_ms"], 3) if pmc and pmc.get("kernel_ms") else None,
                           "source": pmc["file"] if pmc else None})
        out["longer_queries_sw"]["q300"]["full"] = {"ms": round(dt * 1e3, 2), "host_results_gcups": round(300.0 * N * L / dt / 1e9, 1),
                       "alignment_operations": int(res["aln_off"][-1]), "kernels": stages,
                       "note": "VALU-issue-bound passes, two pairs per lane on 16-bit halves since round 5 (4.5 / 8 instructions per "
                               "cell; 8.1 / 15 at 32 bit before): DESIGN.md section 4"}
        del res
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
            dt, k_ms, routing, _ = timed_leg(cdb, q, "score", algo, 3)
            cfg4[algo] = {"ms": round(dt * 1e3, 1), "host_results_gcups": round(cells / dt / 1e9, 1),
                          "targets_on_the_int32_kernel": int(routing[0])}
            n_t = len(lengths)
            alg = float(off[-1]) + 12.0 * n_t + 2000 + 4 * 24 * 24
            packed_cols = 2000.0 * 782      # groups of 128 targets x columns, the 33 longest targets aside
            bnd = 16.0 * 64 * packed_cols * (strips_of(2000, algo == "sw") - 1) if (routing[1] & 15) in (6, 7) else None
            cfg4[algo]["roofline"] = leg_roofline(k_ms, cells, alg, routing, f"cfg4_{algo}", bnd)
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
    model, physical, logical, usable = host_cpu()
    # (ranks started by torch.distributed.run inherit OMP_NUM_THREADS=1: this leg runs after the group has ended,
    # on rank 0 alone, and takes the cores the process may use)
    import _cpu_baseline
    limit = _cpu_baseline.max_threads()
    if os.environ.get("OMP_NUM_THREADS") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        limit = usable   # (the search sets its thread count itself: omp_set_num_threads)
    threads = max(1, min(physical, usable, limit))
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
