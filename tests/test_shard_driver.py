"""The multi-rank shard protocol of bench.py / SURVEY.md section 8e on CPU:
two `gloo` ranks each own a shard of the database, compute their scores (here
with the CPU checker standing in for the per-rank search) and rank 0 gathers
them; the gathered vector equals a single-rank search of the whole database."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
    import _data, _oracle
    from pyopal_amd import shard
    from pyopal_amd.matrices import ScoringMatrix

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(3)
    lengths = rng.integers(1, 120, size=101)
    res, off = _data.random_db(rng, lengths)
    q = _data.random_protein(rng, 30)
    m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
    bounds = shard.balanced_bounds(off, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    local = _oracle.search(q, res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], m, 3, 1, "score", "sw")["score"]
    full = shard.gather_scores(torch.from_numpy(local.copy()), bounds, dst=0)
    if rank == 0:
        want = _oracle.search(q, res, off, m, 3, 1, "score", "sw")["score"]
        assert np.array_equal(full.numpy(), want), "gathered scores differ"
        assert bounds[0] == 0 and bounds[-1] == 101 and all(b > a for a, b in zip(bounds, bounds[1:]))
        print("SHARD_OK", bounds)

    # the overlapped gather of bench.py: two buffers in turn, asynchronous collectives
    bufs = [torch.zeros(1000, dtype=torch.int32) for _ in range(2)]
    pipe = shard.OverlappedGather(bufs, dst=0, on_device=False)
    seen = {}
    for step in range(7):
        b, buf = pipe.acquire()
        buf.fill_(step * 100 + rank)          # "the search" of this step
        pipe.submit(b)
        seen[b] = step
    pipe.drain()
    if rank == 0:
        for b, step in seen.items():
            for r in range(world):
                assert torch.all(pipe.received[b][r] == step * 100 + r), (b, step, r)
        assert pipe.last == 0 and seen == {0: 6, 1: 5}
        print("PIPE_OK")
    dist.destroy_process_group()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "SHARD_OK" in out.stdout and "PIPE_OK" in out.stdout


def test_balanced_bounds_unit():
    import numpy as np
    sys.path.insert(0, ROOT)
    from pyopal_amd import shard
    off = np.array([0, 10, 20, 30, 1030, 1040], dtype=np.int64)
    assert shard.balanced_bounds(off, 1) == [0, 5]
    b = shard.balanced_bounds(off, 2)
    assert b[0] == 0 and b[-1] == 5 and len(b) == 3
    b8 = shard.balanced_bounds(off, 8)   # more ranks than targets: empty shards allowed at the end
    assert b8[0] == 0 and b8[-1] == 5 and len(b8) == 9 and all(y >= x for x, y in zip(b8, b8[1:]))
