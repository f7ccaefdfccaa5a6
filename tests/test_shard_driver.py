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


class _FakeDevices:
    """Two 'GPUs' whose searches are the CPU checker: records what each device was given."""

    def __init__(self, monkeypatch, count=2):
        import numpy as np
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle
        from pyopal_amd import _capi
        self.uploads = []   # (device, targets, residues)
        self.searches = []  # (device, start, end) in shard coordinates
        outer = self

        class FakeLib:
            def miopalDeviceCount(self):
                return count

        class FakeDatabase:
            def __init__(self, residues, offsets, alphabet_length, device=0):
                self.residues = np.array(residues, dtype=np.uint8)
                self.offsets = np.array(offsets, dtype=np.int64)
                self.alphabet_length, self.device, self.count = alphabet_length, device, len(offsets) - 1
                outer.uploads.append((device, self.count, int(self.offsets[-1])))

            def search(self, query, matrix, gap_open, gap_extend, mode, algorithm, start=0, end=None):
                end = self.count if end is None else min(end, self.count)
                outer.searches.append((self.device, start, end))
                off = self.offsets[start:end + 1] - self.offsets[start]
                res = self.residues[self.offsets[start]:self.offsets[end]]
                out = _oracle.search(query, res, off, matrix, gap_open, gap_extend, mode, algorithm)
                if mode == "full":
                    lens = np.array([len(a) for a in out["aln"]], dtype=np.int64)
                    out["aln_off"] = np.concatenate([[0], np.cumsum(lens)])
                    out["aln_flat"] = np.concatenate(out["aln"]) if len(lens) else np.zeros(0, np.uint8)
                return out

            def close(self):
                pass

        monkeypatch.setattr(_capi, "lib", lambda: FakeLib())
        monkeypatch.setattr(_capi, "DeviceDatabase", FakeDatabase)


def test_align_shards_the_database_over_devices(monkeypatch):
    # pyopal_amd.align with two devices: each uploads only its residue-balanced shard, chunks
    # are cut at the shard boundary, indices are absolute, ordered=True keeps database order
    import numpy as np
    fake = _FakeDevices(monkeypatch, 2)
    import pyopal_amd
    from pyopal_amd import shard
    import _data
    rng = np.random.default_rng(21)
    lengths = np.concatenate([rng.integers(200, 400, size=30), rng.integers(5, 40, size=170)])  # skewed
    seqs = ["".join(_data.NCBI[c] for c in _data.random_protein(rng, int(n))) for n in lengths]
    query = "".join(_data.NCBI[c] for c in _data.random_protein(rng, 25))
    db = pyopal_amd.Database(seqs)
    one = pyopal_amd.Aligner()
    for threads in (0, 2, 3, 7):
        fake.uploads.clear(); fake.searches.clear(); db._mirrors.clear()
        for mode in ("score", "full"):
            got = list(pyopal_amd.align(query, db, mode=mode, algorithm="sw", threads=threads, ordered=True))
            assert [r.target_index for r in got] == list(range(len(seqs))), (threads, mode)
            fake_uploads = list(fake.uploads)
            want = one.align(query, db, mode=mode, algorithm="sw")   # one (fake) device, whole database
            del fake.uploads[len(fake_uploads):]
            assert [r.score for r in got] == [r.score for r in want]
            if mode == "full":
                key = lambda r: (r.query_start, r.target_start, r.query_end, r.target_end, r.alignment)  # noqa: E731
                assert [key(r) for r in got] == [key(r) for r in want]
        off = np.concatenate([[0], np.cumsum(lengths)])
        bounds = shard.balanced_bounds(off, 2)
        assert 0 < bounds[1] < 60, bounds   # balanced by residues, not by count (count would give 100)
        shard_uploads = sorted(set(u for u in fake.uploads if u[1] != len(seqs)))
        assert shard_uploads == [(0, bounds[1], int(off[bounds[1]])),
                                 (1, len(seqs) - bounds[1], int(off[-1] - off[bounds[1]]))], fake.uploads
        # every search stayed inside its device's shard, in shard coordinates
        for device, start, end in fake.searches:
            width = bounds[device + 1] - bounds[device]
            assert 0 <= start <= end <= max(width, len(seqs)), (device, start, end)
    # unordered: same multiset of results
    got = sorted((r.target_index, r.score) for r in pyopal_amd.align(query, db, threads=4))
    assert got == [(r.target_index, r.score) for r in one.align(query, db)]
    # a mutation drops every shard mirror
    db.append("ACDE")
    assert [r.target_index for r in pyopal_amd.align(query, db, threads=2, ordered=True)] == list(range(len(seqs) + 1))


def test_align_empty_and_single_target_with_several_devices(monkeypatch):
    # an empty database yields nothing however many GPUs are visible (src/pyopal/_align.py:129-141:
    # threads = 1, the aligner returns an empty list), and one target is one chunk on one GPU
    _FakeDevices(monkeypatch, 2)
    import pyopal_amd
    for threads in (0, 1, 2, 5):
        assert list(pyopal_amd.align("ACDE", pyopal_amd.Database(), threads=threads)) == []
        assert list(pyopal_amd.align("ACDE", [], threads=threads, ordered=True)) == []
    one = pyopal_amd.Database(["ACDEFG"])
    want = pyopal_amd.Aligner().align("ACDE", one, mode="full")
    for threads in (0, 1, 3):
        got = list(pyopal_amd.align("ACDE", one, mode="full", threads=threads, ordered=True))
        assert [(r.target_index, r.score, r.alignment) for r in got] == \
               [(r.target_index, r.score, r.alignment) for r in want]


def test_align_survives_a_mutation_between_sharding_and_search(monkeypatch):
    # the shard bounds are cut under a read lock that is released before the chunks are searched: a
    # search that finds the database mutated since uses the whole-database mirror of its device
    # instead of failing on a shard that no longer exists
    import numpy as np
    fake = _FakeDevices(monkeypatch, 2)
    import pyopal_amd
    import _data
    rng = np.random.default_rng(5)
    seqs = ["".join(_data.NCBI[c] for c in _data.random_protein(rng, int(n))) for n in rng.integers(5, 60, size=40)]
    db = pyopal_amd.Database(seqs)
    aligner = pyopal_amd.Aligner()
    want = [r.score for r in aligner.align("ACDEFGHIKL", db)]
    stale = db._version
    db.append("ACDEFGHIKLMN")    # shrinks nothing, but every shard cut before it is void
    got = aligner.align("ACDEFGHIKL", db, start=10, end=20, device=1, shard=(8, 25), shard_version=stale)
    assert [r.target_index for r in got] == list(range(10, 20))
    assert [r.score for r in got] == want[10:20]
    assert fake.uploads[-1][1] == len(seqs) + 1     # the whole database went to device 1, not the stale shard
    while len(db) > 5:
        del db[len(db) - 1]
    got = aligner.align("ACDEFGHIKL", db, start=2, end=30, device=0, shard=(0, 20), shard_version=stale)
    assert [r.target_index for r in got] == [2, 3, 4]
    try:
        aligner.align("ACDEFGHIKL", db, start=10, end=20, device=0, shard=(8, 25), shard_version=stale)
    except IndexError:
        pass
    else:
        raise AssertionError("a slice past the end of the shrunken database must raise IndexError")
