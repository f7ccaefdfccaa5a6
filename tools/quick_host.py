"""Scratch timing: the headline search with the scores left in HBM (miopalSearchDeviceScores) and
delivered to a host buffer (miopalSearch), with the direct scatter on and off. Usage: quick_host.py [N] [L]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _data
from pyopal_amd import _capi
from pyopal_amd.matrices import ScoringMatrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
m = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
q = _data.encode(_data.README_QUERY)
lengths = np.full(N, L) if L > 0 else np.clip(rng.lognormal(5.55, 0.6, size=N), 20, 8000).astype(np.int64)
res, off = _data.random_db(rng, lengths)
db = _capi.DeviceDatabase(res, off, 24)
out = torch.zeros(N, dtype=torch.int32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
cells = float(len(q)) * float(off[-1])
ref = None
for label, env in (("direct scatter", {}), ("no host scatter", {"MIOPAL_NO_HOST_SCATTER": "1"}),
                   ("no direct scatter", {"MIOPAL_NO_DIRECT_SCATTER": "1"}), ("direct scatter", {})):
    for k in ("MIOPAL_NO_HOST_SCATTER", "MIOPAL_NO_DIRECT_SCATTER"):
        _capi.set_tuning(k, None)
    for name, value in env.items():
        _capi.set_tuning(name, value)
    for _ in range(5):
        db.search_device_scores(q, m, out.data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40):
        db.search_device_scores(q, m, out.data_ptr(), stream, 3, 1, "sw")
    torch.cuda.synchronize(); dev = (time.perf_counter() - t) / 40
    for _ in range(5):
        got = db.search(q, m, 3, 1, "score", "sw")["score"]
    ts = []
    for _ in range(40):
        t = time.perf_counter(); got = db.search(q, m, 3, 1, "score", "sw")["score"]; ts.append(time.perf_counter() - t)
    host = float(np.median(ts))
    reuse = np.empty(N, dtype=np.int32)
    pinned_t = torch.empty(N, dtype=torch.int32).pin_memory()
    pinned = pinned_t.numpy()
    extra = []
    for buf in (reuse, pinned):
        for _ in range(5):
            db.search(q, m, 3, 1, "score", "sw", score_out=buf)
        tt = []
        for _ in range(40):
            t = time.perf_counter(); db.search(q, m, 3, 1, "score", "sw", score_out=buf); tt.append(time.perf_counter() - t)
        assert np.array_equal(buf, got)
        extra.append(float(np.median(tt)))
    if ref is None:
        ref = got.copy()
    same = bool(np.array_equal(got, ref)) and bool(np.array_equal(out.cpu().numpy(), ref))
    print(f"{label:18s}: device results {dev*1e3:.3f} ms = {cells/dev/1e9:7.0f} GCUPS | host results {host*1e3:.3f} ms = "
          f"{cells/host/1e9:7.0f} GCUPS (min {min(ts)*1e3:.3f}) | re-used array {extra[0]*1e3:.3f} ms | pinned array {extra[1]*1e3:.3f} ms = {cells/extra[1]/1e9:7.0f} GCUPS | equal {same}", flush=True)
    assert same
