"""Whole-database checksums of configs[2] (SW full, 53-aa README query vs 1M x 300, seed 1,
BLOSUM62, gap 3/1) from the scalar CPU checker: the constants of
tests/test_gpu_fullsize.py::test_cfg3_every_alignment. No GPU involved.

    python tests/golden/make_cfg3_checksum.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _data  # noqa: E402
import _oracle  # noqa: E402
from pyopal_amd.matrices import ScoringMatrix  # noqa: E402

B62 = np.array(ScoringMatrix.from_name("BLOSUM62").int_array(), dtype=np.int32)
rng = np.random.default_rng(1)
res, off = _data.random_db(rng, np.full(1_000_000, 300))
q = _oracle.encode(_data.README_QUERY)
ref = _oracle.search_parallel(q, res, off, B62, 3, 1, "full", "sw", os.cpu_count() or 1)
print("CFG3_SCORE_SUM =", int(ref["score"].sum()))
print("CFG3_ALIGNMENT_BYTES =", int(ref["aln_off"][-1]))
print("CFG3_OPS_CRC32 =", hex(zlib.crc32(np.ascontiguousarray(ref["aln_flat"]).tobytes())))
