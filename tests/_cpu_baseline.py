"""ctypes binding to oracle/cpu_simd_baseline.c (own AVX2 SW baseline; test/bench only)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "libopal_cpu_simd.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.cpuSimdPrepare.restype = ctypes.c_void_p
        _LIB.cpuSimdPrepare.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        _LIB.cpuSimdFree.argtypes = [ctypes.c_void_p]
        _LIB.cpuSimdSearchSW.restype = ctypes.c_int
        _LIB.cpuSimdSearchSW.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_int]
        _LIB.cpuSimdSearchGlobal.restype = ctypes.c_int
        _LIB.cpuSimdSearchGlobal.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int]
        _LIB.cpuSimdThreads.restype = ctypes.c_int
    return _LIB


class CpuDatabase:
    def __init__(self, residues, offsets, alphabet_length=24):
        self.residues = np.ascontiguousarray(residues, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n = len(offsets) - 1
        self.A = alphabet_length
        self.h = lib().cpuSimdPrepare(self.residues.ctypes.data, self.offsets.ctypes.data, self.n, self.A)

    def search_sw(self, query, matrix, gap_open=3, gap_extend=1, threads=0):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        S = np.ascontiguousarray(matrix, dtype=np.int32)
        out = np.zeros(self.n, dtype=np.int32)
        rc = lib().cpuSimdSearchSW(self.h, q.ctypes.data, len(q), gap_open, gap_extend, S.ctypes.data,
                                   self.A, out.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"cpu baseline failed ({rc})")
        return out

    def search(self, query, matrix, gap_open=3, gap_extend=1, algorithm="sw", threads=0):
        """Scores under any of the four algorithms (nw / hw / ov: 16-bit lanes + 64-bit scalar)."""
        if algorithm == "sw":
            return self.search_sw(query, matrix, gap_open, gap_extend, threads)
        q = np.ascontiguousarray(query, dtype=np.uint8)
        S = np.ascontiguousarray(matrix, dtype=np.int32)
        out = np.zeros(self.n, dtype=np.int32)
        rc = lib().cpuSimdSearchGlobal(self.h, q.ctypes.data, len(q), gap_open, gap_extend, S.ctypes.data,
                                       self.A, {"nw": 0, "hw": 1, "ov": 2}[algorithm], out.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"cpu baseline failed ({rc})")
        return out

    def close(self):
        if self.h:
            lib().cpuSimdFree(self.h)
            self.h = None

    def __del__(self):
        self.close()


def max_threads():
    return int(lib().cpuSimdThreads())
