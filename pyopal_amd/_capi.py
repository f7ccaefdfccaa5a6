"""ctypes binding to libmiopal.so, the C ABI declared in include/opal.h and
include/miopal.h. This is the only way Python code in this repository reaches
the HIP kernels; there is no CPU fallback behind it."""

from __future__ import annotations

import ctypes
import os
import typing

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIOPAL_LIBRARY: another build of the same library (A/B timing of kernel variants, tools/ab_build.sh)
LIB_PATH = os.environ.get("MIOPAL_LIBRARY") or os.path.join(_HERE, "libmiopal.so")

OPAL_ERR_OVERFLOW = 1
OPAL_ERR_NO_SIMD_SUPPORT = 2
OPAL_ERR_INVALID_MODE = 3

SEARCH = {"score": 0, "end": 1, "full": 2}
MODE = {"nw": 0, "hw": 1, "ov": 2, "sw": 3}
OVERFLOW = {"simple": 0, "buckets": 1}


class OpalSearchResult(ctypes.Structure):
    _fields_ = [
        ("scoreSet", ctypes.c_int),
        ("score", ctypes.c_int),
        ("endLocationTarget", ctypes.c_int),
        ("endLocationQuery", ctypes.c_int),
        ("startLocationTarget", ctypes.c_int),
        ("startLocationQuery", ctypes.c_int),
        ("alignment", ctypes.POINTER(ctypes.c_ubyte)),
        ("alignmentLength", ctypes.c_int),
    ]


_lib = None
_libc = None

EXPORTS = [
    # opal.h
    "opalInitSearchResult", "opalSearchResultIsEmpty", "opalSearchResultSetScore",
    "opalSearchDatabase", "opalSearchDatabaseCharSW",
    # miopal.h
    "miopalDeviceCount", "miopalLastError", "miopalDbCreate", "miopalDbCreateFlat", "miopalDbCreateSubset",
    "miopalDbDestroy", "miopalDbCount", "miopalDbTotalLength", "miopalDbDeviceBytes",
    "miopalSearch", "miopalSearchFlat", "miopalSearchFlatInto", "miopalSearchDeviceScores", "miopalSetProfiling", "miopalLastKernelTime",
    "miopalLastRouting", "miopalLastFullRouting", "miopalSearchResults", "miopalReleaseCaches",
    "miopalSetTuning", "miopalGetTuning", "miopalDbSetOption", "miopalDbReleaseWorkspaces",
    # test hooks
    "miopalSelfTest", "miopalTestInjectFault", "miopalTestSetLogicalDevices",
]


def _share_hip_runtime() -> None:
    """Make libmiopal.so and PyTorch-ROCm use ONE HIP runtime.

    PyTorch wheels bundle their own ``libamdhip64.so.7`` (torch/lib). A process
    that loads the system copy first (as libmiopal.so's RUNPATH would) and
    torch's copy second ends up with two runtimes, and the second one reports
    "No HIP GPUs are available". Loading torch's copy first, when torch is
    installed, lets libmiopal.so bind to it by SONAME; torch itself is not
    imported.
    """
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib() -> ctypes.CDLL:
    """Load libmiopal.so (built in-tree by ``__graft_entry__.build()`` or
    ``make -C pyopal_amd/csrc``). Raises if it is missing."""
    global _lib, _libc
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()')")
        _share_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        c_int, c_i64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
        L.miopalDeviceCount.restype = c_int
        L.miopalLastError.restype = ctypes.c_char_p
        L.miopalDbCreate.restype = c_int
        L.miopalDbCreate.argtypes = [ctypes.POINTER(c_vp), c_vp, c_vp, c_i64, c_int, c_int]
        L.miopalDbCreateFlat.restype = c_int
        L.miopalDbCreateFlat.argtypes = [ctypes.POINTER(c_vp), c_vp, c_vp, c_i64, c_int, c_int]
        L.miopalDbCreateSubset.restype = c_int
        L.miopalDbCreateSubset.argtypes = [ctypes.POINTER(c_vp), c_vp, c_vp, c_i64]
        L.miopalSelfTest.restype = c_int
        L.miopalSelfTest.argtypes = [c_int]
        L.miopalTestInjectFault.restype = None
        L.miopalTestInjectFault.argtypes = [c_int, c_int, c_int]
        L.miopalSetTuning.restype = c_int
        L.miopalSetTuning.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.miopalGetTuning.restype = ctypes.c_char_p
        L.miopalGetTuning.argtypes = [ctypes.c_char_p]
        L.miopalDbSetOption.restype = c_int
        L.miopalDbSetOption.argtypes = [c_vp, ctypes.c_char_p, c_i64]
        L.miopalTestSetLogicalDevices.restype = c_int
        L.miopalTestSetLogicalDevices.argtypes = [c_int]
        L.miopalDbDestroy.restype = None
        L.miopalDbDestroy.argtypes = [c_vp]
        for name in ("miopalDbCount", "miopalDbTotalLength", "miopalDbDeviceBytes", "miopalDbReleaseWorkspaces"):
            getattr(L, name).restype = c_i64
            getattr(L, name).argtypes = [c_vp]
        L.miopalSearch.restype = c_int
        L.miopalSearch.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int,
                                   c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.miopalSearchFlat.restype = c_int
        L.miopalSearchFlat.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int,
                                       c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       ctypes.POINTER(c_vp), c_vp]
        L.miopalSearchFlatInto.restype = c_int
        L.miopalSearchFlatInto.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int,
                                           c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), c_vp]
        L.miopalSearchDeviceScores.restype = c_int
        L.miopalSearchDeviceScores.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int,
                                               c_i64, c_i64, c_vp, c_vp]
        L.miopalSetProfiling.restype = None
        L.miopalSetProfiling.argtypes = [c_vp, c_int]
        L.miopalReleaseCaches.restype = None
        L.miopalReleaseCaches.argtypes = []
        L.miopalLastFullRouting.restype = c_int
        L.miopalLastFullRouting.argtypes = []
        L.miopalLastRouting.restype = None
        L.miopalLastRouting.argtypes = [ctypes.POINTER(ctypes.c_int64)]
        L.miopalLastKernelTime.restype = c_int
        L.miopalLastKernelTime.argtypes = [c_vp, ctypes.POINTER(ctypes.c_float)]
        L.miopalSearchResults.restype = c_int
        L.miopalSearchResults.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int,
                                          c_int, c_int, c_i64, c_i64]
        L.opalSearchDatabase.restype = c_int
        L.opalSearchDatabase.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int,
                                         c_vp, c_int, c_int, c_int]
        _lib = L
        _libc = ctypes.CDLL(None)
        _libc.free.argtypes = [c_vp]
        _libc.free.restype = None
    return _lib


def last_error() -> str:
    msg = lib().miopalLastError()
    return msg.decode("utf-8", "replace") if msg else ""


def raise_for(rc: int) -> None:
    """Error mapping of the reference plugin (src/pyopal/platform/pyx.in:102-107)."""
    if rc == 0:
        return
    if rc == OPAL_ERR_NO_SIMD_SUPPORT:
        raise RuntimeError("no supported SIMD backend available")
    if rc == OPAL_ERR_OVERFLOW:
        raise OverflowError("overflow detected while computing alignment scores")
    detail = last_error()
    raise RuntimeError(f"failed to align to Opal database (code={rc})" + (f": {detail}" if detail else ""))


def set_tuning(name: str, value: typing.Optional[str]) -> None:
    """Set (or, with None, unset) a tuning switch of the library for the searches that start from now on
    (include/miopal.h, miopalSetTuning; pyopal_amd/csrc/tuning.h lists the switches). The environment is
    only read once, when the library first looks at a switch: later changes go through here."""
    rc = lib().miopalSetTuning(name.encode(), None if value is None else str(value).encode())
    raise_for(rc)


def get_tuning(name: str) -> typing.Optional[str]:
    v = lib().miopalGetTuning(name.encode())
    return None if v is None else v.decode()


class tuning:
    """``with tuning(NO_BIASED="1", PAIR_STRIPS=None): ...`` - switches set (None: unset) for the block and
    put back afterwards. Process-wide, like the environment they replace: for tests and A/B tools."""

    def __init__(self, **switches):
        self._want = switches
        self._saved = {}

    def __enter__(self):
        for name, value in self._want.items():
            self._saved[name] = get_tuning(name)
            set_tuning(name, value)
        return self

    def __exit__(self, *exc):
        for name, value in self._saved.items():
            set_tuning(name, value)
        return False


def _ptr(a: typing.Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


class _MallocBytes:
    """Owner of a buffer returned by the C ABI; numpy arrays made from it keep it alive."""

    def __init__(self, address: int, size: int, capacity: int = 0):
        self._address = address
        self._capacity = max(capacity, size)
        self.__array_interface__ = {"data": (address, False), "shape": (size,), "typestr": "|u1",
                                    "version": 3}

    def view(self, size: int) -> "_MallocView":
        return _MallocView(self, size)

    def __del__(self):
        if _libc is not None:   # (gone at interpreter shutdown: the process is about to return its memory anyway)
            _libc.free(ctypes.c_void_p(self._address))


class _MallocView:
    """The first ``size`` bytes of a _MallocBytes buffer that was written again (``reuse=``)."""

    def __init__(self, owner: _MallocBytes, size: int):
        self._owner = owner
        self.__array_interface__ = {"data": (owner._address, False), "shape": (size,), "typestr": "|u1",
                                    "version": 3}


class _LazyAlignments:
    """List-like view of the per-target operation arrays inside the flat buffer."""

    def __init__(self, flat: np.ndarray, offsets: np.ndarray):
        self._flat = flat
        self._off = offsets

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        return self._flat[self._off[k]:self._off[k + 1]]

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]


class DeviceDatabase:
    """Owner of a MiopalDb handle (device-resident database)."""

    def __init__(self, residues: np.ndarray, offsets: np.ndarray, alphabet_length: int, device: int = 0):
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        handle = ctypes.c_void_p()
        rc = lib().miopalDbCreateFlat(ctypes.byref(handle), _ptr(residues), _ptr(offsets),
                                      len(offsets) - 1, alphabet_length, device)
        raise_for(rc)
        self._h = handle
        self.count = len(offsets) - 1
        self.offsets = offsets            # host copy: target k is residues[offsets[k]:offsets[k + 1]]
        self.alphabet_length = alphabet_length
        self.device = device

    @property
    def handle(self):
        return self._h

    def subset(self, indices) -> "DeviceDatabase":
        """A new resident database holding the targets ``indices`` of this one, gathered on the device
        from this one's residues (miopalDbCreateSubset: nothing crosses PCIe)."""
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        handle = ctypes.c_void_p()
        rc = lib().miopalDbCreateSubset(ctypes.byref(handle), self._h, _ptr(idx) if len(idx) else None, len(idx))
        raise_for(rc)
        sub = DeviceDatabase.__new__(DeviceDatabase)
        sub._h = handle
        sub.count = len(idx)
        lengths = np.diff(self.offsets)[idx] if len(idx) else np.zeros(0, dtype=np.int64)
        sub.offsets = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(lengths, out=sub.offsets[1:])
        sub.alphabet_length = self.alphabet_length
        sub.device = self.device
        return sub

    def close(self):
        if getattr(self, "_h", None):
            lib().miopalDbDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return int(lib().miopalDbDeviceBytes(self._h))

    def release_workspaces(self) -> int:
        """Free the idle per-search workspaces parked on the handle; returns the device bytes released."""
        return int(lib().miopalDbReleaseWorkspaces(self._h))

    def set_option(self, name: str, value: int) -> None:
        """Per-handle option (include/miopal.h, miopalDbSetOption): "reserve_cus", "small_search"."""
        raise_for(lib().miopalDbSetOption(self._h, name.encode(), int(value)))

    def search(self, query: np.ndarray, matrix: np.ndarray, gap_open: int = 3, gap_extend: int = 1,
               mode: str = "score", algorithm: str = "sw", start: int = 0,
               end: typing.Optional[int] = None,
               score_out: typing.Optional[np.ndarray] = None,
               reuse: typing.Optional[typing.Dict[str, typing.Any]] = None) -> typing.Dict[str, typing.Any]:
        """miopalSearch with numpy outputs (same keys as tests/_oracle.search). ``score_out``: an
        int32 array of end - start entries to receive the scores (a caller that re-uses its result
        array; when it is pinned, device-visible host memory the kernel writes into it directly).
        ``reuse``: the result of an earlier search of the same slice and search type whose per-target
        arrays (scores, locations, operation offsets) and operations buffer are written again instead of
        fresh ones - the earlier result's arrays then hold the new values (a million targets: 36 MB of
        per-target pages and 67-360 MB of operations that need not be faulted in again, nor unmapped when
        the earlier result goes: 1.3 ms of a 11.5-ms `full` search for the arrays, 17 of 113 ms for the
        operations of a 300-residue query). An operations buffer that is too small stays with the earlier
        result, untouched."""
        end = self.count if end is None else min(end, self.count)
        n = max(end - start, 0)
        q = np.ascontiguousarray(query, dtype=np.uint8)
        S = np.ascontiguousarray(matrix, dtype=np.int32)
        st = SEARCH[mode]
        if score_out is not None and (score_out.dtype != np.int32 or score_out.shape != (n,) or
                                      not score_out.flags.c_contiguous):
            raise ValueError("score_out must be a contiguous int32 array with one entry per target of the slice")
        # the C side writes every entry of its outputs (locations of empty alignments are -1)
        def array(key, length, dtype, zeros=False):
            old = reuse.get(key) if reuse else None
            if (isinstance(old, np.ndarray) and old.dtype == dtype and old.shape == (length,) and
                    old.flags.c_contiguous and old.flags.writeable):
                if zeros:
                    old[:] = 0
                return old
            return np.zeros(length, dtype=dtype) if zeros else np.empty(length, dtype=dtype)

        out = {"score": score_out if score_out is not None else array("score", n, np.int32)}
        et = eq = s_t = s_q = aoff = None
        ops_ptr = ctypes.c_void_p()
        ops_cap = ctypes.c_int64(0)
        lent = reuse.get("_ops_owner") if reuse and st == 2 else None
        if isinstance(lent, _MallocBytes):
            ops_ptr = ctypes.c_void_p(lent._address)
            ops_cap = ctypes.c_int64(lent._capacity)
        else:
            lent = None
        if st >= 1:
            et = array("end_t", n, np.int32)
            eq = array("end_q", n, np.int32)
        if st == 2:
            s_t = array("start_t", n, np.int32)
            s_q = array("start_q", n, np.int32)
            aoff = array("aln_off", n + 1, np.int64, zeros=True)
        rc = lib().miopalSearchFlatInto(self._h, _ptr(q), len(q), gap_open, gap_extend, _ptr(S),
                                        self.alphabet_length, st, MODE[algorithm], start, end,
                                        _ptr(out["score"]), _ptr(et), _ptr(eq), _ptr(s_t), _ptr(s_q),
                                        ctypes.byref(ops_ptr), ctypes.byref(ops_cap), _ptr(aoff))
        raise_for(rc)
        if st >= 1:
            out.update(end_t=et, end_q=eq)
        if st == 2:
            total = int(aoff[-1]) if n else 0
            owner = None
            if lent is not None and ops_ptr.value == lent._address:
                # written in place: the earlier result's buffer, owned as before (and seen through its arrays too)
                owner = lent
                owner._capacity = max(owner._capacity, int(ops_cap.value))
                flat = np.asarray(owner.view(total))
            elif ops_ptr.value:
                # the array takes the malloc'ed buffer over (freed with the array) - no copy
                owner = _MallocBytes(ops_ptr.value, total, int(ops_cap.value))
                flat = np.asarray(owner)
            else:
                flat = np.zeros(0, dtype=np.uint8)
            out.update(start_t=s_t, start_q=s_q, aln_flat=flat, aln_off=aoff,
                       aln=_LazyAlignments(flat, aoff), _ops_owner=owner)
        return out

    def search_device_scores(self, query: np.ndarray, matrix: np.ndarray, device_ptr: int,
                             stream: int = 0, gap_open: int = 3, gap_extend: int = 1,
                             algorithm: str = "sw", start: int = 0,
                             end: typing.Optional[int] = None) -> None:
        end = self.count if end is None else min(end, self.count)
        q = np.ascontiguousarray(query, dtype=np.uint8)
        S = np.ascontiguousarray(matrix, dtype=np.int32)
        rc = lib().miopalSearchDeviceScores(self._h, _ptr(q), len(q), gap_open, gap_extend, _ptr(S),
                                            self.alphabet_length, MODE[algorithm], start, end,
                                            ctypes.c_void_p(device_ptr), ctypes.c_void_p(stream))
        raise_for(rc)

    @staticmethod
    def last_routing() -> typing.Tuple[int, int, int, int]:
        """(targets on the wavefront-per-pair kernel, reserved, groups of 128 targets on the
        lane-per-target kernel, targets recomputed after leaving the 16-bit range) for the
        calling thread's most recent search."""
        counts = (ctypes.c_int64 * 4)()
        lib().miopalLastRouting(counts)
        return tuple(int(c) for c in counts)

    @staticmethod
    def last_full_routing() -> int:
        """Bits describing the start-cell and direction passes of the calling thread's most recent
        `full` search (include/miopal.h, miopalLastFullRouting)."""
        return int(lib().miopalLastFullRouting())

    def set_profiling(self, enabled: bool) -> None:
        lib().miopalSetProfiling(self._h, 1 if enabled else 0)

    def last_kernel_time(self) -> typing.Tuple[int, float]:
        ms = ctypes.c_float(0)
        n = lib().miopalLastKernelTime(self._h, ctypes.byref(ms))
        return int(n), float(ms.value)
