"""Host-side mirror of ``pyopal.lib`` for the MI355X search path.

Same names, argument meaning and error behaviour as the reference's Cython
module (``src/pyopal/lib.pyx``); the computation behind `Aligner.align` is the
HIP path reached through the C ABI (``pyopal_amd._capi``), never a CPU
fallback. Per-residue loops of the reference (encoding, result fill) are
numpy table look-ups / bulk array reads here.
"""

from __future__ import annotations

import array
import threading
import typing
import weakref

import numpy as np

from . import _capi
from .matrices import ScoringMatrix

__version__ = "0.1.0"

MAX_ALPHABET_SIZE = 32  # src/pyopal/lib.pxd:28-32
UINT32_MAX = 0xFFFFFFFF

_OPAL_SEARCH_MODES = dict(_capi.SEARCH)        # src/pyopal/lib.pyx:73-77
_OPAL_OVERFLOW_MODES = dict(_capi.OVERFLOW)    # src/pyopal/lib.pyx:85-88
_OPAL_ALGORITHMS = dict(_capi.MODE)            # src/pyopal/lib.pyx:90-95
_OPAL_ALIGNMENT_OPERATION = {"M": 0, "D": 1, "I": 2, "X": 3}  # src/pyopal/lib.pyx:97-102


# --- Read/write lock ------------------------------------------------------------

class SharedMutex:
    """Readers-writer lock with the surface of ``src/pyopal/lib.pyx:153-181``
    (``lock.read`` / ``lock.write`` context managers). Searches hold the read
    side while the C call runs without the GIL; mutators hold the write side."""

    def __init__(self):
        self._cond = threading.Condition(threading.Lock())
        self._readers = 0
        self._writer = False
        self.read = ReadLock(self)
        self.write = WriteLock(self)


class ReadLock:
    def __init__(self, owner: SharedMutex):
        self.owner = owner

    def __enter__(self):
        o = self.owner
        with o._cond:
            while o._writer:
                o._cond.wait()
            o._readers += 1

    def __exit__(self, exc_type, exc_value, traceback):
        o = self.owner
        with o._cond:
            o._readers -= 1
            if o._readers == 0:
                o._cond.notify_all()


class WriteLock:
    def __init__(self, owner: SharedMutex):
        self.owner = owner

    def __enter__(self):
        o = self.owner
        with o._cond:
            while o._writer or o._readers:
                o._cond.wait()
            o._writer = True

    def __exit__(self, exc_type, exc_value, traceback):
        o = self.owner
        with o._cond:
            o._writer = False
            o._cond.notify_all()


# --- Alphabet -------------------------------------------------------------------

class Alphabet:
    """Ordinal encoding of sequences (``src/pyopal/lib.pyx:186-332``).

    Letters outside the alphabet (lower case included: there is no case
    folding) map to the index of ``*``; non-letters are rejected.
    """

    _DEFAULT_LETTERS = "ARNDCQEGHILKMFPSTWYVBZX*"
    __slots__ = ("letters", "length", "_unknown", "_table", "_trans", "_letters")

    def __init__(self, letters: str = _DEFAULT_LETTERS):
        if not isinstance(letters, str):
            raise TypeError(f"expected str, found {type(letters).__name__}")
        if len(letters) != len(set(letters)):
            raise ValueError("duplicate symbols in alphabet letters")
        if any(x != "*" and not x.isupper() for x in letters):
            raise ValueError("alphabet must only contain uppercase characters or wildcard")
        if len(letters) > MAX_ALPHABET_SIZE:
            raise ValueError("Cannot use alphabet of more than 32 symbols")
        self.letters = letters
        self.length = len(letters)
        self._unknown = letters.find("*")
        self._letters = np.frombuffer(letters.encode("ascii"), dtype=np.uint8).copy()
        # 255: not a letter; 254: letter without a code (alphabet has no wildcard)
        table = np.full(256, 255, dtype=np.uint8)
        for c in range(256):
            if (65 <= c <= 90) or (97 <= c <= 122):
                table[c] = self._unknown if self._unknown >= 0 else 254
        for i, x in enumerate(self._letters):
            if (65 <= x <= 90) or (97 <= x <= 122):
                table[x] = i
        self._table = table
        self._trans = table.tobytes()  # 256-entry table for bytes.translate

    def __len__(self):
        return self.length

    def __contains__(self, item):
        return item in self.letters

    def __getitem__(self, index: int):
        index_ = index
        if index_ < 0:
            index_ += self.length
        if index_ < 0 or index_ >= self.length:
            raise IndexError(index)
        return self.letters[index_]

    def __reduce__(self):
        return type(self), (self.letters,)

    def __repr__(self):
        if self.letters == self._DEFAULT_LETTERS:
            return f"{type(self).__name__}()"
        return f"{type(self).__name__}({self.letters!r})"

    def __str__(self):
        return self.letters

    def __eq__(self, item):
        if isinstance(item, str):
            return self.letters == item
        if isinstance(item, Alphabet):
            return self.letters == item.letters
        return False

    def __hash__(self):
        return hash(self.letters)

    # -- bulk forms used by Database --------------------------------------------
    def _encode_bytes(self, sequence) -> bytes:
        """Ordinal encoding of a str / bytes-like sequence in one C-level pass."""
        if isinstance(sequence, str):
            sequence = sequence.encode("ascii", "replace")
        elif not isinstance(sequence, (bytes, bytearray)):
            sequence = memoryview(sequence).cast("B").tobytes()
        codes = sequence.translate(self._trans)
        if b"\xff" in codes or b"\xfe" in codes:
            bad = [p for p in (codes.find(b"\xff"), codes.find(b"\xfe")) if p >= 0]
            pos = min(bad)
            letter = sequence[pos]
            if codes[pos] == 255:
                raise ValueError(f"character outside ASCII range: {letter!r}")
            raise ValueError(f"non-alphabet character in sequence: {chr(letter)!r}")
        return bytes(codes)

    def _encode_array(self, sequence) -> np.ndarray:
        return np.frombuffer(self._encode_bytes(sequence), dtype=np.uint8)

    def encode_into(self, sequence, encoded) -> None:
        src = memoryview(sequence).cast("B")
        dst = memoryview(encoded).cast("B")
        if len(src) != len(dst):
            raise ValueError("Buffers do not have the same dimensions")
        dst[:] = self._encode_bytes(src)

    def decode_into(self, encoded, sequence) -> None:
        src = np.frombuffer(memoryview(encoded).cast("B"), dtype=np.uint8)
        dst = memoryview(sequence).cast("B")
        if len(src) != len(dst):
            raise ValueError("Buffers do not have the same dimensions")
        if src.size and src.max() >= self.length:
            code = int(src[np.argmax(src >= self.length)])
            raise ValueError(f"invalid index in encoded sequence: {code!r}")
        dst[:] = self._letters[src].tobytes()

    def encode(self, sequence) -> bytes:
        return self._encode_bytes(sequence)

    def decode(self, encoded) -> str:
        decoded = bytearray(len(encoded))
        self.decode_into(encoded, decoded)
        return decoded.decode("ascii")


# --- Sequence storage -------------------------------------------------------------

class BaseDatabase:
    """Base class of sequence databases (``src/pyopal/lib.pyx:337-466``).

    The reference's three C-level accessors (``get_sequences``/``get_lengths``/
    ``get_size``, ``src/pyopal/lib.pxd:90-92``) become `_get_size`,
    `_get_lengths` and `_get_encoded`; a subclass that implements them can be
    searched by `Aligner.align`.
    """

    _DEFAULT_ALPHABET = Alphabet()

    def __init__(self, sequences=(), alphabet=None):
        self.lock = SharedMutex()
        self._mirrors: typing.Dict[int, typing.Tuple[int, _capi.DeviceDatabase]] = {}
        self._mirror_guard = threading.Lock()
        self._version = 0
        if alphabet is None:
            self.alphabet = self._DEFAULT_ALPHABET
        elif isinstance(alphabet, Alphabet):
            self.alphabet = alphabet
        else:
            self.alphabet = Alphabet(alphabet)
        if sequences:
            raise TypeError("cannot create a `BaseDatabase` with sequences")

    # -- interface to implement ---------------------------------------------------
    def _get_size(self) -> int:
        return 0

    def _get_lengths(self) -> typing.Sequence[int]:
        raise NotImplementedError("BaseDatabase.get_lengths")

    def _get_encoded(self) -> typing.Sequence[bytes]:
        raise NotImplementedError("BaseDatabase.get_sequences")

    # -- device mirror -----------------------------------
    def _device_mirror(self, device: int = 0,
                       shard: typing.Optional[typing.Tuple[int, int]] = None) -> _capi.DeviceDatabase:
        """Packed copy of the database - or of its targets ``[lo, hi)`` only, when ``shard`` is
        given (`pyopal_amd.align` on several GPUs: every GPU holds 1/N of the residues) - in the
        HBM of ``device``, rebuilt only after a mutation. Called with the read lock held."""
        key = (device, shard)
        with self._mirror_guard:
            # mirrors of an older state of the database are dropped whichever is asked for
            for old in [k for k, (version, _) in self._mirrors.items() if version != self._version]:
                self._mirrors.pop(old)[1].close()
            entry = self._mirrors.get(key)
            if entry is not None:
                return entry[1]
            derived = self._mirror_from_parent(device) if shard is None else None
            if derived is not None:
                self._mirrors[key] = (self._version, derived)
                return derived
            seqs = self._get_encoded()
            lo, hi = (0, len(seqs)) if shard is None else shard
            if not 0 <= lo <= hi <= len(seqs):
                raise IndexError(f"shard [{lo}, {hi}) outside the database")
            if shard is not None:
                seqs = seqs[lo:hi]
            lengths = np.fromiter(self._get_lengths(), dtype=np.int64)[lo:hi]
            offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
            np.cumsum(lengths, out=offsets[1:])
            residues = np.frombuffer(b"".join(seqs), dtype=np.uint8)
            mirror = _capi.DeviceDatabase(residues, offsets, self.alphabet.length, device)
            self._mirrors[key] = (self._version, mirror)
            return mirror

    def _mirror_from_parent(self, device: int):
        """A subset made by `Database.mask` / `Database.extract` whose parent is still resident on
        ``device`` and unchanged gathers its residues there from the parent's mirror instead of
        uploading them again (the reference's subsets share the parent's
        sequence buffers, ``src/pyopal/lib.pyx:694-778``). None: build the mirror the ordinary way."""
        link = getattr(self, "_parent_link", None)
        if link is None:
            return None
        parent_ref, parent_version, my_version, indices = link
        parent = parent_ref()
        if parent is None or my_version != self._version:
            self._parent_link = None     # the parent is gone, or this subset was edited: nothing to share
            return None
        with parent._mirror_guard:
            entry = parent._mirrors.get((device, None))
            if entry is None or entry[0] != parent_version or parent._version != parent_version:
                return None
            return entry[1].subset(indices)

    def _invalidate(self) -> None:
        """Called with the write lock held by every mutator."""
        self._version += 1

    # -- properties -----------------------------------------------------------------
    @property
    def lengths(self) -> typing.List[int]:
        with self.lock.read:
            return [int(x) for x in self._get_lengths()][: self._get_size()]

    @property
    def total_length(self) -> int:
        with self.lock.read:
            if self._get_size() == 0:
                return 0
            return int(sum(self._get_lengths()))

    # -- sequence interface ---------------------------------------------------------
    def __contains__(self, query):
        encoded = self.alphabet.encode(query)
        with self.lock.read:
            if self._get_size() == 0:
                return False
            return any(s == encoded for s in self._get_encoded())

    def __len__(self):
        with self.lock.read:
            return self._get_size()

    def __getitem__(self, index: int):
        with self.lock.read:
            size = self._get_size()
            index_ = index
            if index_ < 0:
                index_ += size
            if index_ < 0 or index_ >= size:
                raise IndexError(index)
            return self.alphabet.decode(self._get_encoded()[index_])


class Database(BaseDatabase):
    """A database of target sequences, stored encoded
    (``src/pyopal/lib.pyx:469-778``)."""

    def __init__(self, sequences=(), alphabet=None):
        super().__init__(alphabet=alphabet)
        self._sequences: typing.List[bytes] = []
        self._lengths: typing.List[int] = []
        self.clear()
        self.extend(sequences)

    def __reduce__(self):
        return (type(self), ((), self.alphabet), None, iter(self))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # -- database interface -----------------------------------------------------------
    def _get_size(self) -> int:
        return len(self._sequences)

    def _get_lengths(self):
        return self._lengths

    def _get_encoded(self):
        return self._sequences

    def _encode(self, sequence) -> bytes:
        return self.alphabet.encode(sequence)

    # -- sequence interface -----------------------------------------------------------
    def __getitem__(self, index):
        if isinstance(index, slice):
            return self.extract(range(*index.indices(len(self))))
        return super().__getitem__(index)

    def __setitem__(self, index: int, sequence):
        encoded = self._encode(sequence)
        with self.lock.write:
            size = len(self._sequences)
            index_ = index
            if index_ < 0:
                index_ += size
            if index_ < 0 or index_ >= size:
                raise IndexError(index)
            self._sequences[index_] = encoded
            self._lengths[index_] = len(encoded)
            self._invalidate()

    def __delitem__(self, index: int):
        with self.lock.write:
            size = len(self._sequences)
            index_ = index
            if index_ < 0:
                index_ += size
            if index_ < 0 or index_ >= size:
                raise IndexError(index)
            del self._sequences[index_]
            del self._lengths[index_]
            self._invalidate()

    def clear(self) -> None:
        with self.lock.write:
            self._sequences.clear()
            self._lengths.clear()
            self._invalidate()

    def extend(self, sequences) -> None:
        # Bulk path (the reference encodes sequence by sequence, lib.pyx:586-636; here one
        # table look-up over the concatenation replaces a Python-level loop per residue):
        # lists of str / bytes are joined, encoded once and cut back at their lengths.
        if isinstance(sequences, (list, tuple)) and len(sequences) > 8:
            if all(type(x) is str for x in sequences):
                joined = "".join(sequences).encode("ascii", "replace")
            elif all(type(x) is bytes for x in sequences):
                joined = b"".join(sequences)
            else:
                joined = None
            if joined is not None:
                lengths = [len(x) for x in sequences]
                try:
                    encoded = self.alphabet.encode(joined)
                except ValueError:
                    encoded = None  # let the per-sequence path raise for the offending one
                if encoded is not None and len(encoded) == sum(lengths):
                    pieces = []
                    offset = 0
                    for n in lengths:
                        pieces.append(encoded[offset:offset + n])
                        offset += n
                    with self.lock.write:
                        self._sequences.extend(pieces)
                        self._lengths.extend(lengths)
                        self._invalidate()
                    return
        for sequence in sequences:
            self.append(sequence)

    def append(self, sequence) -> None:
        encoded = self._encode(sequence)
        with self.lock.write:
            self._sequences.append(encoded)
            self._lengths.append(len(encoded))
            self._invalidate()

    def reverse(self) -> None:
        with self.lock.write:
            self._sequences.reverse()
            self._lengths.reverse()
            self._invalidate()

    def insert(self, index: int, sequence) -> None:
        encoded = self._encode(sequence)
        with self.lock.write:
            size = len(self._sequences)
            index_ = index
            if index_ < 0:
                index_ += size
            if index_ < 0:
                index_ = 0
            elif index_ >= size:
                index_ = size
            self._sequences.insert(index_, encoded)
            self._lengths.insert(index_, len(encoded))
            self._invalidate()

    # -- subsets (share the encoded buffers, like the reference's shared_ptr) ------------
    def _subset(self) -> "Database":
        subdb = Database.__new__(Database)
        BaseDatabase.__init__(subdb, alphabet=self.alphabet)
        subdb._sequences = []
        subdb._lengths = []
        return subdb

    def _link_subset(self, subdb: "Database", picked: typing.List[int]) -> None:
        """(read lock held) Remember where the subset came from: while this database and the subset
        stay as they are, the subset's device mirror is gathered from this one's on the device."""
        picked = np.asarray(picked, dtype=np.int64)
        link = getattr(self, "_parent_link", None)
        if link is not None and link[2] == self._version and link[0]() is not None:
            # a subset of an (unedited) subset: linked to the database at the root, which is the one
            # likely to be resident
            subdb._parent_link = (link[0], link[1], subdb._version, link[3][picked])
        else:
            subdb._parent_link = (weakref.ref(self), self._version, subdb._version, picked)

    def mask(self, bitmask) -> "Database":
        subdb = self._subset()
        picked = []
        with self.lock.read:
            size = self._get_size()
            i = 0
            for b in bitmask:
                if i >= size:
                    raise IndexError(bitmask)
                if b:
                    subdb._sequences.append(self._sequences[i])
                    subdb._lengths.append(self._lengths[i])
                    picked.append(i)
                i += 1
            if i < size:
                raise IndexError(bitmask)
            self._link_subset(subdb, picked)
        return subdb

    def extract(self, indices) -> "Database":
        subdb = self._subset()
        picked = []
        with self.lock.read:
            size = self._get_size()
            for index in indices:
                if index < 0 or index >= size:
                    raise IndexError(index)
                subdb._sequences.append(self._sequences[index])
                subdb._lengths.append(self._lengths[index])
                picked.append(index)
            self._link_subset(subdb, picked)
        return subdb


# --- Results ----------------------------------------------------------------------

# Cython, like the reference's (src/pyopal/lib.pyx:783-1119): see _results.pyx
try:
    from ._results import EndResult, FullResult, ScoreResult  # noqa: E402
except ImportError as err:  # pragma: no cover
    raise ImportError(
        "pyopal_amd._results is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C pyopal_amd/csrc`)") from err

_RESULT_TYPES = {"score": ScoreResult, "end": EndResult, "full": FullResult}


# --- Aligner ------------------------------------------------------------------------

def resolve_scoring_matrix(spec, verb: str = "found") -> ScoringMatrix:
    """``None`` (BLOSUM50), a matrix name or a `ScoringMatrix`: what `Aligner`
    (``src/pyopal/lib.pyx:1202-1212``) and `pyopal.align` (``src/pyopal/_align.py:119-127``)
    accept; their error messages differ by one word."""
    if isinstance(spec, ScoringMatrix):
        return spec
    if spec is None:
        return Aligner._DEFAULT_SCORING_MATRIX
    if isinstance(spec, str):
        return ScoringMatrix.from_name(spec)
    raise TypeError(f"expected str or ScoringMatrix, {verb} {type(spec).__name__}")


class Aligner:
    """The Opal aligner, served by MI355X kernels
    (``src/pyopal/lib.pyx:1122-1383``)."""

    _DEFAULT_SCORING_MATRIX = ScoringMatrix.from_name("BLOSUM50")
    _DEFAULT_GAP_OPEN = 3
    _DEFAULT_GAP_EXTEND = 1

    def __init__(self, scoring_matrix=None, gap_open: int = _DEFAULT_GAP_OPEN,
                 gap_extend: int = _DEFAULT_GAP_EXTEND):
        self.scoring_matrix = resolve_scoring_matrix(scoring_matrix, "found")
        self.alphabet = Alphabet(self.scoring_matrix.alphabet)
        self.gap_open = int(gap_open)
        self.gap_extend = int(gap_extend)

        # the one backend: HIP kernels behind libmiopal.so (the slot where the
        # reference picks SSE2/SSE4/AVX2/NEON, src/pyopal/lib.pyx:1213-1227)
        try:
            _capi.lib()
        except (RuntimeError, OSError) as err:
            raise RuntimeError("no supported SIMD backend available") from err
        from .platform.hip import searchHIP
        self._search = searchHIP

        if not self.scoring_matrix.is_integer():
            raise ValueError("Integer scoring matrix is expected")
        self._int_matrix = self.scoring_matrix.int_array()

    def __repr__(self):
        args = []
        if self.scoring_matrix != self._DEFAULT_SCORING_MATRIX:
            args.append(f"{self.scoring_matrix!r}")
        if self.gap_open != self._DEFAULT_GAP_OPEN:
            args.append(f"gap_open={self.gap_open!r}")
        if self.gap_extend != self._DEFAULT_GAP_EXTEND:
            args.append(f"gap_extend={self.gap_extend!r}")
        return f"{type(self).__name__}({', '.join(args)})"

    def __reduce__(self):
        return type(self), (self.scoring_matrix, self.gap_open, self.gap_extend)

    def __eq__(self, other):
        if not isinstance(other, Aligner):
            return NotImplemented
        return self.__reduce__()[1] == other.__reduce__()[1]

    __hash__ = None

    def align(self, query, database: BaseDatabase, *, mode: str = "score",
              overflow: str = "buckets", algorithm: str = "sw", start: int = 0,
              end: int = UINT32_MAX, device: int = 0,
              shard: typing.Optional[typing.Tuple[int, int]] = None,
              shard_version: typing.Optional[int] = None) -> typing.List[ScoreResult]:
        """Align the query to every target of ``database[start:end]``.

        Same keywords as the reference (``src/pyopal/lib.pyx:1258-1268``);
        ``device`` (extension) selects the GPU holding the database mirror, ``shard`` (extension)
        a mirror of the targets ``[lo, hi)`` only, which must contain ``[start, end)``: what
        `pyopal_amd.align` uses to give every GPU its own part of the database; with
        ``shard_version`` the shard only counts while the database is still in the state it was cut
        in (a database mutated since is searched through the whole-database mirror of ``device``).
        Target indices of the results are absolute either way.
        ``overflow`` is validated and otherwise ignored: the GPU path picks the
        narrowest exact lane width per target, results are identical.
        """
        if query is None:
            raise TypeError("Argument 'query' must not be None")
        if not isinstance(database, BaseDatabase):
            raise TypeError(f"Argument 'database' has incorrect type (expected BaseDatabase, "
                            f"got {type(database).__name__})")
        if mode in _OPAL_SEARCH_MODES:
            _mode = _OPAL_SEARCH_MODES[mode]
        else:
            raise ValueError(f"invalid search mode: {mode!r}")
        if overflow in _OPAL_OVERFLOW_MODES:
            _overflow = _OPAL_OVERFLOW_MODES[overflow]
        else:
            raise ValueError(f"invalid overflow mode: {overflow!r}")
        if algorithm in _OPAL_ALGORITHMS:
            _algo = _OPAL_ALGORITHMS[algorithm]
        else:
            raise ValueError(f"invalid algorithm: {algorithm!r}")
        if start < 0 or end < 0:
            raise OverflowError("can't convert negative value to uint32_t")

        if database.alphabet != self.alphabet:
            raise ValueError("database and score matrix have different alphabets")

        encoded = database.alphabet.encode(query)

        with database.lock.read:
            size = database._get_size()
            if shard is not None and shard_version is not None and getattr(database, "_version", None) != shard_version:
                shard = None
            if end < start:
                raise IndexError("database slice end is lower than start")
            if end > size:
                end = size
            if start > size:
                # the reference does not guard this case (unsigned underflow at
                # src/pyopal/platform/pyx.in:62); an IndexError is raised instead
                raise IndexError("database slice start is past the end of the database")
            if shard is not None and not (shard[0] <= start and end <= shard[1]):
                raise IndexError(f"slice [{start}, {end}) outside the shard [{shard[0]}, {shard[1]})")
            return self._search(encoded, database, _mode, _overflow, _algo, self.gap_open,
                                self.gap_extend, self._int_matrix, start, end, device=device, shard=shard)


    def scores(self, query, database: BaseDatabase, *, algorithm: str = "sw", start: int = 0,
               end: int = UINT32_MAX, device: int = 0) -> np.ndarray:
        """Extension: the scores of ``align(mode="score")`` as
        one ``int32`` array instead of a list of `ScoreResult` objects, which for a
        million targets costs more wall time than the search itself."""
        return self.align_arrays(query, database, mode="score", algorithm=algorithm, start=start,
                                 end=end, device=device).score

    def align_arrays(self, query, database: BaseDatabase, *, mode: str = "score",
                     algorithm: str = "sw", start: int = 0, end: int = UINT32_MAX,
                     device: int = 0) -> "ResultArrays":
        """Extension: the results of `align` as arrays, one entry
        per target of ``database[start:end]``, without a Python object per target. The
        returned `ResultArrays` builds the reference's result objects on demand
        (``arrays[k]``, iteration), so ``list(arrays) == aligner.align(...)``."""
        if mode not in _OPAL_SEARCH_MODES:
            raise ValueError(f"invalid search mode: {mode!r}")
        if algorithm not in _OPAL_ALGORITHMS:
            raise ValueError(f"invalid algorithm: {algorithm!r}")
        if start < 0 or end < 0:
            raise OverflowError("can't convert negative value to uint32_t")
        if database.alphabet != self.alphabet:
            raise ValueError("database and score matrix have different alphabets")
        encoded = database.alphabet.encode(query)
        with database.lock.read:
            size = database._get_size()
            if end < start:
                raise IndexError("database slice end is lower than start")
            end = min(end, size)
            if start > size:
                raise IndexError("database slice start is past the end of the database")
            if end == start:
                out = {"score": np.zeros(0, dtype=np.int32)}
                if mode != "score":
                    out.update(end_q=np.zeros(0, dtype=np.int32), end_t=np.zeros(0, dtype=np.int32))
                if mode == "full":
                    out.update(start_q=np.zeros(0, dtype=np.int32), start_t=np.zeros(0, dtype=np.int32),
                               aln_flat=np.zeros(0, dtype=np.uint8), aln_off=np.zeros(1, dtype=np.int64))
                return ResultArrays(mode, start, len(encoded), [], out)
            if _capi.lib().miopalDeviceCount() < 1:
                raise RuntimeError("no supported SIMD backend available")
            mirror = database._device_mirror(device)
            out = mirror.search(np.frombuffer(encoded, dtype=np.uint8), _int_matrix_array(self._int_matrix),
                                self.gap_open, self.gap_extend, mode, algorithm, start, end)
            lengths = np.diff(mirror.offsets[start:end + 1]) if mode == "full" else None
            return ResultArrays(mode, start, len(encoded), lengths, out)


class ResultArrays:
    """Results of one search as arrays (`Aligner.align_arrays`).

    Attributes (``int32`` arrays, one entry per target of the slice): ``score``; for the
    ``end`` and ``full`` modes ``query_end``, ``target_end``; for ``full`` also
    ``query_start``, ``target_start``, ``target_length`` and the alignments as one ``uint8``
    buffer ``operations`` (codes of ``src/pyopal/opal.pxd:21-24``) with ``operation_offsets``
    (``int64``, n + 1 entries). Indexing and iteration yield `ScoreResult` / `EndResult` /
    `FullResult` objects equal to those of `Aligner.align`.
    """

    def __init__(self, mode: str, start: int, query_length: int, target_lengths, out):
        self.mode = mode
        self.start = start
        self.query_length = query_length
        self.score = out["score"]
        self.query_end = out.get("end_q")
        self.target_end = out.get("end_t")
        self.query_start = out.get("start_q")
        self.target_start = out.get("start_t")
        self.operations = out.get("aln_flat")
        self.operation_offsets = out.get("aln_off")
        self.target_length = target_lengths if mode == "full" else None

    def __len__(self) -> int:
        return len(self.score)

    def alignment(self, k: int) -> str:
        """The operations of target ``k`` of the slice over ``MDIX`` (`FullResult.alignment`)."""
        if self.mode != "full":
            raise ValueError("alignments are only computed in 'full' mode")
        lo, hi = self.operation_offsets[k], self.operation_offsets[k + 1]
        return self.operations[lo:hi].tobytes().translate(_OPS_TO_TEXT).decode("ascii")

    def __getitem__(self, k: int):
        n = len(self)
        if k < 0:
            k += n
        if k < 0 or k >= n:
            raise IndexError(k)
        index = self.start + k
        if self.mode == "score":
            return ScoreResult(index, int(self.score[k]))
        if self.mode == "end":
            return EndResult(index, int(self.score[k]), int(self.query_end[k]), int(self.target_end[k]))
        return FullResult(index, int(self.score[k]), int(self.query_end[k]), int(self.target_end[k]),
                          int(self.query_start[k]), int(self.target_start[k]), self.query_length,
                          int(self.target_length[k]), self.alignment(k))

    def __iter__(self):
        return (self[k] for k in range(len(self)))


_OPS_TO_TEXT = bytes.maketrans(bytes([0, 1, 2, 3]), b"MDIX")  # src/pyopal/lib.pyx:991


def _int_matrix_array(matrix) -> np.ndarray:
    if isinstance(matrix, array.array):
        return np.frombuffer(matrix, dtype=np.int32)
    return np.ascontiguousarray(matrix, dtype=np.int32)
