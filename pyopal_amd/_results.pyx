# cython: language_level=3, boundscheck=False, wraparound=False, embedsignature=True, binding=True
"""Result types of the search path and their bulk constructors.

Cython, like the reference's own result classes (``src/pyopal/lib.pyx:783-1119``,
``lib.pxd:113-126``): a search over a million targets returns a million Python
objects, and the platform plugin's fill loop (``src/pyopal/platform/pyx.in:64-72``)
is the second hot loop of the path once the DP itself runs on the GPU
(SURVEY.md section 3, hot loop ii). `score_results` / `end_results` /
`full_results` build the whole list from the arrays the C ABI fills, without a
Python-level call per target.
"""

cimport cython
from cpython.list cimport PyList_New, PyList_SET_ITEM
from cpython.ref cimport Py_INCREF
from libc.stdint cimport int32_t, int64_t, uint8_t

cdef dict _OPAL_ALIGNMENT_OPERATION = {"M": 0, "D": 1, "I": 2, "X": 3}   # src/pyopal/lib.pyx:97-102
cdef bytes _OPS_TO_TEXT = bytes.maketrans(bytes([0, 1, 2, 3]), b"MDIX")   # src/pyopal/lib.pyx:991


# (no_gc: the only object a result refers to is the shared buffer of operations, which refers to
# no result - a million results tracked by the cyclic collector made building the list 13x slower)
@cython.no_gc
cdef class ScoreResult:
    """Result of a search in ``score`` mode (``src/pyopal/lib.pyx:783-834``)."""

    # (each class carries only its own fields - the reference embeds one OpalSearchResult in the
    # base class, src/pyopal/lib.pxd:113-115; a million 32-byte objects build faster than a
    # million 88-byte ones)
    cdef Py_ssize_t _target_index
    cdef bint       _score_set
    cdef int        _score

    def __cinit__(self):
        self._target_index = -1
        self._score_set = False
        self._score = 0

    def __init__(self, size_t target_index, int score):
        self._target_index = target_index
        self._score = score
        self._score_set = True

    def __repr__(self):
        return f"{type(self).__name__}({self.target_index}, score={self.score!r})"

    def __reduce__(self):
        return type(self), (self.target_index, self.score)

    def __eq__(self, object other):
        if not isinstance(other, ScoreResult):
            return NotImplemented
        return self.__reduce__()[1] == other.__reduce__()[1]

    __hash__ = None

    @property
    def target_index(self):
        """`int`: The index of the target in the database."""
        assert self._target_index >= 0
        return self._target_index

    @property
    def score(self):
        """`int`: The score of the alignment."""
        assert self._score_set
        return self._score


@cython.no_gc
cdef class EndResult(ScoreResult):
    """Result of a search in ``end`` mode (``src/pyopal/lib.pyx:837-881``)."""

    cdef int        _query_end
    cdef int        _target_end

    def __cinit__(self):
        self._query_end = self._target_end = -1

    def __init__(self, size_t target_index, int score, int query_end, int target_end):
        super().__init__(target_index, score)
        self._query_end = query_end
        self._target_end = target_end

    def __repr__(self):
        return (f"{type(self).__name__}({self.target_index}, score={self.score!r}, "
                f"query_end={self.query_end!r}, target_end={self.target_end!r})")

    def __reduce__(self):
        return type(self), (self.target_index, self.score, self.query_end, self.target_end)

    @property
    def query_end(self):
        """`int`: The coordinate where the alignment ends in the query."""
        assert self._query_end >= 0
        return self._query_end

    @property
    def target_end(self):
        """`int`: The coordinate where the alignment ends in the target."""
        assert self._target_end >= 0
        return self._target_end


@cython.no_gc
cdef class FullResult(EndResult):
    """Result of a search in ``full`` mode (``src/pyopal/lib.pyx:884-1119``)."""

    cdef int        _query_start
    cdef int        _target_start
    cdef int        _query_length
    cdef int        _target_length
    cdef bytes      _ops
    # alignments of a bulk search live in one shared buffer until somebody looks at them
    cdef object     _ops_owner
    cdef Py_ssize_t _ops_offset
    cdef Py_ssize_t _ops_length

    def __cinit__(self):
        self._query_start = self._target_start = -1
        self._query_length = self._target_length = -1
        self._ops = None
        self._ops_owner = None
        self._ops_offset = self._ops_length = 0

    def __init__(self, size_t target_index, int score, int query_end, int target_end,
                 int query_start, int target_start, int query_length, int target_length,
                 str alignment not None):
        super().__init__(target_index, score, query_end, target_end)
        self._query_length = query_length
        self._target_length = target_length
        self._query_start = query_start
        self._target_start = target_start
        self._ops = bytes([_OPAL_ALIGNMENT_OPERATION[x] for x in alignment])

    def __repr__(self):
        return (f"{type(self).__name__}({self.target_index}, score={self.score!r}, "
                f"query_end={self.query_end!r}, target_end={self.target_end!r}, "
                f"query_start={self.query_start!r}, target_start={self.target_start!r}, "
                f"query_length={self.query_length!r}, target_length={self.target_length!r}, "
                f"alignment={self.alignment!r})")

    def __reduce__(self):
        return (type(self), (self.target_index, self.score, self.query_end, self.target_end,
                             self.query_start, self.target_start, self.query_length,
                             self.target_length, self.alignment))

    @property
    def query_start(self):
        """`int`: The coordinate where the alignment starts in the query."""
        assert self._query_start >= 0
        return self._query_start

    @property
    def target_start(self):
        """`int`: The coordinate where the alignment starts in the target."""
        assert self._target_start >= 0
        return self._target_start

    @property
    def query_length(self):
        """`int`: The complete length of the query sequence."""
        assert self._query_length >= 0
        return self._query_length

    @property
    def target_length(self):
        """`int`: The complete length of the target sequence."""
        assert self._target_length >= 0
        return self._target_length

    cdef bytes _operations(self):
        # op codes of src/pyopal/opal.pxd:21-24, one byte each
        if self._ops is None and self._ops_owner is not None:
            self._ops = bytes(self._ops_owner[self._ops_offset:self._ops_offset + self._ops_length])
            self._ops_owner = None
        return self._ops

    @property
    def alignment(self):
        """`str`: The operations over ``MDIX`` (D: query residue against a gap,
        I: target residue against a gap)."""
        cdef bytes ops = self._operations()
        if ops is None:
            return ""
        return ops.translate(_OPS_TO_TEXT).decode("ascii")

    cpdef str cigar(self):
        """CIGAR string in SAM convention (``op % 3`` -> ``M, I, D``); `None` when
        the alignment is empty."""
        cdef bytes ops = self._operations()
        cdef Py_ssize_t i, n
        cdef unsigned char symbol, current
        cdef size_t count
        cdef list chunks = []
        if ops is None or len(ops) == 0:
            return None
        n = len(ops)
        count = 0
        current = (<unsigned char> ops[0]) % 3
        for i in range(n):
            symbol = (<unsigned char> ops[i]) % 3
            if symbol == current:
                count += 1
            else:
                chunks.append(f"{count}{'MID'[current]}")
                current = symbol
                count = 1
        chunks.append(f"{count}{'MID'[current]}")
        return "".join(chunks)

    cpdef float identity(self):
        """Fraction of aligned residue pairs that are identical (float32 arithmetic,
        ``src/pyopal/lib.pyx:1039-1052``)."""
        cdef bytes ops = self._operations()
        assert ops is not None
        cdef int matches = ops.count(0)
        cdef int mismatches = ops.count(3)
        return (<float> matches) / (<float> (matches + mismatches))

    cpdef float coverage(self, str reference="query"):
        """Fraction of the reference sequence covered by the alignment; edge
        operations that are gaps in the reference do not count
        (``src/pyopal/lib.pyx:1054-1119``)."""
        cdef bytes ops = self._operations()
        assert ops is not None
        cdef Py_ssize_t i, n = len(ops)
        cdef Py_ssize_t length, reflength
        cdef unsigned char operation
        if reference == "query":
            reflength = self._query_length
            length = self._query_end + 1 - self._query_start
            operation = 1
        elif reference == "target":
            reflength = self._target_length
            length = self._target_end + 1 - self._target_start
            operation = 2
        else:
            raise ValueError(f"Invalid coverage reference: {reference!r}")
        for i in range(n):
            if <unsigned char> ops[i] == operation:
                length -= 1
            else:
                break
        for i in range(n - 1, -1, -1):
            if <unsigned char> ops[i] == operation:
                length -= 1
            else:
                break
        return 0.0 if length < 0 else (<float> length) / (<float> reflength)


# --- bulk constructors (the plugin's fill loop, pyx.in:64-72 and 95-99) -------------------

def score_results(Py_ssize_t start, const int32_t[::1] scores):
    cdef Py_ssize_t k, n = scores.shape[0]
    cdef list out = PyList_New(n)
    cdef ScoreResult r
    for k in range(n):
        r = ScoreResult.__new__(ScoreResult)
        r._target_index = start + k
        r._score = scores[k]
        r._score_set = True
        Py_INCREF(r)
        PyList_SET_ITEM(out, k, r)
    return out


def end_results(Py_ssize_t start, const int32_t[::1] scores, const int32_t[::1] end_q,
                const int32_t[::1] end_t):
    cdef Py_ssize_t k, n = scores.shape[0]
    cdef list out = PyList_New(n)
    cdef EndResult r
    for k in range(n):
        r = EndResult.__new__(EndResult)
        r._target_index = start + k
        r._score = scores[k]
        r._score_set = True
        r._query_end = end_q[k]
        r._target_end = end_t[k]
        Py_INCREF(r)
        PyList_SET_ITEM(out, k, r)
    return out


def full_results(Py_ssize_t start, const int32_t[::1] scores, const int32_t[::1] end_q,
                 const int32_t[::1] end_t, const int32_t[::1] start_q, const int32_t[::1] start_t,
                 int query_length, object target_lengths, const uint8_t[::1] ops,
                 const int64_t[::1] ops_off):
    """`target_lengths` is the database's length sequence (indexed by absolute target
    index); `ops` / `ops_off` the flat alignment buffer of miopalSearchFlat."""
    cdef Py_ssize_t k, n = scores.shape[0]
    cdef list out = PyList_New(n)
    cdef FullResult r
    cdef object owner = memoryview(ops)   # one shared view of the flat buffer
    for k in range(n):
        r = FullResult.__new__(FullResult)
        r._target_index = start + k
        r._score = scores[k]
        r._score_set = True
        r._query_end = end_q[k]
        r._target_end = end_t[k]
        r._query_start = start_q[k]
        r._target_start = start_t[k]
        r._query_length = query_length
        r._target_length = target_lengths[start + k]
        # no bytes object per target here: the operations stay in the shared buffer until used
        r._ops_owner = owner
        r._ops_offset = ops_off[k]
        r._ops_length = ops_off[k + 1] - ops_off[k]
        Py_INCREF(r)
        PyList_SET_ITEM(out, k, r)
    return out
