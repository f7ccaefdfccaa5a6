"""Scoring matrices for the Opal search path.

The reference takes its matrices from the third-party ``scoring-matrices``
package (``src/pyopal/lib.pyx:39,1153,1202-1209,1230-1238``;
``src/pyopal/_align.py:6,120-127``), which is neither vendored in the
reference tree nor installed here. This module is a minimal provider with the
same surface the path uses (`from_name`, `alphabet`, `is_integer`, `size`,
row access, equality, pickling) and the public NCBI tables.

Only the A/C/G/T corner of BLOSUM50 is pinned by the reference's tests
(SURVEY.md appendix B); `_self_check` verifies those entries, symmetry and
shape for every table carried here.
"""

from __future__ import annotations

import array
import typing

_NCBI_ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"

_TABLES: typing.Dict[str, str] = {
    "BLOSUM62": """
 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4
-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4
-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4
-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4
 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4
-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4
-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4
-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4
-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4
-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4
-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4
-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4
-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4
 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4
-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4
 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4
-2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4
-1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4
 0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4
-4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1
""",
    "BLOSUM50": """
 5 -2 -1 -2 -1 -1 -1  0 -2 -1 -2 -1 -1 -3 -1  1  0 -3 -2  0 -2 -1 -1 -5
-2  7 -1 -2 -4  1  0 -3  0 -4 -3  3 -2 -3 -3 -1 -1 -3 -1 -3 -1  0 -1 -5
-1 -1  7  2 -2  0  0  0  1 -3 -4  0 -2 -4 -2  1  0 -4 -2 -3  4  0 -1 -5
-2 -2  2  8 -4  0  2 -1 -1 -4 -4 -1 -4 -5 -1  0 -1 -5 -3 -4  5  1 -1 -5
-1 -4 -2 -4 13 -3 -3 -3 -3 -2 -2 -3 -2 -2 -4 -1 -1 -5 -3 -1 -3 -3 -2 -5
-1  1  0  0 -3  7  2 -2  1 -3 -2  2  0 -4 -1  0 -1 -1 -1 -3  0  4 -1 -5
-1  0  0  2 -3  2  6 -3  0 -4 -3  1 -2 -3 -1 -1 -1 -3 -2 -3  1  5 -1 -5
 0 -3  0 -1 -3 -2 -3  8 -2 -4 -4 -2 -3 -4 -2  0 -2 -3 -3 -4 -1 -2 -2 -5
-2  0  1 -1 -3  1  0 -2 10 -4 -3  0 -1 -1 -2 -1 -2 -3  2 -4  0  0 -1 -5
-1 -4 -3 -4 -2 -3 -4 -4 -4  5  2 -3  2  0 -3 -3 -1 -3 -1  4 -4 -3 -1 -5
-2 -3 -4 -4 -2 -2 -3 -4 -3  2  5 -3  3  1 -4 -3 -1 -2 -1  1 -4 -3 -1 -5
-1  3  0 -1 -3  2  1 -2  0 -3 -3  6 -2 -4 -1  0 -1 -3 -2 -3  0  1 -1 -5
-1 -2 -2 -4 -2  0 -2 -3 -1  2  3 -2  7  0 -3 -2 -1 -1  0  1 -3 -1 -1 -5
-3 -3 -4 -5 -2 -4 -3 -4 -1  0  1 -4  0  8 -4 -3 -2  1  4 -1 -4 -4 -2 -5
-1 -3 -2 -1 -4 -1 -1 -2 -2 -3 -4 -1 -3 -4 10 -1 -1 -4 -3 -3 -2 -1 -2 -5
 1 -1  1  0 -1  0 -1  0 -1 -3 -3  0 -2 -3 -1  5  2 -4 -2 -2  0  0 -1 -5
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  2  5 -3 -2  0  0 -1  0 -5
-3 -3 -4 -5 -5 -1 -3 -3 -3 -3 -2 -3 -1  1 -4 -4 -3 15  2 -3 -5 -2 -3 -5
-2 -1 -2 -3 -3 -1 -2 -3  2 -1 -1 -2  0  4 -3 -2 -2  2  8 -1 -3 -2 -1 -5
 0 -3 -3 -4 -1 -3 -3 -4 -4  4  1 -3  1 -1 -3 -2  0 -3 -1  5 -4 -3 -1 -5
-2 -1  4  5 -3  0  1 -1  0 -4 -4  0 -3 -4 -2  0  0 -5 -3 -4  5  2 -1 -5
-1  0  0  1 -3  4  5 -2  0 -3 -3  1 -1 -4 -1  0 -1 -2 -2 -3  2  5 -1 -5
-1 -1 -1 -1 -2 -1 -1 -2 -1 -1 -1 -1 -1 -2 -2 -1  0 -3 -1 -1 -1 -1 -1 -5
-5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5 -5  1
""",
}


# names the reference resolves through scoring-matrices (src/pyopal/tests/test_aligner.py:10-18 uses
# VTML80) whose tables are not carried here: see ScoringMatrix.from_name
_KNOWN_ELSEWHERE = frozenset(
    ["BENNER6", "BENNER22", "BENNER74", "DAYHOFF", "GONNET", "NUC.4.4", "MATCH", "BLOSUMN"]
    + [f"BLOSUM{n}" for n in (30, 35, 40, 45, 55, 60, 65, 70, 75, 80, 85, 90, 100)]
    + [f"PAM{n}" for n in range(10, 510, 10)]
    + [f"VTML{n}" for n in (10, 20, 40, 80, 120, 160)])


class ScoringMatrix:
    """A square substitution matrix over an alphabet.

    Mirrors the part of ``scoring_matrices.ScoringMatrix`` that the reference
    touches (call sites listed in the module docstring).
    """

    __slots__ = ("_alphabet", "_rows", "_name")

    def __init__(self, matrix, alphabet: str = _NCBI_ORDER, name: typing.Optional[str] = None):
        rows = [tuple(float(x) for x in row) for row in matrix]
        n = len(alphabet)
        if len(rows) != n or any(len(r) != n for r in rows):
            raise ValueError("matrix must be square and match the alphabet length")
        if len(set(alphabet)) != n:
            raise ValueError("alphabet contains duplicate letters")
        self._alphabet = str(alphabet)
        self._rows = tuple(rows)
        self._name = name

    # --- constructors -----------------------------------------------------
    @classmethod
    def from_name(cls, name: str = "BLOSUM62") -> "ScoringMatrix":
        """A matrix by its usual name. Resolution order: the tables carried here (BLOSUM50,
        BLOSUM62: provenance in the module docstring); the ``scoring-matrices`` package the
        reference uses (``src/pyopal/lib.pyx:1202-1207``) when it is installed; an NCBI-format
        file ``<name>``, ``<name>.txt`` or ``<name>.mat`` (any case) in one of the directories of
        ``PYOPAL_AMD_MATRIX_PATH``. The other tables the reference resolves by name (VTML, PAM,
        the rest of BLOSUM, ...) are deliberately NOT transcribed here: their values are not in the
        reference tree, and a table typed from memory cannot be vouched for entry by entry."""
        text = _TABLES.get(name)
        if text is not None:
            rows = [[int(x) for x in line.split()] for line in text.strip().splitlines()]
            return cls(rows, _NCBI_ORDER, name=name)
        try:
            import scoring_matrices  # the reference's own provider, if present
        except ImportError:
            scoring_matrices = None
        if scoring_matrices is not None:
            try:
                theirs = scoring_matrices.ScoringMatrix.from_name(name)
            except ValueError:
                theirs = None
            if theirs is not None:
                return cls([list(row) for row in theirs.matrix], str(theirs.alphabet), name=name)
        import os
        for directory in filter(None, os.environ.get("PYOPAL_AMD_MATRIX_PATH", "").split(os.pathsep)):
            try:
                entries = os.listdir(directory)
            except OSError:
                continue
            wanted = {name.lower(), name.lower() + ".txt", name.lower() + ".mat"}
            for entry in sorted(entries):
                if entry.lower() in wanted:
                    # (a digest beside the table - "<file>.sha256", the first word of it - is enforced)
                    digest = None
                    try:
                        with open(os.path.join(directory, entry + ".sha256")) as f:
                            digest = f.read().split()[0]
                    except (OSError, IndexError):
                        pass
                    return cls.from_file(os.path.join(directory, entry), name=name, sha256=digest)
        known = " (a name of the scoring-matrices package)" if name.upper() in _KNOWN_ELSEWHERE else ""
        raise ValueError(
            f"unknown scoring matrix: {name!r}{known}; built in: {', '.join(sorted(_TABLES))}. Load an "
            "NCBI-format table with ScoringMatrix.from_file(path), put it in a directory listed in "
            "PYOPAL_AMD_MATRIX_PATH, or install scoring-matrices")

    @classmethod
    def from_file(cls, file, name: typing.Optional[str] = None, sha256: typing.Optional[str] = None) -> "ScoringMatrix":
        """Load a matrix in the NCBI text format (``#`` comments, a header line of
        column letters, then one row per letter). `file` is a path or a file object.

        ``sha256``: the hex digest the file's bytes must have (a path only) - a deployment that ships the NCBI
        tables this package does not carry (VTML, PAM, the rest of BLOSUM: see `from_name`) pins them this way;
        a table that differs by one entry raises ``ValueError`` instead of scoring differently."""
        if isinstance(file, (str, bytes)) or hasattr(file, "__fspath__"):
            if sha256 is not None:
                import hashlib
                with open(file, "rb") as raw:
                    digest = hashlib.sha256(raw.read()).hexdigest()
                if digest.lower() != sha256.strip().lower():
                    raise ValueError(f"{file}: sha256 {digest} differs from the expected {sha256}")
            with open(file) as handle:
                return cls.from_file(handle, name=name)
        if sha256 is not None:
            raise ValueError("sha256 can only be checked for a path")
        letters = None
        rows = []
        row_letters = []
        for line in file:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            fields = line.split()
            if letters is None:
                letters = "".join(fields)
                continue
            if fields[0].isalpha() or fields[0] == "*":
                row_letters.append(fields[0])
                fields = fields[1:]
            rows.append([float(x) for x in fields])
        if letters is None:
            raise ValueError("no matrix found")
        if row_letters and "".join(row_letters) != letters:
            raise ValueError("row and column letters differ")
        return cls(rows, letters, name=name)

    @classmethod
    def from_match_mismatch(cls, match: float = 1.0, mismatch: float = -1.0,
                            alphabet: str = "ACGT") -> "ScoringMatrix":
        n = len(alphabet)
        return cls([[match if i == j else mismatch for j in range(n)] for i in range(n)], alphabet)

    # --- accessors --------------------------------------------------------
    @property
    def alphabet(self) -> str:
        return self._alphabet

    @property
    def name(self) -> typing.Optional[str]:
        return self._name

    @property
    def matrix(self):
        return [list(r) for r in self._rows]

    def size(self) -> int:
        return len(self._alphabet)

    def __len__(self) -> int:
        return len(self._alphabet)

    def __getitem__(self, item):
        if isinstance(item, tuple):
            i, j = item
            return self._rows[i][j]
        return self._rows[item]

    def is_integer(self) -> bool:
        return all(float(x).is_integer() for r in self._rows for x in r)

    def is_symmetric(self) -> bool:
        n = len(self)
        return all(self._rows[i][j] == self._rows[j][i] for i in range(n) for j in range(i))

    def min(self) -> float:
        return min(min(r) for r in self._rows)

    def max(self) -> float:
        return max(max(r) for r in self._rows)

    def int_array(self) -> array.array:
        """Row-major ``int`` copy, as `Aligner.__init__` builds it
        (``src/pyopal/lib.pyx:1229-1238``)."""
        return array.array("i", [int(x) for r in self._rows for x in r])

    # --- dunder -----------------------------------------------------------
    def __eq__(self, other):
        if not isinstance(other, ScoringMatrix):
            return NotImplemented
        return self._alphabet == other._alphabet and self._rows == other._rows

    def __hash__(self):
        return hash((self._alphabet, self._rows))

    def __reduce__(self):
        return type(self), (self.matrix, self._alphabet, self._name)

    def __repr__(self):
        if self._name is not None:
            return f"{type(self).__name__}.from_name({self._name!r})"
        return f"{type(self).__name__}({self.matrix!r}, alphabet={self._alphabet!r})"


def available() -> typing.List[str]:
    return sorted(_TABLES)


def _self_check() -> None:
    for name in _TABLES:
        m = ScoringMatrix.from_name(name)
        assert m.size() == 24 and m.is_integer() and m.is_symmetric(), name
    b50 = ScoringMatrix.from_name("BLOSUM50")
    ix = {c: i for i, c in enumerate(_NCBI_ORDER)}
    pinned = {"AA": 5, "CC": 13, "GG": 8, "TT": 5, "AC": -1, "AG": 0, "AT": 0,
              "CG": -3, "CT": -1, "GT": -2}
    for pair, v in pinned.items():
        assert b50[ix[pair[0]], ix[pair[1]]] == v, pair
