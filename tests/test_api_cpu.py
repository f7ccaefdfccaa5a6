"""API parity of the host layer with ``pyopal`` (no GPU needed): the behaviours
pinned by the reference's src/pyopal/tests/test_{alphabet,database,result}.py
and the doctests of src/pyopal/lib.pyx, re-stated against ``pyopal_amd``."""
import pickle
import threading

import pytest

import pyopal_amd as pyopal
from pyopal_amd.matrices import ScoringMatrix


# ---- Alphabet (src/pyopal/tests/test_alphabet.py:9-124) -----------------------
def test_alphabet_basics():
    default = pyopal.Alphabet()
    assert len(default) == 24 == len(default.letters)
    assert repr(default) == "Alphabet()"
    abc = pyopal.Alphabet("ATGC")
    assert (len(abc), str(abc), abc.letters, repr(abc)) == (4, "ATGC", "ATGC", "Alphabet('ATGC')")
    assert "A" in abc and "T" in abc and "X" not in abc
    assert [abc[0], abc[2], abc[-1], abc[-2]] == ["A", "G", "C", "G"]
    for bad in (-5, 4, 5):
        with pytest.raises(IndexError):
            abc[bad]
    assert abc == abc and abc == "ATGC" and abc == pyopal.Alphabet("ATGC")
    assert abc != pyopal.Alphabet("TCGA") and abc != 10
    clone = pickle.loads(pickle.dumps(abc))
    assert clone == abc and clone.letters == "ATGC"


@pytest.mark.parametrize("letters", ["AAAA", "AtgC", "A[]C", "ABCDEFGHIJKLMNOPQRSTUVWXYZ*ABCDEFG"[:33]])
def test_alphabet_rejects(letters):
    with pytest.raises(ValueError):
        pyopal.Alphabet(letters)


def test_alphabet_encode_decode():
    abc = pyopal.Alphabet("ATGC")
    for seq in ("ATGC", b"ATGC"):
        assert abc.encode(seq) == bytes([0, 1, 2, 3])
    assert abc.encode("AAAAA") == bytes(5)
    for buf in (bytes([0, 1, 2, 3]), bytearray([0, 1, 2, 3]), memoryview(bytearray([0, 1, 2, 3]))):
        assert abc.decode(buf) == "ATGC"
    # doctests src/pyopal/lib.pyx:304-306, 325-327
    acgt = pyopal.Alphabet("ACGT")
    assert acgt.encode("GATACA") == b"\x02\x00\x03\x00\x01\x00"
    assert acgt.decode(bytearray([2, 0, 3, 0, 1, 0])) == "GATACA"
    with pytest.raises(ValueError):
        acgt.decode(bytes([4]))
    with pytest.raises(ValueError):
        acgt.encode("AXA")       # letter outside the alphabet, no wildcard
    with pytest.raises(ValueError):
        acgt.encode("A-A")       # not a letter
    with pytest.raises(ValueError):
        acgt.encode_into(b"ACG", bytearray(2))
    # unknown letters, lower case included, take the index of '*' (lib.pyx:211-219)
    default = pyopal.Alphabet()
    assert default.encode("AUa") == bytes([0, 23, 23])
    out = bytearray(3)
    default.encode_into(b"ARN", out)
    assert bytes(out) == bytes([0, 1, 2])


# ---- Database (src/pyopal/tests/test_database.py:9-127) ---------------------------
def test_database_read_interface():
    seqs = ["ATGC", "ATTTAC", "TTACCG"]
    db = pyopal.Database(seqs)
    assert len(db) == 3 and list(db) == seqs
    assert all(s in db for s in seqs) and "TAACCG" not in db and "AAAA" not in db
    with pytest.raises(TypeError):
        1 in db
    assert db.lengths == [4, 6, 6] and db.total_length == 16
    assert [db[i] for i in (0, 1, 2, -1, -2, -3)] == seqs + seqs[::-1]
    for bad in (3, -4, -8):
        with pytest.raises(IndexError):
            db[bad]
    assert list(pyopal.Database([s.encode() for s in seqs])) == seqs
    assert pyopal.Database().total_length == 0
    with pytest.raises(TypeError):
        pyopal.BaseDatabase(["ATGC"])


def test_database_slices_and_subsets():
    seqs = ["ATGC", "ATTC", "TTCG", "TTAT", "AAAC"]
    db = pyopal.Database(seqs)
    assert list(db[:2]) == seqs[:2]
    assert list(db[1:4:2]) == seqs[1:4:2]
    assert list(db[1::-1]) == seqs[1::-1]
    assert list(db.mask([True, False, False, True, False])) == ["ATGC", "TTAT"]
    assert list(db.extract([4, 0])) == ["AAAC", "ATGC"]
    with pytest.raises(IndexError):
        db.mask([True])
    with pytest.raises(IndexError):
        db.mask([True] * 6)
    with pytest.raises(IndexError):
        db.extract([-1])
    with pytest.raises(IndexError):
        db.extract([5])


def test_database_mutation():
    db = pyopal.Database(["ATGC", "ATTC"])
    db.insert(1, "TTCC")
    db.insert(-10, "TTTT")
    db.insert(10, "AAAA")
    assert list(db) == ["TTTT", "ATGC", "TTCC", "ATTC", "AAAA"]
    db.reverse()
    assert list(db) == ["AAAA", "ATTC", "TTCC", "ATGC", "TTTT"]
    db[2] = "AAAT"
    del db[1]
    del db[-1]
    assert list(db) == ["AAAA", "AAAT", "ATGC"]
    for bad in (-8, 5):
        with pytest.raises(IndexError):
            db[bad] = "TCGA"
    db.extend(["GGTG"])
    db.append("CCCC")
    assert list(db)[-2:] == ["GGTG", "CCCC"]
    assert list(pickle.loads(pickle.dumps(db))) == list(db)
    db.clear()
    assert len(db) == 0
    db.reverse()
    for bad in (0, -1):
        with pytest.raises(IndexError):
            del db[bad]
    other = pyopal.Database(["AC"], "ACGT")
    assert other.alphabet == "ACGT" and other.lengths == [2]


def test_database_lock_is_shared_exclusive():
    db = pyopal.Database(["ATGC"])
    order = []
    with db.lock.read:
        with db.lock.read:          # readers share
            order.append("two readers")
        t = threading.Thread(target=lambda: (db.append("AAAA"), order.append("writer done")))
        t.start()
        t.join(0.2)
        assert t.is_alive()         # the writer waits for the readers
        order.append("reader leaves")
    t.join(5)
    assert order == ["two readers", "reader leaves", "writer done"] and len(db) == 2


# ---- Results (src/pyopal/tests/test_result.py:8-99) -----------------------------------
def test_score_result():
    r = pyopal.ScoreResult(10, score=30)
    assert (r.target_index, r.score, repr(r)) == (10, 30, "ScoreResult(10, score=30)")
    clone = pickle.loads(pickle.dumps(r))
    assert clone == r and r == pyopal.ScoreResult(target_index=10, score=30)
    assert r != pyopal.ScoreResult(12, 50) and r != 12
    blank = pyopal.ScoreResult.__new__(pyopal.ScoreResult)
    with pytest.raises(AssertionError):
        blank.score


def test_end_result():
    r = pyopal.EndResult(2, score=30, query_end=10, target_end=20)
    assert (r.target_index, r.score, r.query_end, r.target_end) == (2, 30, 10, 20)
    assert repr(pyopal.EndResult(10, 30, 10, 20)) == "EndResult(10, score=30, query_end=10, target_end=20)"
    assert pickle.loads(pickle.dumps(r)) == r
    assert r != pyopal.EndResult(10, 35, 20, 60) and r != 12
    assert isinstance(r, pyopal.ScoreResult)


def test_full_result():
    kw = dict(score=30, query_end=10, target_end=20, query_start=0, target_start=10,
              query_length=100, target_length=100, alignment="M" * 10)
    r = pyopal.FullResult(10, **kw)
    assert (r.query_start, r.target_start, r.query_length, r.target_length) == (0, 10, 100, 100)
    assert r.alignment == "M" * 10 and r.cigar() == "10M" and r.identity() == 1.0
    assert pickle.loads(pickle.dumps(r)) == r
    other = dict(kw, score=48, target_start=30, query_length=500, target_length=200)
    assert r != pyopal.FullResult(2, **other) and r != 12
    # doctests src/pyopal/lib.pyx:1006-1010, 1076-1082 on the reference's known alignment
    g1 = pyopal.FullResult(0, 44, 5, 7, 0, 0, 6, 8, "IMMMXMIM")
    assert g1.cigar() == "1D5M1D1M"
    assert g1.coverage("query") == 1.0 and g1.coverage("target") == 0.875
    assert g1.identity() == pytest.approx(5 / 6)
    with pytest.raises(ValueError):
        g1.coverage("both")
    assert pyopal.FullResult(0, 0, 0, 0, 0, 0, 1, 1, "").cigar() is None


# ---- Aligner construction (src/pyopal/tests/test_aligner.py:8-21; lib.pyx:1153-1256) ---
def test_aligner_construction():
    a = pyopal.Aligner()
    assert (a.gap_open, a.gap_extend, repr(a)) == (3, 1, "Aligner()")
    assert a.scoring_matrix == ScoringMatrix.from_name("BLOSUM50")
    assert a.alphabet == pyopal.Alphabet()
    b = pyopal.Aligner("BLOSUM62", 11, gap_extend=2)
    assert b.scoring_matrix == ScoringMatrix.from_name("BLOSUM62")
    assert repr(b) == "Aligner(ScoringMatrix.from_name('BLOSUM62'), gap_open=11, gap_extend=2)"
    assert pyopal.Aligner(ScoringMatrix.from_name("BLOSUM62")).scoring_matrix == b.scoring_matrix
    assert pickle.loads(pickle.dumps(b)) == b and a != b
    with pytest.raises(TypeError):
        pyopal.Aligner(1)
    with pytest.raises(ValueError):
        pyopal.Aligner(ScoringMatrix([[0.5, 0], [0, 0.5]], "AC"))
    with pytest.raises(ValueError):
        pyopal.Aligner("NOT_A_MATRIX")


def test_align_argument_validation():
    a = pyopal.Aligner()
    db = pyopal.Database(["AACCGCTG"])
    for kw in (dict(mode="fast"), dict(overflow="none"), dict(algorithm="xx")):
        with pytest.raises(ValueError):
            a.align("ACCTCG", db, **kw)
    with pytest.raises(ValueError):
        a.align("ACCTCG", pyopal.Database(["ACGT"], "ACGT"))       # alphabets differ
    with pytest.raises(ValueError):
        a.align("ACC-CG", db)                                      # bad query character
    with pytest.raises(IndexError):
        a.align("ACCTCG", db, start=1, end=0)
    with pytest.raises(TypeError):
        a.align("ACCTCG", ["AACCGCTG"])
    assert a.align("ACCTCG", pyopal.Database()) == []              # empty slice: no C call
    assert a.align("ACCTCG", db, start=1, end=1) == []
    assert list(pyopal.align("ACCTCG", [])) == []
    with pytest.raises(TypeError):
        list(pyopal.align("ACCTCG", ["AACCGCTG"], 1))


def test_scoring_matrix_provider(tmp_path):
    b62 = ScoringMatrix.from_name("BLOSUM62")
    assert b62.is_symmetric() and b62.is_integer() and b62.size() == 24
    assert b62.alphabet == "ARNDCQEGHILKMFPSTWYVBZX*" and (b62.min(), b62.max()) == (-4, 11)
    text = "# comment\n   " + "  ".join(b62.alphabet) + "\n"
    for letter, row in zip(b62.alphabet, b62.matrix):
        text += letter + " " + " ".join(str(int(x)) for x in row) + "\n"
    path = tmp_path / "m.txt"
    path.write_text(text)
    assert ScoringMatrix.from_file(str(path)) == b62
    assert pickle.loads(pickle.dumps(b62)) == b62
    dna = ScoringMatrix.from_match_mismatch(5, -4)
    assert dna.alphabet == "ACGT" and dna[0, 0] == 5 and dna[0, 1] == -4
    assert pyopal.Aligner(dna).alphabet == "ACGT"


def test_public_names():
    # src/pyopal/__init__.py:4-13
    assert sorted(pyopal.__all__) == sorted(
        ["Alphabet", "Aligner", "BaseDatabase", "Database", "ScoreResult", "EndResult", "FullResult", "align"])
    assert pyopal.lib.Aligner is pyopal.Aligner and isinstance(pyopal.__version__, str)


def test_matrix_names_beyond_the_built_in_tables(tmp_path, monkeypatch):
    # src/pyopal/tests/test_aligner.py:10-18 builds Aligner("VTML80") through scoring-matrices,
    # whose tables are not in the reference tree: here the name resolves from a user-supplied NCBI
    # file, and fails with a message that says how to supply one otherwise
    from pyopal_amd.matrices import ScoringMatrix
    monkeypatch.delenv("PYOPAL_AMD_MATRIX_PATH", raising=False)
    with pytest.raises(ValueError, match="from_file"):
        ScoringMatrix.from_name("VTML80")
    with pytest.raises(ValueError, match="scoring-matrices"):
        pyopal.Aligner("VTML80")
    b62 = ScoringMatrix.from_name("BLOSUM62")
    letters = b62.alphabet
    lines = ["# a stand-in table under another name", "   " + "  ".join(letters)]
    for letter, row in zip(letters, b62.matrix):
        lines.append(letter + " " + " ".join(str(int(x) + (1 if letter == "W" else 0)) for x in row))
    (tmp_path / "vtml80.mat").write_text("\n".join(lines) + "\n")
    monkeypatch.setenv("PYOPAL_AMD_MATRIX_PATH", str(tmp_path))
    m = ScoringMatrix.from_name("VTML80")
    assert m.alphabet == letters and m.is_integer() and m.name == "VTML80"
    assert m[letters.index("W"), letters.index("W")] == 12
    aligner = pyopal.Aligner("VTML80")
    assert aligner.scoring_matrix == m and aligner.alphabet == pyopal.Alphabet(letters)


def test_matrix_files_pinned_by_their_digest(tmp_path, monkeypatch):
    # a deployment that ships the NCBI tables this package does not carry pins them: from_file(path, sha256=...)
    # and a "<file>.sha256" beside a table found through PYOPAL_AMD_MATRIX_PATH; one changed entry is refused
    import hashlib
    from pyopal_amd.matrices import ScoringMatrix
    b62 = ScoringMatrix.from_name("BLOSUM62")
    letters = b62.alphabet
    text = "   " + "  ".join(letters) + "\n" + "".join(
        letter + " " + " ".join(str(int(x)) for x in row) + "\n" for letter, row in zip(letters, b62.matrix))
    path = tmp_path / "pam250.mat"
    path.write_text(text)
    digest = hashlib.sha256(text.encode()).hexdigest()
    assert ScoringMatrix.from_file(path, sha256=digest) == b62
    assert ScoringMatrix.from_file(str(path), sha256=digest.upper()) == b62
    with pytest.raises(ValueError, match="sha256"):
        ScoringMatrix.from_file(path, sha256="0" * 64)
    with pytest.raises(ValueError, match="path"):
        with open(path) as handle:
            ScoringMatrix.from_file(handle, sha256=digest)
    monkeypatch.setenv("PYOPAL_AMD_MATRIX_PATH", str(tmp_path))
    (tmp_path / "pam250.mat.sha256").write_text(digest + "  pam250.mat\n")
    assert ScoringMatrix.from_name("PAM250") == b62
    path.write_text(text.replace(" 11 ", " 12 ", 1))     # W/W of BLOSUM62
    with pytest.raises(ValueError, match="sha256"):
        ScoringMatrix.from_name("PAM250")
