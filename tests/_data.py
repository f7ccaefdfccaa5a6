"""Seeded synthetic inputs shared by the tests."""
import numpy as np

NCBI = "ARNDCQEGHILKMFPSTWYVBZX*"  # default letters of the reference alphabet (src/pyopal/lib.pyx:193)

AA20 = "ACDEFGHIKLMNPQRSTVWY"  # same residue set as src/pyopal/tests/test_aligner.py:30
AA20_CODES = np.array([NCBI.index(c) for c in AA20], dtype=np.uint8)
README_QUERY = "MAGFLKVVQLLAKYGSKAVQWAWANKGKILDWLNAGQAIDWVVSKIKQILGIK"  # README.md:86
README_TARGETS = [  # README.md:87-92
    "MESILDLQELETSEEESALMAASTVSNNC",
    "MKKAVIVENKGCATCSIGAACLVDGPIPDFEIAGATGLFGLWG",
    "MAGFLKVVQILAKYGSKAVQWAWANKGKILDWINAGQAIDWVVEKIKQILGIK",
    "MTQIKVPTALIASVHGEGQHLFEPMAARCTCTTIISSSSTF",
]


def encode(seq):
    """Letters -> ordinals over the NCBI alphabet order (pure Python; no checker involved)."""
    return np.array([NCBI.index(c) for c in seq], dtype=np.uint8)


def random_protein(rng, length):
    return AA20_CODES[rng.integers(0, 20, size=length)]


def random_db(rng, lengths):
    """-> (residues uint8, offsets int64)"""
    lengths = np.asarray(lengths, dtype=np.int64)
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=off[1:])
    res = AA20_CODES[rng.integers(0, 20, size=int(off[-1]))]
    return np.ascontiguousarray(res), off


def mutate(rng, seq, rate=0.15):
    """Noisy copy with substitutions and indels (gives high-scoring hits)."""
    out = []
    for c in seq:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(AA20_CODES[rng.integers(0, 20)])
        out.append(c if rng.random() > rate / 3 else AA20_CODES[rng.integers(0, 20)])
    return np.array(out, dtype=np.uint8)
