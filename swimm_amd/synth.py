"""Deterministic synthetic protein data (queries, databases, FASTA text).

Real UniProt / NCBI data is not available offline, so every benchmark and test
input is produced here from a counter-based splitmix64 stream: the same
(seed, stream, index) always gives the same value on any machine, and any block
of a large database can be generated independently of the rest.

Shapes follow SURVEY.md section 8(d): residues are i.i.d. from the
Robinson-Robinson background frequencies plus 0.1 % each of B, Z, X, U (so the
ambiguity codes and the J/O/U -> 23 dummy code occur), sequence lengths are
normal / log-normal with the means quoted there, and every query gets planted
homologs (0/10/30/50 % substituted copies in random flanks) so the top-r list
is non-trivial and the int16 -> int32 promotion tier fires.
"""
from __future__ import annotations

import numpy as np

AA20 = "ARNDCQEGHILKMFPSTWYV"
_BG = [7.8, 5.1, 4.5, 5.4, 1.9, 4.3, 6.3, 7.4, 2.2, 5.1, 9.0, 5.7, 2.2, 3.9, 5.2, 7.1, 5.8, 1.3, 3.2, 6.4]
_EXTRA = "BZXU"
_EXTRA_P = [0.1, 0.1, 0.1, 0.1]

# the 20-query benchmark set used in the Smith-Waterman literature (lengths only;
# accession names are kept as titles, the residues are synthetic)
QUERY_SET = [
    ("P02232", 144), ("P05013", 189), ("P14942", 222), ("P07327", 375), ("P01008", 464),
    ("P03435", 567), ("P42357", 657), ("P21177", 729), ("Q38941", 850), ("P27895", 1000),
    ("P07756", 1500), ("P04775", 2005), ("P19096", 2504), ("P28167", 3005), ("P0C6B8", 3564),
    ("P20930", 4061), ("P08519", 4548), ("Q7TMA5", 4743), ("P33450", 5147), ("Q9UKN1", 5478),
]

_U64 = np.uint64
_MASK = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser of a uint64 counter array."""
    with np.errstate(over="ignore"):
        z = (x + _U64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _stream_base(seed: int, stream: int) -> np.uint64:
    s = splitmix64(np.array([(seed * 0x100000001B3 + stream * 0x9E3779B1) & _MASK], dtype=np.uint64))[0]
    return s


def u64(seed: int, stream: int, start: int, count: int) -> np.ndarray:
    """count raw 64-bit values; value i depends only on (seed, stream, start+i)."""
    base = _stream_base(seed, stream)
    with np.errstate(over="ignore"):
        ctr = (np.arange(start, start + count, dtype=np.uint64) * _U64(0xD1342543DE82EF95)) + base
    return splitmix64(ctr)


def uniform01(seed: int, stream: int, start: int, count: int) -> np.ndarray:
    return (u64(seed, stream, start, count) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _residue_lut() -> np.ndarray:
    letters = list(AA20) + list(_EXTRA)
    p = np.array(_BG + _EXTRA_P, dtype=np.float64)
    cum = np.cumsum(p / p.sum())
    edges = np.minimum((cum * 65536.0 + 0.5).astype(np.int64), 65536)
    lut = np.empty(65536, dtype=np.uint8)
    lo = 0
    for ch, hi in zip(letters, edges):
        lut[lo:hi] = ord(ch)
        lo = hi
    lut[lo:] = ord(letters[-1])
    return lut


_LUT = _residue_lut()


def residues(seed: int, stream: int, start: int, count: int) -> np.ndarray:
    """count ASCII residue letters (uint8); four residues per 64-bit draw."""
    if count <= 0:
        return np.empty(0, dtype=np.uint8)
    first = start // 4
    last = (start + count + 3) // 4
    raw = u64(seed, stream, first, last - first)
    idx16 = raw.view(np.uint16)  # little-endian: 4 x 16 bit per draw
    off = start - first * 4
    return _LUT[idx16[off:off + count]]


def lengths_normal(seed: int, n: int, mean: float, sd: float, lo: int, hi: int) -> np.ndarray:
    u1 = np.maximum(uniform01(seed, 101, 0, n), 1e-300)
    u2 = uniform01(seed, 102, 0, n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return np.clip(np.rint(mean + sd * z), lo, hi).astype(np.int64)


def lengths_lognormal(seed: int, n: int, mean: float, sigma: float, lo: int, hi: int) -> np.ndarray:
    """log-normal with the requested arithmetic mean (before clipping)."""
    u1 = np.maximum(uniform01(seed, 101, 0, n), 1e-300)
    u2 = uniform01(seed, 102, 0, n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    mu = np.log(mean) - 0.5 * sigma * sigma
    return np.clip(np.rint(np.exp(mu + sigma * z)), lo, hi).astype(np.int64)


def make_query(seed: int, qi: int, length: int) -> np.ndarray:
    return residues(seed, 1000 + qi, 0, length)


def make_queries(seed: int, lengths=None):
    """[(title, letters uint8)] for the given lengths (default: the 20-query set)."""
    if lengths is None:
        items = QUERY_SET
    else:
        items = [(f"SYNQ{i:02d}", int(L)) for i, L in enumerate(lengths)]
    out = []
    for qi, (name, L) in enumerate(items):
        out.append((f"sp|{name}|SYN_{L} synthetic query {L} aa", make_query(seed, qi, L)))
    return out


def mutate(seq: np.ndarray, rate: float, seed: int, stream: int) -> np.ndarray:
    """substitute a fraction `rate` of positions by fresh background residues."""
    if rate <= 0.0:
        return seq.copy()
    u = uniform01(seed, stream, 0, len(seq))
    repl = residues(seed, stream + 1, 0, len(seq))
    return np.where(u < rate, repl, seq).astype(np.uint8)


def planted_homologs(seed: int, queries, rates=(0.0, 0.1, 0.3, 0.5), max_len: int = 65535):
    """[(title, letters)] : for each query one copy per substitution rate, in random flanks."""
    out = []
    for qi, (title, q) in enumerate(queries):
        for ri, rate in enumerate(rates):
            st = 5000 + qi * 100 + ri * 10
            fl = (u64(seed, st, 0, 2) % _U64(81)).astype(np.int64) + 20
            room = max_len - len(q)
            left = int(min(fl[0], max(room // 2, 0)))
            right = int(min(fl[1], max(room - left, 0)))
            body = mutate(q, rate, seed, st + 2)
            seq = np.concatenate([residues(seed, st + 4, 0, left), body, residues(seed, st + 5, 0, right)])
            acc = title.split("|")[1] if "|" in title else f"Q{qi}"
            out.append((f"syn|HOM_{acc}_{int(rate * 100):02d}|planted homolog of {acc} at {int(rate * 100)}% substitution", seq))
    return out


class SynthDB:
    """A database as (titles?, lengths, concatenated letters) in ORIGINAL (unsorted) order."""

    def __init__(self, lengths: np.ndarray, letters: np.ndarray, titles=None):
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.letters = letters
        self.titles = titles

    @property
    def n(self):
        return len(self.lengths)

    @property
    def residues(self):
        return int(self.lengths.sum())


def make_db(seed: int, lengths: np.ndarray, planted=None, with_titles: bool = True, block: int = 1 << 26) -> SynthDB:
    """Random background sequences of the given lengths, plus planted sequences spliced in at
    deterministic positions (so the unsorted order is not trivially 'planted last')."""
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    letters = np.empty(total, dtype=np.uint8)
    for s in range(0, total, block):
        e = min(total, s + block)
        letters[s:e] = residues(seed, 7, s, e - s)
    titles = None
    if with_titles:
        titles = [f"syn|S{seed}_{i:08d}|synthetic protein {i} len {int(L)}" for i, L in enumerate(lengths)]
    if planted:
        n0 = len(lengths)
        pos = (u64(seed, 9, 0, len(planted)) % _U64(n0 + 1)).astype(np.int64)
        order = np.argsort(pos, kind="stable")
        offs = np.concatenate([[0], np.cumsum(lengths)])
        pieces_len, pieces, new_titles = [], [], []
        prev = 0
        for k in order:
            p = int(pos[k])
            pieces.append(letters[offs[prev]:offs[p]])
            pieces_len.append(lengths[prev:p])
            if with_titles:
                new_titles.extend(titles[prev:p])
                new_titles.append(planted[k][0])
            pieces.append(planted[k][1])
            pieces_len.append(np.array([len(planted[k][1])], dtype=np.int64))
            prev = p
        pieces.append(letters[offs[prev]:])
        pieces_len.append(lengths[prev:])
        if with_titles:
            new_titles.extend(titles[prev:])
            titles = new_titles
        letters = np.concatenate(pieces)
        lengths = np.concatenate(pieces_len)
    return SynthDB(lengths, letters, titles)


def write_fasta(path: str, records, width: int = 60) -> None:
    """records: iterable of (title, letters uint8).  60-80 column lines, trailing newline."""
    with open(path, "wb") as f:
        for title, seq in records:
            f.write(b">" + title.encode("ascii") + b"\n")
            b = seq.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


def db_records(db: SynthDB):
    offs = np.concatenate([[0], np.cumsum(db.lengths)])
    for i in range(db.n):
        yield db.titles[i], db.letters[offs[i]:offs[i + 1]]


# ---- the BASELINE.json configurations ------------------------------------------------------

def config_lengths(name: str, scale: float = 1.0) -> np.ndarray:
    """Sequence-length vector of a BASELINE config ('c1'..'c5'); `scale` shrinks N."""
    if name == "c1":
        return lengths_normal(1, max(1, int(1000 * scale)), 300, 100, 30, 1000)
    if name == "c2":
        return lengths_lognormal(2, max(1, int(1_000_000 * scale)), 600.0, 0.55, 30, 5000)
    if name == "c3":
        n = max(1, int(540_000 * scale))
        L = lengths_lognormal(3, n, 360.0, 0.65, 20, 12000)
        ntail = max(1, n // 10000)
        tail = (u64(3, 55, 0, ntail) % _U64(35000 - 12000)).astype(np.int64) + 12000
        L[:ntail] = tail
        return L
    if name in ("c4", "c5"):
        return lengths_lognormal(5, max(1, int(35_500_000 * scale)), 197.0, 0.6, 20, 7500)
    raise ValueError(name)
