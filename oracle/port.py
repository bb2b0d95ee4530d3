"""CPU ORACLE, Python side -- test infrastructure, NOT product code.

numpy restatement of the reference's data formats plus ctypes access to the C restatement
of its DP (oracle/sw_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; nothing under swimm_amd/ does.

Parity status: PINNED -- see the header of sw_oracle.c and tests/test_oracle_golden.py.

Restated reference lines (under /root/reference):
  alphabet recode        sequences.c:164-175, 393-402 ; sequences.h:17-18
  .seq/.info/.desc       sequences.c:128-208  
  stable length sort     sequences.c:770-865  (merge takes left on <=)
  query even-padding     sequences.c:346-409
  single-chunk assembly  sequences.c:666-723
  multi-chunk split      sequences.c:528-602
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libsw_oracle.so")

DUMMY_CODE = 23          # 'Z'+1 recoded (sequences.h:17)
PAD_CODE = 24            # PREPROCESSED_DUMMY_ELEMENT (sequences.h:18)
SEQ_LEN_MULT = 5         # sequences.h:19

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("oracle/libsw_oracle.so missing: run `make -C oracle`")
        L = C.CDLL(_PATH)
        L.sw_oracle_pair.restype = C.c_int
        _lib = L
    return _lib


# ---- alphabet ------------------------------------------------------------------------------

def recode(letters: np.ndarray) -> np.ndarray:
    """'A'..'Z' -> 0..23 : J, O, U become 'Z'+1 first, then letters above J/O/U shift down."""
    x = np.asarray(letters, dtype=np.uint8).astype(np.int16)
    for ch in (ord("J"), ord("O"), ord("U")):
        x = np.where(x == ch, ord("Z") + 1, x)
    diff = ord("A") + (x > ord("J")).astype(np.int16) + (x > ord("O")).astype(np.int16) + (x > ord("U")).astype(np.int16)
    return (x - diff).astype(np.int8)


def parse_fasta(path: str):
    """[(title_with_gt, letters uint8)] -- titles keep the leading '>' like the .desc file."""
    recs, title, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\n")
            if line.startswith(b">"):
                if title is not None:
                    recs.append((title, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
                title, chunks = line.decode("latin1"), []
            else:
                chunks.append(line)
    if title is not None:
        recs.append((title, np.frombuffer(b"".join(chunks), dtype=np.uint8)))
    return recs


def stable_sort_by_length(lengths) -> np.ndarray:
    return np.argsort(np.asarray(lengths), kind="stable")


# ---- preprocess (appendix A) ---------------------------------------------------------------

def preprocess(fasta: str, out_prefix: str) -> None:
    recs = parse_fasta(fasta)
    lens = np.array([len(s) for _, s in recs], dtype=np.int64)
    order = stable_sort_by_length(lens)
    with open(out_prefix + ".desc", "wb") as f:
        for i in order:
            f.write(recs[i][0].encode("latin1") + b"\n")
    D = int(lens.sum())
    max_title = max(len(t) for t, _ in recs) + 2  # '>'+title, newline, +1 (sequences.c:36)
    with open(out_prefix + ".info", "wb") as f:
        f.write(b"%d %d %d" % (len(recs), D, max_title))
    with open(out_prefix + ".seq", "wb") as f:
        f.write(lens[order].astype("<u2").tobytes())
        for i in order:
            f.write(recode(recs[i][1]).tobytes())


def read_preprocessed(prefix: str):
    n, D, mt = (int(x) for x in open(prefix + ".info").read().split())
    raw = np.fromfile(prefix + ".seq", dtype=np.uint8)
    lens = raw[:2 * n].view("<u2").astype(np.int64)
    codes = raw[2 * n:2 * n + D].view(np.int8)
    return {"n": n, "D": D, "max_title_length": mt, "lengths": lens, "codes": codes}


# ---- queries -------------------------------------------------------------------------------

def load_queries(fasta: str, execution_mode: int = 0):
    recs = parse_fasta(fasta)
    lens = np.array([len(s) for _, s in recs], dtype=np.int64)
    order = stable_sort_by_length(lens)
    real = lens[order]
    pad = (execution_mode != 1)  # MIC_ONLY keeps odd lengths (sequences.c:347)
    m = real + (real % 2 if pad else 0)
    disp = np.concatenate([[0], np.cumsum(m)]).astype(np.uint32)
    a = np.empty(int(m.sum()), dtype=np.int8)
    for k, i in enumerate(order):
        a[disp[k]:disp[k] + real[k]] = recode(recs[i][1])
        if m[k] != real[k]:
            a[disp[k] + real[k]] = DUMMY_CODE
    return {"a": a, "m": m.astype(np.uint16), "lengths": real.astype(np.uint16), "disp": disp,
            "titles": [recs[i][0] for i in order], "Q": int(m.sum())}


# ---- DB assembly ---------------------------------------------------------------------------

def _group_lengths(lens: np.ndarray, vl: int) -> np.ndarray:
    n = len(lens)
    vc = -(-n // vl)
    last = np.minimum((np.arange(vc) + 1) * vl - 1, n - 1)
    g = lens[last]
    return (-(-g // SEQ_LEN_MULT)) * SEQ_LEN_MULT


def assemble_single_chunk(lens: np.ndarray, codes: np.ndarray, vl: int, block_size: int):
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    gl = _group_lengths(lens, vl)
    vc = len(gl)
    disp = np.concatenate([[0], np.cumsum(gl * vl)]).astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    b = np.full(int(disp[-1]), PAD_CODE, dtype=np.int8)
    for g in range(vc):
        tile = b[int(disp[g]):int(disp[g + 1])].reshape(int(gl[g]), vl)
        for k in range(vl):
            s = g * vl + k
            if s < n:
                tile[:lens[s], k] = codes[offs[s]:offs[s + 1]]
    nbbs = -(-gl // block_size)
    return {"b": b, "n": gl.astype(np.uint16), "nbbs": nbbs.astype(np.uint16), "disp": disp, "vc": vc,
            "vD": int(disp[-1])}


def assemble_multiple_chunks(lens: np.ndarray, codes: np.ndarray, vl: int, max_chunk_size: int):
    one = assemble_single_chunk(lens, codes, vl, SEQ_LEN_MULT)
    gl = one["n"].astype(np.int64)
    vc = one["vc"]
    counts = []
    i = 0
    while i < vc:  # sequences.c:535-555 : add groups while chunk_size <= max (may overshoot by one)
        j, size = 0, 0
        while i < vc and size <= max_chunk_size:
            size += int(gl[i]) * vl + 2 + 4
            j += 1
            i += 1
        counts.append(j)
    chunks, g0 = [], 0
    for cnt in counts:
        base = int(one["disp"][g0])
        end = int(one["disp"][g0 + cnt])
        chunks.append({"b": one["b"][base:end], "n": one["n"][g0:g0 + cnt],
                       "disp": (one["disp"][g0:g0 + cnt] - np.uint64(base)).astype(np.uint32),
                       "count": cnt, "vD": end - base})
        g0 += cnt
    return {"vc": vc, "vD": one["vD"], "chunks": chunks}


# ---- DP (C restatement) --------------------------------------------------------------------

def _p(x):
    return C.c_void_p(x.ctypes.data)


def pair_score(q_codes, d_codes, submat, open_gap, extend_gap) -> int:
    q = np.ascontiguousarray(q_codes, dtype=np.int8)
    d = np.ascontiguousarray(d_codes, dtype=np.int8)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    return lib().sw_oracle_pair(_p(q), C.c_int(len(q)), _p(d), C.c_int(len(d)), _p(sm), C.c_int(open_gap), C.c_int(extend_gap))


def _search(fn_name, a, m, a_disp, b, n, b_disp, submat, open_gap, extend_gap, vl, threads, want_tiers):
    a = np.ascontiguousarray(a, dtype=np.int8)
    m = np.ascontiguousarray(m, dtype=np.uint16)
    a_disp = np.ascontiguousarray(a_disp, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.int8)
    n = np.ascontiguousarray(n, dtype=np.uint16)
    b_disp = np.ascontiguousarray(b_disp, dtype=np.uint64)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    scores = np.zeros(len(m) * len(n) * vl, dtype=np.int32)
    args = [_p(a), _p(m), C.c_ulong(len(m)), _p(a_disp), _p(b), _p(n), C.c_ulong(len(n)), _p(b_disp), _p(sm),
            C.c_int(open_gap), C.c_int(extend_gap), C.c_int(vl), C.c_int(threads), _p(scores)]
    tiers = np.zeros(3, dtype=np.int64)
    if want_tiers:
        args.append(_p(tiers))
    getattr(lib(), fn_name)(*args)
    scores = scores.reshape(len(m), len(n) * vl)
    return (scores, tiers) if want_tiers else scores


def search_tiered(a, m, a_disp, b, n, b_disp, submat, open_gap, extend_gap, vl, threads=1):
    """literal int8 -> int16 -> int32 saturating tiers; returns (scores, lanes per tier)."""
    return _search("sw_oracle_search_tiered", a, m, a_disp, b, n, b_disp, submat, open_gap, extend_gap, vl, threads, True)


def search_exact(a, m, a_disp, b, n, b_disp, submat, open_gap, extend_gap, vl, threads=None):
    """exact int32 Gotoh on the reference layout (lane loop vectorised)."""
    if threads is None:
        threads = os.cpu_count() or 1
    return _search("sw_oracle_search_exact", a, m, a_disp, b, n, b_disp, submat, open_gap, extend_gap, vl, threads, False)


def topr(scores: np.ndarray, r: int):
    """first r rows of the reference's sorted listing: (scores, sorted-DB indices)."""
    sc = np.ascontiguousarray(scores, dtype=np.int32)
    r = min(r, len(sc))
    out_s = np.zeros(r, dtype=np.int32)
    out_i = np.zeros(r, dtype=np.int64)
    lib().sw_oracle_topr(_p(sc), C.c_long(len(sc)), C.c_long(r), _p(out_s), _p(out_i))
    return out_s, out_i
