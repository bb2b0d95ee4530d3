"""ctypes binding of oracle/_ref/libswimm_ref.so -- the REFERENCE's own CPU hot path.

ORACLE / TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; nothing under swimm_amd/ does.

The .so is compiled by oracle/Makefile from four of the reference's own source files
(CPUsearch.c, sequences.c, utils.c, submat.c) where they lie under /root/reference; no
reference source is stored in this repository.  It is used to (1) generate the golden
vectors under tests/golden/, (2) validate the C restatement in sw_oracle.c, and (3) time
the reference's AVX2 path on the GPU box's host cores (cpu_baseline.kind = "reference").

Each wrapper names the reference function it calls (file:line in /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libswimm_ref.so")

MATRICES = ("blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250")


def available() -> bool:
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libswimm_ref.so missing: run `make -C oracle` where /root/reference exists")
        _lib = C.CDLL(_PATH)
    return _lib


def submat(name: str) -> np.ndarray:
    """768-byte table exported by submat.c:4-227 (24 rows x 32 cols, int8)."""
    arr = (C.c_byte * 768).in_dll(lib(), name)
    return np.frombuffer(bytes(arr), dtype=np.int8).copy()


def _aligned(n, dtype, align=64):
    dtype = np.dtype(dtype)
    raw = np.zeros(n * dtype.itemsize + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * dtype.itemsize].view(dtype)


def preprocess_db(fasta: str, out_prefix: str, threads: int = 1) -> None:
    """preprocess_db, sequences.c:4-220 (writes <out>.seq/.info/.desc; prints a report)."""
    lib().preprocess_db(fasta.encode(), out_prefix.encode(), C.c_int(threads))


def load_queries(fasta: str, execution_mode: int = 0, threads: int = 1):
    """load_query_sequences, sequences.c:223-423.  Returns dict with a (codes, concatenated,
    even-padded in modes 0/2), m (padded lengths), lengths (real), disp, titles, Q."""
    a = C.c_char_p()
    hdr = C.POINTER(C.c_char_p)()
    lens = C.POINTER(C.c_ushort)()
    m = C.POINTER(C.c_ushort)()
    cnt = C.c_ulong()
    Q = C.c_ulong()
    disp = C.POINTER(C.c_uint)()
    lib().load_query_sequences(fasta.encode(), C.c_int(execution_mode), C.byref(a), C.byref(hdr), C.byref(lens),
                               C.byref(m), C.byref(cnt), C.byref(Q), C.byref(disp), C.c_int(threads))
    n = cnt.value
    a_np = np.frombuffer(C.string_at(a, Q.value), dtype=np.int8).copy()
    return {
        "a": a_np,
        "m": np.array([m[i] for i in range(n)], dtype=np.uint16),
        "lengths": np.array([lens[i] for i in range(n)], dtype=np.uint16),
        "disp": np.array([disp[i] for i in range(n + 1)], dtype=np.uint32),
        "titles": [hdr[i].decode("latin1") for i in range(n)],
        "Q": Q.value,
    }


def assemble_single_chunk(db_prefix: str, vector_length: int, block_size: int, threads: int = 1):
    """assemble_single_chunk_db, sequences.c:618-734."""
    cnt = C.c_ulong(); D = C.c_ulong(); maxlen = C.c_ushort(); maxtitle = C.c_int()
    vc = C.c_ulong(); vD = C.c_ulong()
    b = C.c_void_p(); n = C.POINTER(C.c_ushort)(); nbbs = C.POINTER(C.c_ushort)(); disp = C.POINTER(C.c_ulong)()
    lib().assemble_single_chunk_db(db_prefix.encode(), C.c_int(vector_length), C.byref(cnt), C.byref(D), C.byref(maxlen),
                                   C.byref(maxtitle), C.byref(vc), C.byref(vD), C.byref(b), C.byref(n), C.byref(nbbs),
                                   C.byref(disp), C.c_int(threads), C.c_int(block_size))
    k = vc.value
    return {
        "sequences_count": cnt.value, "D": D.value, "max_length": maxlen.value, "max_title_length": maxtitle.value,
        "vc": k, "vD": vD.value,
        "b": np.frombuffer(C.string_at(b, vD.value), dtype=np.int8).copy(),
        "n": np.array([n[i] for i in range(k)], dtype=np.uint16),
        "nbbs": np.array([nbbs[i] for i in range(k)], dtype=np.uint16),
        "disp": np.array([disp[i] for i in range(k + 1)], dtype=np.uint64),
    }


def assemble_multiple_chunks(db_prefix: str, vector_length: int, max_chunk_size: int, threads: int = 1):
    """assemble_multiple_chunks_db, sequences.c:425-616."""
    cnt = C.c_ulong(); D = C.c_ulong(); maxlen = C.c_ushort(); maxtitle = C.c_int()
    vc = C.c_ulong(); vD = C.c_ulong(); cc = C.c_uint()
    cb = C.POINTER(C.c_void_p)(); ccnt = C.POINTER(C.c_uint)(); cvD = C.POINTER(C.c_ulong)()
    cn = C.POINTER(C.POINTER(C.c_ushort))(); cdisp = C.POINTER(C.POINTER(C.c_uint))()
    lib().assemble_multiple_chunks_db(db_prefix.encode(), C.c_int(vector_length), C.c_ulong(max_chunk_size), C.byref(cnt),
                                      C.byref(D), C.byref(maxlen), C.byref(maxtitle), C.byref(vc), C.byref(vD),
                                      C.byref(cb), C.byref(cc), C.byref(ccnt), C.byref(cvD), C.byref(cn), C.byref(cdisp),
                                      C.c_int(threads))
    chunks = []
    for c in range(cc.value):
        k = ccnt[c]
        chunks.append({
            "b": np.frombuffer(C.string_at(cb[c], cvD[c]), dtype=np.int8).copy(),
            "n": np.array([cn[c][i] for i in range(k)], dtype=np.uint16),
            "disp": np.array([cdisp[c][i] for i in range(k)], dtype=np.uint32),
            "count": k, "vD": cvD[c],
        })
    return {"sequences_count": cnt.value, "D": D.value, "max_length": maxlen.value, "max_title_length": maxtitle.value,
            "vc": vc.value, "vD": vD.value, "chunks": chunks}


def cpu_search(a, m, a_disp, b, n, nbbs, b_disp, submat_tbl, open_gap, extend_gap, vector_length=32,
               threads=1, block_size=None):
    """cpu_search_avx2_sp (CPUsearch.c:482-967) for vector_length 32, cpu_search_sse_sp
    (CPUsearch.c:6-479) for 16.  Returns (scores int32 [q_cnt, vc*VL], workTime seconds)."""
    if block_size is None:
        block_size = 60 if vector_length == 32 else 125  # swimm.c:32-35
    qcnt = len(m)
    vc = len(n)
    a_ = _aligned(len(a) + 64, np.int8); a_[:len(a)] = a
    b_ = _aligned(len(b) + 64, np.int8); b_[:len(b)] = b
    m_ = _aligned(qcnt, np.uint16); m_[:] = m
    ad_ = _aligned(len(a_disp), np.uint32); ad_[:] = a_disp
    n_ = _aligned(vc, np.uint16); n_[:] = n
    nb_ = _aligned(vc, np.uint16); nb_[:] = nbbs
    bd_ = _aligned(len(b_disp), np.uint64); bd_[:] = b_disp
    sm_ = _aligned(768, np.int8); sm_[:] = np.asarray(submat_tbl, dtype=np.int8)
    scores = _aligned(qcnt * vc * vector_length, np.int32)
    wt = C.c_double()
    fn = lib().cpu_search_avx2_sp if vector_length == 32 else lib().cpu_search_sse_sp
    fn(C.c_void_p(a_.ctypes.data), C.c_void_p(m_.ctypes.data), C.c_ulong(qcnt), C.c_void_p(ad_.ctypes.data),
       C.c_void_p(b_.ctypes.data), C.c_void_p(n_.ctypes.data), C.c_void_p(nb_.ctypes.data), C.c_ulong(vc),
       C.c_void_p(bd_.ctypes.data), C.c_void_p(sm_.ctypes.data), C.c_int(open_gap), C.c_int(extend_gap),
       C.c_int(threads), C.c_int(block_size), C.c_void_p(scores.ctypes.data), C.byref(wt))
    return scores.reshape(qcnt, vc * vector_length).copy(), wt.value


def sort_scores(scores: np.ndarray, threads: int = 1):
    """sort_scores, utils.c:71-86, on one query's first N scores.  The 'titles' column is an
    array of fake pointers holding the original index, so the permutation comes back too."""
    n = len(scores)
    sc = np.ascontiguousarray(scores, dtype=np.int32).copy()
    idx = np.arange(1, n + 1, dtype=np.uint64)  # non-NULL fake pointers
    lib().sort_scores(C.c_void_p(sc.ctypes.data), C.c_void_p(idx.ctypes.data), C.c_ulong(n), C.c_int(threads))
    return sc, (idx - 1).astype(np.int64)
