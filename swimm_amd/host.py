"""ctypes mirror of swimm_amd/csrc/host/swimm_host.h -- the host-side C half of the build
(file formats, query batch, lane-interleaved layout, top-r, substitution tables).

Same functions the C `swimm` program uses; exposed here so tests read like the reference's own
call sites (preprocess_db / load_query_sequences / assemble_*_db / sort_scores) and so bench.py can
lay out a multi-hundred-MB synthetic database at C speed.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libswimm_host.so")

HOST_SYMBOLS = (
    "swimm_host_last_error", "swimm_recode", "swimm_fasta_read", "swimm_fasta_free", "swimm_preprocess_db",
    "swimm_db_load", "swimm_db_free", "swimm_db_titles", "swimm_queries_load", "swimm_queries_free",
    "swimm_assemble_single_chunk", "swimm_single_chunk_free", "swimm_assemble_chunks", "swimm_chunks_free",
    "swimm_topr", "swimm_topr_merge", "swimm_cpu_search", "swimm_submat", "swimm_submat_label", "swimm_wtime",
    "swimm_affinity_plan", "swimm_affinity_allowed", "swimm_affinity_apply",
)


class SwimmHostError(RuntimeError):
    def __init__(self, msg, status=0):
        super().__init__(msg)
        self.status = status


class _Db(C.Structure):
    _fields_ = [("count", C.c_uint64), ("residues", C.c_uint64), ("max_title_length", C.c_int),
                ("lengths", C.POINTER(C.c_uint16)), ("codes", C.c_void_p), ("map_base", C.c_void_p), ("map_bytes", C.c_uint64)]


class _Queries(C.Structure):
    _fields_ = [("count", C.c_uint64), ("Q", C.c_uint64), ("a", C.c_void_p), ("m", C.POINTER(C.c_uint16)),
                ("lengths", C.POINTER(C.c_uint16)), ("disp", C.POINTER(C.c_uint32)), ("titles", C.POINTER(C.c_char_p)),
                ("arena_", C.c_void_p)]


class _Single(C.Structure):
    _fields_ = [("vc", C.c_uint64), ("vD", C.c_uint64), ("b", C.c_void_p), ("n", C.POINTER(C.c_uint16)),
                ("nbbs", C.POINTER(C.c_uint16)), ("disp", C.POINTER(C.c_uint64))]


class _Chunks(C.Structure):
    _fields_ = [("vc", C.c_uint64), ("vD", C.c_uint64), ("chunk_count", C.c_uint32), ("b_all", C.c_void_p),
                ("chunk_b", C.POINTER(C.c_void_p)), ("chunk_groups", C.POINTER(C.c_uint32)),
                ("chunk_n", C.POINTER(C.POINTER(C.c_uint16))), ("chunk_disp", C.POINTER(C.POINTER(C.c_uint32))),
                ("chunk_vD", C.POINTER(C.c_uint64)), ("chunk_first_group", C.POINTER(C.c_uint64)),
                ("n_all_", C.c_void_p), ("disp_all_", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwimmHostError(f"{LIB_PATH} not built: run `make -C swimm_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        L.swimm_host_last_error.restype = C.c_char_p
        L.swimm_submat.restype = C.c_void_p
        L.swimm_submat_label.restype = C.c_char_p
        L.swimm_wtime.restype = C.c_double
        for name in HOST_SYMBOLS:
            getattr(L, name)
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SwimmHostError(lib().swimm_host_last_error().decode("utf-8", "replace"), rc)


def _np(ptr, n, dtype):
    """copy n items from a C pointer"""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).copy()


def _p(x):
    return C.c_void_p(x.ctypes.data)


def recode(letters) -> np.ndarray:
    x = np.array(letters, dtype=np.uint8, copy=True)
    lib().swimm_recode(_p(x), C.c_size_t(x.size))
    return x.view(np.int8)


def submat(name: str) -> np.ndarray:
    p = lib().swimm_submat(name.encode())
    if not p:
        raise SwimmHostError(f"{name} is not a valid option for substitution matrix.")
    return _np(p, 768, np.int8)


def preprocess_db(fasta: str, out_prefix: str):
    n = C.c_uint64(); d = C.c_uint64()
    _check(lib().swimm_preprocess_db(fasta.encode(), out_prefix.encode(), C.byref(n), C.byref(d)))
    return n.value, d.value


def db_load(prefix: str):
    db = _Db()
    _check(lib().swimm_db_load(prefix.encode(), C.byref(db)))
    out = {"count": db.count, "residues": db.residues, "max_title_length": db.max_title_length,
           "lengths": _np(db.lengths, db.count, np.uint16), "codes": _np(db.codes, db.residues, np.int8)}
    lib().swimm_db_free(C.byref(db))
    return out


def db_titles(prefix: str, count: int, idx):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = (C.c_void_p * len(idx))()
    _check(lib().swimm_db_titles(prefix.encode(), C.c_uint64(count), _p(idx), C.c_uint64(len(idx)), out))
    libc = C.CDLL(None)
    titles = []
    for p in out:
        titles.append(C.string_at(p).decode("latin1"))
        libc.free(C.c_void_p(p))
    return titles


def queries_load(fasta: str, pad_even: bool = True):
    q = _Queries()
    _check(lib().swimm_queries_load(fasta.encode(), C.c_int(1 if pad_even else 0), C.byref(q)))
    n = q.count
    out = {"a": _np(q.a, q.Q, np.int8), "m": _np(q.m, n, np.uint16), "lengths": _np(q.lengths, n, np.uint16),
           "disp": _np(q.disp, n + 1, np.uint32), "titles": [q.titles[i].decode("latin1") for i in range(n)], "Q": q.Q}
    lib().swimm_queries_free(C.byref(q))
    return out


def assemble_single_chunk(lengths, codes, vl: int, block_size: int):
    lengths = np.ascontiguousarray(lengths, dtype=np.uint16)
    codes = np.ascontiguousarray(codes, dtype=np.int8)
    s = _Single()
    _check(lib().swimm_assemble_single_chunk(_p(lengths), _p(codes), C.c_uint64(len(lengths)), C.c_int(vl),
                                             C.c_int(block_size), C.byref(s)))
    out = {"vc": s.vc, "vD": s.vD, "b": _np(s.b, s.vD, np.int8), "n": _np(s.n, s.vc, np.uint16),
           "nbbs": _np(s.nbbs, s.vc, np.uint16), "disp": _np(s.disp, s.vc + 1, np.uint64)}
    lib().swimm_single_chunk_free(C.byref(s))
    return out


class Chunks:
    """Chunked lane-interleaved database; owns the C buffers (numpy views, no copy of the residues)."""

    def __init__(self, lengths, codes, vl: int, max_chunk_size: int):
        lengths = np.ascontiguousarray(lengths, dtype=np.uint16)
        codes = np.ascontiguousarray(codes, dtype=np.int8)
        self._c = _Chunks()
        _check(lib().swimm_assemble_chunks(_p(lengths), _p(codes), C.c_uint64(len(lengths)), C.c_int(vl),
                                           C.c_uint64(max_chunk_size), C.byref(self._c)))
        c = self._c
        self.vl, self.vc, self.vD = vl, c.vc, c.vD
        self.chunks = []
        for i in range(c.chunk_count):
            k = c.chunk_groups[i]
            vd = c.chunk_vD[i]
            b = np.frombuffer((C.c_char * vd).from_address(c.chunk_b[i]), dtype=np.int8)
            n = np.frombuffer((C.c_char * (2 * k)).from_address(C.cast(c.chunk_n[i], C.c_void_p).value), dtype=np.uint16)
            d = np.frombuffer((C.c_char * (4 * k)).from_address(C.cast(c.chunk_disp[i], C.c_void_p).value), dtype=np.uint32)
            self.chunks.append({"b": b, "n": n, "disp": d, "count": int(k), "vD": int(vd),
                                "first_group": int(c.chunk_first_group[i])})

    def close(self):
        if self._c is not None:
            self.chunks = []
            lib().swimm_chunks_free(C.byref(self._c))
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cpu_search(a, m, a_disp, b, n, b_disp, submat_tbl, open_gap, extend_gap, vl, threads=1, block_size=60):
    """execution mode 0 (explicit host-CPU search); argument order of cpu_search_avx2_sp + vl.
    -> (scores int32 [q, vc*vl], work_time)"""
    a = np.ascontiguousarray(a, dtype=np.int8); m = np.ascontiguousarray(m, dtype=np.uint16)
    a_disp = np.ascontiguousarray(a_disp, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.int8)
    n = np.ascontiguousarray(n, dtype=np.uint16); b_disp = np.ascontiguousarray(b_disp, dtype=np.uint64)
    sm = np.ascontiguousarray(submat_tbl, dtype=np.int8)
    scores = np.zeros((len(m), len(n) * vl), dtype=np.int32)
    wt = C.c_double()
    _check(lib().swimm_cpu_search(_p(a), _p(m), C.c_uint64(len(m)), _p(a_disp), _p(b), _p(n), C.c_uint64(len(n)), _p(b_disp),
                                  _p(sm), C.c_int(open_gap), C.c_int(extend_gap), C.c_int(threads), C.c_int(block_size),
                                  C.c_int(vl), _p(scores), C.byref(wt)))
    return scores, wt.value


def topr(scores, r: int):
    sc = np.ascontiguousarray(scores, dtype=np.int32)
    out_s = np.zeros(r, dtype=np.int32)
    out_i = np.zeros(r, dtype=np.int64)
    lib().swimm_topr(_p(sc), C.c_uint64(len(sc)), C.c_uint32(r), _p(out_s), _p(out_i))
    return out_s, out_i


def topr_merge(scores, idx, r: int):
    """scores / idx: [lists, r] per-shard top-r (idx -1 = empty) -> merged (scores[r], idx[r])"""
    sc = np.ascontiguousarray(scores, dtype=np.int32)
    ix = np.ascontiguousarray(idx, dtype=np.int64)
    lists = sc.size // r
    out_s = np.zeros(r, dtype=np.int32)
    out_i = np.zeros(r, dtype=np.int64)
    lib().swimm_topr_merge(_p(sc), _p(ix), C.c_uint32(lists), C.c_uint32(r), _p(out_s), _p(out_i))
    return out_s, out_i


def _cpulist(cpus) -> str:
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    out, i, cpus = [], 0, list(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(f"{cpus[i]}-{cpus[j]}" if j > i else f"{cpus[i]}")
        i = j + 1
    return ",".join(out)


def affinity_plan(pci_bdf, device: int, allowed=None, sysfs_root: str = "/sys"):
    """CPUs for the host threads of device `device` of len(pci_bdf) devices (affinity.h: the device's sysfs local_cpulist,
    shared by whole physical cores among the devices naming the same CPUs; an even share of `allowed` otherwise)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else list(allowed)
    n = len(pci_bdf)
    arr = (C.c_char_p * n)(*[(b or "").encode() for b in pci_bdf])
    al = np.ascontiguousarray(allowed, dtype=np.int32)
    out = np.zeros(4096, dtype=np.int32)
    got = lib().swimm_affinity_plan(sysfs_root.encode(), arr, C.c_int(n), C.c_int(device), _p(al), C.c_int(len(al)), _p(out), C.c_int(len(out)))
    if got <= 0:
        raise SwimmHostError(f"no CPU plan for device {device} of {n}")
    return [int(x) for x in out[:got]]


def affinity_apply(cpus):
    """binds the calling thread (threads started afterwards inherit it)"""
    a = np.ascontiguousarray(list(cpus), dtype=np.int32)
    if lib().swimm_affinity_apply(_p(a), C.c_int(len(a))) != 0:
        raise SwimmHostError("sched_setaffinity failed")
