"""ctypes mirror of include/swimm_hip.h -- the MI355X search back-end.

This is the host-side binding tests and bench.py use; the C `swimm` program binds the same
symbols with dlopen (swimm_amd/csrc/host/hip_loader.c).  There is NO CPU fallback here: if the
library or a gfx950 device is missing, every entry point raises.

Argument meaning follows the reference seam (cpu_search_avx2_sp, CPUsearch.h:37-39;
mic_search_knc_ap_multiple_chunks, MICsearch.h:35-38): queries as (a, m, a_disp), database as
lane-interleaved chunks (b, n, b_disp, vl), scores[q, sorted_index].
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWIMM_HIP_LIB") or os.path.join(_HERE, "lib", "libswimm_hip.so")

# every symbol include/swimm_hip.h declares (checked by tests/test_abi_symbols.py)
ABI_SYMBOLS = (
    "swimm_hip_abi_version", "swimm_hip_last_error", "swimm_hip_device_count", "swimm_hip_create",
    "swimm_hip_destroy", "swimm_hip_set_queries", "swimm_hip_add_chunk", "swimm_hip_add_sequences", "swimm_hip_clear_db",
    "swimm_hip_search", "swimm_hip_search_topr", "swimm_hip_last_stats", "swimm_hip_last_plan", "swimm_hip_last_kernel_name", "swimm_hip_last_launch_ms",
    "swimm_hip_set_option",
    "swimm_hip_search_chunks", "swimm_hip_device_pci_bus_id", "swimm_hip_bind_host_thread",
)


class SwimmHipError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen libswimm_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwimmHipError(f"{LIB_PATH} not built: run `make -C swimm_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.swimm_hip_last_error.restype = C.c_char_p
        for name in ABI_SYMBOLS:
            getattr(L, name)
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SwimmHipError(load_library().swimm_hip_last_error().decode("utf-8", "replace"))


def _p(x):
    return C.c_void_p(x.ctypes.data)


def device_count() -> int:
    n = load_library().swimm_hip_device_count()
    if n <= 0:
        raise SwimmHipError(load_library().swimm_hip_last_error().decode() or "no GPU")
    return n


def device_pci_bus_id(device: int) -> str:
    """sysfs spelling of the device's PCI address ("0000:0c:00.0")"""
    buf = C.create_string_buffer(64)
    _check(load_library().swimm_hip_device_pci_bus_id(C.c_int(device), buf, C.c_size_t(64)))
    return buf.value.decode()


def bind_host_thread(device: int, num_devices: int) -> str:
    """binds the calling thread (and the threads it starts later) to the CPUs local to `device`; -> the CPU list"""
    buf = C.create_string_buffer(1024)
    _check(load_library().swimm_hip_bind_host_thread(C.c_int(device), C.c_int(num_devices), buf, C.c_size_t(1024)))
    return buf.value.decode()


class HipSearcher:
    """One GPU: resident database + queries; mirrors the per-device thread of MICsearch.c:53-346."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        self._ctx = C.c_void_p()
        _check(self._L.swimm_hip_create(C.c_int(device), C.byref(self._ctx)))
        self.device = device
        self.n_queries = 0
        self._keep = []          # host arrays a lazy upload still points at

    def close(self):
        if self._ctx:
            self._L.swimm_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        _check(self._L.swimm_hip_set_option(self._ctx, key.encode(), C.c_int(value)))

    def set_queries(self, a, m, a_disp, submat, open_gap: int, extend_gap: int):
        a = np.ascontiguousarray(a, dtype=np.int8)
        m = np.ascontiguousarray(m, dtype=np.uint16)
        a_disp = np.ascontiguousarray(a_disp, dtype=np.uint32)
        sm = np.ascontiguousarray(submat, dtype=np.int8)
        if sm.size != 768:
            raise ValueError("submat must be 24x32 int8")
        if len(a_disp) < len(m):
            raise ValueError("a_disp shorter than m")
        _check(self._L.swimm_hip_set_queries(self._ctx, _p(a), _p(m), _p(a_disp), C.c_uint32(len(m)), _p(sm),
                                             C.c_int(open_gap), C.c_int(extend_gap)))
        self.n_queries = len(m)

    def add_chunk(self, b, n, b_disp, vl: int, first_group: int):
        b = np.ascontiguousarray(b, dtype=np.int8)
        n = np.ascontiguousarray(n, dtype=np.uint16)
        b_disp = np.ascontiguousarray(b_disp, dtype=np.uint32)
        self._keep.append((b, n, b_disp))
        _check(self._L.swimm_hip_add_chunk(self._ctx, _p(b), C.c_uint64(b.size), _p(n), _p(b_disp), C.c_uint32(len(n)),
                                           C.c_uint32(vl), C.c_uint64(first_group)))

    def add_sequences(self, lengths, codes, first_seq: int = 0):
        """sorted sequences as the .seq file stores them (lengths + concatenated codes): tiled on the device"""
        lengths = np.ascontiguousarray(lengths, dtype=np.uint16)
        codes = np.ascontiguousarray(codes, dtype=np.int8)
        if int(np.sum(lengths, dtype=np.int64)) != codes.size:
            raise ValueError("lengths do not add up to the number of codes")
        self._keep.append((lengths, codes))
        _check(self._L.swimm_hip_add_sequences(self._ctx, _p(lengths), _p(codes), C.c_uint64(len(lengths)), C.c_uint64(first_seq)))

    def clear_db(self):
        _check(self._L.swimm_hip_clear_db(self._ctx))
        self._keep = []

    def search(self, score_stride: int, out: np.ndarray | None = None):
        """-> (scores int32 [n_queries, score_stride], work_time seconds)"""
        if out is None:
            out = np.zeros((self.n_queries, score_stride), dtype=np.int32)
        assert out.dtype == np.int32 and out.flags.c_contiguous and out.shape == (self.n_queries, score_stride)
        wt = C.c_double()
        _check(self._L.swimm_hip_search(self._ctx, _p(out), C.c_uint64(score_stride), C.byref(wt)))
        self._keep = []          # everything is resident now
        return out, wt.value

    def search_topr(self, r: int, n_valid: int):
        """-> (top scores int32 [q, r], top sorted-DB index int64 [q, r], work_time)"""
        ts = np.zeros((self.n_queries, r), dtype=np.int32)
        ti = np.zeros((self.n_queries, r), dtype=np.int64)
        wt = C.c_double()
        _check(self._L.swimm_hip_search_topr(self._ctx, C.c_uint32(r), C.c_uint64(n_valid), _p(ts), _p(ti), C.byref(wt)))
        self._keep = []
        return ts, ti, wt.value

    def last_stats(self):
        ms = C.c_double(); cells = C.c_uint64(); prom = C.c_uint64(); nl = C.c_uint32()
        _check(self._L.swimm_hip_last_stats(self._ctx, C.byref(ms), C.byref(cells), C.byref(prom), C.byref(nl)))
        return {"kernel_ms": ms.value, "cells": cells.value, "promoted": prom.value, "launches": nl.value}


    def last_plan(self, q: int = 0):
        t = C.c_int(); w = C.c_int(); p = C.c_int()
        _check(self._L.swimm_hip_last_plan(self._ctx, C.c_uint32(q), C.byref(t), C.byref(w), C.byref(p)))
        return {"rows_per_wave": t.value, "waves": w.value, "passes": p.value}

    def last_launch_ms(self):
        """(sum of the pipeline launches' own durations in ms, number of launches); needs set_option("time_launches", 1)"""
        ms = C.c_double(); n = C.c_uint32()
        _check(self._L.swimm_hip_last_launch_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_kernel_name(self, q: int = 0) -> str:
        buf = C.create_string_buffer(256)
        _check(self._L.swimm_hip_last_kernel_name(self._ctx, C.c_uint32(q), buf, C.c_size_t(len(buf))))
        return buf.value.decode()


def search_chunks(a, m, a_disp, vc_total: int, chunks, submat, open_gap: int, extend_gap: int, num_gpus: int, vl: int):
    """Whole-call drop-in (swimm_hip_search_chunks == mic_search_knc_ap_multiple_chunks' argument list).
    chunks: list of dicts with b, n, disp (reference chunk layout).  -> (scores [q, vc_total*vl], workTime)"""
    L = load_library()
    a = np.ascontiguousarray(a, dtype=np.int8)
    m = np.ascontiguousarray(m, dtype=np.uint16)
    a_disp = np.ascontiguousarray(a_disp, dtype=np.uint32)
    sm = np.ascontiguousarray(submat, dtype=np.int8)
    cc = len(chunks)
    keep = []
    cb = (C.c_void_p * cc)(); cn = (C.c_void_p * cc)(); cd = (C.c_void_p * cc)()
    ccnt = np.zeros(cc, dtype=np.uint32); cvd = np.zeros(cc, dtype=np.uint64)
    for i, ch in enumerate(chunks):
        b = np.ascontiguousarray(ch["b"], dtype=np.int8)
        n = np.ascontiguousarray(ch["n"], dtype=np.uint16)
        d = np.ascontiguousarray(ch["disp"], dtype=np.uint32)
        keep += [b, n, d]
        cb[i], cn[i], cd[i] = b.ctypes.data, n.ctypes.data, d.ctypes.data
        ccnt[i], cvd[i] = len(n), b.size
    scores = np.zeros((len(m), vc_total * vl), dtype=np.int32)
    wt = C.c_double()
    _check(L.swimm_hip_search_chunks(_p(a), _p(m), C.c_uint32(len(m)), _p(a_disp), C.c_uint64(vc_total), cb, C.c_uint32(cc),
                                     _p(ccnt), cn, cd, _p(cvd), _p(sm), C.c_int(open_gap), C.c_int(extend_gap),
                                     C.c_int(num_gpus), C.c_uint32(vl), _p(scores), C.byref(wt)))
    return scores, wt.value
