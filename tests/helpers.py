"""Shared input builders for the parity tests (test infrastructure: may use oracle/)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import port, ref


def matrix(name: str) -> np.ndarray:
    """768-byte substitution table.  Product tables once built, else the reference's (oracle/_ref)."""
    try:
        from swimm_amd import submat as product_submat
        return product_submat.table(name)
    except ImportError:
        return ref.submat(name)


def golden_inputs(tmpdir, golden, vl=128, max_chunk=None):
    """queries + reference-layout DB chunks built (by the oracle's numpy restatement) from the fixtures."""
    prefix = os.path.join(str(tmpdir), "gdb")
    port.preprocess(os.path.join(GOLDEN, golden["db_fasta"]), prefix)
    pp = port.read_preprocessed(prefix)
    q = port.load_queries(os.path.join(GOLDEN, golden["query_fasta"]), 0)
    return q, pp, make_chunks(pp["lengths"], pp["codes"], vl, max_chunk)


def make_chunks(lengths, codes, vl, max_chunk=None):
    if max_chunk is None:
        one = port.assemble_single_chunk(lengths, codes, vl, 5)
        return {"vc": one["vc"], "chunks": [{"b": one["b"], "n": one["n"], "disp": one["disp"][:-1].astype(np.uint32),
                                             "count": one["vc"], "vD": one["vD"]}]}
    return port.assemble_multiple_chunks(lengths, codes, vl, max_chunk)


def load_chunks(searcher, chunked, vl):
    first = 0
    for ch in chunked["chunks"]:
        searcher.add_chunk(ch["b"], ch["n"], ch["disp"], vl, first)
        first += ch["count"]
    return first


def oracle_matrix(w, go=10, ge=2, budget_s=40.0, threads=None):
    """Score matrix of a workloads.build() dict from the CPU checker: the reference's own AVX2 path
    (oracle/_ref, cpu_search_avx2_sp CPUsearch.c:482-967) when it was built, else the C restatement.
    The whole matrix when it fits ~budget_s of the host's cores (about 1 GCUPS per hardware thread), else
    every k-th sequence.  -> (scores int32 [queries, len(idx)], idx = sorted-database indices scored)"""
    import os
    if threads is None:
        threads = len(os.sched_getaffinity(0))
    sm = matrix(w["matrix"])
    cells = float(w["query_residues"]) * float(w["residues"])
    rate = (1.0e9 if ref.available() else 0.25e9) * threads
    stride = max(1, int(np.ceil(cells / (rate * budget_s))))
    idx = np.arange(0, w["n"], stride, dtype=np.int64)
    if stride == 1:
        lens, codes = w["lengths"], w["codes"]
    else:
        lens = w["lengths"][idx]
        codes = np.concatenate([w["codes"][w["offs"][i]:w["offs"][i + 1]] for i in idx]) if len(idx) else np.zeros(0, np.int8)
    # the reference pads odd queries with one dummy residue (code 23, sequences.c:382): scores are unaffected
    real = w["m"].astype(np.int64)
    mp = real + (real % 2)
    dp = np.concatenate([[0], np.cumsum(mp)]).astype(np.uint32)
    a = np.full(int(mp.sum()), 23, dtype=np.int8)
    for k in range(len(real)):
        a[dp[k]:dp[k] + real[k]] = w["a"][w["disp"][k]:w["disp"][k] + real[k]]
    from swimm_amd import host
    one = host.assemble_single_chunk(lens, codes, 32, 60)
    if ref.available():
        sc, _ = ref.cpu_search(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["nbbs"], one["disp"], sm, go, ge, 32, threads=threads)
    else:
        sc = port.search_exact(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["disp"], sm, go, ge, 32, threads=threads)
    return sc[:, :len(idx)], idx
