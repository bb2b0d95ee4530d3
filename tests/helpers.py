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
