#!/usr/bin/env python3
"""Batches of short queries against a small database (1e8 residues, c2-shaped): the regime where a single long
sequence's serial chain is longer than a query's whole bulk work.  usage: [SQ_SCALE=0.17] [SQ_MEDIUM=1] [SWIMM_HIP_OPTIONS=...] python tools/short_query_bench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, host, submat, synth  # noqa: E402

L = np.sort(synth.config_lengths("c2", float(os.environ.get("SQ_SCALE", "0.17")))).astype(np.uint16)
total = int(L.astype(np.int64).sum())
codes = host.recode(synth.residues(2, 7, 0, total))
rng = np.random.default_rng(1)
SETS = ((100, 600, 700), (60, 1200, 1400), (200, 50, 1500)) if os.environ.get('SQ_MEDIUM') else ((300, 80, 120), (100, 280, 320), (1000, 20, 60), (100, 600, 700), (60, 1200, 1400), (200, 50, 1500), (3, 80, 120), (1, 80, 120))
if os.environ.get('SQ_SET'):            # e.g. SQ_SET=100,400,440: 100 queries of 400-440 residues
    SETS = (tuple(int(x) for x in os.environ['SQ_SET'].split(',')),)
elif os.environ.get('SQ_ONLY'):
    SETS = tuple(SETS[int(i)] for i in os.environ['SQ_ONLY'].split(','))
for (nq, lo, hi) in SETS:
    ms = np.sort(rng.integers(lo, hi, nq)).astype(np.uint16)
    a = host.recode(synth.residues(2, 11, 0, int(ms.sum())))
    disp = np.concatenate([[0], np.cumsum(ms.astype(np.int64))]).astype(np.uint32)
    with hip_backend.HipSearcher(0) as s:
        s.add_sequences(L, codes, 0)
        if os.environ.get('SQ_TAIL_FRAC'):
            s.set_option('tail_frac', int(os.environ['SQ_TAIL_FRAC']))
        s.set_queries(a, ms, disp, submat.table("blosum62"), 10, 2)
        s.search_topr(10, len(L))
        ts, ti, wt = s.search_topr(10, len(L))
        st = s.last_stats()
        cells = float(ms.astype(np.int64).sum()) * total
        print(f"{nq} queries of {lo}-{hi} residues x {total} residues: {wt:.3f} s (device {st['kernel_ms'] / 1e3:.3f} s) -> "
              f"{cells / wt / 1e9:.0f} GCUPS, {st['launches']} launches", flush=True)
