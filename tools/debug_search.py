#!/usr/bin/env python3
"""One search of a BASELINE configuration with the library's debug lines (SWIMM_HIP_DEBUG) on stderr: launch shapes,
streams, work lists per query.  usage: python tools/debug_search.py c3 1.0 [key=val,...]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, submat, workloads  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
opts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
w = workloads.build(cfg, scale)
with hip_backend.HipSearcher(0) as s:
    for k, v in opts.items():
        s.set_option(k, int(v))
    s.set_queries(w["a"], w["m"], w["disp"], submat.table(w["matrix"]), 10, 2)
    s.add_sequences(w["lengths"], w["codes"], 0)
    s.search_topr(20, w["n"])
    os.environ["SWIMM_HIP_DEBUG"] = "1"
    ts, ti, wt = s.search_topr(20, w["n"])
    os.environ.pop("SWIMM_HIP_DEBUG")
    print(f"{cfg} x {scale}: {float(w['query_residues']) * w['residues'] / wt / 1e9:.0f} GCUPS, {s.last_stats()['launches']} launches", file=sys.stderr)
