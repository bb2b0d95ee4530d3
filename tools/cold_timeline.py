#!/usr/bin/env python3
"""Where does the first search after a cold upload spend its time?  One BASELINE configuration (c3 / c4 / c5, .seq slabs as
bench.py uploads them): two resident searches, then two cold ones (clear_db, lazy_upload, add_sequences, search_topr) --
the second with SWIMM_HIP_DEBUG=1, so stderr carries the library's own timeline (ranges, copies, launches, device time).
usage: python tools/cold_timeline.py <c3|c4|c5> <scale> [library options k=v,...]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, submat, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
opts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
t0 = time.time()
db = workloads.SortedDb(name, scale)
slabs = db.slabs(8)
codes = [db.codes(s0, s1) for s0, s1, _ in slabs]
print(f"{name} at scale {scale}: {db.n} sequences, {db.residues} residues, {len(db.m)} queries ({db.query_residues} rows); generated in {time.time() - t0:.1f} s", file=sys.stderr)
sm = submat.table(db.matrix)
cells = float(db.query_residues) * db.residues


def upload(s):
    for (s0, s1, _), c in zip(slabs, codes):
        s.add_sequences(db.lengths[s0:s1], c, first_seq=s0)


with hip_backend.HipSearcher(0) as s:
    s.set_queries(db.a, db.m, db.disp, sm, 10, 2)
    for k, v in opts.items():
        s.set_option(k, int(v))
    upload(s)
    for rep in range(2):
        if rep == 1 and os.environ.get("SWIMM_TL_DEBUG_RESIDENT"):
            os.environ["SWIMM_HIP_DEBUG"] = "1"
        t = time.time()
        s.search_topr(20, db.n)
        dt = time.time() - t
        os.environ.pop("SWIMM_HIP_DEBUG", None)
        print(f"resident search {rep}: {dt * 1e3:.2f} ms -> {cells / dt / 1e9:.0f} GCUPS; device {s.last_stats()['kernel_ms']:.2f} ms, {s.last_stats()['launches']} launches", file=sys.stderr)
    for rep in range(2):
        s.clear_db()
        s.set_option("lazy_upload", 1)
        if rep == 1 or os.environ.get("SWIMM_TL_DEBUG_ALL"):
            os.environ["SWIMM_HIP_DEBUG"] = "1"
        t = time.time()
        upload(s)
        s.search_topr(20, db.n)
        dt = time.time() - t
        os.environ.pop("SWIMM_HIP_DEBUG", None)
        print(f"cold search {rep}: add + search {dt * 1e3:.2f} ms -> {cells / dt / 1e9:.0f} GCUPS incl. upload; device {s.last_stats()['kernel_ms']:.2f} ms, {s.last_stats()['launches']} launches", file=sys.stderr)
