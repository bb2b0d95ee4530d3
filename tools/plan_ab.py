#!/usr/bin/env python3
"""One BASELINE configuration, resident, under several forced launch shapes beside the planner's own choice: ms per search and
GCUPS (3 searches each, the best).  usage: python tools/plan_ab.py <c2|c3|c4|c5> <scale> "T,W;T,W;..."   (0,0 = the planner)"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swimm_amd import hip_backend, host, submat, workloads  # noqa: E402

if os.environ.get("SWIMM_HIP_LIB"):      # (A/B against an older build of the library: it may lack the newest entry points)
    hip_backend.ABI_SYMBOLS = tuple(n for n in hip_backend.ABI_SYMBOLS if n not in ("swimm_hip_device_pci_bus_id", "swimm_hip_bind_host_thread"))

name, scale = sys.argv[1], float(sys.argv[2])
shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[3].split(";")]
if name == "c2":
    shard = bench.build_shard(2, scale)
    chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
    q = shard["query"]
    a, m, disp, sm, n, res = q, np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32), submat.table("blosum62"), shard["n"], shard["residues"]

    def upload(s):
        for ch in chunks.chunks:
            s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
else:
    db = workloads.SortedDb(name, scale)
    slabs = db.slabs(8)
    codes = [db.codes(s0, s1) for s0, s1, _ in slabs]
    a, m, disp, sm, n, res = db.a, db.m, db.disp, submat.table(db.matrix), db.n, db.residues

    def upload(s):
        for (s0, s1, _), c in zip(slabs, codes):
            s.add_sequences(db.lengths[s0:s1], c, first_seq=s0)
cells = float(m.astype(np.int64).sum()) * res
with hip_backend.HipSearcher(0) as s:
    s.set_queries(a, m, disp, sm, 10, 2)
    upload(s)
    for T, W in shapes:
        s.set_option("rows_per_wave", T)
        s.set_option("waves", W)
        if (T, W) == (0, 0):
            os.environ["SWIMM_HIP_DEBUG_PLAN"] = "1"
        best = 1e9
        for rep in range(3):
            t = time.time()
            s.search_topr(20, n)
            best = min(best, time.time() - t)
            os.environ.pop("SWIMM_HIP_DEBUG_PLAN", None)
        p = s.last_plan(len(m) - 1)
        print(f"{name} scale {scale}: forced {T} x {W} -> longest query runs {p['waves']} x {p['rows_per_wave']} rows, {p['passes']} passes: {best * 1e3:.2f} ms, {cells / best / 1e9:.0f} GCUPS, device {s.last_stats()['kernel_ms']:.2f} ms", file=sys.stderr)
