#!/usr/bin/env python3
"""Timeline of ONE first search after a cold upload (option lazy_upload), c2 shard: per range, when the host copies
returned, when the launches were issued, when every pipeline launch started on the device and how long it ran
(SWIMM_HIP_DEBUG lines on stderr).
usage: python tools/upload_timeline.py [chunks|slabs] [scale] [library options k=v,...]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swimm_amd import hip_backend, host, submat  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "slabs"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
opts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {}
shard = bench.build_shard(2, scale)
L, codes, q = shard["lengths"], shard["codes"], shard["query"]
sm = submat.table("blosum62")
m, disp = np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32)
offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
ch = host.Chunks(L, codes, 128, 96 << 20) if path == "chunks" else None
slab = 1 << 17
with hip_backend.HipSearcher(0) as s:
    s.set_queries(q, m, disp, sm, 10, 2)
    s.set_option("time_launches", 1)
    for k, v in opts.items():
        s.set_option(k, int(v))
    for rep in range(3):
        s.clear_db()
        s.set_option("lazy_upload", 1)
        if rep == 2:
            os.environ["SWIMM_HIP_DEBUG"] = "1"
        t0 = time.time()
        if ch is not None:
            for c in ch.chunks:
                s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
        else:
            for first in range(0, len(L), slab):
                e = min(len(L), first + slab)
                s.add_sequences(L[first:e], codes[offs[first]:offs[e]], first)
        ts, ti, wt = s.search_topr(20, len(L))
        t1 = time.time()
        os.environ.pop("SWIMM_HIP_DEBUG", None)
        print(f"rep {rep}: add + first search {1e3 * (t1 - t0):.2f} ms -> {len(q) * shard['residues'] / (t1 - t0) / 1e9:.0f} GCUPS incl. upload; device {s.last_stats()['kernel_ms']:.2f} ms",
              file=sys.stderr)
if ch is not None:
    ch.close()
