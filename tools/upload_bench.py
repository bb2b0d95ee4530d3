#!/usr/bin/env python3
"""Database upload paths on one MI355X, c2 shard (1M sequences, 0.6 GB):
  eager  add_chunk (reference chunk layout: H2D + re-tile kernel) / add_sequences (.seq slabs: H2D + tile kernel)
  lazy   the same calls only record the buffers; the first search streams chunk k+1 in while chunk k is aligned
plus the raw H2D rates of pageable and pinned host memory (torch) for reference.
usage: python tools/upload_bench.py [scale of the c2 shard, default 1.0]"""
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()        # torch first: it brings its own HIP runtime, which must be the one the process initialises

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swimm_amd import hip_backend, host, submat  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
shard = bench.build_shard(2, scale)
L, codes, q = shard["lengths"], shard["codes"], shard["query"]
total = shard["residues"]
sm = submat.table("blosum62")
m, disp = np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32)
res = {"sequences": len(L), "residues": total}
t0 = time.time(); ch = host.Chunks(L, codes, 128, 96 << 20); res["host_interleave_s"] = round(time.time() - t0, 3)
offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
slab = 1 << 17


def add_chunks(s):
    for c in ch.chunks:
        s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])


def add_slabs(s):
    first = 0
    while first < len(L):
        e = min(len(L), first + slab)
        s.add_sequences(L[first:e], codes[offs[first]:offs[e]], first)
        first = e


with hip_backend.HipSearcher(0) as s:
    s.set_queries(q, m, disp, sm, 10, 2)
    ref = None
    for name, add in (("chunks", add_chunks), ("slabs", add_slabs)):
        for lazy in (0, 1):
            for rep in range(2):
                s.clear_db()
                s.set_option("lazy_upload", lazy)
                t0 = time.time(); add(s); t1 = time.time()
                ts, ti, wt = s.search_topr(20, len(L)); t2 = time.time()
                st = s.last_stats()
                ts2, ti2, wt2 = s.search_topr(20, len(L))
                key = f"{name}_{'lazy' if lazy else 'eager'}_{'first' if rep == 0 else 'again'}"
                res[key] = {"add_ms": round((t1 - t0) * 1e3, 2), "first_search_ms": round((t2 - t1) * 1e3, 2), "total_ms": round((t2 - t0) * 1e3, 2),
                            "device_ms": round(st["kernel_ms"], 2), "resident_search_ms": round(wt2 * 1e3, 2),
                            "gcups_incl_h2d": round(len(q) * total / (t2 - t0) / 1e9, 1)}
                if ref is None:
                    ref = (ts.copy(), ti.copy())
                assert np.array_equal(ts, ref[0]) and np.array_equal(ti, ref[1]) and np.array_equal(ts2, ref[0])
x = torch.from_numpy(codes[:96 << 20].view(np.uint8).copy())
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.time(); y = x.cuda(); torch.cuda.synchronize(); res[f"torch_pageable_96MB_GBps_{rep}"] = round(x.numel() / (time.time() - t0) / 1e9, 1)
t0 = time.time(); xp = x.pin_memory(); res["torch_pin_96MB_ms"] = round((time.time() - t0) * 1e3, 2)
for rep in range(2):
    t0 = time.time(); y = xp.cuda(non_blocking=True); torch.cuda.synchronize(); res[f"torch_pinned_96MB_GBps_{rep}"] = round(x.numel() / (time.time() - t0) / 1e9, 1)
ch.close()
print(json.dumps(res, indent=1))
