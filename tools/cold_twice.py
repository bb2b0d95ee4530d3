#!/usr/bin/env python3
"""What bench.py's cold measurement does -- resident upload and searches first, then clear + lazy upload + search, twice -- with the
library's debug timeline for both cold searches: what the first one pays that the second does not.
usage: python tools/cold_twice.py [chunks|slabs] 2> timeline.txt"""
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

shard = bench.build_shard(2, 1.0)
L, codes, q = shard["lengths"], shard["codes"], shard["query"]
sm = submat.table("blosum62")
m, disp = np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32)
offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
slab = 1 << 17


path = sys.argv[1] if len(sys.argv) > 1 else "chunks"          # bench.py hands c2 over in the reference's chunk layout
ch = host.Chunks(L, codes, 128, 96 << 20) if path == "chunks" else None


def upload(s):
    if ch is not None:
        for c in ch.chunks:
            s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
        return
    for first in range(0, len(L), slab):          # .seq slabs, as the `swimm` program hands a database over
        e = min(len(L), first + slab)
        s.add_sequences(L[first:e], codes[offs[first]:offs[e]], first)


with hip_backend.HipSearcher(0) as s:
    s.set_queries(q, m, disp, sm, 10, 2)
    s.set_option("time_launches", 1)
    upload(s)
    for _ in range(5):
        s.search_topr(20, len(L))
    os.environ["SWIMM_HIP_DEBUG"] = "1"
    for rep in range(3):
        s.clear_db()
        s.set_option("lazy_upload", 1)
        print(f"==== cold {rep}", file=sys.stderr, flush=True)
        if os.environ.get("COLD_IDLE_S"):          # an idle GPU first (what the bench's first cold search finds after its CPU check)
            time.sleep(float(os.environ["COLD_IDLE_S"]))
        t0 = time.perf_counter()
        upload(s)
        s.search_topr(20, len(L))
        dt = time.perf_counter() - t0
        print(f"cold {rep}: {dt * 1e3:.2f} ms", flush=True)
if ch is not None:
    ch.close()
