#!/usr/bin/env python3
"""The library's own timeline (SWIMM_HIP_DEBUG) of ONE resident search of the c2 shard through search_topr: when the launches were
issued, when the streams drained, what the promotion ladder and the top-r selection add behind the kernels.
usage: python tools/resident_timeline.py [scale]"""
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

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
shard = bench.build_shard(2, scale)
L, codes, q = shard["lengths"], shard["codes"], shard["query"]
ch = host.Chunks(L, codes, 128, 96 << 20)
with hip_backend.HipSearcher(0) as s:
    s.set_queries(q, np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32), submat.table("blosum62"), 10, 2)
    s.set_option("time_launches", 1)
    for c in ch.chunks:
        s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
    for _ in range(5):
        s.search_topr(20, len(L))
    os.environ["SWIMM_HIP_DEBUG"] = "1"
    t0 = time.perf_counter()
    s.search_topr(20, len(L))
    dt = time.perf_counter() - t0
    os.environ.pop("SWIMM_HIP_DEBUG", None)
    print(f"resident search: {dt * 1e3:.3f} ms through the wrapper", file=sys.stderr)
ch.close()
