#!/usr/bin/env python3
"""Measures the pipeline kernel's throughput for every (rows per wave T, waves per workgroup W) launch shape on
one c2-shaped shard: query length m = T*W (one pass, no padding rows), so the figure is the shape's own
efficiency.  The launch-plan model in plan.cpp (choose_plan) is calibrated against this table.

    python tools/plan_sweep.py [--scale 0.3] [--ts 8,12,...] [--ws 4,8,12,16]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, host, submat, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.3)
ap.add_argument("--ts", default="8,12,16,20,24,28,32,36")
ap.add_argument("--ws", default="2,4,6,8,10,12,14,16")
ap.add_argument("--lengths", default="c2")
args = ap.parse_args()

L = np.sort(synth.config_lengths(args.lengths, args.scale)).astype(np.uint16)
total = int(L.astype(np.int64).sum())
codes = host.recode(synth.residues(2, 7, 0, total))
chunks = host.Chunks(L, codes, 128, 96 << 20)
sm = submat.table("blosum62")
qfull = host.recode(synth.residues(2, 11, 0, 36 * 16))
print(f"# {len(L)} sequences, {total} residues", flush=True)
ref_top = {}
with hip_backend.HipSearcher(0) as s:
    for c in chunks.chunks:
        s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
    for T in [int(x) for x in args.ts.split(",")]:
        for W in [int(x) for x in args.ws.split(",")]:
            if T > 28 and W > 12:
                continue
            m = T * W
            s.set_option("rows_per_wave", T)
            s.set_option("waves", W)
            s.set_queries(qfull[:m], np.array([m], np.uint16), np.array([0, m], np.uint32), sm, 10, 2)
            s.search_topr(20, len(L))
            best = None
            for _ in range(2):
                ts, ti, wt = s.search_topr(20, len(L))
                ms = s.last_stats()["kernel_ms"]
                best = ms if best is None else min(best, ms)
            plan = s.last_plan(0)
            assert plan["rows_per_wave"] == T and plan["waves"] == W and plan["passes"] == 1, plan
            key = (ts[0].tobytes(), ti[0].tobytes())
            if m in ref_top:
                assert ref_top[m] == key, f"top-20 differs between launch shapes at m={m}"
            ref_top[m] = key
            print(json.dumps({"T": T, "W": W, "m": m, "kernel_ms": round(best, 3), "gcups": round(m * total / best / 1e6, 1)}), flush=True)
chunks.close()
