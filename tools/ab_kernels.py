#!/usr/bin/env python3
"""A/B of kernel configurations on the c2 workload, interleaved rounds in ONE process on ONE device
(cdna_hip_programming.md rule 24).  usage: python tools/ab_kernels.py [--scale 1.0] [--rounds 7] "name:key=val,key=val" ..."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swimm_amd import hip_backend, host, submat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("variants", nargs="*", default=["t32:", "t24:rows_per_wave=24", "t16:rows_per_wave=16"])
    args = ap.parse_args()
    shard = bench.build_shard(2, args.scale)
    chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
    q = shard["query"]
    sm = submat.table("blosum62")
    searchers = []
    for v in args.variants:
        name, _, opts = v.partition(":")
        s = hip_backend.HipSearcher(0)
        for kv in filter(None, opts.split(",")):
            k, val = kv.split("=")
            s.set_option(k, int(val))
        s.set_queries(q, np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32), sm, 10, 2)
        for ch in chunks.chunks:
            s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
        s.search_topr(20, shard["n"])
        searchers.append((name, s))
    times = {n: [] for n, _ in searchers}
    ref = None
    for _ in range(args.rounds):
        for name, s in searchers:
            ts, ti, wt = s.search_topr(20, shard["n"])
            if ref is None:
                ref = (ts.copy(), ti.copy())
            assert np.array_equal(ts, ref[0]) and np.array_equal(ti, ref[1]), name
            times[name].append(s.last_stats()["kernel_ms"])
    cells = len(q) * shard["residues"]
    for name, _ in searchers:
        t = np.array(times[name])
        print(json.dumps({"variant": name, "kernel_ms_median": round(float(np.median(t)), 3), "kernel_ms_min": round(float(t.min()), 3),
                          "gcups_median": round(cells / np.median(t) / 1e6, 1)}), flush=True)
    for _, s in searchers:
        s.close()
    chunks.close()


if __name__ == "__main__":
    main()
