#!/usr/bin/env python3
"""A/B of library options on one of the BASELINE configurations, interleaved rounds in ONE process on ONE device.
usage: python tools/ab_configs.py --config c4 --scale 0.05 [--only 19] [--rounds 5] "name:key=val,key=val" ..."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, host, submat, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--scale", type=float, default=0.05)
    ap.add_argument("--only", type=str, default="", help="comma-separated indices into the config's query list")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("variants", nargs="*", default=["resident:", "per_pass:resident=0"])
    args = ap.parse_args()
    w = workloads.build(args.config, args.scale, queries=[int(i) for i in args.only.split(",")] if args.only else None)
    sm = submat.table(w["matrix"])
    print(f"# {args.config} scale {args.scale}: {w['n']} sequences, {w['residues']} residues, {len(w['m'])} queries ({w['query_residues']} aa)", flush=True)
    searchers = []
    for v in args.variants:
        name, _, opts = v.partition(":")
        s = hip_backend.HipSearcher(0)
        for kv in filter(None, opts.split(",")):
            k, val = kv.split("=")
            s.set_option(k, int(val))
        s.set_queries(w["a"], w["m"], w["disp"], sm, 10, 2)
        s.add_sequences(w["lengths"], w["codes"], 0)
        s.search_topr(20, w["n"])
        searchers.append((name, s))
    times = {n: [] for n, _ in searchers}
    wall = {n: [] for n, _ in searchers}
    ref = None
    for _ in range(args.rounds):
        for name, s in searchers:
            ts, ti, wt = s.search_topr(20, w["n"])
            if ref is None:
                ref = (ts.copy(), ti.copy())
            assert np.array_equal(ts, ref[0]) and np.array_equal(ti, ref[1]), name
            times[name].append(s.last_stats()["kernel_ms"])
            wall[name].append(wt * 1e3)
    cells = float(w["query_residues"]) * w["residues"]
    for name, s in searchers:
        t, tw = np.array(times[name]), np.array(wall[name])
        print(json.dumps({"variant": name, "device_ms_median": round(float(np.median(t)), 3), "search_ms_median": round(float(np.median(tw)), 3),
                          "gcups_device": round(cells / np.median(t) / 1e6, 1), "gcups_search": round(cells / np.median(tw) / 1e6, 1),
                          "launches": s.last_stats()["launches"], "plan_longest": s.last_plan(len(w["m"]) - 1)}), flush=True)
        s.close()


if __name__ == "__main__":
    main()
