#!/usr/bin/env python3
"""Cold searches in the SECOND and THIRD context of a process (bench.py runs one context per configuration): c2 through chunks, then
a BASELINE configuration through slabs (default c5 at 10 %); prints add + search ms and the resident ms beside them.
usage: python tools/exp_ctx.py [c3|c4|c5] [scale]     (EXP_DEBUG=1: the library's timeline of the last cold search of the slab contexts)"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
from swimm_amd import hip_backend, host, submat, workloads

shard = bench.build_shard(2, 1.0)
chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
q = shard["query"]; sm = submat.table("blosum62")
m, disp = np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32)
NAME = sys.argv[1] if len(sys.argv) > 1 else "c5"
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
db = workloads.SortedDb(NAME, SCALE)
slabs = db.slabs(8); codes = [db.codes(s0, s1) for s0, s1, _ in slabs]
sm5 = submat.table(db.matrix)


def c2_ctx(tag):
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(q, m, disp, sm, 10, 2)
        for ch in chunks.chunks: s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
        t = time.time(); s.search_topr(20, shard["n"]); t = time.time(); s.search_topr(20, shard["n"]); res = time.time() - t
        out = []
        for rep in range(3):
            s.clear_db(); s.set_option("lazy_upload", 1)
            t = time.time()
            for ch in chunks.chunks: s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
            s.search_topr(20, shard["n"]); out.append((time.time() - t) * 1e3)
        print(f"{tag}: c2 resident {res * 1e3:.2f} ms, cold {[round(x, 2) for x in out]}", file=sys.stderr)


def c5_ctx(tag):
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(db.a, db.m, db.disp, sm5, 10, 2)
        for (s0, s1, _), c in zip(slabs, codes): s.add_sequences(db.lengths[s0:s1], c, first_seq=s0)
        s.search_topr(20, db.n); t = time.time(); s.search_topr(20, db.n); res = time.time() - t
        out = []
        for rep in range(2):
            s.clear_db(); s.set_option("lazy_upload", 1)
            if rep == 1 and os.environ.get("EXP_DEBUG"): os.environ["SWIMM_HIP_DEBUG"] = "1"
            t = time.time()
            for (s0, s1, _), c in zip(slabs, codes): s.add_sequences(db.lengths[s0:s1], c, first_seq=s0)
            s.search_topr(20, db.n); out.append((time.time() - t) * 1e3)
            os.environ.pop("SWIMM_HIP_DEBUG", None)
        print(f"{tag}: {NAME} at {SCALE} resident {res * 1e3:.1f} ms, cold {[round(x, 1) for x in out]}", file=sys.stderr)


c2_ctx("context 1"); c5_ctx("context 2"); c2_ctx("context 3"); c5_ctx("context 4")
chunks.close()
