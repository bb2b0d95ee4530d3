#!/usr/bin/env python3
"""Database upload: reference chunk layout (host interleave + retile kernel) vs the direct path (raw .seq content,
tiled on the device).  usage: python tools/upload_bench.py [scale of the c2 shard, default 1.0]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import hip_backend, host, submat, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
L = np.sort(synth.config_lengths("c2", scale)).astype(np.uint16)
total = int(L.astype(np.int64).sum())
codes = host.recode(synth.residues(2, 7, 0, total))
q = host.recode(synth.residues(2, 11, 0, 375))
sm = submat.table("blosum62")
res = {"sequences": len(L), "residues": total}
with hip_backend.HipSearcher(0) as s:
    s.set_queries(q, np.array([375], np.uint16), np.array([0, 375], np.uint32), sm, 10, 2)
    t0 = time.time(); ch = host.Chunks(L, codes, 128, 96 << 20); res["host_interleave_s"] = round(time.time() - t0, 3)
    t0 = time.time()
    for c in ch.chunks:
        s.add_chunk(c["b"], c["n"], c["disp"], 128, c["first_group"])
    res["add_chunk_s"] = round(time.time() - t0, 3)
    a, _ = s.search(ch.vc * 128)
    s.clear_db()
    t0 = time.time()
    first = 0
    slab = 1 << 17                                        # 131 072 sequences per slab
    offs = np.concatenate([[0], np.cumsum(L.astype(np.int64))])
    while first < len(L):
        e = min(len(L), first + slab)
        s.add_sequences(L[first:e], codes[offs[first]:offs[e]], first)
        first = e
    res["add_sequences_s"] = round(time.time() - t0, 3)
    b, _ = s.search(ch.vc * 128)
    res["scores_identical"] = bool(np.array_equal(a[:, :len(L)], b[:, :len(L)]))
    ch.close()
print(json.dumps(res))
