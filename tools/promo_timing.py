import sys, json, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from swimm_amd import hip_backend, host, submat
gdir = "/root/repo/tests/golden"
g = json.load(open(os.path.join(gdir, "golden.json")))
import tempfile
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "db")
host.preprocess_db(os.path.join(gdir, g["db_fasta"]), prefix)
db = host.db_load(prefix)
q = host.queries_load(os.path.join(gdir, g["query_fasta"]), False)
chunks = host.Chunks(db["lengths"], db["codes"], 128, 30000)
sm = submat.table("blosum62")
for opts in ({}, {"tail_mode": 2}, {"tail_mode": 1}, {"force_i32": 1}):
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items(): s.set_option(k, v)
        for ch in chunks.chunks: s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
        for qi in range(4):
            s.set_queries(q["a"][q["disp"][qi]:q["disp"][qi+1]], q["m"][qi:qi+1], np.array([0, q["m"][qi]], np.uint32), sm, 10, 2)
            s.search(chunks.vc*128)
            sc, wt = s.search(chunks.vc*128)
            print(opts, "m=%d" % q["m"][qi], "wt %.2f ms" % (wt*1e3), s.last_stats())
