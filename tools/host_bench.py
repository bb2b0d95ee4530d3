#!/usr/bin/env python3
"""Times the host-side rows either side of the GPU path against the reference's own
functions (oracle/_ref, when built) on one synthetic FASTA, and checks the outputs are byte-identical.

usage: python tools/host_bench.py [n_sequences, default 300000]
Runs on the CPU only; the reference legs are skipped when oracle/_ref is absent (GPU box)."""
import filecmp
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
res = {"n_sequences": n}
with tempfile.TemporaryDirectory() as tmp:
    L = synth.lengths_lognormal(2, n, 600.0, 0.55, 30, 5000)
    db = synth.make_db(2, L, planted=[])
    fa = os.path.join(tmp, "db.fa")
    synth.write_fasta(fa, synth.db_records(db))
    res["fasta_MB"] = round(os.path.getsize(fa) / 1e6, 1)

    t0 = time.time(); host.preprocess_db(fa, os.path.join(tmp, "mine")); res["preprocess_s"] = round(time.time() - t0, 3)
    t0 = time.time(); d = host.db_load(os.path.join(tmp, "mine")); res["db_load_s"] = round(time.time() - t0, 3)
    t0 = time.time(); ch = host.Chunks(d["lengths"], d["codes"], 128, 96 << 20); res["assemble_chunks_s"] = round(time.time() - t0, 3)
    res["chunks"] = len(ch.chunks)
    idx = np.arange(d["count"] - 20, d["count"], dtype=np.int64)
    t0 = time.time(); host.db_titles(os.path.join(tmp, "mine"), d["count"], idx); res["titles_top20_s"] = round(time.time() - t0, 3)

    try:
        from oracle import ref
        have_ref = ref.available()
    except Exception:
        have_ref = False
    if have_ref:
        t0 = time.time(); ref.preprocess_db(fa, os.path.join(tmp, "ref"), threads=8); res["ref_preprocess_s"] = round(time.time() - t0, 3)
        same = all(filecmp.cmp(os.path.join(tmp, "mine" + e), os.path.join(tmp, "ref" + e), shallow=False) for e in (".seq", ".info"))
        # .desc: the reference leaves one uninitialised byte after some titles (tests/test_host_formats.py); compare line starts
        with open(os.path.join(tmp, "mine.desc"), "rb") as a, open(os.path.join(tmp, "ref.desc"), "rb") as b:
            la, lb = a.read().split(b"\n"), b.read().split(b"\n")
        same = same and len(la) == len(lb) and all(y.startswith(x) and len(y) - len(x) <= 1 for x, y in zip(la, lb))
        res["identical_to_reference"] = bool(same)
        t0 = time.time(); r = ref.assemble_multiple_chunks(os.path.join(tmp, "ref"), 16, 96 << 20, threads=8); res["ref_load_and_assemble_s"] = round(time.time() - t0, 3)
        del r
print(json.dumps(res))
