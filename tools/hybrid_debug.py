#!/usr/bin/env python3
"""mode 2 (host CPU + GPU) on a synthetic 400 000-sequence database with the split's probe figures on stderr
(SWIMM_DEBUG=1): what hybrid_split measured and what the two legs then took.  usage: python tools/hybrid_debug.py [threads]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import synth  # noqa: E402

SWIMM = os.path.join(ROOT, "swimm_amd", "bin", "swimm")
threads = sys.argv[1] if len(sys.argv) > 1 else str(min(64, len(os.sched_getaffinity(0))))
with tempfile.TemporaryDirectory() as tmp:
    qs = synth.make_queries(11, [375, 729, 1500])
    lens = synth.lengths_lognormal(11, 400_000, 300.0, 0.55, 30, 4000)
    db = synth.make_db(11, lens, planted=synth.planted_homologs(11, qs), with_titles=True)
    fa, qfa, prefix = os.path.join(tmp, "db.fa"), os.path.join(tmp, "q.fa"), os.path.join(tmp, "db")
    synth.write_fasta(fa, synth.db_records(db))
    synth.write_fasta(qfa, qs)
    subprocess.check_call([SWIMM, "-S", "preprocess", "-i", fa, "-o", prefix], stdout=subprocess.DEVNULL)
    for rep in range(2):
        p = subprocess.run([SWIMM, "-S", "search", "-q", qfa, "-d", prefix, "-m", "2", "-c", threads, "-r", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, env=dict(os.environ, SWIMM_DEBUG="1"))
        print(p.stderr.strip())
        print("\n".join(l for l in p.stdout.splitlines() if "share" in l or "Search time" in l or "Kernel time" in l))
    p = subprocess.run([SWIMM, "-S", "search", "-q", qfa, "-d", prefix, "-m", "0", "-c", threads, "-r", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print("\n".join(l for l in p.stdout.splitlines() if "Search time" in l or "Search speed" in l or "Execution" in l))
