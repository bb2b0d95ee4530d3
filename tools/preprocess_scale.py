#!/usr/bin/env python3
"""SURVEY 8(f2), preprocess at scale: `swimm -S preprocess` on an Env-NR-shaped synthetic FASTA of >= 1e9 residues -- wall
time, peak resident memory of the process (ru_maxrss of the child) against the size of its output, sha256 of the .seq file.
The preprocess walks the mapped file once (swimm_amd/csrc/host/seqio.c): peak memory ~ output size, not 2x the input.

usage: python tools/preprocess_scale.py [residues, default 1.0e9] [out.json]
Runs on the host CPU only (no GPU, no reference)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import synth  # noqa: E402

target = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0e9
SWIMM = os.path.join(ROOT, "swimm_amd", "bin", "swimm")
res = {"target_residues": target}
tmpdir = os.environ.get("SWIMM_SCALE_TMP") or tempfile.mkdtemp(prefix="swimm_scale_")
fa, prefix = os.path.join(tmpdir, "db.fa"), os.path.join(tmpdir, "db")
t0 = time.time()
L = synth.config_lengths("c5", target / 6.99e9)
offs = np.concatenate([[0], np.cumsum(L)])
total = int(offs[-1])
letters = np.frombuffer(b"ARNDCQEGHILKMFPSTWYVBZXU", dtype=np.uint8)
with open(fa, "wb", buffering=1 << 24) as f:
    blk = 200_000                                      # sequences per block: residues drawn block by block
    for s0 in range(0, len(L), blk):
        s1 = min(len(L), s0 + blk)
        codes = synth.residues(5, 7, int(offs[s0]), int(offs[s1] - offs[s0]))      # letters (uint8), counter-based generator
        body = codes.tobytes()
        o = offs[s0:s1 + 1] - offs[s0]
        for i in range(s1 - s0):
            f.write(b">ENV%09d synthetic environmental sequence %d len=%d\n" % (s0 + i, s0 + i, L[s0 + i]))
            b = body[o[i]:o[i + 1]]
            f.write(b"\n".join(b[k:k + 70] for k in range(0, len(b), 70)) + b"\n")
res["fasta_bytes"] = os.path.getsize(fa)
res["sequences"], res["residues"] = int(len(L)), total
res["generate_s"] = round(time.time() - t0, 1)
t0 = time.time()
p = subprocess.run([SWIMM, "-S", "preprocess", "-i", fa, "-o", prefix], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SWIMM_DEBUG="1"))
res["preprocess_s"] = round(time.time() - t0, 2)
res["rc"] = p.returncode
# VmHWM of the swimm process image itself (ru_maxrss of a child also carries what the parent held when it forked)
res["peak_rss_bytes"] = next((int(l.split()[2]) * 1024 for l in p.stderr.splitlines() if l.startswith("swimm: VmHWM:")), None)
res["rss_anon_at_exit_bytes"] = next((int(l.split()[2]) * 1024 for l in p.stderr.splitlines() if l.startswith("swimm: RssAnon:")), None)
if p.returncode == 0:
    res["seq_bytes"], res["desc_bytes"] = os.path.getsize(prefix + ".seq"), os.path.getsize(prefix + ".desc")
    res["peak_rss_over_output"] = round(res["peak_rss_bytes"] / (res["seq_bytes"] + res["desc_bytes"]), 3)
    res["peak_rss_over_input"] = round(res["peak_rss_bytes"] / res["fasta_bytes"], 3)
    h = hashlib.sha256()
    with open(prefix + ".seq", "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    res["seq_sha256"] = h.hexdigest()
    res["info"] = open(prefix + ".info").read()
    # the .seq file against the generator: sorted lengths, and the residues of the first and the last sequence
    lens = np.fromfile(prefix + ".seq", dtype=np.uint16, count=len(L))
    res["lengths_sorted_and_complete"] = bool(np.array_equal(lens, np.sort(L).astype(np.uint16)))
else:
    res["stdout_tail"] = p.stdout[-400:]
for e in (".seq", ".desc", ".info"):
    if os.path.exists(prefix + e):
        os.remove(prefix + e)
os.remove(fa)
line = json.dumps(res)
print(line)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(line + "\n")
