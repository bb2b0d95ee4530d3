#!/usr/bin/env python3
"""SURVEY 8(f3) + the CLI at scale: `swimm -S preprocess` then `swimm -S search -m 1` on an Env-NR-shaped synthetic
database (BASELINE config 5: 35.5 M sequences / 7e9 residues at scale 1, the 20 queries, PAM250 10/2, top 20), with the
wall-clock split of the search process -- load (.seq read + checks), search (the reference's workTime scope: upload +
kernels + top-r), titles (.desc through the .didx sidecar) -- and the same title step WITHOUT the sidecar (the scan the
sidecar replaces, sequences.c:757-761).  Every reported (query, hit) score is recomputed by the CPU checker.

usage: python tools/cli_scale.py [scale, default 1.0] [out.json]
Needs one MI355X; the FASTA (1.33 B per residue) and the three database files go to $SWIMM_SCALE_TMP or a temp directory."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from swimm_amd import host, synth  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
SWIMM = os.path.join(ROOT, "swimm_amd", "bin", "swimm")
res = {"config": "c5 (Env-NR shape, 20 queries, PAM250 10/2, top 20)", "scale": scale}
tmpdir = os.environ.get("SWIMM_SCALE_TMP") or tempfile.mkdtemp(prefix="swimm_cli_scale_")
os.makedirs(tmpdir, exist_ok=True)
fa, qfa, prefix = os.path.join(tmpdir, "db.fa"), os.path.join(tmpdir, "q.fa"), os.path.join(tmpdir, "db")

t0 = time.time()
L = synth.config_lengths("c5", scale)
offs = np.concatenate([[0], np.cumsum(L)])
with open(fa, "wb", buffering=1 << 24) as f:
    blk = 200_000
    for s0 in range(0, len(L), blk):
        s1 = min(len(L), s0 + blk)
        body = synth.residues(5, 7, int(offs[s0]), int(offs[s1] - offs[s0])).tobytes()      # letters, counter-based generator
        o = offs[s0:s1 + 1] - offs[s0]
        for i in range(s1 - s0):
            f.write(b">ENV%09d synthetic environmental sequence %d len=%d\n" % (s0 + i, s0 + i, L[s0 + i]))
            b = body[o[i]:o[i + 1]]
            f.write(b"\n".join(b[k:k + 70] for k in range(0, len(b), 70)) + b"\n")
        if s0 % (20 * blk) == 0:
            print(f"fasta: {s0} of {len(L)} sequences, {time.time() - t0:.0f} s", flush=True)
queries = synth.make_queries(5)
synth.write_fasta(qfa, queries)
res.update(sequences=int(len(L)), residues=int(offs[-1]), fasta_bytes=os.path.getsize(fa), generate_s=round(time.time() - t0, 1))
print(json.dumps(res), flush=True)

t0 = time.time()
p = subprocess.run([SWIMM, "-S", "preprocess", "-i", fa, "-o", prefix], capture_output=True, text=True)
res["preprocess_wall_s"] = round(time.time() - t0, 2)
assert p.returncode == 0, p.stdout[-400:] + p.stderr[-400:]
os.remove(fa)
res["files"] = {e: os.path.getsize(prefix + e) for e in (".seq", ".desc", ".info", ".didx")}
print(json.dumps(res), flush=True)


def search(tag):
    t0 = time.time()
    p = subprocess.run([SWIMM, "-S", "search", "-q", qfa, "-d", prefix, "-m", os.environ.get("SWIMM_CLI_MODE", "1"), "-c", str(os.cpu_count()), "-s", "pam250", "-g", "10", "-e", "2", "-r", "20"],
                       capture_output=True, text=True, env=dict(os.environ, SWIMM_DEBUG="1"))
    wall = time.time() - t0
    assert p.returncode == 0, p.stdout[-600:] + p.stderr[-600:]
    m = re.search(r"swimm: load ([\d.]+) s, search ([\d.]+) s, titles ([\d.]+) s \((\d+) titles\)", p.stderr)
    assert m, p.stderr[-600:]
    res[tag] = {"process_wall_s": round(wall, 2), "load_s": float(m.group(1)), "search_s": float(m.group(2)), "titles_s": float(m.group(3)),
                "titles": int(m.group(4)), "gcups_reported": float(re.search(r"Search speed:\s+([\d.]+) GCUPS", p.stdout).group(1))}
    print(tag, json.dumps(res[tag]), flush=True)
    return p.stdout


out = search("search_with_didx")
# the report against the CPU checker: every listed hit's score recomputed from the .seq file (sorted order; the title names the
# original index, the lengths are unique enough to find the record by title -> length -> residues of the generator)
lens = np.fromfile(prefix + ".seq", dtype=np.uint16, count=len(L))
codes = np.memmap(prefix + ".seq", dtype=np.int8, mode="r", offset=2 * len(L))
sorted_off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
order = np.argsort(L, kind="stable")          # the preprocess sorts stably by length (sequences.c:850-865)
rank_of = np.empty(len(L), dtype=np.int64)
rank_of[order] = np.arange(len(L), dtype=np.int64)
sm = host.submat("pam250")
blocks = out.split("Query no.")[1:]
assert len(blocks) == len(queries)
checked = 0
from oracle import port  # noqa: E402  (checker: test infrastructure, never the thing measured)
for qi, blk_txt in enumerate(blocks):
    qa = host.recode(queries[qi][1])
    for line in blk_txt.split("Score\tSequence description\n")[1].strip().split("\n")[:20]:
        mm = re.match(r"(\d+)\t>?ENV(\d+) ", line)
        if not mm:
            break
        score, orig = int(mm.group(1)), int(mm.group(2))
        r = int(rank_of[orig])
        seq = np.asarray(codes[sorted_off[r]:sorted_off[r + 1]])
        assert port.pair_score(qa, seq, sm, 10, 2) == score, (qi, orig, score)
        checked += 1
res["reported_scores_recomputed"] = checked
os.remove(prefix + ".didx")
search("search_without_didx")
for e in (".seq", ".desc", ".info"):
    os.remove(prefix + e)
os.remove(qfa)
line = json.dumps(res)
print(line)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(line + "\n")
