"""BASELINE.json's full-size headline workload (c2: 375-aa query x 1 000 004 synthetic proteins,
6e8 residues) on the GPU, checked through size-independent properties plus a sampled comparison with
the CPU oracle (a full oracle pass would take minutes on the test box):

  * decomposition invariance -- the score vector does not depend on how the DP is cut into strips /
    waves / passes / kernels (rows_per_wave 16, single-wave workgroups, every group through the lane-systolic kernel,
    packed-int16 first tier, static partition instead of the dynamic queue, 4-wave workgroups with the boundary buffer cut
    into a dozen runs): seven different schedules of the same recurrence must agree bit for bit on all 1 000 004 scores;
  * known answers -- the planted exact copy scores the query's self score, mutated copies score less,
    in order of mutation rate;
  * sampled oracle -- 400 randomly chosen sequences + the top-20 against sw_oracle_pair;
  * top-r -- device top-20 == host selection over the full vector == reference order (score, index desc).
"""
import numpy as np
import pytest

import bench
from oracle import port
from swimm_amd import hip_backend, host, submat

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    shard = bench.build_shard(2, 1.0)
    chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
    yield shard, chunks
    chunks.close()


def _search(shard, chunks, opts, topr=False):
    q = shard["query"]
    with hip_backend.HipSearcher(0) as s:
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_queries(q, np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32), submat.table("blosum62"), 10, 2)
        for ch in chunks.chunks:
            s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
        full, _ = s.search(chunks.vc * 128)
        top = s.search_topr(20, shard["n"]) if topr else None
    return full[0, :shard["n"]].copy(), top


def test_c2_full_size(c2):
    shard, chunks = c2
    assert shard["n"] == 1_000_004 and 5.9e8 < shard["residues"] < 6.1e8
    base, top = _search(shard, chunks, {}, topr=True)
    for opts in ({"rows_per_wave": 16}, {"max_waves": 1}, {"tail_mode": 1}, {"f16": 0}, {"dynamic": 0},
                 {"bnd_mib": 64, "rows_per_wave": 16, "waves": 4}):
        other, _ = _search(shard, chunks, opts)
        assert np.array_equal(base, other), opts
    # the first search after a cold upload -- ONE launch walks the whole item list while the chunks are still travelling
    # (PipeParams::avail) -- must give the same 1 000 004 scores as the resident database, through chunks and through slabs,
    # also with a handful of workgroups that wait for every part
    for opts in ({"lazy_upload": 1}, {"lazy_upload": 1, "wg_limit": 40}, {"lazy_upload": 1, "upload_piece_kib": 8192}):
        other, _ = _search(shard, chunks, opts)
        assert np.array_equal(base, other), opts
    q = shard["query"]
    offs = np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))])
    with hip_backend.HipSearcher(0) as s:
        s.set_option("lazy_upload", 1)
        s.set_queries(q, np.array([len(q)], np.uint16), np.array([0, len(q)], np.uint32), submat.table("blosum62"), 10, 2)
        for first in range(0, shard["n"], 1 << 17):
            e = min(shard["n"], first + (1 << 17))
            s.add_sequences(shard["lengths"][first:e], shard["codes"][offs[first]:offs[e]], first)
        streamed, _ = s.search((shard["n"] + 127) // 128 * 128)
        assert s.last_stats()["launches"] <= 5          # the ONE pipeline launch + the promotion ladder's re-runs of the planted copies
    assert np.array_equal(base, streamed[0, :shard["n"]])
    sm = submat.table("blosum62")
    self_score = port.pair_score(q, q, sm, 10, 2)
    order = np.argsort(-base.astype(np.int64), kind="stable")
    assert base[order[0]] == self_score                       # planted 0 % copy
    assert base[order[0]] > base[order[1]] > base[order[2]] > base[order[3]] > base[order[4]]   # 10/30/50 % copies, then background
    rng = np.random.default_rng(7)
    picks = list(rng.integers(0, shard["n"], 400)) + [0, shard["n"] - 1] + list(order[:20])
    for i in picks:
        assert base[i] == port.pair_score(q, shard["codes"][offs[i]:offs[i + 1]], sm, 10, 2), i
    ts, ti, _ = top
    hs, hi = host.topr(base, 20)
    assert np.array_equal(ts[0], hs) and np.array_equal(ti[0], hi)
    os_, oi = port.topr(base, 20)
    assert np.array_equal(hs, os_) and np.array_equal(hi, oi)


def test_bench_line_contract():
    """bench.py at a small scale prints ONE JSON line with every field of the contract (and checks itself against
    the CPU reference / oracle on the way)"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--scale", "0.06", "--secondary-scale", "0.02"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "GCUPS" and d["value"] > 100 and "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["bound"] in ("hbm", "mfma") and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["matches_gpu"] is True
    assert d["roofline"]["frac"] <= 1.0 and d["bit_exact_vs_reference"] is True
    # the other BASELINE configurations ride on the same line, each with its own parity sample, roofline and CPU baseline
    assert [r["workload"] for r in d["secondary"]] == ["c3", "c4", "c5"]
    for r in d["secondary"]:
        assert r["bit_exact_vs_reference"] is True and r["merged_top20_matches_full_vectors"] is True and r["value"] > 100
        assert r["cpu_baseline"]["matches_gpu"] is True and r["cpu_baseline"]["kind"] in ("reference", "port")
        assert r["roofline"] is None and "roofline_note" in r or (0 < r["roofline"]["frac"] <= 1.0 and "sw_" in r["roofline"]["kernel"])
        assert r["value_incl_h2d"] > 10 and r["value_incl_h2d_pooled"] > 10
    assert [r["config"]["queries"] for r in d["secondary"]] == [20, 1, 20]
    assert d["cpu_baseline"]["product_m0"] > 0
    # the first cold search of the process under value_incl_h2d, the pooled one beside it; where the rank's threads sit
    assert d["value_incl_h2d"] > 10 and d["value_incl_h2d_pooled"] > 10 and len(d["value_incl_h2d_ms"]) == 2
    assert d["placement"][0]["rank"] == 0 and d["placement"][0]["pci"]
    # ... and the line ENDS with the short summary of every record
    assert list(d.keys())[-1] == "summary" and [x["workload"] for x in d["summary"]] == ["c2", "c3", "c4", "c5"]
    assert all(x["bit_exact_vs_reference"] is True for x in d["summary"]) and len(json.dumps(d["summary"])) < 2500
