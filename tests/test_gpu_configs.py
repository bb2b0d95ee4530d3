"""BASELINE.json configs c3, c4 and c5 on the GPU against the CPU oracle, at shapes that stress what they stress.

  c3  20 queries (144-5478 aa) x the Swiss-Prot-shaped database AT FULL SIZE (540 080 sequences, 1.96e8 residues, the
      35 000-residue tail present), BLOSUM50: bulk pipeline + lane-systolic tail on a second stream + chained lane
      passes + the binary16 -> int16 -> int32 promotion ladder of 20 queries in flight on three streams.
  c4  the 5478-aa query x 1 000 000 Env-NR-shaped sequences, BLOSUM62: 25 passes through HBM boundary rows.
  c5  20 queries x an Env-NR-shaped database, PAM250, statically sharded over 8 (virtual) GPUs: every workgroup
      aligns more than 32 groups back to back (the 32-entry ring of item ids in LDS wraps), through group-resident passes
      on the even devices and through one launch per pass with the boundary buffer cut into >= 2 runs on the odd ones,
      and the per-device top-20 lists are merged on the host.

The checker is the reference's own AVX2 path (oracle/_ref/libswimm_ref.so = cpu_search_avx2_sp, CPUsearch.c:482-967,
compiled by oracle/Makefile) when it is present, else the C restatement oracle/sw_oracle.c; the WHOLE score matrix is
compared when the host's cores can produce it in ~40 s, else every k-th sequence (helpers.oracle_matrix).
Top-r order: utils.c:71-86 (score descending, larger sorted index first).
"""
import os

import numpy as np
import pytest

from helpers import matrix, oracle_matrix
from oracle import port
from swimm_amd import hip_backend, host, sharding, workloads

pytestmark = pytest.mark.gpu


def _load(s, chunks):
    for ch in chunks:
        s.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])


def _check_matrix(got, want, idx, label):
    sub = got[:, idx]
    if not np.array_equal(sub, want):
        bad = np.argwhere(sub != want)
        q, k = bad[0]
        raise AssertionError(f"{label}: {len(bad)} of {want.size} scores differ; first: query {q}, sequence {idx[k]}: "
                             f"GPU {sub[q, k]} vs oracle {want[q, k]}")


def test_c3_swissprot_shape_full_size():
    w = workloads.build("c3", 1.0)
    assert w["n"] == 540_080 and int(w["lengths"][-1]) > 30_000 and len(w["m"]) == 20
    chunks = host.Chunks(w["lengths"], w["codes"], 128, 96 << 20)
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(w["a"], w["m"], w["disp"], matrix(w["matrix"]), 10, 2)
        _load(s, chunks.chunks)
        got, _ = s.search(chunks.vc * 128)
        st = s.last_stats()
        ts, ti, _ = s.search_topr(20, w["n"])
        # the same database streamed in while it is searched (uploader thread, the end with the 35 000-residue sequence
        # first, "long" judged against the whole database, a launch shape per range beside the tail kernels): .seq slabs
        s.clear_db()
        s.set_option("lazy_upload", 1)
        offs = np.concatenate([[0], np.cumsum(w["lengths"].astype(np.int64))])
        cuts = [0] + [int(x) // 128 * 128 for x in np.linspace(0, w["n"], 7)[1:-1]] + [w["n"]]
        for s0, s1 in zip(cuts[:-1], cuts[1:]):
            s.add_sequences(w["lengths"][s0:s1], w["codes"][offs[s0]:offs[s1]], first_seq=s0)
        streamed, _ = s.search((w["n"] + 127) // 128 * 128)
    chunks.close()
    assert np.array_equal(streamed[:, :w["n"]], got[:, :w["n"]])
    want, idx = oracle_matrix(w)
    _check_matrix(got[:, :w["n"]], want, idx, "c3")
    assert st["promoted"] >= 1                      # the planted copies of the long queries leave the int16 range (5478 x 5+)
    for q in range(len(w["m"])):                    # device top-20 == the reference's listing order over the GPU's own vector
        os_, oi = port.topr(got[q, :w["n"]], 20)
        assert np.array_equal(ts[q], os_) and np.array_equal(ti[q], oi), q


def test_c4_long_query_envnr_shape():
    w = workloads.build("c4", 0, n_sequences=1_000_000)
    assert int(w["m"][0]) == 5478 and w["n"] >= 1_000_000
    chunks = host.Chunks(w["lengths"], w["codes"], 128, 96 << 20)
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(w["a"], w["m"], w["disp"], matrix(w["matrix"]), 10, 2)
        _load(s, chunks.chunks)
        got, _ = s.search(chunks.vc * 128)         # one query: one launch per pass (group-resident batches need two queries or the option)
        plan = s.last_plan(0)
        launches = s.last_stats()["launches"]
        # the same recurrence cut differently: group-resident passes (one launch, every workgroup takes a group through all
        # its passes back to back); 4-wave workgroups; the boundary buffer of the per-pass launches in 3+ runs
        s.set_option("resident", 1)
        res, _ = s.search(chunks.vc * 128)
        name = s.last_kernel_name(0)
        assert s.last_stats()["launches"] < 8 <= plan["passes"] <= launches
        s.set_option("waves", 4)
        other, _ = s.search(chunks.vc * 128)
        s.set_option("resident", 0)
        s.set_option("bnd_mib", 300)
        third, _ = s.search(chunks.vc * 128)
        assert s.last_stats()["launches"] >= 3 * s.last_plan(0)["passes"]
    chunks.close()
    assert plan["passes"] >= 12 and name.endswith("true, true, false>(swimm::PipeParams)"), name
    assert np.array_equal(got, res) and np.array_equal(got, other) and np.array_equal(got, third)
    want, idx = oracle_matrix(w)
    _check_matrix(got[:, :w["n"]], want, idx, "c4")


def test_c4_envnr_quarter_of_the_target_database():
    """north_star's target configuration (BASELINE config 4: the 5 478-residue query x the Env-NR-shaped database, BLOSUM62
    10/2) at a QUARTER of its 7e9 residues -- 8.9 M sequences / 1.75e9 residues, 69 000 device groups -- the first search
    streamed in from .seq slabs while it runs, the second on the resident copy: both must give the same 8.9 M scores, the
    planted copies of the query their known answers, and a block sample of the database (every k-th block of 64 consecutive
    sequences, k from the host's cores) the reference's own scores."""
    db = workloads.SortedDb("c4", 0.25)
    assert int(db.m[0]) == 5478 and db.residues > 0.25 * 6.9e9 and db.n > 8_800_000
    sm = matrix(db.matrix)
    slabs = db.slabs(8)
    stride_slots = (db.n + 127) // 128 * 128
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(db.a, db.m, db.disp, sm, 10, 2)
        s.set_option("lazy_upload", 1)
        keep = []
        for s0, s1, _ in slabs:
            keep.append(db.codes(s0, s1))
            s.add_sequences(db.lengths[s0:s1], keep[-1], first_seq=s0)
        cold, _ = s.search(stride_slots)
        cold_launches = s.last_stats()["launches"]
        warm, _ = s.search(stride_slots)
        ts, ti, _ = s.search_topr(20, db.n)
        plan = s.last_plan(0)
    assert np.array_equal(cold, warm), "the streamed first search and the resident search differ"
    got = warm[0, :db.n]
    assert plan["passes"] >= 12 and cold_launches >= 2
    # known answers: the exact copy of the query scores its self score and leads the listing, ahead of the mutated copies
    self_score = port.pair_score(db.a, db.a, sm, 10, 2)
    assert int(ts[0][0]) == self_score == int(got.max()) and ts[0][0] > ts[0][1] > ts[0][2] > ts[0][3]
    os_, oi = port.topr(got, 20)
    assert np.array_equal(ts[0], os_) and np.array_equal(ti[0], oi)
    # block sample against the reference: ~20 s of the host's cores at about 1 GCUPS per hardware thread
    threads = len(os.sched_getaffinity(0))
    k = max(1, int(np.ceil(5478.0 * db.residues / (1.0e9 * threads * 20.0))))
    starts = np.arange(0, db.n, 64 * k, dtype=np.int64)
    idx = np.concatenate([np.arange(b, min(b + 64, db.n), dtype=np.int64) for b in starts])
    sub = {"lengths": db.lengths[idx], "codes": np.concatenate([db.codes(int(b), int(min(b + 64, db.n))) for b in starts]), "n": len(idx),
           "residues": int(db.lengths[idx].astype(np.int64).sum()), "a": db.a, "m": db.m, "disp": db.disp, "query_residues": 5478, "matrix": db.matrix}
    sub["offs"] = np.concatenate([[0], np.cumsum(sub["lengths"].astype(np.int64))])
    want, widx = oracle_matrix(sub, budget_s=1e9)
    assert len(widx) == len(idx) and sub["residues"] > 1e6
    _check_matrix(got[None, idx], want, np.arange(len(idx)), f"c4 at a quarter of 7e9 residues (every {k}th block of 64)")


def test_c5_envnr_pam250_eight_way_shard(monkeypatch):
    G = 8
    w = workloads.build("c5", 0, n_sequences=320_000)
    sm = matrix(w["matrix"])
    chunks = host.Chunks(w["lengths"], w["codes"], 128, 2 << 20)
    assert len(chunks.chunks) >= 3 * G
    monkeypatch.setenv("SWIMM_HIP_VIRTUAL_GPUS", str(G))      # test hook: device d runs on physical device d % real count
    assert hip_backend.device_count() >= G
    owner = sharding.assign_chunks([c["vD"] for c in chunks.chunks], G)
    q = {"a": w["a"], "m": w["m"], "disp": w["disp"]}
    got = np.full((len(w["m"]), chunks.vc * 128), -7, dtype=np.int32)
    got_default = np.full((len(w["m"]), chunks.vc * 128), -7, dtype=np.int32)
    lists_s, lists_i = [], []
    for d in range(G):
        mine = [c for i, c in enumerate(chunks.chunks) if owner[i] == d]
        groups = sum(c["count"] for c in mine)
        cols = sum(int(c["n"].astype(np.int64).sum()) for c in mine)
        with hip_backend.HipSearcher(d) as s:
            # four persistent workgroups: each aligns groups / 4 > 32 items back to back; no lane-systolic tail (as at
            # full size, where no Env-NR group is long beside a CU's load); boundary buffer: two runs
            s.set_option("wg_limit", 4)
            s.set_option("tail_mode", 2)
            if d % 2:    # odd devices: one launch per pass, the boundary rows through HBM in two runs; even: group-resident passes
                s.set_option("resident", 0)
                s.set_option("bnd_mib", max(1, int(cols * 512 * 0.55) >> 20))
            else:
                s.set_option("resident", 1)
            s.set_queries(q["a"], q["m"], q["disp"], sm, 10, 2)
            _load(s, mine)
            assert groups >= 4 * 40, groups
            s.search(chunks.vc * 128, out=got)
            st = s.last_stats()
            plans = [s.last_plan(k) for k in range(len(w["m"]))]
            multi = [p["passes"] for p in plans if p["passes"] > 1]
            if d % 2:
                assert multi and st["launches"] >= sum(p["passes"] for p in plans) + sum(multi), (st, plans)   # >= 2 boundary runs per multi-pass query
            else:
                assert multi and 1 <= st["launches"] < len(plans) + 20, (st, plans)                            # one launch per launch shape + promotion re-runs
            s.set_option("wg_limit", 64)                       # (the listing itself does not need the slow four-workgroup shape again)
            ts, ti, _ = s.search_topr(20, w["n"])
        lists_s.append(ts)
        lists_i.append(ti)
        # ... and the same shard with the planner's own choices (no workgroup cap, its own tail rule, batch or per-pass
        # launches as it sees fit): what a real 8-GPU run of this database executes on device d
        with hip_backend.HipSearcher(d) as s:
            s.set_queries(q["a"], q["m"], q["disp"], sm, 10, 2)
            _load(s, mine)
            s.search(chunks.vc * 128, out=got_default)
    assert not (got[:, :w["n"]] == -7).any()                   # every sequence was resident on exactly one device
    assert np.array_equal(got_default[:, :w["n"]], got[:, :w["n"]])
    want, idx = oracle_matrix(w)
    _check_matrix(got[:, :w["n"]], want, idx, "c5 (8-way shard)")
    for k in range(len(w["m"])):                               # host merge of the 8 per-device lists == the reference's listing
        ms, mi = host.topr_merge(np.stack([l[k] for l in lists_s]), np.stack([l[k] for l in lists_i]), 20)
        os_, oi = port.topr(got[k, :w["n"]], 20)
        assert np.array_equal(ms, os_) and np.array_equal(mi, oi), k
    # the whole-call drop-in (argument list of mic_search_knc_ap_multiple_chunks, MICsearch.h:35-38) over the same 8 devices
    monkeypatch.setenv("SWIMM_HIP_OPTIONS", "wg_limit=64,tail_mode=2,bnd_mib=4")
    allsc, _ = hip_backend.search_chunks(w["a"], w["m"], w["disp"], chunks.vc, chunks.chunks, sm, 10, 2, G, 128)
    assert np.array_equal(allsc[:, :w["n"]], got[:, :w["n"]])
    chunks.close()


def test_c5_ring_wrap_at_natural_occupancy():
    """4.3 M Env-NR-shaped sequences = 33 000+ groups: with the launch shapes the planner really picks (up to 1 024 persistent
    workgroups) every workgroup takes more than 32 groups.  No CPU pass over 3.7e13 cells: the dynamic queue must agree
    with the static partition (no ring at all) on every score, and with the oracle on sampled pairs + every best hit."""
    w = workloads.build("c5", 0.12)
    sm = matrix(w["matrix"])
    chunks = host.Chunks(w["lengths"], w["codes"], 128, 96 << 20)
    assert chunks.vc >= 33_000
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(w["a"], w["m"], w["disp"], sm, 10, 2)
        _load(s, chunks.chunks)
        dyn, _ = s.search(chunks.vc * 128)
        ts, ti, _ = s.search_topr(20, w["n"])
        s.set_option("dynamic", 0)
        sta, _ = s.search(chunks.vc * 128)
    chunks.close()
    assert np.array_equal(dyn, sta)
    rng = np.random.default_rng(5)
    picks = [(int(rng.integers(len(w["m"]))), int(rng.integers(w["n"]))) for _ in range(150)]
    picks += [(k, int(ti[k, 0])) for k in range(len(w["m"]))]
    for k, i in picks:
        qa = w["a"][w["disp"][k]:w["disp"][k + 1]]
        assert dyn[k, i] == port.pair_score(qa, w["codes"][w["offs"][i]:w["offs"][i + 1]], sm, 10, 2), (k, i)
    for k in range(len(w["m"])):
        os_, oi = port.topr(dyn[k, :w["n"]], 20)
        assert np.array_equal(ts[k], os_) and np.array_equal(ti[k], oi), k


def test_query_batch_group_resident_launch():
    """40 queries of 300-1 000 residues against a 170 000-sequence database (c2-shaped, 1.0e8 residues, with its long-sequence
    tail): by default ONE group-resident launch whose items are (group, query) pairs -- every group through the pipeline kernel,
    queries of 3 to 8 passes side by side -- against the score matrix of the reference, and against one launch per query and pass."""
    import bench
    from swimm_amd import synth
    shard = bench.build_shard(7, 0.17)
    rng = np.random.default_rng(11)
    ms = np.sort(rng.integers(300, 1000, 40)).astype(np.uint16)
    qa = [host.recode(synth.residues(7, 2000 + k, 0, int(m))) for k, m in enumerate(ms)]
    a = np.concatenate(qa)
    disp = np.concatenate([[0], np.cumsum(ms.astype(np.int64))]).astype(np.uint32)
    w = {"lengths": shard["lengths"], "codes": shard["codes"], "offs": np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))]),
         "a": a, "m": ms, "disp": disp, "matrix": "blosum62", "residues": shard["residues"], "n": shard["n"],
         "query_residues": int(ms.astype(np.int64).sum())}
    sm = matrix("blosum62")
    with hip_backend.HipSearcher(0) as s:
        s.set_queries(a, ms, disp, sm, 10, 2)
        s.add_sequences(shard["lengths"], shard["codes"], 0)
        stride = (shard["n"] + 127) // 128 * 128
        got, _ = s.search(stride)
        st, name = s.last_stats(), s.last_kernel_name(len(ms) - 1)
        ts, ti, _ = s.search_topr(20, shard["n"])
        s.set_option("resident", 0)
        other, _ = s.search(stride)
        st0 = s.last_stats()
    assert name.endswith("true, true, false>(swimm::PipeParams)") and st["launches"] < 20 < st0["launches"], (name, st, st0)
    assert np.array_equal(got, other)
    want, idx = oracle_matrix(w)
    _check_matrix(got[:, :shard["n"]], want, idx, "query batch")
    for k in range(len(ms)):
        os_, oi = port.topr(got[k, :shard["n"]], 20)
        assert np.array_equal(ts[k], os_) and np.array_equal(ti[k], oi), k
