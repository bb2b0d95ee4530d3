#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SWIMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c5] [--scale S]

--workload c2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): one 375-residue query
    (P07327-shaped, synthetic) against 1 000 004 synthetic proteins (~6e8 residues, log-normal lengths), BLOSUM62,
    gap 10/2, top-20.  With N > 1 every rank holds its own 1M-sequence shard (seed differs per rank) of an
    N-million-sequence database: WEAK scaling.
--workload c5 (BASELINE.json configs[4], north_star's multi-GPU case): the 20-query set against ONE Env-NR-shaped
    database (35.5 M sequences / 7e9 residues at --scale 1; default 0.25 so that generating it stays within minutes),
    PAM250, cut into 8 N slabs of equal padded size that are dealt statically to the N ranks
    (sharding.assign_chunks): STRONG scaling, the total work does not depend on N.

A "step" is one complete search of the resident database shard: DP kernels, promotion re-runs, device top-20; with
N > 1 the ranks' top-20 lists are all-gathered (RCCL; 20 x 16 bytes per query and rank) and merged on the host --
the path has no other exchange step.  The database is resident in HBM before the timed region (288 GB holds every
configuration); the first search after a cold upload is reported beside it as `value_incl_h2d`.

N > 1: one process per GPU.  Under torch.distributed.run the environment carries RANK / LOCAL_RANK / WORLD_SIZE;
invoked plainly (`python bench.py --gpus N`) this process starts the N ranks itself, before it touches the GPU, and
relays rank 0's line.

Prints ONE JSON line on rank 0: metric GCUPS = Q_real * D / t / 1e9 (swimm.c:163), `roofline` (dominant kernel,
HIP-event timed inside the library on the streams the kernels run on), `valu_roofline` (the ceiling that binds this
kernel, against both the guide's SIMD issue peak and the measured rate of its instruction class) and, at N = 1,
`cpu_baseline` (the reference's own AVX2 path from oracle/_ref timed on the host cores).  The oracle is only ever
the thing compared against, never the thing measured as `value`; every rank checks a sample of its own scores
against it and a mismatch is a non-zero exit.
"""
import argparse
import glob
import datetime
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from swimm_amd import hip_backend, host, sharding, submat, synth, workloads  # noqa: E402

METRIC = "GCUPS (whole node) + bit-exact top-r scores vs CPUsearch.c"   # BASELINE.json's metric, verbatim
QUERY_INDEX = 3          # P07327, 375 aa, in synth.QUERY_SET
TOP_R = 20
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
# VALU ceilings in wave64 instructions per second (256 CUs x 4 SIMDs x 2.4 GHz):
#   the guide's issue peak -- a SIMD issues a wave64 VALU instruction over 2 cycles (157.3 TFLOPS fp32 vector);
#   the instruction class of this kernel -- every VOP3P packed / 3-source op takes 4 cycles on gfx950, and measures
#   4.40 with 4 waves per SIMD in isolation (tools/microbench/valu_rate, profiles/r01_valu_issue_rates.txt)
VALU_PEAK_SIMD_ISSUE = 256 * 4 * 2.4e9 / 2 / 1e9
VALU_PEAK_CLASS = 256 * 4 * 2.4e9 / 4 / 1e9
VALU_MEASURED_CLASS = 256 * 4 * 2.4e9 / 4.40 / 1e9
INSTR_PER_ROW, INSTR_PER_COLUMN = 8.5, 10          # model when no PMC profile matches: 7.5 packed-f16 ops + 1 v_perm per packed row; per-column overhead


def build_shard(seed: int, scale: float):
    """sorted lengths + recoded residues of one c2 shard, with planted homologs of the query"""
    t0 = time.time()
    base_len = synth.lengths_lognormal(seed, max(256, int(1_000_000 * scale)), 600.0, 0.55, 30, 5000)
    queries = synth.make_queries(2)
    q_title, q_letters = queries[QUERY_INDEX]
    planted = synth.planted_homologs(seed, [(q_title, q_letters)])
    lens = np.concatenate([base_len, np.array([len(s) for _, s in planted], dtype=np.int64)])
    order = np.argsort(lens, kind="stable")
    lens_sorted = lens[order].astype(np.uint16)
    total = int(lens.sum())
    codes = np.empty(total, dtype=np.int8)
    # background residues are i.i.d., so they can be generated directly in sorted order;
    # planted sequences are dropped into their sorted slots afterwards
    blk = 1 << 26
    for s in range(0, total, blk):
        e = min(total, s + blk)
        codes[s:e] = host.recode(synth.residues(seed, 7, s, e - s))
    offs = np.concatenate([[0], np.cumsum(lens_sorted.astype(np.int64))])
    pos_of = np.empty(len(lens), dtype=np.int64)
    pos_of[order] = np.arange(len(lens))
    for k, (_, seq) in enumerate(planted):
        p = pos_of[len(base_len) + k]
        codes[offs[p]:offs[p] + len(seq)] = host.recode(seq)
    qa = host.recode(q_letters)
    return {"lengths": lens_sorted, "codes": codes, "residues": total, "n": len(lens_sorted),
            "query": qa, "gen_s": time.time() - t0}


def host_cpus():
    """(hardware threads this process may run on, physical cores they belong to)"""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            cores.add(("?", str(c)))
    return len(cpus), len(cores)


def oracle_scores(qa_list, lens, codes, sm, threads):
    """CPU checker on a set of sequences: the reference's AVX2 path (oracle/_ref: cpu_search_avx2_sp,
    CPUsearch.c:482-967) when it was built, else the C restatement.  -> (scores [q, n], seconds, kind)"""
    from oracle import port, ref
    real = np.array([len(q) for q in qa_list], dtype=np.int64)
    mp = real + (real % 2)                                    # even padding with the dummy residue, sequences.c:382
    dp = np.concatenate([[0], np.cumsum(mp)]).astype(np.uint32)
    a = np.full(int(mp.sum()), 23, dtype=np.int8)
    for k, q in enumerate(qa_list):
        a[dp[k]:dp[k] + real[k]] = q
    one = host.assemble_single_chunk(lens, codes, 32, 60)
    if ref.available():
        sc, wt = ref.cpu_search(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["nbbs"], one["disp"], sm, 10, 2, 32, threads=threads)
        kind = "reference"
    else:
        t0 = time.time()
        sc = port.search_exact(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["disp"], sm, 10, 2, 32, threads=threads)
        wt = time.time() - t0
        kind = "port"
    return sc[:, :len(lens)], wt, kind


def sample_check(qa_list, lens, codes_of, sm, threads, gpu_scores_of, budget_s, n_total):
    """every k-th sequence of a shard through the CPU checker, k chosen so that the CPU work fits ~budget_s.
    codes_of(i) -> residues of local sequence i; gpu_scores_of(idx) -> [q, len(idx)] GPU scores of those sequences.
    -> (ok, stride, sampled sequences, sampled residues, cpu seconds, kind)"""
    q_res = sum(len(q) for q in qa_list)
    residues = float(np.asarray(lens, dtype=np.int64).sum())
    est = q_res * residues / (1.0e9 * max(threads, 1))       # the reference runs at about 1 GCUPS per hardware thread
    stride = max(1, int(np.ceil(est / budget_s)))
    idx = np.arange(0, n_total, stride, dtype=np.int64)
    sub_lens = np.asarray(lens)[idx]
    sub_codes = np.concatenate([codes_of(int(i)) for i in idx]) if stride > 1 else codes_of(None)
    sc, wt, kind = oracle_scores(qa_list, sub_lens, sub_codes, sm, threads)
    ok = bool(np.array_equal(gpu_scores_of(idx), sc))
    return ok, stride, len(idx), int(sub_lens.astype(np.int64).sum()), wt, kind


def pmc_profile(workload, scale, plan, kernel, launches):
    """PMC summary of exactly this configuration + launch plan, if one is committed under profiles/ (rocprofv3 --pmc
    passes, tools/profile_bench.sh + tools/summarize_profile.py); None otherwise -- a number from another
    configuration is not this run's traffic."""
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if (d.get("workload_key") == workload and abs(float(d.get("scale", -1)) - scale) < 1e-9 and d.get("plan") == plan and d.get("kernel") == kernel
                and d.get("kernel_launches_per_search") == launches):
            d["file"] = os.path.relpath(fn, ROOT)
            best = d
    return best


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has not
    touched the GPU and never will), relay rank 0's output, fail if any rank fails."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 3300
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0:
                rc = rc or r
                for o in procs:          # one rank failed: the others would wait in a collective forever
                    o.terminate()
        if time.time() > deadline:
            for o in procs:
                o.kill()
            return 124
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("c2", "c5"), default="c2")
    ap.add_argument("--scale", type=float, default=None, help="fraction of the configuration's database (default: c2 1.0, c5 0.25)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the first-search-after-a-cold-upload measurement (profiling runs: only the resident search's launches in the trace)")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--max-waves", type=int, default=0)
    ap.add_argument("--wgs-per-cu", type=int, default=0)
    args = ap.parse_args()
    if args.scale is None:
        args.scale = 1.0 if args.workload == "c2" else 0.25

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    # stdout carries the ONE JSON line and nothing else: whatever the libraries print there (gloo's connection notes, ...)
    # goes to stderr from here on
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    share = os.environ.get("SWIMM_BENCH_SHARE_DEVICE") == "1"   # rehearsal only: every rank on GPU 0, gloo only
    visible = torch.cuda.device_count()
    if not share and world > 1 and visible < world:
        # fewer visible devices than ranks (e.g. a launcher that gives every rank its own HIP_VISIBLE_DEVICES): take what is
        # there; when ranks end up on one device the line says so
        share = visible <= 1 and os.environ.get("HIP_VISIBLE_DEVICES") is None and os.environ.get("ROCR_VISIBLE_DEVICES") is None
    dev_index = 0 if share else local_rank % max(visible, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    rccl = None
    rccl_ranks = 0
    if world > 1:
        import torch.distributed as dist
        # control plane (barrier, max-over-ranks) on gloo; the only data exchange of the path -- the ranks' top-20
        # lists, once per step -- goes over RCCL
        dist.init_process_group(backend="gloo")
        if not share:
            try:
                # (a rank whose RCCL set-up fails leaves the others inside this collective: they give up after a minute and
                # everybody meets again in the gloo all_reduce below)
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
                rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=60))
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, group=rccl)
                torch.cuda.synchronize()
                rccl_ranks = int(t.item())
                assert rccl_ranks == world
            except Exception as e:   # result path only: fall back to gloo, say so in the output
                print(f"[rank {rank}] RCCL group unavailable ({e}); top-r lists go over gloo", file=sys.stderr)
                rccl = None
            ok = torch.tensor([1 if rccl is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank takes the same result path
            if int(ok.item()) == 0:
                rccl = None
                rccl_ranks = 0

    threads_all, cores_all = host_cpus()
    my_threads = max(1, threads_all // world)
    searcher = hip_backend.HipSearcher(dev_index)
    for k, v in (("rows_per_wave", args.rows_per_wave), ("max_waves", args.max_waves), ("wgs_per_cu", args.wgs_per_cu)):
        if v:
            searcher.set_option(k, v)
    searcher.set_option("time_launches", 1)      # HIP events around every pipeline launch, on the stream it runs on

    # ---- the rank's shard, resident in HBM ------------------------------------------------------------------
    t_gen = time.time()
    chunks = None
    if args.workload == "c2":
        sm = submat.table("blosum62")
        shard = build_shard(2 + 1000 * rank, args.scale)
        chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
        qa_list = [shard["query"]]
        a, m, disp = shard["query"], np.array([len(shard["query"])], np.uint16), np.array([0, len(shard["query"])], np.uint32)
        my_lens, my_n, n_valid, score_stride = shard["lengths"], shard["n"], shard["n"], chunks.vc * 128
        my_residues, my_padded = shard["residues"], chunks.vD
        offs = np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))])
        local_index = np.arange(my_n, dtype=np.int64)            # position of the rank's sequences in its score rows
        codes_of = lambda i: shard["codes"] if i is None else shard["codes"][offs[i]:offs[i + 1]]   # noqa: E731
        index_base = rank * (1 << 40)                            # ranks hold disjoint databases: make the merged indices distinct
        workload_txt = "c2: 375-aa query x 1M synthetic proteins per GPU, BLOSUM62 g10 e2, top-20"

        def upload():
            for ch in chunks.chunks:
                searcher.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
    else:
        db = workloads.SortedDb("c5", args.scale)
        sm = submat.table(db.matrix)
        slabs = db.slabs(8 * world)
        owner = sharding.assign_chunks([s[2] for s in slabs], world)
        mine = [s for s, o in zip(slabs, owner) if o == rank]
        a, m, disp = db.a, db.m, db.disp
        qa_list = [a[disp[k]:disp[k + 1]] for k in range(len(m))]
        slab_codes = [db.codes(s0, s1) for s0, s1, _ in mine]
        local_index = np.concatenate([np.arange(s0, s1, dtype=np.int64) for s0, s1, _ in mine])
        my_lens = db.lengths[local_index]
        my_n, n_valid, score_stride = len(local_index), db.n, (db.n + 127) // 128 * 128
        my_residues, my_padded = int(my_lens.astype(np.int64).sum()), sum(s[2] for s in mine)
        codes_of = lambda i: np.concatenate(slab_codes) if i is None else db.codes(int(local_index[i]), int(local_index[i]) + 1)   # noqa: E731
        index_base = 0                                           # one database: indices are global already
        workload_txt = (f"c5: 20-query set (144-5478 aa) x ONE Env-NR-shaped database at scale {args.scale} ({db.n} sequences, "
                        f"{db.residues} residues), PAM250 g10 e2, top-20; {len(slabs)} slabs dealt statically to {world} rank(s)")

        def upload():
            for (s0, s1, _), codes in zip(mine, slab_codes):
                searcher.add_sequences(db.lengths[s0:s1], codes, first_seq=s0)
    t_gen = time.time() - t_gen
    searcher.set_queries(a, m, disp, sm, 10, 2)
    t_up = time.time()
    upload()
    t_up = time.time() - t_up

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    nq = len(m)

    def exchange(ts, ti):
        """the path's only exchange step: every rank's top-20 per query -> merged listing (host merge, utils.c order)"""
        if dist is None:
            return ts, ti
        mine_t = torch.from_numpy(np.concatenate([ts.astype(np.int64).ravel(), np.where(ti >= 0, ti + index_base, -1).ravel()]))
        if rccl is not None:
            mine_t = mine_t.cuda()
            allv = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(allv, mine_t, group=rccl)
        else:
            allv = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(allv, mine_t)
        g = torch.stack(allv).cpu().numpy()
        ms = np.zeros((nq, TOP_R), np.int32); mi = np.zeros((nq, TOP_R), np.int64)
        for k in range(nq):
            ms[k], mi[k] = host.topr_merge(g[:, k * TOP_R:(k + 1) * TOP_R].astype(np.int32), g[:, (nq + k) * TOP_R:(nq + k + 1) * TOP_R], TOP_R)
        return ms, mi

    def one_step():
        ts, ti, wt = searcher.search_topr(TOP_R, n_valid)
        ms, mi = exchange(ts, ti)
        return ms, mi, wt

    for _ in range(args.warmup):
        one_step()
    kernel_ms, wts, launch_ms = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        top_s, top_i, wt = one_step()
        kernel_ms.append(searcher.last_stats()["kernel_ms"])
        launch_ms.append(searcher.last_launch_ms())
        wts.append(wt)
    barrier()
    elapsed = time.perf_counter() - t0
    total_residues = float(my_residues)
    k_ms_mean = float(np.mean(kernel_ms))
    k_ms_max = k_ms_mean
    if dist is not None:
        tmax = torch.tensor([elapsed, k_ms_mean], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, k_ms_max = float(tmax[0].item()), float(tmax[1].item())
        res = torch.tensor([my_residues], dtype=torch.float64)
        dist.all_reduce(res, op=dist.ReduceOp.SUM)
        total_residues = float(res.item())
    stats = searcher.last_stats()
    plans = [searcher.last_plan(k) for k in range(nq)]
    kernel_name = searcher.last_kernel_name(nq - 1)          # the longest query's kernel dominates the device time

    # ---- parity: every rank checks a sample of its own scores, and the merged listing, against the CPU checker ----
    q_real = int(m.astype(np.int64).sum())
    full = np.zeros((nq, score_stride), dtype=np.int32)
    searcher.search(score_stride, out=full)
    mine_scores = full[:, local_index]
    budget = 25.0 if (world == 1 and not args.no_cpu_baseline) else 6.0
    ok, stride, n_s, res_s, cpu_s, kind = sample_check(qa_list, my_lens, codes_of, sm, my_threads, lambda idx: mine_scores[:, idx], budget, my_n)
    # the merged top-20 of the timed path (device top-r + exchange) against a host selection over the ranks' full vectors
    hs = np.zeros((nq, TOP_R), np.int32); hi = np.zeros((nq, TOP_R), np.int64)
    for k in range(nq):
        s_, i_ = host.topr(mine_scores[k], TOP_R)
        hs[k], hi[k] = s_, np.where(i_ >= 0, local_index[np.maximum(i_, 0)], -1)
    ms2, mi2 = exchange(hs, hi)
    ok_top = bool(np.array_equal(ms2, top_s) and np.array_equal(mi2, top_i))
    flags = torch.tensor([1 if ok else 0, 1 if ok_top else 0], dtype=torch.int32)
    if dist is not None:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    all_ok, all_top_ok = bool(flags[0].item()), bool(flags[1].item())

    # ---- first search after a cold upload (the reference's workTime brackets the transfers, MICsearch.c:51,350) ----
    cold_s = float("nan")
    if not args.no_cold:
        searcher.clear_db()
        searcher.set_option("lazy_upload", 1)        # chunks stream in while the search runs (what swimm_hip_search_chunks does)
        barrier()
        t0 = time.perf_counter()
        upload()
        exchange(*searcher.search_topr(TOP_R, n_valid)[:2])
        barrier()
        cold_s = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([cold_s], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        cold_s = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gcups = q_real * total_residues * args.steps / elapsed / 1e9
        launches = max(1, stats["launches"])
        # Algorithmic HBM bytes of one search of this rank's shard (w = 2, T_eff = rows per pass): per pass
        # the tiled database bytes are read once (1/m B per cell of the whole query); between two passes the strip
        # boundary (H and F, 2 B each, per sequence and column = 4x the tiled residue bytes) is written once and read
        # once; 8 B per sequence and query for the scores.  For c2 (one query, `passes` launches of one kernel) divided
        # by the launches this is the per-launch figure of DESIGN.md; for a query batch it is the sum over the queries.
        alg_bytes = sum(float(my_padded) * (p["passes"] + 8.0 * (p["passes"] - 1)) + 8.0 * my_n for p in plans)
        # The dominant kernel's launches: the library brackets every pipeline launch with HIP events on the stream it is
        # launched on and reports their sum and number (two launches that share the chip on two streams each count with
        # their own, longer, duration -- exactly what a kernel trace lists per dispatch).
        pipe_launches = max(1, int(round(np.mean([n for _, n in launch_ms]))))
        per_launch_ms = max(float(np.mean([ms_ / max(n, 1) for ms_, n in launch_ms])), 1e-6)   # (a shard so small that every group runs on the lane-systolic kernel has no pipeline launch)
        single_kernel = nq == 1                                    # one query = one kernel instantiation: per-launch figures are meaningful
        achieved = (alg_bytes / pipe_launches) / (per_launch_ms * 1e-3) / 1e9 if single_kernel else alg_bytes / (k_ms_mean * 1e-3) / 1e9
        plan_key = {"rows_per_wave": plans[-1]["rows_per_wave"], "waves": plans[-1]["waves"], "passes": plans[-1]["passes"]}
        prof = pmc_profile(args.workload, args.scale, plan_key, kernel_name, pipe_launches) if (world == 1 and single_kernel) else None
        traffic = prof["hbm_bytes_per_launch"] if prof else None
        cells_real = q_real * float(my_residues)
        kernel_gcups = cells_real / (k_ms_mean * 1e-3) / 1e9
        cells_padded = float(stats["cells"])
        if prof and prof.get("sq_insts_valu_per_launch"):
            n_instr = float(prof["sq_insts_valu_per_launch"]) * pipe_launches
            instr_src = f"SQ_INSTS_VALU of {prof['file']}"
        else:
            # wave-columns of query p = padded columns x waves x passes; each costs T x 8.5 + 10 instructions
            n_instr = sum(float(my_padded) / 128 * p["waves"] * p["passes"] * (p["rows_per_wave"] * INSTR_PER_ROW + INSTR_PER_COLUMN) for p in plans)
            instr_src = "model: 8.5 VALU instructions per packed row + 10 per column (no PMC profile of this configuration and plan under profiles/)"
        ginstr = n_instr / (k_ms_mean * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if args.workload == "c2" else "strong",
            "vs_baseline": None, "dtype": "f16 (exact integers < 2048) -> int16 -> int32", "data": "synthetic",
            "config": {"workload": workload_txt, "query_residues": q_real, "queries": nq,
                       "db_sequences_rank0": my_n, "db_residues_rank0": my_residues, "db_residues_total": int(total_residues),
                       "parallelism": f"db-shard x{world}", "plan": plans[-1], "scale": args.scale,
                       "topr_exchange": "none" if world == 1 else ("rccl all_gather" if rccl is not None else "gloo all_gather"),
                       "rccl_ranks": rccl_ranks},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": (prof["file"] if prof else "none: no PMC profile of this workload, scale and launch plan is committed (profiles/*_pmc_traffic*.json)"),
                         "kernel": kernel_name, "kernel_ms": round(per_launch_ms, 4), "kernel_launches_per_search": pipe_launches,
                         "device_ms_per_search": round(k_ms_mean, 4), "launches_per_search": launches,
                         "achieved_chip_GBps": round(alg_bytes / (k_ms_mean * 1e-3) / 1e9, 3),
                         "alg_bytes_per_search": alg_bytes, "alg_bytes_per_launch": alg_bytes / pipe_launches if single_kernel else None,
                         "note": ("VALU-bound kernel (see valu_roofline); HBM carries the database residues once per pass and the strip boundary between passes; "
                                  "achieved = alg_bytes_per_launch / kernel_ms (kernel_ms = average duration of the kernel's launches by HIP events on their own streams; "
                                  "a query of three or more passes runs two launches side by side on two streams, each at about half the chip: achieved_chip_GBps is bytes per search / device time)"
                                  if single_kernel else "query batch: several kernel instantiations; achieved = algorithmic bytes per search / device time of the search")},
            "valu_roofline": {"achieved": round(ginstr, 1), "unit": "G wave64-instr/s", "instructions": instr_src,
                              "peak": round(VALU_PEAK_SIMD_ISSUE, 1), "frac": round(ginstr / VALU_PEAK_SIMD_ISSUE, 4),
                              "peak_source": "MI355X_MICROARCH.md: 4 SIMDs per CU, one wave64 VALU instruction issued over 2 cycles",
                              "class_peak": round(VALU_PEAK_CLASS, 1), "class_frac": round(ginstr / VALU_PEAK_CLASS, 4),
                              "class_measured": round(VALU_MEASURED_CLASS, 1), "class_measured_frac": round(ginstr / VALU_MEASURED_CLASS, 4),
                              "class_source": "VOP3P packed / 3-source ops issue over 4 cycles on gfx950; 4.40 measured in isolation with 4 waves per SIMD (profiles/r01_valu_issue_rates.txt)",
                              "kernel_only_gcups": round(kernel_gcups, 2), "padded_cells": cells_padded},
            "search_call_ms": round(float(np.mean(wts)) * 1e3, 4),
            "value_incl_h2d": None if args.no_cold else round(q_real * total_residues / cold_s / 1e9, 2),
            "value_incl_h2d_note": f"first search after a cold upload of the shard (pageable host memory, {'reference chunk layout' if args.workload == 'c2' else '.seq slabs'}; chunk k+1 copied and tiled while chunk k is aligned), {cold_s * 1e3:.1f} ms",
            "h2d_upload_s": round(t_up, 3), "datagen_s": round(t_gen, 2),
            "top1": [int(top_s[0][0]), int(top_i[0][0])],
            "bit_exact_vs_reference": all_ok, "merged_top20_matches_full_vectors": all_top_ok,
            "parity_sample": f"every rank: every {stride}th sequence of its shard x all queries vs the CPU {kind} ({n_s} sequences, {res_s} residues on rank 0)",
        }
        if share:
            out["shared_device"] = True
            out["note"] = "REHEARSAL: all ranks share GPU 0 (SWIMM_BENCH_SHARE_DEVICE=1); the value is not a multi-GPU figure"
        if world > 1:
            out["per_rank_kernel_gcups_min"] = round(q_real * float(my_residues) / (k_ms_max * 1e-3) / 1e9, 2)
        if world == 1 and not args.no_cpu_baseline:
            try:
                model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                model = "unknown"
            out["cpu_baseline"] = {"value": round(q_real * res_s / cpu_s / 1e9, 2), "unit": "GCUPS", "cores": cores_all, "threads": my_threads,
                                   "kind": kind, "cpu_model": model, "matches_gpu": ok,
                                   "sample": f"{args.workload} shard, every {stride}th sequence ({n_s} sequences, {res_s} residues), {nq} quer{'y' if nq == 1 else 'ies'}, {cpu_s:.2f} s"}
        print(json.dumps(out), file=json_out, flush=True)
    searcher.close()
    if chunks is not None:
        chunks.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not (all_ok and all_top_ok):
        raise SystemExit(f"[rank {rank}] GPU scores differ from the CPU checker (sample ok: {all_ok}, merged top-20 ok: {all_top_ok})")


if __name__ == "__main__":
    main()
