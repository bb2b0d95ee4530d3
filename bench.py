#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SWIMM hot path on MI355X.

Workload (BASELINE.json configs[1], "c2"): one 375-residue query (P07327-shaped, synthetic) against
1 000 000 synthetic proteins (~6e8 residues, log-normal lengths), BLOSUM62, gap 10/2, top-20.
A "step" is one complete search of the resident database shard: DP kernels, int32 promotion if any,
top-r.  The database is resident in HBM before the timed region (that is the design: 288 GB holds
every configuration; the PCIe-inclusive figure is in DESIGN.md).

Multi-GPU (--gpus N, launched by torch.distributed.run): WEAK scaling, every rank holds its own
1M-sequence shard (seed differs per rank) of an N-million-sequence database, no data-path collective;
per step the ranks' top-20 lists are all-gathered (RCCL, 20 x 12 bytes) and merged on the host.

Prints ONE JSON line on rank 0 (see the repo's task contract): metric GCUPS = Q_real * D / t / 1e9,
plus `roofline` (dominant kernel, HIP-event timed) and, at N=1, `cpu_baseline` (the reference's own
AVX2 path from oracle/_ref timed on the host cores; the oracle is only ever the thing compared
against, never the thing measured as `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from swimm_amd import hip_backend, host, submat, synth  # noqa: E402

METRIC = "GCUPS (whole node) + bit-exact top-r scores vs CPUsearch.c"   # BASELINE.json's metric, verbatim
QUERY_INDEX = 3          # P07327, 375 aa, in synth.QUERY_SET
TOP_R = 20
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# every instruction of the inner loop is VOP3P / 3-source: 4 cycles per wave64 instruction on a SIMD
# (tools/microbench/valu_rate, profiles/r01_valu_issue_rates.txt) -> 1024 SIMDs x 2.4 GHz / 4
VALU_PEAK_GINSTR = 256 * 4 * 2.4e9 / 4 / 1e9
INSTR_PER_ROW, INSTR_PER_COLUMN = 8.5, 10          # 7.5 packed-f16 ops + 1 v_perm per packed row; per-column overhead


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


def cpu_baseline(shard, sm, threads, budget_s=30.0):
    """reference AVX2 path (oracle/_ref) on the same shard, or on a strided subsample if the full shard
    would take longer than ~budget_s at ~10 GCUPS/thread"""
    from oracle import port, ref
    q = shard["query"]
    m_real = len(q)
    a = np.concatenate([q, np.array([23], dtype=np.int8)]) if m_real % 2 else q.copy()   # even padding, sequences.c:382
    m = np.array([len(a)], dtype=np.uint16)
    disp = np.array([0, len(a)], dtype=np.uint32)
    est = m_real * shard["residues"] / (8e9 * max(threads, 1))
    stride = max(1, int(np.ceil(est / budget_s)))
    lens = shard["lengths"][::stride]
    if stride > 1:
        offs = np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))])
        idx = np.arange(0, shard["n"], stride)
        codes = np.concatenate([shard["codes"][offs[i]:offs[i + 1]] for i in idx])
    else:
        codes = shard["codes"]
    residues = int(lens.astype(np.int64).sum())
    if ref.available():
        one = host.assemble_single_chunk(lens, codes, 32, 60)
        sc, wt = ref.cpu_search(a, m, disp, one["b"], one["n"], one["nbbs"], one["disp"], sm, 10, 2, 32, threads=threads)
        kind = "reference"
    else:
        one = host.assemble_single_chunk(lens, codes, 32, 60)
        t0 = time.time()
        sc = port.search_exact(a, m, disp, one["b"], one["n"], one["disp"], sm, 10, 2, 32, threads=threads)
        wt = time.time() - t0
        kind = "port"
    gcups = m_real * residues / wt / 1e9
    sample = f"c2 shard, every {stride}th sequence ({len(lens)} sequences, {residues} residues), 1 pass, {wt:.2f} s"
    return {"value": round(gcups, 2), "unit": "GCUPS", "cores": threads, "kind": kind, "sample": sample}, sc[0, :len(lens)], stride


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the 1M-sequence shard (tests only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--max-waves", type=int, default=0)
    ap.add_argument("--wgs-per-cu", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    share = os.environ.get("SWIMM_BENCH_SHARE_DEVICE") == "1"   # rehearsal only: every rank on GPU 0, gloo only
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    rccl = None
    if world > 1:
        import torch.distributed as dist
        # control plane (barrier, max-over-ranks) on gloo; the only data exchange of the path -- 20 (score, index)
        # pairs per rank and step -- goes over RCCL
        dist.init_process_group(backend="gloo")
        if not share:
            try:
                rccl = dist.new_group(backend="nccl")
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, group=rccl)
                torch.cuda.synchronize()
                assert int(t.item()) == world
            except Exception as e:   # result path only: fall back to gloo, say so in the output
                print(f"[rank {rank}] RCCL group unavailable ({e}); top-r lists go over gloo", file=sys.stderr)
                rccl = None
            ok = torch.tensor([1 if rccl is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank takes the same result path
            if int(ok.item()) == 0:
                rccl = None

    sm = submat.table("blosum62")
    shard = build_shard(2 + 1000 * rank, args.scale)
    chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
    q = shard["query"]
    m = np.array([len(q)], dtype=np.uint16)
    disp = np.array([0, len(q)], dtype=np.uint32)

    searcher = hip_backend.HipSearcher(dev_index)
    if args.rows_per_wave:
        searcher.set_option("rows_per_wave", args.rows_per_wave)
    if args.max_waves:
        searcher.set_option("max_waves", args.max_waves)
    if args.wgs_per_cu:
        searcher.set_option("wgs_per_cu", args.wgs_per_cu)
    searcher.set_queries(q, m, disp, sm, 10, 2)
    t_up = time.time()
    for ch in chunks.chunks:
        searcher.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
    t_up = time.time() - t_up
    padded_bytes = chunks.vD

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        ts, ti, wt = searcher.search_topr(TOP_R, shard["n"])
        if dist is not None:   # result path only: 20 (score, index) pairs per rank
            mine = torch.from_numpy(np.concatenate([ts[0].astype(np.int64), ti[0] + rank * (1 << 40)]))
            if rccl is not None:
                mine = mine.cuda()
                allv = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(allv, mine, group=rccl)
            else:
                allv = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(allv, mine)
            g = torch.stack(allv).cpu().numpy()
            ms, mi = host.topr_merge(g[:, :TOP_R].astype(np.int32), g[:, TOP_R:], TOP_R)
            return ms, mi, wt
        return ts[0], ti[0], wt

    for _ in range(args.warmup):
        one_step()
    kernel_ms, wts = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        top_s, top_i, wt = one_step()
        kernel_ms.append(searcher.last_stats()["kernel_ms"])
        wts.append(wt)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        res = torch.tensor([shard["residues"]], dtype=torch.float64)
        dist.all_reduce(res, op=dist.ReduceOp.SUM)
        total_residues = float(res.item())
    else:
        total_residues = float(shard["residues"])
    stats = searcher.last_stats()

    if rank == 0:
        m_real = len(q)
        ms_per_step = elapsed / args.steps * 1e3
        gcups = m_real * total_residues * args.steps / elapsed / 1e9
        # dominant kernel: sw_pipe_kernel<T, f16 tier, dynamic queue>; one launch per step for this query (single pass)
        k_ms = float(np.mean(kernel_ms))
        launches = max(1, stats["launches"])
        plan_now = searcher.last_plan(0)
        cells_real = m_real * float(shard["residues"])
        # algorithmic HBM bytes of one launch (= one pass): the tiled database residues are read once per pass
        # (1/m B per cell per pass) and the scores are updated (8 B per sequence); between two passes the strip
        # boundary (H and F, 2 B each, per sequence and column = 4x the tiled residue bytes) is written once and
        # read once -- SURVEY 8(d): 4*w/T_eff with w = 2 and T_eff = rows per pass; a one-pass query has no such term
        passes = max(1, plan_now["passes"])
        alg_bytes = float(padded_bytes) * (1.0 + 8.0 * (passes - 1) / passes) + 8.0 * shard["n"]
        achieved = alg_bytes / (k_ms / launches * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kernel_gcups = cells_real / (k_ms * 1e-3) / 1e9
        # VALU ceiling (SURVEY 8d ceiling (1)): wave-instructions issued per second vs the half-rate issue peak
        cells_padded = float(stats["cells"])
        plan = searcher.last_plan(0)
        T = plan["rows_per_wave"]
        n_instr = cells_padded / (128 * T) * (T * INSTR_PER_ROW + INSTR_PER_COLUMN)
        ginstr = n_instr / (k_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 (exact integers < 2048) -> int16 -> int32", "data": "synthetic",
            "config": {"workload": "c2: 375-aa query x 1M synthetic proteins per GPU, BLOSUM62 g10 e2, top-20",
                       "query_len": m_real, "db_sequences_per_gpu": shard["n"], "db_residues_per_gpu": shard["residues"],
                       "parallelism": f"db-shard x{world}", "plan": searcher.last_plan(0), "scale": args.scale,
                       "topr_exchange": "none" if world == 1 else ("rccl all_gather" if rccl is not None else "gloo all_gather")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": f"swimm::sw_pipe_kernel<{T}, 2, true>", "kernel_ms": round(k_ms / launches, 4),
                         "alg_bytes_per_launch": alg_bytes,
                         "note": "VALU-bound kernel: see valu_roofline; HBM carries the DB residues once per pass and the strip boundary between passes"},
            "valu_roofline": {"achieved": round(ginstr, 1), "peak": round(VALU_PEAK_GINSTR, 1), "unit": "G wave-instr/s",
                              "frac": round(ginstr / VALU_PEAK_GINSTR, 4), "kernel_only_gcups": round(kernel_gcups, 2),
                              "padded_cells": cells_padded, "instr_per_wave_column": T * INSTR_PER_ROW + INSTR_PER_COLUMN},
            "search_call_ms": round(float(np.mean(wts)) * 1e3, 4),
            "h2d_upload_s": round(t_up, 3), "datagen_s": round(shard["gen_s"], 2),
            "top1": [int(top_s[0]), int(top_i[0])],
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = len(os.sched_getaffinity(0))
            cb, cpu_scores, stride = cpu_baseline(shard, sm, threads)
            out["cpu_baseline"] = cb
            # the oracle doubles as a checker here: GPU scores of the sampled sequences must agree
            full, _ = searcher.search(chunks.vc * 128)
            ok = bool(np.array_equal(full[0, :shard["n"]][::stride], cpu_scores))
            out["cpu_baseline"]["matches_gpu"] = ok
            out["bit_exact_vs_reference"] = ok           # the second half of the metric: every score of the shard, not only the top-r
            try:
                model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                model = "unknown"
            out["cpu_baseline"]["cpu_model"] = model
            if not ok:
                raise SystemExit("GPU scores differ from the CPU reference on the benchmark shard")
        print(json.dumps(out), flush=True)
    searcher.close()
    chunks.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
