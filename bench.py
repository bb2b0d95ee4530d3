#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SWIMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--scale S]

What one invocation measures (no --workload given):

  top level   c2 (BASELINE.json configs[1], the configuration the metric is quoted on): one 375-residue query
              (P07327-shaped, synthetic) against 1 000 004 synthetic proteins (~6e8 residues, log-normal lengths),
              BLOSUM62, gap 10/2, top-20; K timed steps after W warm-up steps.  With N > 1 every rank holds its own
              1M-sequence shard (seed differs per rank): WEAK scaling -- the same per-GPU work at every N, so that the
              N = 1 point of a scaling run is the single-GPU bench line.
  N = 1       `secondary`: the other BASELINE.json configurations, each with its own parity sample against the reference,
              kernel name, `roofline`, `cpu_baseline` and first-search-after-a-cold-upload figures:
                c3  the 20-query set x the Swiss-Prot-shaped database (540 080 sequences, 0.01 % tail to 35 000), BLOSUM50
                c4  the 5 478-residue query (Q9UKN1-shaped) x the WHOLE Env-NR-shaped database (35.5 M sequences, 7e9
                    residues), BLOSUM62 10/2 -- the configuration north_star's ">= 10 x 350 GCUPS" sentence is written about
                c5  the 20-query set x the Env-NR-shaped database at scale 0.25 (1.75e9 residues), PAM250
              (2 timed steps each after 1 warm-up step); `summary`, the last object of the line, repeats their headline
              figures in a few hundred bytes (a driver that keeps only the tail of stdout still sees them).
  N > 1       `strong_scaling`: north_star's multi-GPU case, BASELINE.json configs[4] -- ONE Env-NR-shaped database
              (35.5 M sequences / 7e9 residues), 20 queries, PAM250, cut into 8 N slabs of equal padded size that are
              dealt statically to the N ranks (sharding.assign_chunks); per-rank kernel GCUPS and HBM fraction, which
              path carried the top-20 lists, and the parity flags of every rank.  The total work does not depend on N.

--workload X runs X alone as the top-level record (profiling runs, rehearsals); --scale applies to it.

A "step" is one complete search of the resident database shard: DP kernels, promotion re-runs, device top-20; with
N > 1 the ranks' top-20 lists are all-gathered (RCCL; 20 x 16 bytes per query and rank) and merged on the host --
the path has no other exchange step.  The database is resident in HBM before the timed region (288 GB holds every
configuration); the first search after a cold upload is reported beside it as `value_incl_h2d` (the very first cold upload of
the process: what a one-shot `swimm -S search` pays) and `value_incl_h2d_pooled` (a later re-upload, device buffers reused).

N > 1: one process per GPU.  Under torch.distributed.run the environment carries RANK / LOCAL_RANK / WORLD_SIZE;
invoked plainly (`python bench.py --gpus N`) this process starts the N ranks itself, before it touches the GPU, and
relays rank 0's line.

Prints ONE JSON line on rank 0: metric GCUPS = Q_real * D / t / 1e9 (swimm.c:163), `roofline` (dominant kernel,
HIP-event timed inside the library on the streams the kernels run on), `valu_roofline` (the ceiling that binds this
kernel, against the guide's SIMD issue peak and its instruction class's 4-cycle rate) and, at N = 1, `cpu_baseline`
(the reference's own AVX2 path from oracle/_ref timed on the host cores).  The oracle is only ever the thing compared
against, never the thing measured as `value`; every rank checks a sample of its own scores against it and a mismatch
is a non-zero exit.
"""
import argparse
import datetime
import glob
import json
import os
import socket
import subprocess
import sys
import time

# (the CPU checker's OpenMP team -- every hardware thread of the host -- parks when a parallel region ends instead of spinning
# on all cores into the measurement that follows it; read by libgomp when it is loaded)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from swimm_amd import hip_backend, host, sharding, submat, synth, workloads  # noqa: E402

METRIC = "GCUPS (whole node) + bit-exact top-r scores vs CPUsearch.c"   # BASELINE.json's metric, verbatim
QUERY_INDEX = 3          # P07327, 375 aa, in synth.QUERY_SET
TOP_R = 20
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
NUM_CU, SHADER_HZ = 256, 2.4e9
# VALU ceilings in wave64 instructions per second (256 CUs x 4 SIMDs x 2.4 GHz):
#   the guide's issue peak -- a SIMD issues a wave64 VALU instruction over 2 cycles (157.3 TFLOPS fp32 vector);
#   the instruction class of this kernel -- every VOP3P packed / 3-source op takes 4 cycles on gfx950
#   (tools/microbench/valu_rate, profiles/r01_valu_issue_rates.txt: 4.2-4.5 measured in isolation, by occupancy)
VALU_PEAK_SIMD_ISSUE = NUM_CU * 4 * SHADER_HZ / 2 / 1e9
VALU_PEAK_CLASS = NUM_CU * 4 * SHADER_HZ / 4 / 1e9
VALU_CLASS_MEASURED = NUM_CU * 4 * SHADER_HZ / 4.40 / 1e9     # that class in isolation, four waves per SIMD (profiles/r01_valu_issue_rates.txt: 4.40 cycles)
INSTR_PER_ROW, INSTR_PER_COLUMN = 6.5, 8           # model when no PMC profile matches: 6.5 packed-f16 ops per packed row (column-offset form, fused pair score); per-column overhead
C4_SCALE = 1.3e9 / 6.99e9                          # the 1.3e9-residue Env-NR subsample of SURVEY 8d
DEFAULT_SCALE = {"c2": 1.0, "c3": 1.0, "c4": 1.0, "c5": 0.25}


def build_shard(seed: int, scale: float):
    """sorted lengths + recoded residues of one c2 shard, with planted homologs of the query"""
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
    return {"lengths": lens_sorted, "codes": codes, "residues": total, "n": len(lens_sorted), "query": host.recode(q_letters)}


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs quota); None = unlimited"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except (OSError, ValueError):
        return None


def host_cpus():
    """(threads the CPU legs should run, physical cores they amount to): the hardware threads this process may run on -- unless the
    container's CPU quota is smaller: the GPU boxes of this pool show 256 hardware threads and grant 16 CPUs' worth of time
    (cpu.max 1600000 100000); 256 threads then take turns on 16 CPUs (the reference: 239 GCUPS with 256 threads, 455 with 16;
    tools/cpu_threads_probe.py) and a short run bursts past the quota (1 467 GCUPS for 30 ms with 128 threads)"""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            cores.add(("?", str(c)))
    quota = cpu_quota()
    if quota is not None and quota < len(cores):
        n = max(1, int(round(quota)))
        return n, n
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


SAMPLE_BLOCK = 64


def sample_check(qa_list, lens, codes_of, sm, threads, gpu_scores_of, budget_s, n_total):
    """blocks of 64 consecutive sequences, every k-th block of a shard, through the CPU checker; k chosen so that the CPU
    work fits ~budget_s (the database is length-sorted: evenly spaced blocks see every length class).
    codes_of(i0, i1) -> residues of the local sequences [i0, i1); gpu_scores_of(idx) -> [q, len(idx)] GPU scores.
    -> (ok, stride, sampled sequences, sampled residues, cpu seconds, kind, inputs of the checker run)"""
    q_res = sum(len(q) for q in qa_list)
    residues = float(np.asarray(lens, dtype=np.int64).sum())
    est = q_res * residues / (12.0e9 * max(threads, 1))      # the reference runs at 14-28 GCUPS per core that is really there (host_cpus)
    stride = max(1, int(np.ceil(est / budget_s)))
    if stride == 1:
        idx = np.arange(n_total, dtype=np.int64)
        sub_codes = codes_of(0, n_total)
    else:
        starts = np.arange(0, n_total, SAMPLE_BLOCK * stride, dtype=np.int64)
        idx = np.concatenate([np.arange(b, min(b + SAMPLE_BLOCK, n_total), dtype=np.int64) for b in starts])
        sub_codes = np.concatenate([codes_of(int(b), int(min(b + SAMPLE_BLOCK, n_total))) for b in starts])
    sub_lens = np.asarray(lens)[idx]
    sc, wt, kind = oracle_scores(qa_list, sub_lens, sub_codes, sm, threads)
    ok = bool(np.array_equal(gpu_scores_of(idx), sc))
    return ok, stride, len(idx), int(sub_lens.astype(np.int64).sum()), wt, kind, (sub_lens, sub_codes)


def product_m0_gcups(qa_list, sub_lens, sub_codes, sm, threads):
    """the product's own `-m 0` (swimm_cpu_search, libswimm_host.so: what `swimm -m 0` and the host leg of `-m 2` run) on the
    checker's sample with the checker's thread count -> GCUPS of real cells; None when it disagrees with itself"""
    real = np.array([len(q) for q in qa_list], dtype=np.int64)
    mp = real + (real % 2)
    dp = np.concatenate([[0], np.cumsum(mp)]).astype(np.uint32)
    a = np.full(int(mp.sum()), 23, dtype=np.int8)
    for k, q in enumerate(qa_list):
        a[dp[k]:dp[k] + real[k]] = q
    one = host.assemble_single_chunk(sub_lens, sub_codes, 32, 60)
    _, wt = host.cpu_search(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["disp"], sm, 10, 2, 32, threads=threads)
    return float(real.sum()) * float(np.asarray(sub_lens, dtype=np.int64).sum()) / max(wt, 1e-9) / 1e9


def pmc_profiles(workload, scale):
    """the PMC summaries committed under profiles/ for this configuration (rocprofv3 --pmc passes, tools/profile_bench.sh
    + tools/summarize_profile.py), latest round last"""
    out = []
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            d = json.load(open(fn))
        except Exception:
            continue
        if d.get("workload_key") == workload and abs(float(d.get("scale", -1)) - scale) < 1e-6:
            d["file"] = os.path.relpath(fn, ROOT)
            out.append(d)
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has not
    touched the GPU and never will), relay rank 0's output, fail if any rank fails."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
        # buffer exchange between the ranks fails in hipIpcGetMemHandle.  The image exports it already: a launcher's
        # own value wins, the default only covers an environment that lost it.
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


class Env:
    """this rank: its device, the process groups, the host threads it may use for the checker"""
    rank = 0; local_rank = 0; world = 1
    dist = None; rccl = None; rccl_ranks = 0; rccl_error = None
    share = False; dev_index = 0; physical_gpus = 1
    threads = 1; cores = 1; threads_all = 1
    cpus = ""; pci = ""          # the CPUs this rank's threads are bound to, the PCI address of its device


def setup(args) -> Env:
    import torch
    env = Env()
    env.rank = int(os.environ.get("RANK", "0"))
    env.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    if env.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env.world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    env.share = os.environ.get("SWIMM_BENCH_SHARE_DEVICE") == "1"   # rehearsal only: every rank on GPU 0, gloo only
    visible = torch.cuda.device_count()
    env.dev_index = 0 if env.share else env.local_rank % max(visible, 1)
    torch.cuda.set_device(env.dev_index)
    env.all_cpus = sorted(os.sched_getaffinity(0))
    env.pci = hip_backend.device_pci_bus_id(env.dev_index)
    bdfs = [env.pci]
    if env.world > 1:
        import torch.distributed as dist
        env.dist = dist
        # control plane (barrier, max-over-ranks) on gloo; the only data exchange of the path -- the ranks' top-20
        # lists, once per step -- goes over RCCL
        dist.init_process_group(backend="gloo")
        # which physical devices do the ranks sit on?  (a launcher may give every rank its own HIP_VISIBLE_DEVICES, so
        # the index says nothing: compare the devices' identities)
        prop = torch.cuda.get_device_properties(env.dev_index)
        ident = (socket.gethostname(), str(getattr(prop, "uuid", "")), getattr(prop, "pci_bus_id", -1), getattr(prop, "pci_device_id", -1),
                 getattr(prop, "pci_domain_id", -1))
        if visible >= env.world and not env.share:
            # every rank sees all the node's devices and took the one of its local rank: distinct by index, whatever the runtime
            # reports as identity (a build that leaves uuid / PCI ids empty must not turn a real multi-GPU run into an error)
            ident = ident + ("index", env.dev_index)
        idents = [None] * env.world
        dist.all_gather_object(idents, ident)
        env.physical_gpus = len(set(idents))
        if env.physical_gpus < env.world and not env.share:
            if env.rank == 0:
                print(f"bench.py: {env.world} ranks on {env.physical_gpus} physical GPU(s); set SWIMM_BENCH_SHARE_DEVICE=1 to rehearse "
                      f"on a shared device (the value is then not a multi-GPU figure)", file=sys.stderr)
            raise SystemExit(3)
        if not env.share:
            err = None
            try:
                # (a rank whose RCCL set-up fails leaves the others inside this collective: they give up after three minutes and
                # everybody meets again in the gloo exchange below)
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
                env.rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, group=env.rccl)
                torch.cuda.synchronize()
                env.rccl_ranks = int(t.item())
                if env.rccl_ranks != env.world:
                    raise RuntimeError(f"RCCL all_reduce saw {env.rccl_ranks} of {env.world} ranks")
            except Exception as e:   # result path only: fall back to gloo, and SAY so in the JSON line
                err = f"rank {env.rank}: {type(e).__name__}: {e}"
                print(f"[rank {env.rank}] RCCL group unavailable ({e}); top-r lists go over gloo", file=sys.stderr)
            errs = [None] * env.world
            dist.all_gather_object(errs, err)             # every rank takes the same result path
            errs = [e for e in errs if e]
            if errs:
                env.rccl = None
                env.rccl_ranks = 0
                env.rccl_error = " | ".join(errs)[:2000]
        else:
            env.rccl_error = "not attempted: all ranks share one device (SWIMM_BENCH_SHARE_DEVICE=1)"
        bdfs = [None] * env.world
        dist.all_gather_object(bdfs, env.pci)
    # Placement: this rank's threads -- the library's uploader thread (copies out of pageable memory), the checker's OpenMP
    # team, numpy -- on the CPUs local to its GPU, shared by whole cores with the ranks whose GPUs name the same CPUs
    # (swimm_amd/csrc/host/affinity.h; plain sched_setaffinity, threads started from here on inherit it).  SWIMM_HIP_BIND=0 skips it.
    if os.environ.get("SWIMM_HIP_BIND") != "0":
        mine = host.affinity_plan(bdfs, env.rank, env.all_cpus)
        host.affinity_apply(mine)
        env.cpus = host._cpulist(mine)
    env.threads_all, env.cores = host_cpus()
    env.threads = env.threads_all            # (the bound set IS the rank's share of the host)
    return env


class all_host_cpus:
    """N = 1 only: the CPU baseline is what the WHOLE host can do, not the socket next to the GPU"""

    def __init__(self, env):
        self.env = env

    def __enter__(self):
        self.before = sorted(os.sched_getaffinity(0))
        if self.env.world == 1:
            os.sched_setaffinity(0, self.env.all_cpus)
        return host_cpus()

    def __exit__(self, *exc):
        os.sched_setaffinity(0, self.before)


def make_workload(env: Env, name: str, scale: float):
    """the rank's shard of a configuration: queries, matrix, how to upload it, how to check it"""
    w = {"name": name, "scale": scale, "chunks": None}
    t0 = time.time()
    if name == "c2":
        w["sm"] = submat.table("blosum62")
        shard = build_shard(2 + 1000 * env.rank, scale)
        chunks = host.Chunks(shard["lengths"], shard["codes"], 128, 96 << 20)
        w["chunks"] = chunks
        w["qa_list"] = [shard["query"]]
        w["a"], w["m"], w["disp"] = shard["query"], np.array([len(shard["query"])], np.uint16), np.array([0, len(shard["query"])], np.uint32)
        w["my_lens"], w["my_n"], w["n_valid"], w["score_stride"] = shard["lengths"], shard["n"], shard["n"], chunks.vc * 128
        w["my_residues"], w["my_padded"] = shard["residues"], chunks.vD
        offs = np.concatenate([[0], np.cumsum(shard["lengths"].astype(np.int64))])
        w["local_index"] = np.arange(shard["n"], dtype=np.int64)      # position of the rank's sequences in its score rows
        w["codes_of"] = lambda i0, i1: shard["codes"][offs[i0]:offs[i1]]
        w["index_base"] = env.rank * (1 << 40)                         # ranks hold disjoint databases: make the merged indices distinct
        w["text"] = "c2: 375-aa query x 1M synthetic proteins per GPU, BLOSUM62 g10 e2, top-20"
        w["scaling"] = "weak"
        w["upload_kind"] = "reference chunk layout"

        def upload(searcher):
            for ch in chunks.chunks:
                searcher.add_chunk(ch["b"], ch["n"], ch["disp"], 128, ch["first_group"])
    else:
        db = workloads.SortedDb(name, scale)
        w["sm"] = submat.table(db.matrix)
        slabs = db.slabs(8 * env.world)
        owner = sharding.assign_chunks([s[2] for s in slabs], env.world)
        mine = [s for s, o in zip(slabs, owner) if o == env.rank]
        w["a"], w["m"], w["disp"] = db.a, db.m, db.disp
        w["qa_list"] = [db.a[db.disp[k]:db.disp[k + 1]] for k in range(len(db.m))]
        slab_codes = [db.codes(s0, s1) for s0, s1, _ in mine]
        local_index = np.concatenate([np.arange(s0, s1, dtype=np.int64) for s0, s1, _ in mine])
        w["local_index"] = local_index
        w["my_lens"] = db.lengths[local_index]
        w["my_n"], w["n_valid"], w["score_stride"] = len(local_index), db.n, (db.n + 127) // 128 * 128
        w["my_residues"], w["my_padded"] = int(w["my_lens"].astype(np.int64).sum()), sum(s[2] for s in mine)
        slab_first = np.concatenate([[0], np.cumsum([s1 - s0 for s0, s1, _ in mine])]).astype(np.int64)       # local index of every slab's first sequence

        def codes_of(i0, i1):
            """residues of the rank's local sequences [i0, i1), out of the slabs it holds in memory"""
            out = []
            for k, (s0, s1, _) in enumerate(mine):
                a0, a1 = max(i0, int(slab_first[k])), min(i1, int(slab_first[k + 1]))
                if a0 < a1:
                    g0, g1 = s0 + a0 - int(slab_first[k]), s0 + a1 - int(slab_first[k])
                    out.append(slab_codes[k][int(db.offs[g0] - db.offs[s0]):int(db.offs[g1] - db.offs[s0])])
            return out[0] if len(out) == 1 else np.concatenate(out)
        w["codes_of"] = codes_of
        w["index_base"] = 0                                            # one database: indices are global already
        if name == "c4":
            w["text"] = (f"c4: 5478-aa query x Env-NR-shaped database at scale {scale:.4f} ({db.n} sequences, {db.residues} residues), "
                         f"BLOSUM62 g10 e2, top-20; {len(slabs)} slabs dealt statically to {env.world} rank(s)")
        elif name == "c3":
            w["text"] = (f"c3: 20-query set (144-5478 aa) x Swiss-Prot-shaped database at scale {scale} ({db.n} sequences, {db.residues} residues, "
                         f"longest {int(db.lengths[-1])}), BLOSUM50 g10 e2, top-20; {len(slabs)} slabs dealt statically to {env.world} rank(s)")
        else:
            w["text"] = (f"c5: 20-query set (144-5478 aa) x ONE Env-NR-shaped database at scale {scale} ({db.n} sequences, "
                         f"{db.residues} residues), PAM250 g10 e2, top-20; {len(slabs)} slabs dealt statically to {env.world} rank(s)")
        w["scaling"] = "strong"
        w["upload_kind"] = ".seq slabs"

        def upload(searcher):
            for (s0, s1, _), codes in zip(mine, slab_codes):
                searcher.add_sequences(db.lengths[s0:s1], codes, first_seq=s0)
    w["upload"] = upload
    w["datagen_s"] = time.time() - t0
    return w


def run_workload(env: Env, args, name: str, scale: float, steps: int, warmup: int, cpu_budget: float, want_cold: bool, want_cpu_baseline: bool):
    """one configuration on this rank's GPU: upload, warm-up, the timed steps, parity against the CPU checker, the first
    search after a cold upload.  -> (record (rank 0; None elsewhere), ok on every rank)"""
    import torch
    dist, world, rank = env.dist, env.world, env.rank
    w = make_workload(env, name, scale)
    searcher = hip_backend.HipSearcher(env.dev_index)
    for k, v in (("rows_per_wave", args.rows_per_wave), ("max_waves", args.max_waves)):
        if v:
            searcher.set_option(k, v)
    searcher.set_option("time_launches", 1)      # HIP events around every pipeline launch, on the stream it runs on
    sm, m = w["sm"], w["m"]
    nq = len(m)
    searcher.set_queries(w["a"], m, w["disp"], sm, 10, 2)
    t_up = time.time()
    w["upload"](searcher)
    t_up = time.time() - t_up

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange(ts, ti):
        """the path's only exchange step: every rank's top-20 per query -> merged listing (sharding.allgather_merge: the
        function the world-size-2 gloo test on the CPU covers; here over the RCCL group when there is one)"""
        return sharding.allgather_merge(ts, ti, dist, group=env.rccl, index_base=w["index_base"])

    def one_step():
        ts, ti, wt = searcher.search_topr(TOP_R, w["n_valid"])
        ms, mi = exchange(ts, ti)
        return ms, mi, wt

    n_searches = 0
    for _ in range(warmup):
        one_step(); n_searches += 1
    kernel_ms, wts, launch_ms = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        top_s, top_i, wt = one_step(); n_searches += 1
        kernel_ms.append(searcher.last_stats()["kernel_ms"])
        launch_ms.append(searcher.last_launch_ms())
        wts.append(wt)
    barrier()
    elapsed = time.perf_counter() - t0
    my_residues, my_padded, my_n = w["my_residues"], w["my_padded"], w["my_n"]
    total_residues = float(my_residues)
    k_ms_mean = float(np.mean(kernel_ms))
    stats = searcher.last_stats()
    plans = [searcher.last_plan(k) for k in range(nq)]
    kernel_name = searcher.last_kernel_name(nq - 1)          # the longest query's kernel dominates the device time
    q_real = int(m.astype(np.int64).sum())
    # Algorithmic HBM bytes of one search of this rank's shard (w = 2, T_eff = rows per pass): per pass the
    # tiled database bytes are read once (1/m B per cell of the whole query); between two passes the strip boundary (H
    # and F, 2 B each, per sequence and column = 4x the tiled residue bytes) is written once and read once; 8 B per
    # sequence and query for the scores.  For c2 (one query, `passes` launches of one kernel) divided by the launches
    # this is the per-launch figure of DESIGN.md; for a query batch it is the sum over the queries.
    alg_bytes = sum(float(my_padded) * (p["passes"] + 8.0 * (p["passes"] - 1)) + 8.0 * my_n for p in plans)
    my_kernel_gcups = q_real * float(my_residues) / (k_ms_mean * 1e-3) / 1e9
    my_hbm_frac = alg_bytes / (k_ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBS
    per_rank = [(my_kernel_gcups, my_hbm_frac)]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        res = torch.tensor([my_residues], dtype=torch.float64)
        dist.all_reduce(res, op=dist.ReduceOp.SUM)
        total_residues = float(res.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, (my_kernel_gcups, my_hbm_frac))

    # ---- parity: every rank checks a sample of its own scores, and the merged listing, against the CPU checker ----
    full = np.zeros((nq, w["score_stride"]), dtype=np.int32)
    searcher.search(w["score_stride"], out=full); n_searches += 1
    local_index = w["local_index"]
    mine_scores = full[:, local_index]
    del full
    with all_host_cpus(env) as (chk_threads, chk_cores):
        ok, stride, n_s, res_s, cpu_s, kind, chk_in = sample_check(w["qa_list"], w["my_lens"], w["codes_of"], sm, chk_threads, lambda idx: mine_scores[:, idx], cpu_budget, my_n)
        m0_gcups = None
        if world == 1 and want_cpu_baseline:
            # the product's own -m 0 on the same sample when the checker took under 2 s on it, else on every third block of it
            # (keeps the default run within minutes), same threads
            o = np.concatenate([[0], np.cumsum(chk_in[0].astype(np.int64))])
            blocks = range(0, len(chk_in[0]), SAMPLE_BLOCK * (1 if cpu_s < 2.0 else 3))
            sub_l = np.concatenate([chk_in[0][b:b + SAMPLE_BLOCK] for b in blocks])
            sub_c = np.concatenate([chk_in[1][o[b]:o[min(b + SAMPLE_BLOCK, len(chk_in[0]))]] for b in blocks])
            m0_gcups = product_m0_gcups(w["qa_list"], sub_l, sub_c, sm, chk_threads)
            m0_sample = (len(sub_l), int(sub_l.astype(np.int64).sum()))
        del chk_in
    # the merged top-20 of the timed path (device top-r + exchange) against a host selection over the ranks' full vectors
    hs = np.zeros((nq, TOP_R), np.int32); hi = np.zeros((nq, TOP_R), np.int64)
    for k in range(nq):
        s_, i_ = host.topr(mine_scores[k], TOP_R)
        hs[k], hi[k] = s_, np.where(i_ >= 0, local_index[np.maximum(i_, 0)], -1)
    ms2, mi2 = exchange(hs, hi)
    ok_top = bool(np.array_equal(ms2, top_s) and np.array_equal(mi2, top_i))
    del mine_scores
    flags = torch.tensor([1 if ok else 0, 1 if ok_top else 0], dtype=torch.int32)
    if dist is not None:
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    all_ok, all_top_ok = bool(flags[0].item()), bool(flags[1].item())

    # ---- first search after a cold upload (the reference's workTime brackets the transfers, MICsearch.c:51,350) ----
    # Twice: the first one also pays for the device buffers of the streamed parts (hipMalloc; they go to the library's pool when the
    # database is cleared), the second is what every later re-upload in the process costs.  Both are reported.
    # want_cold = how many times: the FIRST one also pays for whatever the library allocates for a streamed search (what a
    # one-shot `swimm -S search` pays: value_incl_h2d); a second one finds the device buffers in the library's pool
    # (value_incl_h2d_pooled).  N > 1 strong-scaling record: once.
    colds = []
    time.sleep(0.25)          # (the checker's threads have parked)
    for _ in range(int(want_cold)):
        searcher.clear_db()
        searcher.set_option("lazy_upload", 1)        # chunks stream in while the search runs (what swimm_hip_search_chunks does)
        barrier()
        t0 = time.perf_counter()
        w["upload"](searcher)
        exchange(*searcher.search_topr(TOP_R, w["n_valid"])[:2])
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        colds.append(dt)
    cold_first_s = colds[0] if colds else float("nan")
    cold_pooled_s = colds[1] if len(colds) > 1 else float("nan")
    placement = [(env.rank, env.pci, env.cpus, round(t_up, 3))]
    if dist is not None:
        placement = [None] * world
        dist.all_gather_object(placement, (env.rank, env.pci, env.cpus, round(t_up, 3)))

    rec = None
    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        gcups = q_real * total_residues * steps / elapsed / 1e9
        launches = max(1, stats["launches"])
        # The dominant kernel's launches: the library brackets every pipeline launch with HIP events on the stream it is
        # launched on and reports their sum and number (two launches that share the chip on two streams each count with
        # their own, longer, duration -- exactly what a kernel trace lists per dispatch).
        pipe_launches = int(round(np.mean([n for _, n in launch_ms])))
        single_kernel = nq == 1                                    # one query = one kernel instantiation: per-launch figures are meaningful
        plan_key = {"rows_per_wave": plans[-1]["rows_per_wave"], "waves": plans[-1]["waves"], "passes": plans[-1]["passes"]}
        profs = pmc_profiles(name, scale) if world == 1 else []
        cells_real = q_real * float(my_residues)
        kernel_gcups = cells_real / (k_ms_mean * 1e-3) / 1e9
        if pipe_launches < 1:
            # a shard so small that every group went through the lane-systolic kernel: the pipeline kernel never ran, and a
            # per-launch figure of a kernel that did not run would be fiction
            roofline = None
            roofline_note = ("no pipeline-kernel launch in this search: every group of the shard was aligned by sw_lane_kernel (lane-systolic, one wave per "
                             "alignment; DESIGN.md 3.2), whose launches the library does not bracket one by one; device time of the search "
                             f"{k_ms_mean:.3f} ms, algorithmic bytes {alg_bytes:.0f}")
        else:
            per_launch_ms = float(np.mean([ms_ / max(n, 1) for ms_, n in launch_ms]))
            achieved = (alg_bytes / pipe_launches) / (per_launch_ms * 1e-3) / 1e9 if single_kernel else alg_bytes / (k_ms_mean * 1e-3) / 1e9
            prof, traffic = None, None
            for d in profs:          # a number from another launch plan is not this run's traffic
                if single_kernel and d.get("plan") == plan_key and d.get("kernel") == kernel_name and d.get("kernel_launches_per_search") == pipe_launches:
                    prof, traffic = d, d["hbm_bytes_per_launch"]
                elif not single_kernel and d.get("plans") == plans and d.get("hbm_bytes_per_search"):
                    prof, traffic = d, d["hbm_bytes_per_search"]       # a query batch: all DP kernels of a search together, like `achieved`
            roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_per": "launch" if single_kernel else "search",
                        "traffic_source": (prof["file"] if prof else "none: no PMC profile of this workload, scale and launch plan is committed (profiles/*_pmc_traffic*.json)"),
                        "kernel": kernel_name, "kernel_ms": round(per_launch_ms, 4), "kernel_launches_per_search": pipe_launches,
                        "device_ms_per_search": round(k_ms_mean, 4), "launches_per_search": launches,
                        "achieved_chip_GBps": round(alg_bytes / (k_ms_mean * 1e-3) / 1e9, 3),
                        "alg_bytes_per_search": alg_bytes, "alg_bytes_per_launch": alg_bytes / pipe_launches if single_kernel else None,
                        "note": ("VALU-bound kernel (see valu_roofline); HBM carries the database residues once per pass and the strip boundary between passes; "
                                 "achieved = alg_bytes_per_launch / kernel_ms (kernel_ms = average duration of the kernel's launches by HIP events on their own streams; "
                                 "a query of three or more short passes runs two launches side by side on two streams, each at about half the chip: achieved_chip_GBps is bytes per search / device time)"
                                 if single_kernel else "query batch: several kernel instantiations; achieved = algorithmic bytes per search / device time of the search")}
            if prof and prof.get("lds_idx_active_per_launch") and prof.get("kernel_avg_ns"):
                # the LDS side of the same profile (north_star: bank conflicts against the LDS pipe's busy cycles)
                cyc = float(prof["kernel_avg_ns"]) * 1e-9 * SHADER_HZ * NUM_CU
                roofline["lds"] = {"bank_conflict_frac": round(float(prof["lds_bank_conflict_per_launch"]) / float(prof["lds_idx_active_per_launch"]), 4),
                                   "busy_frac": round(float(prof["lds_idx_active_per_launch"]) / cyc, 4),
                                   "source": f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, SQ_LDS_IDX_ACTIVE / ({NUM_CU} CUs x kernel cycles) of {prof['file']} (a committed profile of this plan, not this run)"}
            roofline_note = None
        # VALU instructions: from a committed SQ_INSTS_VALU summary of this configuration -- per launch when the plan and
        # kernel are this run's (one query), per search for a query batch -- else from the 6.5-per-row model
        n_instr, instr_src = None, None
        for d in profs:
            if single_kernel and pipe_launches >= 1 and d.get("plan") == plan_key and d.get("kernel") == kernel_name and d.get("sq_insts_valu_per_launch"):
                n_instr = float(d["sq_insts_valu_per_launch"]) * pipe_launches
                instr_src = f"SQ_INSTS_VALU per launch of {d['file']} (a committed profile of this plan and kernel, not this run) x {pipe_launches} launches"
            elif not single_kernel and d.get("sq_insts_valu_per_search") and d.get("plans") == plans:
                n_instr = float(d["sq_insts_valu_per_search"])
                instr_src = f"SQ_INSTS_VALU per search, all kernels, of {d['file']} (a committed profile of this configuration with the same launch plans, not this run)"
        if n_instr is None:
            # wave-columns of query p = padded columns x waves x passes; each costs T x 6.5 + 8 instructions
            n_instr = sum(float(my_padded) / 128 * p["waves"] * p["passes"] * (p["rows_per_wave"] * INSTR_PER_ROW + INSTR_PER_COLUMN) for p in plans)
            instr_src = "model: 6.5 VALU instructions per packed row + 8 per column (no PMC profile of this configuration and plan under profiles/)"
        ginstr = n_instr / (k_ms_mean * 1e-3) / 1e9
        rec = {
            "value": round(gcups, 2), "unit": "GCUPS", "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
            "scaling": w["scaling"],
            "config": {"workload": w["text"], "query_residues": q_real, "queries": nq,
                       "db_sequences_rank0": my_n, "db_residues_rank0": my_residues, "db_residues_total": int(total_residues),
                       "parallelism": f"db-shard x{world}", "plan": plans[-1], "scale": scale,
                       "topr_exchange": "none" if world == 1 else ("rccl all_gather" if env.rccl is not None else "gloo all_gather"),
                       "rccl_ranks": env.rccl_ranks},
            "roofline": roofline,
            "valu_roofline": {"achieved": round(ginstr, 1), "unit": "G wave64-instr/s", "instructions": instr_src,
                              "peak": round(VALU_PEAK_SIMD_ISSUE, 1), "frac": round(ginstr / VALU_PEAK_SIMD_ISSUE, 4),
                              "peak_source": "MI355X_MICROARCH.md: 4 SIMDs per CU, one wave64 VALU instruction issued over 2 cycles",
                              "class_peak": round(VALU_PEAK_CLASS, 1), "class_frac": round(ginstr / VALU_PEAK_CLASS, 4),
                              "class_measured_peak": round(VALU_CLASS_MEASURED, 1), "class_measured_frac": round(ginstr / VALU_CLASS_MEASURED, 4),
                              "class_source": ("VOP3P packed / 3-source ops issue over 4 cycles on gfx950 (profiles/r01_valu_issue_rates.txt); SQ_INSTS_VALU also counts the "
                                               "few 2-cycle scalar-operand ops of a column's overhead, so class_frac is a slight over-estimate of the packed pipe's use"),
                              "kernel_only_gcups": round(kernel_gcups, 2), "padded_cells": float(stats["cells"])},
            "search_call_ms": round(float(np.mean(wts)) * 1e3, 4),
            "value_incl_h2d": round(q_real * total_residues / cold_first_s / 1e9, 2) if colds else None,
            "value_incl_h2d_pooled": round(q_real * total_residues / cold_pooled_s / 1e9, 2) if len(colds) > 1 else None,
            "value_incl_h2d_ms": [round(x * 1e3, 2) for x in colds],
            "value_incl_h2d_note": ((f"search_topr right after a cold upload of the shard (pageable host memory, {w['upload_kind']}; the chunks stream in while the search runs), "
                                     "clock around add_* + search: value_incl_h2d = the FIRST cold search of the process (rounds 1-3 reported the pooled one under this key), "
                                     "value_incl_h2d_pooled = the second (device buffers from the library's pool)") if colds else None),
            "h2d_upload_s": round(t_up, 3), "datagen_s": round(w["datagen_s"], 2), "searches_in_run": n_searches + len(colds),
            "plans": plans if nq > 1 else None,
            "top1": [int(top_s[0][0]), int(top_i[0][0])],
            "bit_exact_vs_reference": all_ok, "merged_top20_matches_full_vectors": all_top_ok,
            "parity_sample": (f"every rank: " + ("every sequence" if stride == 1 else f"every {stride}th block of {SAMPLE_BLOCK} consecutive sequences") +
                              f" of its shard x all queries vs the CPU {kind} ({n_s} sequences, {res_s} residues on rank 0)"),
        }
        if roofline_note:
            rec["roofline_note"] = roofline_note
        rec["placement"] = [{"rank": r_, "pci": pci_, "cpus": cpus_, "eager_upload_s": up_} for r_, pci_, cpus_, up_ in placement]
        if world > 1:
            rec["per_rank_kernel_gcups"] = [round(x[0], 2) for x in per_rank]
            rec["per_rank_kernel_gcups_min"] = round(min(x[0] for x in per_rank), 2)
            rec["per_rank_kernel_gcups_max"] = round(max(x[0] for x in per_rank), 2)
            rec["per_rank_hbm_frac"] = [round(x[1], 5) for x in per_rank]
        if world == 1 and want_cpu_baseline:
            try:
                model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                model = "unknown"
            rec["cpu_baseline"] = {"value": round(q_real * res_s / cpu_s / 1e9, 2), "unit": "GCUPS", "cores": chk_cores, "threads": chk_threads,
                                   "kind": kind, "cpu_model": model, "cpu_quota": cpu_quota(), "matches_gpu": ok,
                                   "sample": (f"{name} shard, " + ("every sequence" if stride == 1 else f"every {stride}th block of {SAMPLE_BLOCK} consecutive sequences") +
                                              f" ({n_s} sequences, {res_s} residues), {nq} quer{'y' if nq == 1 else 'ies'}, {cpu_s:.2f} s"),
                                   "product_m0": round(m0_gcups, 2) if m0_gcups else None,
                                   "product_m0_note": (f"the product's own -m 0 (swimm_cpu_search: AVX2 int8 -> int16 -> int32 tiers, one sequence per lane, column-major sweep) on {m0_sample[0]} sequences / {m0_sample[1]} residues of the same sample, "
                                                       f"{chk_threads} threads: the host leg of `swimm -m 2` and all of BASELINE config 1; the reference's AVX2 path beside it is `value`")}
    searcher.close()
    if w["chunks"] is not None:
        w["chunks"].close()
    if not (all_ok and all_top_ok):
        print(f"[rank {rank}] {name}: GPU scores differ from the CPU checker (sample ok: {all_ok}, merged top-20 ok: {all_top_ok})", file=sys.stderr)
    return rec, all_ok and all_top_ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("c2", "c3", "c4", "c5"), default=None,
                    help="run this configuration alone (default: c2 at the top level + the secondary / strong-scaling records)")
    ap.add_argument("--scale", type=float, default=None, help="fraction of the configuration's database (default: c2 1.0, c3 1.0, c4 1.0, c5 0.25); with --workload")
    ap.add_argument("--strong-scale", type=float, default=1.0, help="scale of the c5 database of the strong_scaling record at N > 1")
    ap.add_argument("--secondary-steps", type=int, default=2)
    ap.add_argument("--secondary-scale", type=float, default=1.0, help="multiplies the default scales of the secondary records (tests: small databases)")
    ap.add_argument("--no-secondary", action="store_true", help="top-level record only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the first-search-after-a-cold-upload measurement (profiling runs: only the resident search's launches in the trace)")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--max-waves", type=int, default=0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    # stdout carries the ONE JSON line and nothing else: whatever the libraries print there (gloo's connection notes, ...)
    # goes to stderr from here on
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    env = setup(args)
    world = env.world

    primary = args.workload or "c2"
    scale = args.scale if (args.scale is not None and args.workload) else DEFAULT_SCALE[primary]
    if args.scale is not None and not args.workload:
        scale = args.scale                  # (--scale without --workload: a smaller c2, for rehearsals)
    baseline = world == 1 and not args.no_cpu_baseline
    n_cold = 0 if args.no_cold else 2
    rec, ok = run_workload(env, args, primary, scale, args.steps, args.warmup, 25.0 if baseline else 6.0, n_cold, baseline)
    all_ok = ok
    extra = {}
    if not args.workload and not args.no_secondary:
        if world == 1:
            secondary = []
            for name in ("c3", "c4", "c5"):
                r2, ok2 = run_workload(env, args, name, DEFAULT_SCALE[name] * args.secondary_scale, args.secondary_steps, 1, 10.0, n_cold, baseline)
                all_ok = all_ok and ok2
                r2["workload"] = name
                # (the line must stay well inside what a caller keeps of stdout -- gpurun: 24 000 characters: the texts that the top-level
                # record already carries are not repeated)
                for obj, keys in ((r2.get("roofline") or {}, ("note",)), (r2["valu_roofline"], ("peak_source", "class_source", "instructions")), (r2, ("value_incl_h2d_note", "placement")),
                                  (r2.get("cpu_baseline") or {}, ("product_m0_note", "cpu_model"))):
                    for k in keys:
                        obj.pop(k, None)
                secondary.append(r2)
            extra["secondary"] = secondary
            c5 = secondary[2]
            # the N = 1 point of north_star's strong-scaling series (GCUPS and HBM fraction at 1, 2, 4, 8 GPUs): the c5 record above, at a
            # quarter of the database so that the line stays within minutes (`--workload c5 --scale 1` runs the whole database on one GPU)
            extra["strong_scaling"] = {
                "workload": c5["config"]["workload"], "scale": DEFAULT_SCALE["c5"] * args.secondary_scale, "value": c5["value"], "unit": "GCUPS", "n_gpus": 1,
                "steps": c5["steps"], "warmup": c5["warmup"], "ms_per_step": c5["ms_per_step"],
                "per_rank_kernel_gcups": [c5["valu_roofline"]["kernel_only_gcups"]],
                "hbm_frac": c5["roofline"]["frac"] if c5["roofline"] else None,
                "rccl_ranks": 0, "topr_exchange": "none", "bit_exact": c5["bit_exact_vs_reference"] and c5["merged_top20_matches_full_vectors"],
                "note": "N = 1 at a quarter of the database (at N > 1 the whole 7e9-residue database is sharded: --strong-scale 1); GCUPS does not depend on the scale from 10 % up (profiles/r04_bench_c5_full.json: the whole database on one GPU, 10 215 GCUPS)"}
        else:
            r2, ok2 = run_workload(env, args, "c5", args.strong_scale, args.secondary_steps, 1, 6.0, 0 if args.no_cold else 1, False)
            all_ok = all_ok and ok2
            if env.rank == 0:
                extra["strong_scaling"] = {
                    "workload": r2["config"]["workload"], "scale": args.strong_scale, "value": r2["value"], "unit": "GCUPS", "n_gpus": world,
                    "steps": r2["steps"], "warmup": r2["warmup"], "ms_per_step": r2["ms_per_step"],
                    "per_rank_kernel_gcups_min": r2["per_rank_kernel_gcups_min"], "per_rank_kernel_gcups_max": r2["per_rank_kernel_gcups_max"],
                    "per_rank_kernel_gcups": r2["per_rank_kernel_gcups"],
                    "hbm_frac": round(float(np.mean(r2["per_rank_hbm_frac"])), 5), "per_rank_hbm_frac": r2["per_rank_hbm_frac"],
                    "hbm_frac_note": "per rank: algorithmic HBM bytes of its shard's search (SURVEY 8d: database bytes once per pass + the strip boundary between passes) / its device time / 8 TB/s",
                    "rccl_ranks": env.rccl_ranks, "topr_exchange": r2["config"]["topr_exchange"],
                    "bit_exact": r2["bit_exact_vs_reference"] and r2["merged_top20_matches_full_vectors"], "parity_sample": r2["parity_sample"],
                    "plans": r2["plans"], "db_residues_total": r2["config"]["db_residues_total"], "datagen_s": r2["datagen_s"], "h2d_upload_s": r2["h2d_upload_s"],
                    "value_incl_h2d": r2["value_incl_h2d"], "value_incl_h2d_ms": r2["value_incl_h2d_ms"], "placement": r2["placement"],
                }
    if env.rank == 0:
        out = {"metric": METRIC, "value": rec["value"], "unit": "GCUPS", "n_gpus": world, "steps": rec["steps"], "warmup": rec["warmup"],
               "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": rec["scaling"], "vs_baseline": None,
               "dtype": "f16 (packed binary16 integers, exact below 1920 with extend 2) -> int16 -> int32", "data": "synthetic"}
        for k, v in rec.items():
            if k not in out:
                out[k] = v
        out.update(extra)
        if world > 1:
            out["physical_gpus"] = env.physical_gpus
            if env.rccl_error:
                out["rccl_error"] = env.rccl_error
        if env.share:
            out["shared_device"] = True
            out["note"] = "REHEARSAL: all ranks share GPU 0 (SWIMM_BENCH_SHARE_DEVICE=1); the value is not a multi-GPU figure"
        # the headline figures of every record once more, LAST in the line and short: a caller that keeps only the tail of
        # stdout still sees the configurations, their sizes and whether they matched the reference
        def brief(r, key):
            return {"workload": key, "scale": r["config"]["scale"], "db_residues": r["config"]["db_residues_total"], "queries": r["config"]["queries"],
                    "value": r["value"], "kernel_only": r["valu_roofline"]["kernel_only_gcups"], "value_incl_h2d": r.get("value_incl_h2d"),
                    "value_incl_h2d_pooled": r.get("value_incl_h2d_pooled"), "ms_per_step": r["ms_per_step"],
                    "hbm_frac": (r["roofline"] or {}).get("frac"), "valu_class_frac": r["valu_roofline"]["class_frac"],
                    "bit_exact_vs_reference": r["bit_exact_vs_reference"] and r["merged_top20_matches_full_vectors"],
                    "cpu_reference_gcups": (r.get("cpu_baseline") or {}).get("value"), "cpu_product_m0_gcups": (r.get("cpu_baseline") or {}).get("product_m0")}
        out["summary"] = [brief(rec, primary)] + [brief(r, r["workload"]) for r in extra.get("secondary", [])]
        if world > 1 and "strong_scaling" in extra:
            ss = extra["strong_scaling"]
            out["summary"].append({"workload": "c5 strong scaling", "scale": ss["scale"], "n_gpus": world, "value": ss["value"], "value_incl_h2d": ss.get("value_incl_h2d"),
                                   "rccl_ranks": ss["rccl_ranks"], "bit_exact": ss["bit_exact"], "hbm_frac": ss["hbm_frac"]})
        line = json.dumps(out)
        print(line, file=json_out, flush=True)
        try:                             # (kept beside the profiles of a gpurun call; never read back)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"bench_last_n{world}.json"), "w") as f:
                f.write(line + "\n")
        except OSError:
            pass
    if env.dist is not None:
        env.dist.barrier()
        env.dist.destroy_process_group()
    if not all_ok:
        raise SystemExit(f"[rank {env.rank}] GPU scores differ from the CPU checker")


if __name__ == "__main__":
    main()
