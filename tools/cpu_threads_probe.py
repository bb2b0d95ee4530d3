#!/usr/bin/env python3
"""How many host threads should the CPU checker / `-m 0` use on this box?  Prints the cgroup CPU quota and the affinity mask, then
times the reference's AVX2 path (oracle/_ref, checker) and the product's own `-m 0` (swimm_cpu_search) on a tenth of the c2 shard
with 8 ... 256 threads.  CPU only.  usage: python tools/cpu_threads_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from swimm_amd import host, submat  # noqa: E402
from oracle import ref  # noqa: E402  (checker: test infrastructure)

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
shard = bench.build_shard(2, 0.1)
L, codes, q = shard["lengths"], shard["codes"], shard["query"]
sm = submat.table("blosum62")
real = np.array([len(q)], dtype=np.int64)
mp = real + (real % 2)
dp = np.concatenate([[0], np.cumsum(mp)]).astype(np.uint32)
a = np.full(int(mp.sum()), 23, dtype=np.int8)
a[:len(q)] = q
one = host.assemble_single_chunk(L, codes, 32, 60)
cells = float(real.sum()) * float(L.astype(np.int64).sum())
for threads in (8, 16, 24, 32, 64, 128, 256):
    if threads > (os.cpu_count() or 1):
        break
    best_r = best_p = 0.0
    for rep in range(2):
        _, wt = ref.cpu_search(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["nbbs"], one["disp"], sm, 10, 2, 32, threads=threads)
        best_r = max(best_r, cells / wt / 1e9)
        _, wt2 = host.cpu_search(a, mp.astype(np.uint16), dp, one["b"], one["n"], one["disp"], sm, 10, 2, 32, threads=threads)
        best_p = max(best_p, cells / wt2 / 1e9)
    print(f"{threads:4d} threads: reference {best_r:7.1f} GCUPS, product -m 0 {best_p:7.1f} GCUPS", flush=True)
