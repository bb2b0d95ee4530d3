/*
 * affinity.h -- where the host threads that serve one GPU run.
 *
 * The reference starts one host thread per device (`#pragma omp parallel num_threads(num_mics)`, MICsearch.c:53) and leaves
 * their placement to the OpenMP runtime.  On an 8-GPU MI355X node each device hangs off one of two sockets (several NUMA
 * domains); a device thread, the uploader thread it owns (copies out of pageable memory: the staging memcpy runs on that
 * thread's core) and -- in bench.py -- the checker's OpenMP team should sit on CPUs local to the device, and the eight of
 * them should not sit on each other.  The plan, a pure function of sysfs and the caller's allowed CPUs:
 *
 *   1. the device's CPUs = <sysfs>/bus/pci/devices/<bdf>/local_cpulist, intersected with the allowed set;
 *   2. devices that name the SAME CPUs share them evenly, by whole physical cores (SMT siblings stay together:
 *      <sysfs>/devices/system/cpu/cpuN/topology/thread_siblings_list), in device order;
 *   3. a device sysfs says nothing about (no file, empty list, nothing allowed) takes its even share of ALL allowed CPUs.
 *
 * Applied with plain sched_setaffinity on the calling thread -- threads created afterwards inherit it -- no re-exec.
 * Header-only (static functions) because both libswimm_host.so (C, gcc: bench.py ranks, the `swimm` program, the CPU test
 * with a fake sysfs tree) and libswimm_hip.so (C++, hipcc: swimm_hip_bind_host_thread, swimm_hip_search_chunks) compile it.
 */
#ifndef SWIMM_AFFINITY_H_INCLUDED
#define SWIMM_AFFINITY_H_INCLUDED

#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SWIMM_AFF_MAX_CPUS 4096

/* "0-3,8,10-11\n" -> sorted CPU numbers; returns how many (0 when the file is missing or empty) */
static int swimm_aff_read_list(const char *path, int *out, int cap)
{
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    char buf[8192];
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    int n = 0;
    const char *p = buf;
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\n' || *p == '\t') ++p;
        if (*p < '0' || *p > '9') break;
        char *end;
        long a = strtol(p, &end, 10), b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && n < cap; ++c)
            if (c >= 0 && c < SWIMM_AFF_MAX_CPUS) out[n++] = (int)c;
    }
    return n;
}

static int swimm_aff_contains(const int *v, int n, int x)
{
    for (int i = 0; i < n; ++i) if (v[i] == x) return 1;
    return 0;
}

/* the CPUs sysfs calls local to a PCI function, restricted to `allowed`; 0 = unknown */
static int swimm_aff_local(const char *sysfs_root, const char *bdf, const int *allowed, int n_allowed, int *out, int cap)
{
    if (!bdf || !*bdf) return 0;
    char path[512];
    snprintf(path, sizeof path, "%s/bus/pci/devices/%s/local_cpulist", sysfs_root, bdf);
    int raw[SWIMM_AFF_MAX_CPUS];
    const int nr = swimm_aff_read_list(path, raw, SWIMM_AFF_MAX_CPUS);
    int n = 0;
    for (int i = 0; i < nr && n < cap; ++i)
        if (swimm_aff_contains(allowed, n_allowed, raw[i])) out[n++] = raw[i];
    return n;
}

/* the lowest CPU number of cpu's physical core (its SMT siblings name the same one) */
static int swimm_aff_core_of(const char *sysfs_root, int cpu)
{
    char path[512];
    snprintf(path, sizeof path, "%s/devices/system/cpu/cpu%d/topology/thread_siblings_list", sysfs_root, cpu);
    int sib[64];
    const int n = swimm_aff_read_list(path, sib, 64);
    int lo = cpu;
    for (int i = 0; i < n; ++i) if (sib[i] < lo) lo = sib[i];
    return lo;
}

/* share k of s of the CPU set `set` (n entries), by whole physical cores -> out; returns the count (>= 1 when n >= 1) */
static int swimm_aff_share(const char *sysfs_root, const int *set, int n, int k, int s, int *out, int cap)
{
    if (n <= 0 || s <= 0) return 0;
    int core[SWIMM_AFF_MAX_CPUS], cores[SWIMM_AFF_MAX_CPUS], nc = 0;
    for (int i = 0; i < n; ++i) {
        core[i] = swimm_aff_core_of(sysfs_root, set[i]);
        if (!swimm_aff_contains(cores, nc, core[i])) cores[nc++] = core[i];
    }
    for (int i = 1; i < nc; ++i)          /* ascending (insertion sort: a few hundred entries at most) */
        for (int j = i; j > 0 && cores[j] < cores[j - 1]; --j) { const int t = cores[j]; cores[j] = cores[j - 1]; cores[j - 1] = t; }
    int c0 = (int)((long)nc * k / s), c1 = (int)((long)nc * (k + 1) / s);
    if (nc < s) { c0 = k % nc; c1 = c0 + 1; }        /* more sharers than cores: they double up, round-robin */
    int m = 0;
    for (int i = 0; i < n && m < cap; ++i)
        for (int c = c0; c < c1; ++c)
            if (core[i] == cores[c]) { out[m++] = set[i]; break; }
    return m;
}

/* The CPUs for the host threads of device `device` of `n_devices` (pci_bdf[d] = "0000:0c:00.0" or NULL / "" when unknown).
 * Returns the number of CPUs written to out_cpus (ascending within what sysfs lists), or -1 on bad arguments. */
__attribute__((unused)) static int swimm_affinity_plan_impl(const char *sysfs_root, const char *const *pci_bdf, int n_devices, int device, const int *allowed, int n_allowed,
                                    int *out_cpus, int cap)
{
    if (!sysfs_root || n_devices <= 0 || device < 0 || device >= n_devices || !allowed || n_allowed <= 0 || !out_cpus || cap <= 0) return -1;
    if (n_allowed > SWIMM_AFF_MAX_CPUS) n_allowed = SWIMM_AFF_MAX_CPUS;
    int mine[SWIMM_AFF_MAX_CPUS];
    const int nm = pci_bdf ? swimm_aff_local(sysfs_root, pci_bdf[device], allowed, n_allowed, mine, SWIMM_AFF_MAX_CPUS) : 0;
    if (nm == 0) return swimm_aff_share(sysfs_root, allowed, n_allowed, device, n_devices, out_cpus, cap);      /* rule 3 */
    /* rule 2: who else names exactly these CPUs? */
    int k = 0, s = 0;
    for (int d = 0; d < n_devices; ++d) {
        int other[SWIMM_AFF_MAX_CPUS];
        const int no = d == device ? nm : swimm_aff_local(sysfs_root, pci_bdf[d], allowed, n_allowed, other, SWIMM_AFF_MAX_CPUS);
        int same = no == nm;
        for (int i = 0; same && d != device && i < nm; ++i) same = other[i] == mine[i];
        if (!same) continue;
        if (d < device) ++k;
        ++s;
    }
    return swimm_aff_share(sysfs_root, mine, nm, k, s, out_cpus, cap);
}

/* the CPUs the calling thread may run on now */
__attribute__((unused)) static int swimm_affinity_allowed_impl(int *out, int cap)
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return -1;
    int n = 0;
    for (int c = 0; c < CPU_SETSIZE && n < cap; ++c) if (CPU_ISSET(c, &set)) out[n++] = c;
    return n;
}

/* binds the calling thread (threads it creates later inherit) */
__attribute__((unused)) static int swimm_affinity_apply_impl(const int *cpus, int n)
{
    if (!cpus || n <= 0) return -1;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int i = 0; i < n; ++i) if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE) CPU_SET(cpus[i], &set);
    return sched_setaffinity(0, sizeof set, &set);
}

/* "0-3,8" form of a sorted CPU list, for logs and records */
__attribute__((unused)) static void swimm_affinity_format(const int *cpus, int n, char *buf, size_t len)
{
    size_t at = 0;
    if (len) buf[0] = 0;
    for (int i = 0; i < n;) {
        int j = i;
        while (j + 1 < n && cpus[j + 1] == cpus[j] + 1) ++j;
        char part[48];
        if (j > i) snprintf(part, sizeof part, "%s%d-%d", at ? "," : "", cpus[i], cpus[j]);
        else snprintf(part, sizeof part, "%s%d", at ? "," : "", cpus[i]);
        const size_t l = strlen(part);
        if (at + l + 1 > len) break;
        memcpy(buf + at, part, l + 1);
        at += l;
        i = j + 1;
    }
}

#endif /* SWIMM_AFFINITY_H_INCLUDED */
