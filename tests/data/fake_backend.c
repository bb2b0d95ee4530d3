/* fake_backend.c -- TEST INFRASTRUCTURE, never shipped, never loaded by the product unless a test points SWIMM_HIP_LIB at it.
 *
 * A stand-in for libswimm_hip.so that exports the entry points the `swimm` program binds (swimm_amd/csrc/cli/hip_loader.c)
 * and answers them on the host CPU (through libswimm_host.so's mode-0 search), at an INJECTED rate: it lets the CPU-only
 * test suite drive the host-side logic of mode 2 -- the two-ended work queue of HETsearch.c:57,96-104 -- with a "device"
 * of any speed and start-up delay, deterministically and without a GPU.  FAKE_GPU_GCUPS (default 2) = the rate every
 * search is throttled to, FAKE_GPU_CREATE_MS (default 0) = what creating a context costs. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "swimm_host.h"
#include "swimm_hip.h"

struct swimm_hip_ctx {
    char *a; uint16_t *m; uint32_t *disp; uint32_t nq; uint64_t Q; char submat[768]; int go, ge;
    const uint16_t *lengths; const char *codes; uint64_t n_seq, first_seq; int have_db;
    double kernel_ms;
};

static __thread char g_err[256];
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return 1; }
static double env_d(const char *k, double dflt) { const char *v = getenv(k); return v ? atof(v) : dflt; }
static void sleep_s(double s) { if (s > 0) { struct timespec ts = {(time_t)s, (long)((s - (time_t)s) * 1e9)}; nanosleep(&ts, NULL); } }

int swimm_hip_abi_version(void) { return SWIMM_HIP_ABI_VERSION; }
const char *swimm_hip_last_error(void) { return g_err; }
int swimm_hip_device_count(void) { const char *v = getenv("FAKE_GPU_COUNT"); return v ? atoi(v) : 1; }

int swimm_hip_create(int device, swimm_hip_ctx **out)
{
    (void)device;
    sleep_s(env_d("FAKE_GPU_CREATE_MS", 0) * 1e-3);
    *out = (swimm_hip_ctx *)calloc(1, sizeof(swimm_hip_ctx));
    return *out ? 0 : fail("fake backend: out of memory");
}

void swimm_hip_destroy(swimm_hip_ctx *c) { if (c) { free(c->a); free(c->m); free(c->disp); free(c); } }

int swimm_hip_set_queries(swimm_hip_ctx *c, const char *a, const uint16_t *m, const uint32_t *a_disp, uint32_t nq, const char *submat, int go, int ge)
{
    uint64_t total = 0;
    c->Q = 0;
    for (uint32_t q = 0; q < nq; ++q) { if (a_disp[q] + (uint64_t)m[q] > total) total = a_disp[q] + (uint64_t)m[q]; c->Q += m[q]; }
    c->a = (char *)malloc(total ? total : 1); memcpy(c->a, a, total);
    c->m = (uint16_t *)malloc(nq * sizeof(uint16_t)); memcpy(c->m, m, nq * sizeof(uint16_t));
    c->disp = (uint32_t *)malloc((nq + 1) * sizeof(uint32_t)); memcpy(c->disp, a_disp, nq * sizeof(uint32_t));
    c->nq = nq; memcpy(c->submat, submat, 768); c->go = go; c->ge = ge;
    return 0;
}

int swimm_hip_add_chunk(swimm_hip_ctx *c, const char *b, uint64_t vD, const uint16_t *n, const uint32_t *d, uint32_t gc, uint32_t vl, uint64_t fg)
{
    (void)c; (void)b; (void)vD; (void)n; (void)d; (void)gc; (void)vl; (void)fg;
    return fail("fake backend: add_chunk is not part of the stand-in");
}

int swimm_hip_add_sequences(swimm_hip_ctx *c, const uint16_t *lengths, const char *codes, uint64_t n_seq, uint64_t first_seq)
{
    if (c->have_db) return fail("fake backend: one slab per database");
    c->lengths = lengths; c->codes = codes; c->n_seq = n_seq; c->first_seq = first_seq; c->have_db = 1;
    return 0;
}

int swimm_hip_clear_db(swimm_hip_ctx *c) { c->have_db = 0; return 0; }

int swimm_hip_search(swimm_hip_ctx *c, int32_t *s, uint64_t stride, double *wt) { (void)c; (void)s; (void)stride; (void)wt; return fail("fake backend: search is not part of the stand-in"); }

int swimm_hip_search_topr(swimm_hip_ctx *c, uint32_t r, uint64_t n_valid, int32_t *top_scores, int64_t *top_index, double *work_time)
{
    if (!c->have_db || !c->nq) return fail("fake backend: nothing to search");
    const double t0 = swimm_wtime();
    swimm_single_chunk sc;
    if (swimm_assemble_single_chunk(c->lengths, c->codes, c->n_seq, 32, 60, &sc)) return fail(swimm_host_last_error());
    const uint64_t stride = sc.vc * 32;
    int32_t *scores = (int32_t *)malloc(c->nq * stride * sizeof(int32_t));
    double wt = 0;
    int rc = swimm_cpu_search(c->a, c->m, c->nq, c->disp, sc.b, sc.n, sc.vc, sc.disp, c->submat, c->go, c->ge, 2, 60, 32, scores, &wt);
    swimm_single_chunk_free(&sc);
    if (rc) { free(scores); return fail(swimm_host_last_error()); }
    uint64_t keep = c->n_seq, residues = 0;
    if (c->first_seq + keep > n_valid) keep = n_valid > c->first_seq ? n_valid - c->first_seq : 0;
    for (uint64_t i = 0; i < c->n_seq; ++i) residues += c->lengths[i];
    for (uint32_t q = 0; q < c->nq; ++q) {
        swimm_topr(scores + q * stride, keep, r, top_scores + (size_t)q * r, top_index + (size_t)q * r);
        for (uint32_t k = 0; k < r; ++k) if (top_index[(size_t)q * r + k] >= 0) top_index[(size_t)q * r + k] += (int64_t)c->first_seq;
    }
    free(scores);
    /* the injected device rate: this search lasts cells / rate (plus a fixed cost per search, like a launch) */
    const double want = (double)residues * (double)c->Q / (env_d("FAKE_GPU_GCUPS", 2.0) * 1e9) + env_d("FAKE_GPU_SEARCH_MS", 0.2) * 1e-3;
    sleep_s(want - (swimm_wtime() - t0));
    c->kernel_ms = (swimm_wtime() - t0) * 1e3;
    if (work_time) *work_time = swimm_wtime() - t0;
    return 0;
}

int swimm_hip_last_stats(swimm_hip_ctx *c, double *kernel_ms, uint64_t *cells, uint64_t *promoted, uint32_t *launches)
{
    if (kernel_ms) *kernel_ms = c->kernel_ms;
    if (cells) *cells = 0;
    if (promoted) *promoted = 0;
    if (launches) *launches = 1;
    return 0;
}

int swimm_hip_last_plan(swimm_hip_ctx *c, uint32_t q, int *t, int *w, int *p) { (void)c; (void)q; if (t) *t = 0; if (w) *w = 0; if (p) *p = 0; return 0; }
int swimm_hip_set_option(swimm_hip_ctx *c, const char *key, int value) { (void)c; (void)key; (void)value; return 0; }
int swimm_hip_bind_host_thread(int device, int num_devices, char *cpulist_out, size_t len) { (void)device; (void)num_devices; if (cpulist_out && len) cpulist_out[0] = 0; return 0; }
