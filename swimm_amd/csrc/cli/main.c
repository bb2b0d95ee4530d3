/* main.c -- the `swimm` program: `-S preprocess` and `-S search` (reference: swimm.c:9-207).
 *
 * Same command line, same preprocessed-database files, same report text; the search itself runs on
 * MI355X GPUs through libswimm_hip.so (dlopen, include/swimm_hip.h) in the default mode 1, or on the
 * host CPU when mode 0 is asked for explicitly.  Structure of the GPU path follows
 * mic_search_knc_ap_multiple_chunks (MICsearch.c:53-346): one host thread per device, each device
 * gets its chunks, queries and matrix are replicated, results are merged on the host -- except that
 * the chunk-to-device assignment is static (longest-first onto the least-loaded GPU) and only the
 * top-r rows per query come back instead of every score.  Mode 2 (host + GPUs) is the reference's dynamic scheme
 * (HETsearch.c:57,96-104): one work queue, the host team at its short end and the GPU workers at its long end.
 */
#define _GNU_SOURCE
#include <omp.h>
#include <pthread.h>

#define SWIMM_MAX_WARM 64      /* contexts the start-up thread makes ahead of the search (one per device) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../host/swimm_host.h"
#include "hip_loader.h"
#include "options.h"

static void die_host(int status)
{
    printf("%s\n", swimm_host_last_error());
    /* exit codes of the reference: 1 memory, 2 files, 3 description file (sequences.c:16-19,54,748-751) */
    exit(status == SWIMM_E_NOMEM ? 1 : status == SWIMM_E_FILE ? 2 : status == SWIMM_E_DESC ? 3 : 4);
}

static int do_preprocess(const swimm_options *o)
{
    const double tick = swimm_wtime();
    uint64_t n = 0, d = 0;
    int rc = swimm_preprocess_db(o->input_filename, o->output_filename, &n, &d);
    if (rc) die_host(rc);
    printf("\nSWIMM v%s\n\n", SWIMM_VERSION);
    printf("Database file:\t\t\t %s\n", o->input_filename);
    printf("Database size:\t\t\t%ld sequences (%ld residues) \n", (long)n, (long)d);
    printf("Preprocessed database name:\t%s\n", o->output_filename);
    printf("Preprocessing time:\t\t%lf seconds\n\n", swimm_wtime() - tick);
    if (getenv("SWIMM_DEBUG")) {          /* peak resident memory of this process image (VmHWM), for tools/preprocess_scale.py */
        FILE *st = fopen("/proc/self/status", "r");
        char line[256];
        while (st && fgets(line, sizeof line, st))
            if (strncmp(line, "VmHWM:", 6) == 0 || strncmp(line, "RssAnon:", 8) == 0) fprintf(stderr, "swimm: %s", line);
        if (st) fclose(st);
    }
    return 0;
}

/* A slab = a run of consecutive sequences of the sorted database that goes to one GPU in one piece: the
 * counterpart of the reference's chunks (sequences.c:533-557: greedy, a chunk may exceed max_chunk_size by one lane
 * group), cut at multiples of 128 sequences.  The residues travel as the .seq file stores them; the device builds
 * its own layout (swimm_hip_add_sequences), so no host-side interleave is needed for the GPU modes. */
typedef struct { uint64_t first, count, residues, offset; int owner; } slab_t;

static slab_t *make_slabs(const uint16_t *lengths, uint64_t count, uint64_t max_bytes, uint32_t *n_out)
{
    uint32_t cap = 16, n = 0;
    slab_t *v = (slab_t *)malloc(cap * sizeof(slab_t));
    uint64_t i = 0, offset = 0;
    if (max_bytes > 0xE0000000ull) max_bytes = 0xE0000000ull;       /* a slab stays below 4 GiB */
    while (i < count) {
        slab_t sl = {i, 0, 0, offset, 0};
        while (i < count && (sl.count == 0 || sl.residues <= max_bytes)) {
            const uint64_t e = i + 128 < count ? i + 128 : count;
            for (; i < e; ++i) sl.residues += lengths[i];
            sl.count = i - sl.first;
        }
        offset += sl.residues;
        if (n == cap) { cap *= 2; v = (slab_t *)realloc(v, cap * sizeof(slab_t)); }
        v[n++] = sl;
    }
    *n_out = n;
    return v;
}

/* static shard: slabs longest-first onto the least-loaded GPU (cost = residues) */
static void shard_slabs(slab_t *v, uint32_t n, int gpus)
{
    uint64_t *load = (uint64_t *)calloc((size_t)gpus, sizeof(uint64_t));
    uint32_t *order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    for (uint32_t i = 1; i < n; ++i) {   /* insertion sort, stable, descending size */
        uint32_t x = order[i];
        int64_t j = (int64_t)i - 1;
        while (j >= 0 && v[order[j]].residues < v[x].residues) { order[j + 1] = order[j]; --j; }
        order[j + 1] = x;
    }
    for (uint32_t k = 0; k < n; ++k) {
        int best = 0;
        for (int g = 1; g < gpus; ++g) if (load[g] < load[best]) best = g;
        v[order[k]].owner = best;
        load[best] += v[order[k]].residues;
    }
    free(load);
    free(order);
}

typedef struct { double kernel_ms, seconds; uint64_t promoted; uint32_t chunk_count; } leg_stats;

/* -p / -u (swimm.c:81-85, MICsearch.c:39-43): queries of at least `threshold` rows are aligned with the SCORE profile, shorter
 * ones with the QUERY profile.  -p Q: no query; -p S: every query; -p A (the default): the reference takes the score profile
 * from -u rows on because it wins there on its hardware; on gfx950 it is the slower technique at every query length (a
 * workgroup has too few query rows in flight to amortise the table), so the adaptive choice resolves to
 * the query profile for every query and -u only names where the switch WOULD be considered. */
static int sp_threshold_of(const swimm_options *o) { return o->profile == 'S' ? 0 : 65536; }

/* Host leg (mode 0, and the host's share in mode 2): `count` sequences starting at the short end of the sorted
 * database; per query the first `top` rows of the listing, indices = sorted-database indices.  The lane-interleaved
 * copy of the database is built by the caller beforehand (untimed, like swimm.c:46 before the search call). */
static void cpu_leg(const swimm_options *o, const swimm_queries *q, const char *submat, const swimm_single_chunk *sc,
                    uint64_t count, unsigned long top, int32_t *out_s, int64_t *out_i, leg_stats *st)
{
    const uint64_t stride = sc->vc * (uint64_t)o->vector_length;
    int32_t *scores = (int32_t *)malloc(q->count * stride * sizeof(int32_t));
    if (!scores) { printf("SWIMM: An error occurred while allocating memory.\n"); exit(1); }
    int rc = swimm_cpu_search(q->a, q->m, q->count, q->disp, sc->b, sc->n, sc->vc, sc->disp, submat, o->open_gap, o->extend_gap,
                              o->cpu_threads, o->cpu_block_size, o->vector_length, scores, &st->seconds);
    if (rc) die_host(rc);
    for (uint64_t i = 0; i < q->count; ++i)   /* sort_scores + first `top` rows, swimm.c:151-160 */
        swimm_topr(scores + i * stride, count, (uint32_t)top, out_s + i * top, out_i + i * top);
    free(scores);
}

/* GPU leg: `count` sequences of the sorted database starting at sequence `first` (lengths / codes point at them),
 * cut into slabs and dealt to o->num_gpus devices; per device and query the first `top` rows, lists laid out
 * [device][query][top], indices global. */
static void gpu_leg(const swimm_hip_api *api, const swimm_options *o, const swimm_queries *q, const char *submat,
                    const uint16_t *lengths, const char *codes, uint64_t count, uint64_t first, unsigned long top,
                    int32_t *part_s, int64_t *part_i, leg_stats *st, swimm_hip_ctx **keep_ctx, swimm_hip_ctx **pre_ctx)
{
    const double tick = swimm_wtime();
    const int G = o->num_gpus;
    uint32_t n_slabs = 0;
    slab_t *slabs = make_slabs(lengths, count, (uint64_t)o->max_chunk_size, &n_slabs);
    shard_slabs(slabs, n_slabs, G);
    st->chunk_count = n_slabs;
    char (*gerr)[512] = calloc((size_t)G, 512);
    double *g_kms = (double *)calloc((size_t)G, sizeof(double));
    uint64_t *g_prom = (uint64_t *)calloc((size_t)G, sizeof(uint64_t));
    int *g_done = (int *)calloc((size_t)G, sizeof(int));
    /* one iteration per device (one host thread each when the runtime grants them, MICsearch.c:53; fewer threads only
     * serialise devices, they cannot drop one) */
#pragma omp parallel for num_threads(G) schedule(static, 1)
    for (int g = 0; g < G; ++g) {
        swimm_hip_ctx *ctx = NULL;
        int mine = 0;
        for (uint32_t c = 0; c < n_slabs; ++c) mine += slabs[c].owner == g;
        int bad = 0;
        if (mine == 0) bad = -1;   /* nothing to do on this device */
        const double t0 = swimm_wtime();
        /* this device's host thread, and the uploader thread its context starts, on the CPUs local to the device (best effort) */
        if (!bad) (void)api->bind_host_thread(g, G, NULL, 0);
        if (!bad && pre_ctx && g < SWIMM_MAX_WARM && pre_ctx[g]) { ctx = pre_ctx[g]; pre_ctx[g] = NULL; }      /* made while the database was read */
        else if (!bad && api->create(g, &ctx)) bad = 1;
        const double t1 = swimm_wtime();
        /* the preprocessed database stays in host memory for the whole run: slabs stream in while the search runs
         * (transfer overlapped with compute, MICsearch.c:85-91) */
        if (!bad && api->set_option(ctx, "lazy_upload", 1)) bad = 1;
        if (!bad && api->set_option(ctx, "sp_threshold", sp_threshold_of(o))) bad = 1;
        if (!bad && api->set_queries(ctx, q->a, q->m, q->disp, (uint32_t)q->count, submat, o->open_gap, o->extend_gap)) bad = 1;
        for (uint32_t c = 0; !bad && c < n_slabs; ++c)
            if (slabs[c].owner == g && api->add_sequences(ctx, lengths + slabs[c].first, codes + slabs[c].offset, slabs[c].count,
                                                          slabs[c].first)) bad = 1;
        const double t2 = swimm_wtime();
        int32_t *ps = part_s + (size_t)g * q->count * top;
        int64_t *pi = part_i + (size_t)g * q->count * top;
        if (!bad && api->search_topr(ctx, (uint32_t)top, count, ps, pi, NULL)) bad = 1;
        if (getenv("SWIMM_DEBUG"))
            fprintf(stderr, "swimm: GPU %d: context %.3f s, upload %.3f s, search %.3f s\n", g, t1 - t0, t2 - t1, swimm_wtime() - t2);
        if (bad > 0) snprintf(gerr[g], 512, "%s", api->last_error());
        if (!bad) {
            api->last_stats(ctx, &g_kms[g], NULL, &g_prom[g], NULL);
            for (size_t i = 0; i < q->count * top; ++i) if (pi[i] >= 0) pi[i] += (int64_t)first;
        }
        /* (tearing a context down returns tens of GB to the driver -- about a second for the 7e9-residue database: the caller does
         * it after the search's clock has stopped) */
        if (ctx && keep_ctx) keep_ctx[g] = ctx;
        else if (ctx) api->destroy(ctx);
        g_done[g] = 1;
    }
    for (int g = 0; g < G; ++g) if (gerr[g][0]) { printf("SWIMM: GPU %d: %s\n", g, gerr[g]); exit(5); }
    for (int g = 0; g < G; ++g) if (!g_done[g]) { printf("SWIMM: GPU %d: its share of the database was not searched.\n", g); exit(5); }
    free(g_done);
    for (int g = 0; g < G; ++g) { if (g_kms[g] > st->kernel_ms) st->kernel_ms = g_kms[g]; st->promoted += g_prom[g]; }
    free(gerr); free(g_kms); free(g_prom); free(slabs);
    st->seconds = swimm_wtime() - tick;
}

/* ---- Mode 2: host and devices pull work from ONE queue (the reference's het_search_*, HETsearch.c:57 `num_threads(num_mics+1)`,
 * :96-104 `omp for schedule(dynamic)` over the chunks, :337-342 the per-side chunk counts).
 *
 * The queue is the length-sorted database itself with a cursor at either end: the host team claims blocks of whole
 * 128-sequence groups from the SHORT end (where a CPU core is at its best and a GPU wave at its worst), the GPU workers
 * claim slabs from the LONG end, and the search is over when the two cursors meet -- no probe, no split fixed in
 * advance: whichever side is faster simply ends up with more of the database, and both legs end within one grain of
 * each other.  Grains: a GPU slab is min(-k, half the worker's fair share of what is left), never below a few
 * milliseconds of GPU work, so the slabs shrink towards the end; a host block is sized for about 4 ms of the team's measured rate.
 * Every device is served by two workers (two contexts, two host threads), so that one slab's copy and tiling overlap
 * the other's alignment (the transfer the reference leaves synchronous, MICsearch.c:85-91). */
typedef struct {
    pthread_mutex_t mu;
    const uint16_t *lengths;
    uint64_t lo, hi;              /* unclaimed sequences: [lo, hi) of the sorted database; lo is a multiple of 128 */
    uint64_t res_lo, res_hi;      /* residue offsets of lo and hi in the .seq codes */
    uint64_t host_cap, gpu_floor; /* test hook (a fixed host share): the host stops at host_cap, the GPUs at gpu_floor */
    uint64_t left_res;            /* residues still unclaimed */
    double host_rate;             /* residues per second the host team has been seen to do (0 = not yet known) ... */
    double gpu_rate[64];          /* ... and every GPU worker (slab after slab; 0 = not yet known) */
    int workers;
} work_queue;

typedef struct { uint64_t first, count, offset, residues; } claim_t;

static int claim_host(work_queue *wq, uint64_t want_residues, claim_t *c)
{
    pthread_mutex_lock(&wq->mu);
    const uint64_t end = wq->hi < wq->host_cap ? wq->hi : wq->host_cap;
    int got = 0;
    if (wq->lo < end) {
        c->first = wq->lo; c->offset = wq->res_lo; c->residues = 0;
        uint64_t i = wq->lo;
        do {
            const uint64_t e = i + 128 < end ? i + 128 : end;
            for (; i < e; ++i) c->residues += wq->lengths[i];
        } while (i < end && c->residues < want_residues);
        c->count = i - wq->lo;
        wq->lo = i; wq->res_lo += c->residues; wq->left_res -= c->residues;
        got = 1;
    }
    pthread_mutex_unlock(&wq->mu);
    return got;
}

/* Guided self-scheduling between workers of very different speed: a worker's slab is HALF its share of what is left, the
 * shares in proportion to the rates measured so far (its own last slabs, the other workers', the host team's).  A worker
 * whose rate is not known yet takes the smallest slab -- a few milliseconds of GPU work -- and is measured on it. */
static int claim_gpu(work_queue *wq, uint64_t max_residues, uint64_t min_residues, int me, double my_last_rate, claim_t *c)
{
    pthread_mutex_lock(&wq->mu);
    const uint64_t floor_ = wq->lo > wq->gpu_floor ? wq->lo : wq->gpu_floor;
    int got = 0;
    if (my_last_rate > 0) wq->gpu_rate[me] = wq->gpu_rate[me] > 0 ? 0.5 * (wq->gpu_rate[me] + my_last_rate) : my_last_rate;
    if (wq->hi > floor_) {
        double total = wq->host_cap > wq->lo ? wq->host_rate : 0;
        for (int i = 0; i < wq->workers; ++i) total += wq->gpu_rate[i] > 0 ? wq->gpu_rate[i] : wq->gpu_rate[me];   /* (unknown: like me) */
        uint64_t want = wq->gpu_rate[me] > 0 ? (uint64_t)(0.5 * (double)wq->left_res * wq->gpu_rate[me] / total) : 0;
        if (want > max_residues) want = max_residues;
        if (want < min_residues) want = min_residues;
        uint64_t b = wq->hi, res = 0;
        do {                                                          /* slab boundaries are multiples of 128 sequences */
            uint64_t nb = (b - 1) / 128 * 128;
            if (nb < floor_) nb = floor_;
            for (uint64_t i = nb; i < b; ++i) res += wq->lengths[i];
            b = nb;
        } while (b > floor_ && res < want && res < 0xE0000000ull);
        c->first = b; c->count = wq->hi - b; c->residues = res; c->offset = wq->res_hi - res;
        wq->hi = b; wq->res_hi -= res; wq->left_res -= res;
        got = 1;
    }
    pthread_mutex_unlock(&wq->mu);
    return got;
}

/* running top-r list of one worker: new = merge(old, part) */
static void fold_lists(int32_t *run_s, int64_t *run_i, const int32_t *part_s, const int64_t *part_i, uint64_t queries, unsigned long top)
{
    int32_t *ls = (int32_t *)malloc(2 * top * sizeof(int32_t)), *os = (int32_t *)malloc(top * sizeof(int32_t));
    int64_t *li = (int64_t *)malloc(2 * top * sizeof(int64_t)), *oi = (int64_t *)malloc(top * sizeof(int64_t));
    for (uint64_t q = 0; q < queries; ++q) {
        memcpy(ls, run_s + q * top, top * sizeof(int32_t)); memcpy(ls + top, part_s + q * top, top * sizeof(int32_t));
        memcpy(li, run_i + q * top, top * sizeof(int64_t)); memcpy(li + top, part_i + q * top, top * sizeof(int64_t));
        swimm_topr_merge(ls, li, 2, (uint32_t)top, os, oi);
        memcpy(run_s + q * top, os, top * sizeof(int32_t)); memcpy(run_i + q * top, oi, top * sizeof(int64_t));
    }
    free(ls); free(os); free(li); free(oi);
}

typedef struct {
    const swimm_hip_api *api; const swimm_options *o; const swimm_queries *q; const char *submat; const swimm_db *db;
    work_queue *wq; int device, index; unsigned long top; double tick; uint64_t min_slab;
    int32_t *run_s; int64_t *run_i;          /* [query][top], global indices */
    uint64_t sequences, claims, promoted; double kernel_ms, end_s; char err[512];
} gpu_worker;

static void *gpu_worker_main(void *p)
{
    gpu_worker *w = (gpu_worker *)p;
    const swimm_hip_api *api = w->api;
    const swimm_queries *q = w->q;
    swimm_hip_ctx *ctx = NULL;
    int32_t *ps = (int32_t *)malloc(q->count * w->top * sizeof(int32_t));
    int64_t *pi = (int64_t *)malloc(q->count * w->top * sizeof(int64_t));
    /* (no CPU binding here, unlike mode 1: the host leg's OpenMP team runs on every core the process has, and a device worker
     * -- its uploader thread copies out of pageable memory -- is better off wherever the kernel finds it a free hardware thread
     * than pinned among spinning OpenMP threads: measured, the GPU leg of the 400 000-sequence test ended 28 % after the host's) */
    int bad = api->create(w->device, &ctx);
    if (!bad) bad = api->set_option(ctx, "lazy_upload", 1);     /* the slab streams in piece by piece while the search runs */
    if (!bad) bad = api->set_option(ctx, "sp_threshold", sp_threshold_of(w->o));
    if (!bad) bad = api->set_queries(ctx, q->a, q->m, q->disp, (uint32_t)q->count, w->submat, w->o->open_gap, w->o->extend_gap);
    claim_t c;
    double rate = 0;
    while (!bad && claim_gpu(w->wq, (uint64_t)w->o->max_chunk_size, w->min_slab, w->index, rate, &c)) {
        const double t0 = swimm_wtime();
        if (w->claims && api->clear_db(ctx)) { bad = 1; break; }
        if (api->add_sequences(ctx, w->db->lengths + c.first, w->db->codes + c.offset, c.count, c.first)) { bad = 1; break; }
        if (api->search_topr(ctx, (uint32_t)w->top, w->db->count, ps, pi, NULL)) { bad = 1; break; }
        fold_lists(w->run_s, w->run_i, ps, pi, q->count, w->top);
        double kms = 0; uint64_t prom = 0;
        api->last_stats(ctx, &kms, NULL, &prom, NULL);
        w->kernel_ms += kms; w->promoted += prom;
        w->sequences += c.count; w->claims++;
        w->end_s = swimm_wtime() - w->tick;
        rate = (double)c.residues / (swimm_wtime() - t0 > 1e-6 ? swimm_wtime() - t0 : 1e-6);
        if (getenv("SWIMM_DEBUG"))
            fprintf(stderr, "swimm: GPU %d: slab of %llu sequences from %llu (%.1f MB) done at %.4f s\n", w->device, (unsigned long long)c.count,
                    (unsigned long long)c.first, c.residues / 1e6, w->end_s);
    }
    if (bad) snprintf(w->err, sizeof w->err, "%s", api->last_error());
    if (ctx) api->destroy(ctx);
    free(ps); free(pi);
    return NULL;
}

typedef struct { uint64_t n_cpu, n_gpu, gpu_claims, cpu_claims; } hybrid_counts;

static void hybrid_search(const swimm_hip_api *api, const swimm_options *o, const swimm_queries *q, const char *submat, const swimm_db *db,
                          unsigned long top, int32_t *top_scores, int64_t *top_idx, leg_stats *gst, leg_stats *cst, hybrid_counts *hc)
{
    const int G = o->num_gpus;
    work_queue wq;
    pthread_mutex_init(&wq.mu, NULL);
    wq.lengths = db->lengths; wq.lo = 0; wq.hi = db->count; wq.res_lo = 0; wq.res_hi = db->residues; wq.left_res = db->residues;
    wq.host_cap = db->count; wq.gpu_floor = 0; wq.host_rate = 0;
    memset(wq.gpu_rate, 0, sizeof wq.gpu_rate);
    const char *forced = getenv("SWIMM_HYBRID_CPU_SEQUENCES");   /* test hook: a fixed host share (the first n sequences) */
    if (forced) {
        uint64_t n = strtoull(forced, NULL, 10) / 128 * 128;
        if (n >= db->count) n = 0;
        wq.host_cap = wq.gpu_floor = n;
    }
    /* the smallest slab: what a GPU aligns in a few milliseconds (2e10 cells), so that a slab's fixed costs -- buffers, the
     * first piece's copy, launches, the top-r pass -- stay a small part of it */
    uint64_t min_slab = 20000000000ull / (q->Q ? q->Q : 1);
    if (min_slab < (256u << 10)) min_slab = 256u << 10;
    if (min_slab > (64u << 20)) min_slab = 64u << 20;
    if (getenv("SWIMM_HYBRID_MIN_SLAB")) min_slab = strtoull(getenv("SWIMM_HYBRID_MIN_SLAB"), NULL, 10);   /* test hook */
    /* two workers per device when the database is more than a few slabs' worth (else one: a second context only costs) */
    const int per_dev = db->residues > 4 * min_slab ? 2 : 1;
    const int NW = G * per_dev > 64 ? 64 : G * per_dev;
    wq.workers = NW;
    gpu_worker *gw = (gpu_worker *)calloc((size_t)NW, sizeof(gpu_worker));
    pthread_t *th = (pthread_t *)malloc((size_t)NW * sizeof(pthread_t));
    const size_t list = q->count * top;
    const double tick = swimm_wtime();
    for (int i = 0; i < NW; ++i) {
        gpu_worker *w = &gw[i];
        w->api = api; w->o = o; w->q = q; w->submat = submat; w->db = db; w->wq = &wq; w->device = i % G; w->index = i; w->top = top; w->tick = tick; w->min_slab = min_slab;
        w->run_s = (int32_t *)malloc(list * sizeof(int32_t)); w->run_i = (int64_t *)malloc(list * sizeof(int64_t));
        for (size_t k = 0; k < list; ++k) { w->run_s[k] = -1; w->run_i[k] = -1; }
        if (pthread_create(&th[i], NULL, gpu_worker_main, w)) { printf("SWIMM: cannot start a GPU thread.\n"); exit(1); }
    }
    /* the host leg on this thread, with the ordinary (top-level, warm) OpenMP team */
    int32_t *hs = (int32_t *)malloc(list * sizeof(int32_t)), *bs = (int32_t *)malloc(list * sizeof(int32_t));
    int64_t *hi_ = (int64_t *)malloc(list * sizeof(int64_t)), *bi = (int64_t *)malloc(list * sizeof(int64_t));
    for (size_t k = 0; k < list; ++k) { hs[k] = -1; hi_[k] = -1; }
    double host_rate = 0;                            /* cells per second, measured block by block */
    claim_t c;
    uint64_t want = 200000000ull / (q->Q ? q->Q : 1) + 1;        /* first block: 2e8 cells */
    while (claim_host(&wq, want, &c)) {
        const double t0 = swimm_wtime();
        swimm_single_chunk sc;
        int rc = swimm_assemble_single_chunk(db->lengths + c.first, db->codes + c.offset, c.count, o->vector_length, o->cpu_block_size, &sc);
        if (rc) die_host(rc);
        leg_stats st = {0, 0, 0, 0};
        cpu_leg(o, q, submat, &sc, c.count, top, bs, bi, &st);
        swimm_single_chunk_free(&sc);
        for (size_t k = 0; k < list; ++k) if (bi[k] >= 0) bi[k] += (int64_t)c.first;
        fold_lists(hs, hi_, bs, bi, q->count, top);
        const double dt = swimm_wtime() - t0;
        const double rate = (double)c.residues * (double)q->Q / (dt > 1e-6 ? dt : 1e-6);
        host_rate = host_rate > 0 ? 0.5 * (host_rate + rate) : rate;
        pthread_mutex_lock(&wq.mu);
        wq.host_rate = host_rate / (double)(q->Q ? q->Q : 1);
        pthread_mutex_unlock(&wq.mu);
        want = (uint64_t)(host_rate * 0.004 / (double)(q->Q ? q->Q : 1)) + 1;      /* about 4 ms of the team's time */
        hc->n_cpu += c.count; hc->cpu_claims++;
        cst->seconds = swimm_wtime() - tick;
    }
    for (int i = 0; i < NW; ++i) pthread_join(th[i], NULL);
    for (int i = 0; i < NW; ++i) if (gw[i].err[0]) { printf("SWIMM: GPU %d: %s\n", gw[i].device, gw[i].err); exit(5); }
    if (wq.lo < wq.hi) { printf("SWIMM: %llu sequences were not searched.\n", (unsigned long long)(wq.hi - wq.lo)); exit(5); }
    /* host k-way merge of the workers' lists and the host's */
    const int lists = NW + 1;
    int32_t *ls = (int32_t *)malloc((size_t)lists * top * sizeof(int32_t));
    int64_t *li = (int64_t *)malloc((size_t)lists * top * sizeof(int64_t));
    for (uint64_t k = 0; k < q->count; ++k) {
        for (int g = 0; g < NW; ++g) {
            memcpy(ls + (size_t)g * top, gw[g].run_s + k * top, top * sizeof(int32_t));
            memcpy(li + (size_t)g * top, gw[g].run_i + k * top, top * sizeof(int64_t));
        }
        memcpy(ls + (size_t)NW * top, hs + k * top, top * sizeof(int32_t));
        memcpy(li + (size_t)NW * top, hi_ + k * top, top * sizeof(int64_t));
        swimm_topr_merge(ls, li, (uint32_t)lists, (uint32_t)top, top_scores + k * top, top_idx + k * top);
    }
    for (int i = 0; i < NW; ++i) {
        gpu_worker *w = &gw[i];
        hc->n_gpu += w->sequences; hc->gpu_claims += w->claims; gst->promoted += w->promoted;
        if (w->end_s > gst->seconds) gst->seconds = w->end_s;
        free(w->run_s); free(w->run_i);
    }
    /* kernel time per device = the sum over its workers' slabs (they share the device) */
    for (int d = 0; d < G; ++d) {
        double kms = 0;
        for (int i = d; i < NW; i += G) kms += gw[i].kernel_ms;
        if (kms > gst->kernel_ms) gst->kernel_ms = kms;
    }
    gst->chunk_count = (uint32_t)hc->gpu_claims;
    if (getenv("SWIMM_DEBUG"))
        fprintf(stderr, "swimm: hybrid queue: host %llu sequences in %llu blocks (%.3f s, %.1f GCUPS at the end), GPUs %llu sequences in %llu slabs (%.3f s)\n",
                (unsigned long long)hc->n_cpu, (unsigned long long)hc->cpu_claims, cst->seconds, host_rate / 1e9, (unsigned long long)hc->n_gpu,
                (unsigned long long)hc->gpu_claims, gst->seconds);
    free(ls); free(li); free(hs); free(hi_); free(bs); free(bi); free(gw); free(th);
    pthread_mutex_destroy(&wq.mu);
}

/* The HIP runtime's start-up and a device's first context (0.2 s on one MI355X: the runtime, the code object, the streams) do not
 * depend on the database: a thread of its own gets them out of the way while the main thread reads the queries and the .seq file,
 * instead of the search's clock starting with them (a c2-sized search is 0.12 s).  The contexts the search uses are created by
 * its device threads as before -- in milliseconds now. */
typedef struct { swimm_hip_api api; char err[1024]; int loaded, devices, want; swimm_hip_ctx *ctx[SWIMM_MAX_WARM]; } gpu_warmup;
static void *gpu_warmup_main(void *arg)
{
    gpu_warmup *w = (gpu_warmup *)arg;
    w->devices = w->api.device_count();
    /* (a context is 40 ms of hardware queues even when the runtime is up: the ones made here are handed to the search, mode 1) */
    for (int g = 0; g < w->want && g < w->devices && g < SWIMM_MAX_WARM; ++g)
        if (w->api.create(g, &w->ctx[g]) != 0) w->ctx[g] = NULL;
    return NULL;
}

int main(int argc, char **argv)
{
    swimm_options o;
    swimm_parse_options(argc, argv, &o);
    if (strcmp(o.op, "preprocess") == 0) return do_preprocess(&o);

    time_t current_time = time(NULL);
    printf("\nSWIMM v%s \n\n", SWIMM_VERSION);
    printf("Database file:\t\t\t%s\n", o.db_prefix);

    const char *submat = swimm_submat(o.submat_name);
    const int gpu_mode = o.execution_mode != MODE_CPU_ONLY;
    if (o.cpu_block_size == 0) o.cpu_block_size = (o.vector_length == 32 ? 64 : 128) / SWIMM_SEQ_LEN_MULT * SWIMM_SEQ_LEN_MULT;   /* swimm.c:32-35 */

    /* mode 0 pads odd queries to even length (sequences.c:378-387); the accelerator mode does not (347-364) */
    gpu_warmup warm;
    pthread_t warm_thread;
    int warm_started = 0;
    memset(&warm, 0, sizeof warm);
    if (gpu_mode) {
        warm.loaded = swimm_hip_load(&warm.api, warm.err, sizeof warm.err) == 0;
        warm.want = o.num_gpus;
        if (warm.loaded) warm_started = pthread_create(&warm_thread, NULL, gpu_warmup_main, &warm) == 0;
    }
    swimm_queries q;
    const double t_load0 = swimm_wtime();
    int rc = swimm_queries_load(o.queries_filename, gpu_mode ? 0 : 1, &q);
    if (rc) die_host(rc);
    swimm_db db;
    if ((rc = swimm_db_load(o.db_prefix, &db))) die_host(rc);
    const double t_load = swimm_wtime() - t_load0;
    unsigned long top = db.count < o.top ? db.count : o.top;   /* swimm.c:51 */

    printf("Database size:\t\t\t%ld sequences (%ld residues) \n", (long)db.count, (long)db.residues);
    printf("Longest database sequence: \t%d residues\n", (int)db.lengths[db.count - 1]);
    printf("Substitution matrix:\t\t%s\n", swimm_submat_label(o.submat_name));
    printf("Gap open penalty:\t\t%d\n", o.open_gap);
    printf("Gap extend penalty:\t\t%d\n", o.extend_gap);
    printf("Query filename:\t\t\t%s\n", o.queries_filename);
    fflush(stdout);

    int32_t *top_scores = (int32_t *)malloc(q.count * top * sizeof(int32_t));
    int64_t *top_idx = (int64_t *)malloc(q.count * top * sizeof(int64_t));
    if (!top_scores || !top_idx) { printf("SWIMM: An error occurred while allocating memory.\n"); exit(1); }
    double workTime = 0;
    leg_stats gst = {0, 0, 0, 0}, cst = {0, 0, 0, 0};
    uint64_t n_cpu = 0;          /* sequences (from the short end of the sorted database) searched on the host */
    hybrid_counts hc = {0, 0, 0, 0};
    omp_set_num_threads(o.cpu_threads);

    if (o.execution_mode == MODE_CPU_ONLY) {
        swimm_single_chunk sc;
        if ((rc = swimm_assemble_single_chunk(db.lengths, db.codes, db.count, o.vector_length, o.cpu_block_size, &sc))) die_host(rc);
        cpu_leg(&o, &q, submat, &sc, db.count, top, top_scores, top_idx, &cst);
        swimm_single_chunk_free(&sc);
        workTime = cst.seconds;   /* the search call only, like CPUsearch.c:530,960 */
    } else {
        if (!warm.loaded) { printf("%s\n", warm.err); exit(5); }
        if (warm_started) pthread_join(warm_thread, NULL);
        swimm_hip_api api = warm.api;
        const int avail = warm_started ? warm.devices : api.device_count();
        if (avail <= 0) { printf("SWIMM: no MI355X visible: %s\n", api.last_error()); exit(5); }
        if (o.num_gpus > avail) { printf("SWIMM: %d GPUs requested, %d visible.\n", o.num_gpus, avail); exit(5); }
        const int G = o.num_gpus;
        if (o.execution_mode == MODE_HYBRID) {
            for (int g = 0; g < SWIMM_MAX_WARM; ++g) if (warm.ctx[g]) { api.destroy(warm.ctx[g]); warm.ctx[g] = NULL; }      /* (its workers make their own, two per device) */
            const double tick = swimm_wtime();   /* brackets context creation + transfers + kernels + host blocks + merge */
            hybrid_search(&api, &o, &q, submat, &db, top, top_scores, top_idx, &gst, &cst, &hc);
            workTime = swimm_wtime() - tick;
            n_cpu = hc.n_cpu;
        } else {
            int32_t *part_s = (int32_t *)malloc((size_t)G * q.count * top * sizeof(int32_t));
            int64_t *part_i = (int64_t *)malloc((size_t)G * q.count * top * sizeof(int64_t));
            if (!part_s || !part_i) { printf("SWIMM: An error occurred while allocating memory.\n"); exit(1); }
            for (size_t i = 0; i < (size_t)G * q.count * top; ++i) { part_s[i] = -1; part_i[i] = -1; }
            const double tick = swimm_wtime();   /* brackets transfers + kernels + merge, like MICsearch.c:51,350 */
            swimm_hip_ctx **ctxs = (swimm_hip_ctx **)calloc((size_t)G, sizeof *ctxs);
            gpu_leg(&api, &o, &q, submat, db.lengths, db.codes, db.count, 0, top, part_s, part_i, &gst, ctxs, warm.ctx);
            /* host k-way merge of the per-device lists ([device][query][top]) */
            int32_t *ls = (int32_t *)malloc((size_t)G * top * sizeof(int32_t));
            int64_t *li = (int64_t *)malloc((size_t)G * top * sizeof(int64_t));
            for (uint64_t i = 0; i < q.count; ++i) {
                for (int g = 0; g < G; ++g) {
                    memcpy(ls + (size_t)g * top, part_s + ((size_t)g * q.count + i) * top, top * sizeof(int32_t));
                    memcpy(li + (size_t)g * top, part_i + ((size_t)g * q.count + i) * top, top * sizeof(int64_t));
                }
                swimm_topr_merge(ls, li, (uint32_t)G, (uint32_t)top, top_scores + i * top, top_idx + i * top);
            }
            workTime = swimm_wtime() - tick;
            const double t_down = swimm_wtime();
            for (int g = 0; ctxs && g < G; ++g) if (ctxs[g]) api.destroy(ctxs[g]);
            for (int g = 0; g < SWIMM_MAX_WARM; ++g) if (warm.ctx[g]) { api.destroy(warm.ctx[g]); warm.ctx[g] = NULL; }      /* (devices that got no slab) */
            if (getenv("SWIMM_DEBUG")) fprintf(stderr, "swimm: contexts torn down in %.3f s\n", swimm_wtime() - t_down);
            free(ctxs);
            free(ls); free(li); free(part_s); free(part_i);
        }
    }

    /* titles of the reported hits only (the reference loads all N, sequences.c:757-761) */
    char **titles = (char **)malloc(q.count * top * sizeof(char *));
    const double t_titles0 = swimm_wtime();
    if ((rc = swimm_db_titles(o.db_prefix, db.count, top_idx, q.count * top, titles))) die_host(rc);
    if (getenv("SWIMM_DEBUG"))          /* the wall-clock split of a search, for tools/cli_scale.py */
        fprintf(stderr, "swimm: load %.4f s, search %.4f s, titles %.4f s (%lu titles)\n", t_load, workTime, swimm_wtime() - t_titles0,
                (unsigned long)(q.count * top));
    for (uint64_t i = 0; i < q.count; ++i) {
        printf("\nQuery no.\t\t\t%d\n", (int)i + 1);
        printf("Query description: \t\t%s\n", q.titles[i] + 1);
        printf("Query length:\t\t\t%d residues\n", q.lengths[i]);
        printf("\nScore\tSequence description\n");
        for (unsigned long j = 0; j < top; ++j) printf("%d\t%s\n", top_scores[i * top + j], titles[i * top + j]);
    }
    /* GCUPS as the reference prints it: Q (as stored: even-padded in mode 0, real in mode 1) x D / time, swimm.c:163 */
    printf("\nSearch date:\t\t\t%s", ctime(&current_time));
    printf("Search time:\t\t\t%lf seconds\n", workTime);
    printf("Search speed:\t\t\t%.2lf GCUPS\n", ((double)q.Q * (double)db.residues) / (workTime * 1000000000));
    if (o.execution_mode == MODE_CPU_ONLY) {
        printf("Execution mode:\t\t\tHost CPU only (%d threads, block width = %d)\n", o.cpu_threads, o.cpu_block_size);
        printf("Profile technique:\t\tSubstitution row per query residue\n");
        printf("Instruction set:\t\tAVX2 int8 -> int16 -> int32 ladder, one sequence per lane (vector length = %d)\n", o.vector_length);
    } else {
        if (o.execution_mode == MODE_HYBRID)
            printf("Execution mode:\t\t\tConcurrent host CPU and MI355X (%d CPU threads and %d GPUs)\n", o.cpu_threads, o.num_gpus);
        else
            printf("Execution mode:\t\t\tMI355X only (%d GPUs)\n", o.num_gpus);
        if (o.profile == 'S') printf("Profile technique:\t\tScore Profile in LDS\n");
        else if (o.profile == 'Q') printf("Profile technique:\t\tQuery Profile in LDS\n");
        else printf("Profile technique:\t\tAdaptive Profile (threshold = %d; on gfx950 the query profile is the faster one at every length: Query Profile in LDS)\n", o.query_length_threshold);
        printf("Instruction set:\t\tgfx950 packed binary16 -> int16 -> int32 ladder (vector length = 128)\n");
        printf("Max. chunk size:\t\t%ld bytes\n", o.max_chunk_size);
        printf("Chunk count:\t\t\t%ld \n", (long)gst.chunk_count);
        printf("Kernel time:\t\t\t%lf seconds\n", gst.kernel_ms / 1000.0);
        printf("Promoted to int32:\t\t%ld alignments\n", (long)gst.promoted);
        if (o.execution_mode == MODE_HYBRID) {   /* the reference prints "%d chunks in CPU and %d in MICs" (HETsearch.c:337-342) */
            printf("Host CPU share:\t\t\t%ld sequences (%.3lf seconds), MI355X %ld sequences (%.3lf seconds)\n", (long)n_cpu, cst.seconds,
                   (long)(db.count - n_cpu), gst.seconds);
            printf("Work queue:\t\t\t%ld blocks in CPU and %ld slabs in MI355X\n", (long)hc.cpu_claims, (long)hc.gpu_claims);
        }
    }
    for (uint64_t i = 0; i < q.count * top; ++i) free(titles[i]);
    free(titles); free(top_scores); free(top_idx);
    swimm_db_free(&db);
    swimm_queries_free(&q);
    return 0;
}
