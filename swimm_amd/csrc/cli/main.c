/* main.c -- the `swimm` program: `-S preprocess` and `-S search` (reference: swimm.c:9-207).
 *
 * Same command line, same preprocessed-database files, same report text; the search itself runs on
 * MI355X GPUs through libswimm_hip.so (dlopen, include/swimm_hip.h) in the default mode 1, or on the
 * host CPU when mode 0 is asked for explicitly.  Structure of the GPU path follows
 * mic_search_knc_ap_multiple_chunks (MICsearch.c:53-346): one host thread per device, each device
 * gets its chunks, queries and matrix are replicated, results are merged on the host -- except that
 * the chunk-to-device assignment is static (longest-first onto the least-loaded GPU) and only the
 * top-r rows per query come back instead of every score.
 */
#define _GNU_SOURCE
#include <omp.h>
#include <pthread.h>
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
                    int32_t *part_s, int64_t *part_i, leg_stats *st)
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
        if (!bad && api->create(g, &ctx)) bad = 1;
        const double t1 = swimm_wtime();
        /* the preprocessed database stays in host memory for the whole run: slabs stream in while the search runs
         * (transfer overlapped with compute, MICsearch.c:85-91) */
        if (!bad && api->set_option(ctx, "lazy_upload", 1)) bad = 1;
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
        if (ctx) api->destroy(ctx);
        g_done[g] = 1;
    }
    for (int g = 0; g < G; ++g) if (gerr[g][0]) { printf("SWIMM: GPU %d: %s\n", g, gerr[g]); exit(5); }
    for (int g = 0; g < G; ++g) if (!g_done[g]) { printf("SWIMM: GPU %d: its share of the database was not searched.\n", g); exit(5); }
    free(g_done);
    for (int g = 0; g < G; ++g) { if (g_kms[g] > st->kernel_ms) st->kernel_ms = g_kms[g]; st->promoted += g_prom[g]; }
    free(gerr); free(g_kms); free(g_prom); free(slabs);
    st->seconds = swimm_wtime() - tick;
}

typedef struct {
    const swimm_hip_api *api; const swimm_options *o; const swimm_queries *q; const char *submat;
    const uint16_t *lengths; const char *codes; uint64_t count, first; unsigned long top;
    int32_t *part_s; int64_t *part_i; leg_stats *st;
} gpu_leg_args;

static void *gpu_leg_thread(void *p)
{
    gpu_leg_args *a = (gpu_leg_args *)p;
    gpu_leg(a->api, a->o, a->q, a->submat, a->lengths, a->codes, a->count, a->first, a->top, a->part_s, a->part_i, a->st);
    return NULL;
}

/* Mode 2 (the reference's het_search_*, HETsearch.c:57,96-104: host and devices pull chunks from one queue): here
 * the split is fixed before the search, because the GPUs keep their share resident and search it in one go.  Both
 * rates are MEASURED before the clock starts: the host's on a sample of the shortest sequences with the shortest
 * query, the GPU's by a probe search of the real query batch against a sample from the long end of the database
 * on device 0 (context creation and upload are timed too).  The host gets the shortest sequences it can finish in
 * the time the GPUs need for the rest.  Returns the number of sequences for the host (0: not worth it). */
static uint64_t hybrid_split(const swimm_hip_api *api, const swimm_options *o, const swimm_queries *q, const char *submat, const swimm_db *db, int G)
{
    const uint64_t vl = (uint64_t)o->vector_length;
    const char *forced = getenv("SWIMM_HYBRID_CPU_SEQUENCES");   /* test hook: fixed host share */
    if (forced) {
        uint64_t n = strtoull(forced, NULL, 10) / 128 * 128;
        return n < db->count ? n : 0;
    }
    uint64_t sample = 0, sample_res = 0;
    while (sample < db->count && sample_res * q->Q < 2000000000ull) sample_res += db->lengths[sample++];
    sample = (sample + vl - 1) / vl * vl;
    if (sample == 0 || sample >= db->count) return 0;
    swimm_queries q0 = *q;                          /* the whole batch: the host's rate depends on the query length */
    leg_stats st = {0, 0, 0, 0};
    int32_t *s1 = (int32_t *)malloc(q->count * sizeof(int32_t));
    int64_t *i1 = (int64_t *)malloc(q->count * sizeof(int64_t));
    swimm_single_chunk sc;
    int rc = swimm_assemble_single_chunk(db->lengths, db->codes, sample, o->vector_length, o->cpu_block_size, &sc);
    if (rc) die_host(rc);
    cpu_leg(o, &q0, submat, &sc, sample, 1, s1, i1, &st);      /* wakes the thread team */
    st.seconds = 0;
    cpu_leg(o, &q0, submat, &sc, sample, 1, s1, i1, &st);
    swimm_single_chunk_free(&sc);
    sample_res = 0;
    for (uint64_t i = 0; i < sample; ++i) sample_res += db->lengths[i];
    free(s1); free(i1);
    const double host_rate = (double)sample_res * (double)q->Q / (st.seconds > 1e-6 ? st.seconds : 1e-6);   /* cells per second */

    /* GPU probe: the last sequences of the sorted database (the GPUs' end), at most 64 MB of residues */
    uint64_t pn = 0, pres = 0;
    while (pn < db->count - sample && pres < (64ull << 20)) pres += db->lengths[db->count - 1 - pn++];
    pn = pn / 128 * 128;
    if (pn == 0) return 0;
    pres = 0;
    for (uint64_t i = db->count - pn; i < db->count; ++i) pres += db->lengths[i];
    const double Qrows = (double)q->Q;
    double t_ctx = 0, t_up = 0, t_search = 0, t_first = 0;
    {
        swimm_hip_ctx *ctx = NULL;
        const double t0 = swimm_wtime();
        int bad = api->create(0, &ctx);
        t_ctx = swimm_wtime() - t0;
        if (!bad) bad = api->set_queries(ctx, q->a, q->m, q->disp, (uint32_t)q->count, submat, o->open_gap, o->extend_gap);
        const double t1 = swimm_wtime();
        if (!bad) bad = api->add_sequences(ctx, db->lengths + (db->count - pn), db->codes + (db->residues - pres), pn, 0);
        t_up = swimm_wtime() - t1;
        int32_t *ps = (int32_t *)malloc(q->count * sizeof(int32_t));
        int64_t *pi = (int64_t *)malloc(q->count * sizeof(int64_t));
        if (!bad) bad = api->search_topr(ctx, 1, pn, ps, pi, NULL);          /* builds the work lists, warms the code objects */
        const double t2 = swimm_wtime();
        t_first = t2 - (t1 + t_up);
        if (!bad) bad = api->search_topr(ctx, 1, pn, ps, pi, NULL);
        t_search = swimm_wtime() - t2;
        free(ps); free(pi);
        if (bad) { printf("SWIMM: GPU probe failed: %s\n", api->last_error()); exit(5); }
        api->destroy(ctx);
    }
    const double gpu_rate = Qrows * (double)pres / (t_search > 1e-6 ? t_search : 1e-6) * G;     /* cells per second, all devices */
    const double up_rate = (double)pres / (t_up > 1e-6 ? t_up : 1e-6);                            /* bytes per second per device */
    /* the GPUs' time for x cells: context + the longer of upload and search (slabs stream in while the search runs) */
    const double total = Qrows * (double)db->residues;
    const double t_upload = (double)db->residues / G / up_rate;
    /* what a first search costs beyond the alignment itself (work lists, buffers), per database byte */
    const double t_setup = (t_first > t_search ? t_first - t_search : 0.0) / (double)pres * (double)db->residues / G;
    const double fixed = t_ctx + t_setup;
    /* host_cells / host_rate = fixed + max(t_upload, (total - host_cells) / gpu_rate) */
    double host_cells = (fixed + total / gpu_rate) / (1.0 / host_rate + 1.0 / gpu_rate);
    if ((total - host_cells) / gpu_rate < t_upload) host_cells = (fixed + t_upload) * host_rate;
    if (host_cells > 0.5 * total) host_cells = 0.5 * total;
    if (getenv("SWIMM_DEBUG"))
        fprintf(stderr, "swimm: hybrid probe: host %.2f GCUPS; GPU context %.3f s, setup %.3f s, upload %.1f GB/s, search %.1f GCUPS per device -> host share %.3g of %.3g cells\n",
                host_rate / 1e9, t_ctx, t_setup, up_rate / 1e9, gpu_rate / G / 1e9, host_cells, total);
    uint64_t n = 0, res = 0;
    while (n < db->count && (double)(res + db->lengths[n]) * Qrows <= host_cells) res += db->lengths[n++];
    /* The host's rate depends on the sequence length (the sample above was the very shortest sequences), and most of its
     * share's cells sit at the long end of the share: measure again there and size the share with that rate. */
    if (n > 2 * sample) {
        /* long enough (about 0.4 s at the first probe's rate) to see the rate the host SUSTAINS: on the test box a burst of
         * 10 ms ran 3.5x faster than a second of the same work (shared host, CPU quota), and the share is sized for a leg
         * that lasts as long as the GPUs' */
        const double target_cells = 0.4 * host_rate;
        uint64_t w = 0, wres = 0;
        while (w < n && (double)wres * (double)q->Q < target_cells) wres += db->lengths[n - 1 - w++];
        w = w / vl * vl;
        if (w >= vl) {
            const uint64_t first = (n - w) / vl * vl;
            uint64_t off = 0;
            for (uint64_t i = 0; i < first; ++i) off += db->lengths[i];
            wres = 0;
            for (uint64_t i = first; i < first + w; ++i) wres += db->lengths[i];
            rc = swimm_assemble_single_chunk(db->lengths + first, db->codes + off, w, o->vector_length, o->cpu_block_size, &sc);
            if (rc) die_host(rc);
            int32_t *s2 = (int32_t *)malloc(q->count * sizeof(int32_t));
            int64_t *i2 = (int64_t *)malloc(q->count * sizeof(int64_t));
            st.seconds = 0;
            cpu_leg(o, &q0, submat, &sc, w, 1, s2, i2, &st);
            free(s2); free(i2);
            swimm_single_chunk_free(&sc);
            const double rate2 = (double)wres * (double)q->Q / (st.seconds > 1e-6 ? st.seconds : 1e-6);
            host_cells = (fixed + total / gpu_rate) / (1.0 / rate2 + 1.0 / gpu_rate);
            if ((total - host_cells) / gpu_rate < t_upload) host_cells = (fixed + t_upload) * rate2;
            if (host_cells > 0.5 * total) host_cells = 0.5 * total;
            if (getenv("SWIMM_DEBUG")) fprintf(stderr, "swimm: hybrid probe: host at the far end of its share %.2f GCUPS -> %.3g cells\n", rate2 / 1e9, host_cells);
            n = 0; res = 0;
            while (n < db->count && (double)(res + db->lengths[n]) * Qrows <= host_cells) res += db->lengths[n++];
        }
    }
    n = n / 128 * 128;                               /* the GPU part keeps whole lane groups */
    return n >= vl && n < db->count ? n : 0;
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
    swimm_queries q;
    int rc = swimm_queries_load(o.queries_filename, gpu_mode ? 0 : 1, &q);
    if (rc) die_host(rc);
    swimm_db db;
    if ((rc = swimm_db_load(o.db_prefix, &db))) die_host(rc);
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
    omp_set_num_threads(o.cpu_threads);

    if (o.execution_mode == MODE_CPU_ONLY) {
        swimm_single_chunk sc;
        if ((rc = swimm_assemble_single_chunk(db.lengths, db.codes, db.count, o.vector_length, o.cpu_block_size, &sc))) die_host(rc);
        cpu_leg(&o, &q, submat, &sc, db.count, top, top_scores, top_idx, &cst);
        swimm_single_chunk_free(&sc);
        workTime = cst.seconds;   /* the search call only, like CPUsearch.c:530,960 */
    } else {
        swimm_hip_api api;
        char err[1024];
        if (swimm_hip_load(&api, err, sizeof err)) { printf("%s\n", err); exit(5); }
        const int avail = api.device_count();
        if (avail <= 0) { printf("SWIMM: no MI355X visible: %s\n", api.last_error()); exit(5); }
        if (o.num_gpus > avail) { printf("SWIMM: %d GPUs requested, %d visible.\n", o.num_gpus, avail); exit(5); }
        const int G = o.num_gpus;
        if (o.execution_mode == MODE_HYBRID) n_cpu = hybrid_split(&api, &o, &q, submat, &db, G);
        const uint64_t n_gpu = db.count - n_cpu;
        uint64_t cpu_residues = 0;
        for (uint64_t i = 0; i < n_cpu; ++i) cpu_residues += db.lengths[i];
        const int lists = G + (n_cpu ? 1 : 0);
        int32_t *part_s = (int32_t *)malloc((size_t)lists * q.count * top * sizeof(int32_t));
        int64_t *part_i = (int64_t *)malloc((size_t)lists * q.count * top * sizeof(int64_t));
        if (!part_s || !part_i) { printf("SWIMM: An error occurred while allocating memory.\n"); exit(1); }
        for (size_t i = 0; i < (size_t)lists * q.count * top; ++i) { part_s[i] = -1; part_i[i] = -1; }
        swimm_single_chunk sc;      /* the host's lane layout is built before the clock starts (swimm.c:46 precedes the search call) */
        if (n_cpu && (rc = swimm_assemble_single_chunk(db.lengths, db.codes, n_cpu, o.vector_length, o.cpu_block_size, &sc))) die_host(rc);
        const double tick = swimm_wtime();   /* brackets transfers + kernels + merge, like MICsearch.c:51,350 */
        /* The GPU leg on a thread of its own, the host's share on this one with the ordinary (warm, top-level) OpenMP team
         * -- the same conditions under which hybrid_split measured the host's rate; a nested team started cold ran at half
         * of it. */
        gpu_leg_args ga = {&api, &o, &q, submat, db.lengths + n_cpu, db.codes + cpu_residues, n_gpu, n_cpu, top, part_s, part_i, &gst};
        pthread_t gpu_thread;
        if (pthread_create(&gpu_thread, NULL, gpu_leg_thread, &ga)) { printf("SWIMM: cannot start the GPU thread.\n"); exit(1); }
        if (n_cpu) cpu_leg(&o, &q, submat, &sc, n_cpu, top, part_s + (size_t)G * q.count * top, part_i + (size_t)G * q.count * top, &cst);
        pthread_join(gpu_thread, NULL);
        if (n_cpu) swimm_single_chunk_free(&sc);
        /* host k-way merge of the per-device lists ([lists][query][top]) */
        int32_t *ls = (int32_t *)malloc((size_t)lists * top * sizeof(int32_t));
        int64_t *li = (int64_t *)malloc((size_t)lists * top * sizeof(int64_t));
        for (uint64_t i = 0; i < q.count; ++i) {
            for (int g = 0; g < lists; ++g) {
                memcpy(ls + (size_t)g * top, part_s + ((size_t)g * q.count + i) * top, top * sizeof(int32_t));
                memcpy(li + (size_t)g * top, part_i + ((size_t)g * q.count + i) * top, top * sizeof(int64_t));
            }
            swimm_topr_merge(ls, li, (uint32_t)lists, (uint32_t)top, top_scores + i * top, top_idx + i * top);
        }
        workTime = swimm_wtime() - tick;
        free(ls); free(li); free(part_s); free(part_i);
    }

    /* titles of the reported hits only (the reference loads all N, sequences.c:757-761) */
    char **titles = (char **)malloc(q.count * top * sizeof(char *));
    if ((rc = swimm_db_titles(o.db_prefix, db.count, top_idx, q.count * top, titles))) die_host(rc);
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
        printf("Instruction set:\t\tcompiler-vectorised int32 lanes (vector length = %d)\n", o.vector_length);
    } else {
        if (o.execution_mode == MODE_HYBRID)
            printf("Execution mode:\t\t\tConcurrent host CPU and MI355X (%d CPU threads and %d GPUs)\n", o.cpu_threads, o.num_gpus);
        else
            printf("Execution mode:\t\t\tMI355X only (%d GPUs)\n", o.num_gpus);
        printf("Profile technique:\t\tQuery Profile in LDS\n");
        printf("Instruction set:\t\tgfx950 packed binary16 -> int16 -> int32 ladder (vector length = 128)\n");
        printf("Max. chunk size:\t\t%ld bytes\n", o.max_chunk_size);
        printf("Chunk count:\t\t\t%ld \n", (long)gst.chunk_count);
        printf("Kernel time:\t\t\t%lf seconds\n", gst.kernel_ms / 1000.0);
        printf("Promoted to int32:\t\t%ld alignments\n", (long)gst.promoted);
        if (o.execution_mode == MODE_HYBRID)   /* the reference prints "%d chunks in CPU and %d in MICs" (HETsearch.c:337-342) */
            printf("Host CPU share:\t\t\t%ld sequences (%.3lf seconds), MI355X %ld sequences (%.3lf seconds)\n", (long)n_cpu, cst.seconds,
                   (long)(db.count - n_cpu), gst.seconds);
    }
    for (uint64_t i = 0; i < q.count * top; ++i) free(titles[i]);
    free(titles); free(top_scores); free(top_idx);
    swimm_db_free(&db);
    swimm_queries_free(&q);
    return 0;
}
