/*
 * sw_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the SWIMM search hot path, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for
 * the HIP path.  Nothing under swimm_amd/ may link, import or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against golden vectors produced by the reference's own CPU path
 * (oracle/_ref/libswimm_ref.so, built from /root/reference by oracle/Makefile;
 * generator: tests/golden/make_golden.py) and, when oracle/_ref is present,
 * against the reference itself on fresh seeded inputs.
 *
 * Reference lines restated (all under /root/reference):
 *   recurrence            CPUsearch.c:622-636 (int8), 766-780 (int16), 907-921 (int32)
 *   saturating tiers      CPUsearch.c:678-691 (==127 -> int16), 819-832 (==32767 -> int32)
 *   task -> score slot    CPUsearch.c:543-548, 670-676
 *   interleaved DB layout sequences.c:703-723  (byte = disp[s] + j*VL + lane, pad = 24)
 *   query padding         sequences.c:378-387  (odd length -> one trailing code 23)
 *   score lookup          CPUsearch.c:593,611  (submat[q*32 + d], 24 rows x 32 cols)
 *   top-r order           utils.c:3-86         (score desc, then sorted-DB index desc)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define SUBMAT_COLS 32

/* ---- (1) exact scalar Gotoh on two plain residue-code strings -------------
 * H = max(0, Hdiag + S, E, F); E = max(E - ge, H - goe); F = max(F - ge, H - goe)
 * (CPUsearch.c:622-636 with E = "maxRow"/aux1, F = "maxCol").  int32, no tiers. */
int sw_oracle_pair(const int8_t *q, int m, const int8_t *d, int n,
                   const int8_t *submat, int open_gap, int extend_gap)
{
    const int goe = open_gap + extend_gap, ge = extend_gap;
    int *Hrow = (int *)calloc((size_t)n + 1, sizeof(int)); /* H[i-1][*]            */
    int *Fcol = (int *)calloc((size_t)n + 1, sizeof(int)); /* F per column (maxCol) */
    int best = 0;
    for (int i = 0; i < m; i++) {
        const int8_t *srow = submat + (int)q[i] * SUBMAT_COLS;
        int E = 0, hdiag = 0, hleft = 0;
        for (int j = 1; j <= n; j++) {
            int h = hdiag + srow[(int)d[j - 1]];
            if (h < E) h = E;
            if (h < Fcol[j]) h = Fcol[j];
            if (h < 0) h = 0;
            int u = h - goe;
            E = (E - ge > u) ? E - ge : u;
            Fcol[j] = (Fcol[j] - ge > u) ? Fcol[j] - ge : u;
            hdiag = Hrow[j];
            Hrow[j] = h;
            hleft = h;
            if (h > best) best = h;
        }
        (void)hleft;
    }
    free(Hrow);
    free(Fcol);
    return best;
}

/* ---- (2) one lane of the reference layout, at a given saturating width -----
 * width 8/16: signed saturating add/sub as _mm256_adds/subs_epi8/16;
 * width 32 : plain wrapping add/sub as _mm256_add/sub_epi32.               */
static inline int sat(long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

static int lane_score_width(const char *a, int m, const char *b_grp, int npad, int vl, int lane,
                            const char *submat, int open_gap, int extend_gap, int width,
                            int *Hrow, int *Fcol)
{
    const int hi = width == 8 ? 127 : (width == 16 ? 32767 : INT32_MAX);
    const int lo = width == 8 ? -128 : (width == 16 ? -32768 : INT32_MIN);
    /* the reference stores goe/ge in a lane of the tier's width (CPUsearch.c:518-520) */
    int goe = open_gap + extend_gap, ge = extend_gap;
    if (width == 8) { goe = (int8_t)goe; ge = (int8_t)ge; }
    if (width == 16) { goe = (int16_t)goe; ge = (int16_t)ge; }
    memset(Hrow, 0, sizeof(int) * ((size_t)npad + 1));
    memset(Fcol, 0, sizeof(int) * ((size_t)npad + 1));
    int best = 0;
    for (int i = 0; i < m; i++) {
        const int8_t *srow = (const int8_t *)submat + (int)a[i] * SUBMAT_COLS;
        int E = 0, hdiag = 0;
        for (int j = 1; j <= npad; j++) {
            int dres = (unsigned char)b_grp[(size_t)(j - 1) * vl + lane];
            /* row 23 of the score profile is forced to zero (CPUsearch.c:602);
             * submat row 23 is all zero too, so the lookup below is identical. */
            int h = sat((long)hdiag + srow[dres], lo, hi);
            if (h < E) h = E;
            if (h < Fcol[j]) h = Fcol[j];
            if (h < 0) h = 0;
            int u = sat((long)h - goe, lo, hi);
            int e2 = sat((long)E - ge, lo, hi);
            int f2 = sat((long)Fcol[j] - ge, lo, hi);
            E = e2 > u ? e2 : u;
            Fcol[j] = f2 > u ? f2 : u;
            hdiag = Hrow[j];
            Hrow[j] = h;
            if (h > best) best = h;
        }
    }
    return best;
}

/* ---- (3) tiered search on the reference's single-chunk layout --------------
 * Same argument meaning as cpu_search_avx2_sp (CPUsearch.h:37-39) plus `vl`.
 * scores[(q*vc + s)*vl + lane]; tiers[0..2] count lanes finishing in 8/16/32 bit. */
void sw_oracle_search_tiered(const char *a, const uint16_t *m, unsigned long qcnt, const uint32_t *a_disp,
                             const char *b, const uint16_t *n, unsigned long vc, const unsigned long *b_disp,
                             const char *submat, int open_gap, int extend_gap, int vl, int n_threads,
                             int *scores, long *tiers)
{
    long t8 = 0, t16 = 0, t32 = 0;
    int nmax = 0;
    for (unsigned long s = 0; s < vc; s++) if (n[s] > nmax) nmax = n[s];
#pragma omp parallel num_threads(n_threads) reduction(+ : t8, t16, t32)
    {
        int *Hrow = (int *)malloc(sizeof(int) * ((size_t)nmax + 1));
        int *Fcol = (int *)malloc(sizeof(int) * ((size_t)nmax + 1));
#pragma omp for schedule(dynamic)
        for (unsigned long t = 0; t < qcnt * vc; t++) {
            unsigned long q = (qcnt - 1) - (t % qcnt); /* CPUsearch.c:543 */
            unsigned long s = (vc - 1) - (t / qcnt);   /* CPUsearch.c:544 */
            for (int lane = 0; lane < vl; lane++) {
                int sc = lane_score_width(a + a_disp[q], m[q], b + b_disp[s], n[s], vl, lane, submat,
                                          open_gap, extend_gap, 8, Hrow, Fcol);
                if (sc == 127) { /* CPUsearch.c:679-683 */
                    sc = lane_score_width(a + a_disp[q], m[q], b + b_disp[s], n[s], vl, lane, submat,
                                          open_gap, extend_gap, 16, Hrow, Fcol);
                    if (sc == 32767) { /* CPUsearch.c:820-824 */
                        sc = lane_score_width(a + a_disp[q], m[q], b + b_disp[s], n[s], vl, lane, submat,
                                              open_gap, extend_gap, 32, Hrow, Fcol);
                        t32++;
                    } else t16++;
                } else t8++;
                scores[(q * vc + s) * vl + lane] = sc;
            }
        }
        free(Hrow);
        free(Fcol);
    }
    if (tiers) { tiers[0] = t8; tiers[1] = t16; tiers[2] = t32; }
}

/* ---- (4) exact int32 search, all lanes of a group in the inner loop ---------
 * Same result as (3) (every tier is exact below its saturation point); the lane
 * loop auto-vectorises, so this is the oracle used for the larger parity cases. */
void sw_oracle_search_exact(const char *a, const uint16_t *m, unsigned long qcnt, const uint32_t *a_disp,
                            const char *b, const uint16_t *n, unsigned long vc, const unsigned long *b_disp,
                            const char *submat, int open_gap, int extend_gap, int vl, int n_threads,
                            int *scores)
{
    const int goe = open_gap + extend_gap, ge = extend_gap;
    int nmax = 0;
    for (unsigned long s = 0; s < vc; s++) if (n[s] > nmax) nmax = n[s];
#pragma omp parallel num_threads(n_threads)
    {
        int *Hrow = (int *)malloc(sizeof(int) * ((size_t)nmax + 1) * vl);
        int *Fcol = (int *)malloc(sizeof(int) * ((size_t)nmax + 1) * vl);
        int *E = (int *)malloc(sizeof(int) * vl), *hd = (int *)malloc(sizeof(int) * vl);
        int *best = (int *)malloc(sizeof(int) * vl);
#pragma omp for schedule(dynamic)
        for (unsigned long t = 0; t < qcnt * vc; t++) {
            unsigned long q = (qcnt - 1) - (t % qcnt);
            unsigned long s = (vc - 1) - (t / qcnt);
            const char *qa = a + a_disp[q];
            const unsigned char *bg = (const unsigned char *)b + b_disp[s];
            const int np = n[s];
            memset(Hrow, 0, sizeof(int) * ((size_t)np + 1) * vl);
            memset(Fcol, 0, sizeof(int) * ((size_t)np + 1) * vl);
            memset(best, 0, sizeof(int) * vl);
            for (int i = 0; i < m[q]; i++) {
                const int8_t *srow = (const int8_t *)submat + (int)qa[i] * SUBMAT_COLS;
                memset(E, 0, sizeof(int) * vl);
                memset(hd, 0, sizeof(int) * vl);
                for (int j = 1; j <= np; j++) {
                    int *Hj = Hrow + (size_t)j * vl, *Fj = Fcol + (size_t)j * vl;
                    const unsigned char *dj = bg + (size_t)(j - 1) * vl;
#pragma omp simd
                    for (int l = 0; l < vl; l++) {
                        int h = hd[l] + srow[dj[l]];
                        h = h < E[l] ? E[l] : h;
                        h = h < Fj[l] ? Fj[l] : h;
                        h = h < 0 ? 0 : h;
                        int u = h - goe;
                        int e2 = E[l] - ge, f2 = Fj[l] - ge;
                        E[l] = e2 > u ? e2 : u;
                        Fj[l] = f2 > u ? f2 : u;
                        hd[l] = Hj[l];
                        Hj[l] = h;
                        best[l] = h > best[l] ? h : best[l];
                    }
                }
            }
            memcpy(scores + (q * vc + s) * vl, best, sizeof(int) * vl);
        }
        free(Hrow); free(Fcol); free(E); free(hd); free(best);
    }
}

/* ---- (5) top-r order of sort_scores (utils.c:3-86): score descending, ties by
 * LARGER sorted-DB index first (merge takes left only if strictly greater,
 * utils.c:12; 2-element base swaps on <=, utils.c:52).  Writes r (score,index). */
typedef struct { int score; long idx; } oracle_hit;
static int hit_cmp(const void *x, const void *y)
{
    const oracle_hit *a = (const oracle_hit *)x, *b = (const oracle_hit *)y;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}
void sw_oracle_topr(const int *scores, long n_seq, long r, int *out_scores, long *out_idx)
{
    oracle_hit *h = (oracle_hit *)malloc(sizeof(oracle_hit) * (size_t)n_seq);
    for (long i = 0; i < n_seq; i++) { h[i].score = scores[i]; h[i].idx = i; }
    qsort(h, (size_t)n_seq, sizeof(oracle_hit), hit_cmp);
    if (r > n_seq) r = n_seq;
    for (long i = 0; i < r; i++) { out_scores[i] = h[i].score; out_idx[i] = h[i].idx; }
    free(h);
}
