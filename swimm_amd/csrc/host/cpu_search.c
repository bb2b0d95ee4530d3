/* cpu_search.c -- execution mode 0: the search on the host CPU (explicitly selected with -m 0;
 * the GPU path never falls back to it).
 *
 * Same inter-task scheme as cpu_search_avx2_sp (CPUsearch.c:482-967): tasks = (query, lane group),
 * longest first, dynamic OpenMP schedule (CPUsearch.c:540-544); every lane of a group aligns one
 * database sequence.  Instead of the int8 -> int16 -> int32 ladder the lanes are int32 from the
 * start (the result is identical: each tier of the ladder is exact below its saturation point) and
 * the lane loop is left to the compiler's vectoriser.  Column-blocked like the reference
 * (CPUsearch.c:562-569) so the per-block state stays in L1/L2. */
#include "swimm_host.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

int swimm_host_fail_(int code, const char *fmt, ...);

int swimm_cpu_search(const char *a, const uint16_t *m, uint64_t query_count, const uint32_t *a_disp, const char *b,
                     const uint16_t *n, uint64_t vc, const uint64_t *b_disp, const char *submat, int open_gap,
                     int extend_gap, int n_threads, int block_size, int vl, int32_t *scores, double *work_time)
{
    if (!a || !m || !a_disp || !b || !n || !b_disp || !submat || !scores || vl <= 0 || block_size <= 0)
        return swimm_host_fail_(SWIMM_E_ARG, "SWIMM: invalid argument to the CPU search.");
    const int goe = open_gap + extend_gap, ge = extend_gap;
    const double t0 = swimm_wtime();
    int mmax = 0;
    for (uint64_t q = 0; q < query_count; ++q) if (m[q] > mmax) mmax = m[q];
    int failed = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        const size_t W = (size_t)vl;
        int32_t *Hblk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(block_size + 1) * W);   /* H of the previous row, this block */
        int32_t *Fblk = (int32_t *)malloc(sizeof(int32_t) * (size_t)(block_size + 1) * W);   /* F per column ("maxCol") */
        int32_t *Erow = (int32_t *)malloc(sizeof(int32_t) * (size_t)mmax * W);               /* E per row across blocks ("maxRow") */
        int32_t *Hlast = (int32_t *)malloc(sizeof(int32_t) * (size_t)(mmax + 1) * W);        /* H of the previous block's last column ("lastCol") */
        int32_t *best = (int32_t *)malloc(sizeof(int32_t) * W);
        int32_t *hd = (int32_t *)malloc(sizeof(int32_t) * W), *e = (int32_t *)malloc(sizeof(int32_t) * W);
        int32_t *hnew_last = (int32_t *)malloc(sizeof(int32_t) * W);
        /* score profile of the block: sp[r][j][lane] = submat[r][db residue], r = 0..23 (the technique of
         * CPUsearch.c:582-603, built with a plain table lookup instead of pshufb) */
        signed char *sp = (signed char *)malloc((size_t)24 * block_size * W);
        if (!Hblk || !Fblk || !Erow || !Hlast || !best || !hd || !e || !hnew_last || !sp) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp barrier
        if (!failed) {
#pragma omp for schedule(dynamic) nowait
            for (uint64_t t = 0; t < query_count * vc; ++t) {
                const uint64_t q = (query_count - 1) - (t % query_count);
                const uint64_t s = (vc - 1) - (t / query_count);
                const char *qa = a + a_disp[q];
                const unsigned char *bg = (const unsigned char *)b + b_disp[s];
                const int mq = m[q], np = n[s];
                memset(Erow, 0, sizeof(int32_t) * (size_t)mq * W);
                memset(Hlast, 0, sizeof(int32_t) * (size_t)(mq + 1) * W);
                memset(best, 0, sizeof(int32_t) * W);
                for (int c0 = 0; c0 < np; c0 += block_size) {
                    const int dim = np - c0 < block_size ? np - c0 : block_size;
                    memset(Hblk, 0, sizeof(int32_t) * (size_t)(dim + 1) * W);
                    memset(Fblk, 0, sizeof(int32_t) * (size_t)(dim + 1) * W);
                    for (int r = 0; r < 24; ++r) {
                        const signed char *srow = (const signed char *)submat + r * 32;
                        signed char *dst = sp + (size_t)r * dim * W;
                        const unsigned char *src = bg + (size_t)c0 * W;
                        for (size_t x = 0; x < (size_t)dim * W; ++x) dst[x] = srow[src[x]];
                    }
                    for (int i = 0; i < mq; ++i) {
                        const signed char *sprow = sp + (size_t)qa[i] * dim * W;
                        /* diagonal for column 0 of the block = H[i-1][last column of previous block];
                         * Hlast[i] holds that, Hlast[i+1] is being produced for the next block */
                        memcpy(hd, Hlast + (size_t)i * W, sizeof(int32_t) * W);
                        memcpy(e, Erow + (size_t)i * W, sizeof(int32_t) * W);
                        for (int j = 1; j <= dim; ++j) {
                            int32_t *Hj = Hblk + (size_t)j * W, *Fj = Fblk + (size_t)j * W;
                            const signed char *sj = sprow + (size_t)(j - 1) * W;
#pragma omp simd
                            for (size_t l = 0; l < W; ++l) {
                                int32_t h = hd[l] + sj[l];
                                h = h < e[l] ? e[l] : h;
                                h = h < Fj[l] ? Fj[l] : h;
                                h = h < 0 ? 0 : h;
                                const int32_t u = h - goe;
                                const int32_t e2 = e[l] - ge, f2 = Fj[l] - ge;
                                e[l] = e2 > u ? e2 : u;
                                Fj[l] = f2 > u ? f2 : u;
                                hd[l] = Hj[l];
                                Hj[l] = h;
                                best[l] = h > best[l] ? h : best[l];
                            }
                        }
                        memcpy(Erow + (size_t)i * W, e, sizeof(int32_t) * W);
                        /* Hblk[dim] now holds H[i][last column]; it becomes row i+1's diagonal in the next block.
                         * Row i's own entry may only be replaced after it was consumed above, so rotate through a temp. */
                        if (i > 0) memcpy(Hlast + (size_t)i * W, hnew_last, sizeof(int32_t) * W);
                        memcpy(hnew_last, Hblk + (size_t)dim * W, sizeof(int32_t) * W);
                    }
                    memcpy(Hlast + (size_t)mq * W, hnew_last, sizeof(int32_t) * W);
                }
                memcpy(scores + (q * vc + s) * W, best, sizeof(int32_t) * W);
            }
        }
        free(Hblk); free(Fblk); free(Erow); free(Hlast); free(best); free(hd); free(e); free(hnew_last); free(sp);
    }
    if (failed) return swimm_host_fail_(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory.");
    if (work_time) *work_time = swimm_wtime() - t0;
    return SWIMM_OK;
}
