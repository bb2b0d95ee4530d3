/* cpu_search.c -- execution mode 0: the search on the host CPU (explicitly selected with -m 0, the host leg of -m 2;
 * the GPU path never falls back to it).
 *
 * What the reference does here (cpu_search_avx2_sp, CPUsearch.c:482-967): inter-task SIMD -- every vector lane aligns one
 * database sequence of a lane group -- in saturating int8, the lanes that saturate again in int16, then in int32
 * (CPUsearch.c:605-668, 678-817, 820-957), tasks = (query, group) handed out dynamically, longest first (CPUsearch.c:540-544).
 * That technique class is kept; the code is this build's own and is organised the other way round:
 *
 *   - the sweep is COLUMN-major: for one database column (32 residues, one per lane) the 24 possible substitution vectors
 *     are formed once with byte shuffles (two vpshufb + a blend per query residue code: a 768-byte table that lives in L1),
 *     then the query is walked top to bottom with H and E of every query row in a per-thread array (2 x m vectors) and F,
 *     the diagonal and the running best in registers.  No column blocks, no per-block score profile, no maxRow / lastCol
 *     hand-over arrays: one pass over the column touches each row's H and E exactly once;
 *   - a group of any lane width that is a multiple of 32 is swept 32 lanes at a time; saturated lanes are found after the
 *     int8 sweep and the 16-lane half (then the 8-lane quarter) that holds them is swept again one tier up -- the DB bytes
 *     are widened on the fly (vpmovsxbw / vpmovsxbd), the same shuffle-made substitution bytes widened with them.
 * Lane widths that are not multiples of 32 take the plain int32 loop at the end of this file (exact, slow, never the default).
 * Every tier is exact below its saturation point, so the scores equal the reference's bit for bit
 * (tests/test_oracle_golden.py, tests/test_cli.py mode 0, tests/test_host_formats.py). */
#include "swimm_host.h"

#include <immintrin.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int swimm_host_fail_(int code, const char *fmt, ...);

/* substitution bytes of one query residue code for 32 (or 16, in the low half) database residues: row = 32 table bytes */
static inline __m256i sub_bytes(const __m256i row_lo, const __m256i row_hi, const __m256i d)
{
    const __m256i lo = _mm256_shuffle_epi8(row_lo, d), hi = _mm256_shuffle_epi8(row_hi, d);      /* (codes 0..31: bit 7 clear, low 4 bits index) */
    return _mm256_blendv_epi8(lo, hi, _mm256_slli_epi16(d, 3));                                   /* bit 4 of the code -> bit 7: codes 16..31 take row_hi */
}

typedef struct {
    __m256i rows_lo[24], rows_hi[24];      /* the matrix, every row's bytes 0..15 / 16..31 in both 128-bit lanes */
    __m256i *H, *E;                        /* per query row, one vector each (any tier) */
    __m256i P[24];                         /* the current column's substitution vectors, by query residue code */
} sweep_state;

/* ---- the three tiers of one sweep: `cols` columns of `lanes` sequences (column stride `stride` bytes) against query qa[0..m) ---- */
#define SWEEP(NAME, LOAD_D, WIDEN, ADDS, SUBS, MAX, SET1)                                                              \
    static __m256i NAME(sweep_state *st, const signed char *qa, int m, const unsigned char *col0, size_t stride,       \
                        int cols, int goe, int ge)                                                                     \
    {                                                                                                                  \
        const __m256i vgoe = SET1(goe), vge = SET1(ge), zero = _mm256_setzero_si256();                                 \
        __m256i best = zero;                                                                                           \
        for (int i = 0; i < m; ++i) { st->H[i] = zero; st->E[i] = zero; }                                              \
        for (int j = 0; j < cols; ++j) {                                                                               \
            const __m256i d = LOAD_D(col0 + (size_t)j * stride);                                                       \
            for (int r = 0; r < 24; ++r) st->P[r] = WIDEN(sub_bytes(st->rows_lo[r], st->rows_hi[r], d));               \
            __m256i diag = zero, F = zero;                                                                             \
            for (int i = 0; i < m; ++i) {                                                                              \
                /* The one loop-carried value is F.  h0 = everything of H that does not come from above; H = max(h0, F);       \
                 * F' = max(F - ge, H - goe) = max(F - ge, h0 - goe) because F - goe <= F - ge (goe >= ge; the saturating      \
                 * subtract is monotone): the chain from row to row is ONE subtract and ONE max, H hangs off it to the side.   \
                 * (With F inside H the chain was max, max, sub, max: 20 GCUPS per Zen 5 core against the reference's 28.) */  \
                const __m256i left = st->H[i], e = st->E[i];                                                           \
                const __m256i h0 = MAX(MAX(ADDS(diag, st->P[(int)qa[i]]), e), zero);                                   \
                const __m256i h = MAX(h0, F);                                                                          \
                F = MAX(SUBS(F, vge), SUBS(h0, vgoe));                                                                 \
                best = MAX(best, h);                                                                                   \
                st->E[i] = MAX(SUBS(e, vge), SUBS(h, vgoe));                                                           \
                st->H[i] = h;                                                                                          \
                diag = left;                                                                                           \
            }                                                                                                          \
        }                                                                                                              \
        return best;                                                                                                   \
    }

#define LOAD32(p) _mm256_loadu_si256((const __m256i *)(p))
#define LOAD16(p) _mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(p)))
#define LOAD8(p) _mm256_castsi128_si256(_mm_loadl_epi64((const __m128i *)(p)))
#define KEEP(x) (x)
#define WIDEN16(x) _mm256_cvtepi8_epi16(_mm256_castsi256_si128(x))
#define WIDEN32(x) _mm256_cvtepi8_epi32(_mm256_castsi256_si128(x))
#define SET1_8(x) _mm256_set1_epi8((char)(x))
#define SET1_16(x) _mm256_set1_epi16((short)(x))
#define SET1_32(x) _mm256_set1_epi32(x)
/* (int32: plain arithmetic; E and F may go negative, H is floored at 0 by the MAX with zero above) */
SWEEP(sweep8, LOAD32, KEEP, _mm256_adds_epi8, _mm256_subs_epi8, _mm256_max_epi8, SET1_8)
SWEEP(sweep16, LOAD16, WIDEN16, _mm256_adds_epi16, _mm256_subs_epi16, _mm256_max_epi16, SET1_16)
SWEEP(sweep32, LOAD8, WIDEN32, _mm256_add_epi32, _mm256_sub_epi32, _mm256_max_epi32, SET1_32)

/* one query against 32 lanes of one group: int8, then int16 for the halves with a saturated lane, then int32 for the quarters */
static void align32(sweep_state *st, const signed char *qa, int m, const unsigned char *col0, size_t stride, int cols, int goe, int ge,
                    int32_t *out)
{
    signed char b8[32];
    _mm256_storeu_si256((__m256i *)b8, sweep8(st, qa, m, col0, stride, cols, goe, ge));
    for (int l = 0; l < 32; ++l) out[l] = b8[l];
    for (int half = 0; half < 2; ++half) {
        int sat = 0;
        for (int l = 16 * half; l < 16 * half + 16; ++l) sat |= b8[l] == 127;
        if (!sat) continue;
        short b16[16];
        _mm256_storeu_si256((__m256i *)b16, sweep16(st, qa, m, col0 + 16 * half, stride, cols, goe, ge));
        for (int l = 0; l < 16; ++l) if (b8[16 * half + l] == 127) out[16 * half + l] = b16[l];
        for (int quarter = 0; quarter < 2; ++quarter) {
            int sat16 = 0;
            for (int l = 8 * quarter; l < 8 * quarter + 8; ++l) sat16 |= b16[l] == 32767;
            if (!sat16) continue;
            int32_t b32[8];
            _mm256_storeu_si256((__m256i *)b32, sweep32(st, qa, m, col0 + 16 * half + 8 * quarter, stride, cols, goe, ge));
            for (int l = 0; l < 8; ++l) if (b16[8 * quarter + l] == 32767) out[16 * half + 8 * quarter + l] = b32[l];
        }
    }
}

/* lane widths that are not multiples of 32: int32 lanes, left to the compiler (exact; the `swimm` program never assembles such a width) */
static void align_any(const signed char *qa, int m, const unsigned char *col0, int vl, int cols, const char *submat, int goe, int ge,
                      int32_t *H, int32_t *E, int32_t *out)
{
    for (int l = 0; l < vl; ++l) {
        int32_t best = 0;
        for (int i = 0; i < m; ++i) { H[i] = 0; E[i] = 0; }
        for (int j = 0; j < cols; ++j) {
            const int d = col0[(size_t)j * vl + l];
            int32_t diag = 0, F = 0;
            for (int i = 0; i < m; ++i) {
                const int32_t left = H[i];
                int32_t h = diag + submat[(int)qa[i] * 32 + d];
                if (h < E[i]) h = E[i];
                if (h < F) h = F;
                if (h < 0) h = 0;
                if (h > best) best = h;
                const int32_t u = h - goe, e2 = E[i] - ge, f2 = F - ge;
                E[i] = e2 > u ? e2 : u;
                F = f2 > u ? f2 : u;
                H[i] = h;
                diag = left;
            }
        }
        out[l] = best;
    }
}

int swimm_cpu_search(const char *a, const uint16_t *m, uint64_t query_count, const uint32_t *a_disp, const char *b,
                     const uint16_t *n, uint64_t vc, const uint64_t *b_disp, const char *submat, int open_gap,
                     int extend_gap, int n_threads, int block_size, int vl, int32_t *scores, double *work_time)
{
    (void)block_size;      /* (the reference's column block: this sweep has none) */
    if (!a || !m || !a_disp || !b || !n || !b_disp || !submat || !scores || vl <= 0)
        return swimm_host_fail_(SWIMM_E_ARG, "SWIMM: invalid argument to the CPU search.");
    if (open_gap < 0 || extend_gap < 0 || open_gap + extend_gap > 127)
        return swimm_host_fail_(SWIMM_E_ARG, "SWIMM: gap penalties must be >= 0 and open + extend <= 127.");
    const int goe = open_gap + extend_gap, ge = extend_gap;
    const double t0 = swimm_wtime();
    int mmax = 1;
    for (uint64_t q = 0; q < query_count; ++q) if (m[q] > mmax) mmax = m[q];
    int failed = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        sweep_state *st = (sweep_state *)aligned_alloc(32, (sizeof(sweep_state) + 31) / 32 * 32);
        __m256i *H = (__m256i *)aligned_alloc(32, (size_t)mmax * sizeof(__m256i)), *E = (__m256i *)aligned_alloc(32, (size_t)mmax * sizeof(__m256i));
        if (!st || !H || !E) {
#pragma omp atomic write
            failed = 1;
        } else {
            st->H = H; st->E = E;
            for (int r = 0; r < 24; ++r) {
                const __m128i lo = _mm_loadu_si128((const __m128i *)(submat + r * 32)), hi = _mm_loadu_si128((const __m128i *)(submat + r * 32 + 16));
                st->rows_lo[r] = _mm256_broadcastsi128_si256(lo);
                st->rows_hi[r] = _mm256_broadcastsi128_si256(hi);
            }
        }
#pragma omp barrier
        if (!failed) {
            /* tasks: (query, group), the longest queries and the longest groups first (the database is length-sorted) */
#pragma omp for schedule(dynamic) nowait
            for (uint64_t t = 0; t < query_count * vc; ++t) {
                const uint64_t q = (query_count - 1) - (t % query_count);
                const uint64_t s = (vc - 1) - (t / query_count);
                const signed char *qa = (const signed char *)a + a_disp[q];
                const unsigned char *bg = (const unsigned char *)b + b_disp[s];
                int32_t *out = scores + (q * vc + s) * (uint64_t)vl;
                if (vl % 32 == 0) {
                    for (int l0 = 0; l0 < vl; l0 += 32) align32(st, qa, m[q], bg + l0, (size_t)vl, n[s], goe, ge, out + l0);
                } else {
                    align_any(qa, m[q], bg, vl, n[s], submat, goe, ge, (int32_t *)H, (int32_t *)E, out);
                }
            }
        }
        free(st); free(H); free(E);
    }
    if (failed) return swimm_host_fail_(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory.");
    if (work_time) *work_time = swimm_wtime() - t0;
    return SWIMM_OK;
}
