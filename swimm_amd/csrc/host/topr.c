/* topr.c -- first r rows of the reference's sorted listing without sorting all N (see swimm_host.h).
 *
 * sort_scores (utils.c:71-86) is a merge sort whose merge takes the LEFT element only when it is
 * strictly greater (utils.c:12) and whose 2-element base swaps on <= (utils.c:52); net effect: score
 * descending, equal scores ordered by LARGER index first.  That is the order of the 64-bit key
 * (score << 32 | index) descending for scores >= 0, which is what both functions below use. */
#include "swimm_host.h"

#include <stdlib.h>

typedef struct { int32_t score; int64_t idx; } hit_t;

static inline int hit_less(hit_t a, hit_t b)   /* a ranks BELOW b in the listing */
{
    return a.score < b.score || (a.score == b.score && a.idx < b.idx);
}

static void sift_down(hit_t *h, uint32_t n, uint32_t i)   /* min-heap on listing rank */
{
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, s = i;
        if (l < n && hit_less(h[l], h[s])) s = l;
        if (r < n && hit_less(h[r], h[s])) s = r;
        if (s == i) return;
        hit_t t = h[i]; h[i] = h[s]; h[s] = t;
        i = s;
    }
}

static void heap_to_listing(hit_t *h, uint32_t k, uint32_t r, int32_t *out_scores, int64_t *out_idx)
{
    /* pop the minimum repeatedly: fills the output from the last row to the first */
    for (uint32_t n = k; n > 0; --n) {
        out_scores[n - 1] = h[0].score;
        out_idx[n - 1] = h[0].idx;
        h[0] = h[n - 1];
        sift_down(h, n - 1, 0);
    }
    for (uint32_t i = k; i < r; ++i) { out_scores[i] = -1; out_idx[i] = -1; }
}

void swimm_topr(const int32_t *scores, uint64_t n, uint32_t r, int32_t *out_scores, int64_t *out_idx)
{
    uint32_t k = (uint32_t)(n < r ? n : r);
    hit_t *h = (hit_t *)malloc((k ? k : 1) * sizeof(hit_t));
    uint32_t fill = 0;
    for (uint64_t i = 0; i < n; ++i) {
        hit_t x = {scores[i], (int64_t)i};
        if (fill < k) {
            h[fill++] = x;
            if (fill == k) for (int32_t j = (int32_t)k / 2 - 1; j >= 0; --j) sift_down(h, k, (uint32_t)j);
        } else if (hit_less(h[0], x)) {
            h[0] = x;
            sift_down(h, k, 0);
        }
    }
    heap_to_listing(h, k, r, out_scores, out_idx);
    free(h);
}

void swimm_topr_merge(const int32_t *scores, const int64_t *idx, uint32_t lists, uint32_t r, int32_t *out_scores,
                      int64_t *out_idx)
{
    hit_t *h = (hit_t *)malloc((r ? r : 1) * sizeof(hit_t));
    uint32_t fill = 0;
    for (uint64_t i = 0; i < (uint64_t)lists * r; ++i) {
        if (idx[i] < 0) continue;
        hit_t x = {scores[i], idx[i]};
        if (fill < r) {
            h[fill++] = x;
            if (fill == r) for (int32_t j = (int32_t)r / 2 - 1; j >= 0; --j) sift_down(h, r, (uint32_t)j);
        } else if (hit_less(h[0], x)) {
            h[0] = x;
            sift_down(h, r, 0);
        }
    }
    if (fill < r) for (int32_t j = (int32_t)fill / 2 - 1; j >= 0; --j) sift_down(h, fill, (uint32_t)j);
    heap_to_listing(h, fill, r, out_scores, out_idx);
    free(h);
}
