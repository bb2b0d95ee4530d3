/* assemble.c -- lane-interleaved database layout, single chunk and chunked (see swimm_host.h). */
#include "swimm_host.h"

#include <stdlib.h>
#include <string.h>

int swimm_host_fail_(int code, const char *fmt, ...);
#define FAIL swimm_host_fail_

/* n[g]: length of the group's last (longest) member, rounded up to a multiple of 5
 * (sequences.c:677-684); the last group ends at the last sequence */
static uint16_t *group_lengths(const uint16_t *lengths, uint64_t count, int vl, uint64_t *vc_out)
{
    uint64_t vc = (count + (uint64_t)vl - 1) / (uint64_t)vl;
    uint16_t *n = (uint16_t *)malloc((vc ? vc : 1) * sizeof(uint16_t));
    if (!n) return NULL;
    for (uint64_t g = 0; g < vc; ++g) {
        uint64_t last = (g + 1) * (uint64_t)vl - 1;
        if (last >= count) last = count - 1;
        uint32_t L = lengths[last];
        L = (L + SWIMM_SEQ_LEN_MULT - 1) / SWIMM_SEQ_LEN_MULT * SWIMM_SEQ_LEN_MULT;
        n[g] = (uint16_t)L;
    }
    *vc_out = vc;
    return n;
}

/* fill one group's tile: byte (j, k) = residue j of sequence g*vl+k, or 24 (sequences.c:703-723) */
static void fill_group(char *tile, uint32_t npad, int vl, const uint16_t *lengths, const char *const *seq_ptr,
                       uint64_t first, uint64_t count)
{
    memset(tile, SWIMM_PAD_CODE, (size_t)npad * vl);
    for (int k = 0; k < vl; ++k) {
        uint64_t s = first + (uint64_t)k;
        if (s >= count) break;
        const char *src = seq_ptr[s];
        uint32_t L = lengths[s];
        char *dst = tile + k;
        for (uint32_t j = 0; j < L; ++j) dst[(size_t)j * vl] = src[j];
    }
}

static int check_args(const uint16_t *lengths, const char *codes, uint64_t count, int vl)
{
    if (!lengths || !codes || count == 0) return FAIL(SWIMM_E_ARG, "SWIMM: empty database.");
    if (vl <= 0 || vl > 4096) return FAIL(SWIMM_E_ARG, "SWIMM: %d is not a valid lane width.", vl);
    if (lengths[count - 1] > 65535 - SWIMM_SEQ_LEN_MULT)
        return FAIL(SWIMM_E_FORMAT, "SWIMM: longest sequence (%u) cannot be rounded up to a multiple of 5 in 16 bits.", lengths[count - 1]);
    return SWIMM_OK;
}

static const char **sequence_pointers(const uint16_t *lengths, const char *codes, uint64_t count)
{
    const char **p = (const char **)malloc(count * sizeof(char *));
    if (!p) return NULL;
    const char *c = codes;
    for (uint64_t i = 0; i < count; ++i) { p[i] = c; c += lengths[i]; }
    return p;
}

int swimm_assemble_single_chunk(const uint16_t *lengths, const char *codes, uint64_t count, int vl, int block_size,
                                swimm_single_chunk *out)
{
    memset(out, 0, sizeof *out);
    int rc = check_args(lengths, codes, count, vl);
    if (rc) return rc;
    if (block_size <= 0) return FAIL(SWIMM_E_ARG, "SWIMM: block size must be positive.");
    uint64_t vc = 0;
    uint16_t *n = group_lengths(lengths, count, vl, &vc);
    uint16_t *nbbs = (uint16_t *)malloc(vc * sizeof(uint16_t));
    uint64_t *disp = (uint64_t *)malloc((vc + 1) * sizeof(uint64_t));
    const char **sp = sequence_pointers(lengths, codes, count);
    if (!n || !nbbs || !disp || !sp) { free(n); free(nbbs); free(disp); free(sp); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); }
    disp[0] = 0;
    for (uint64_t g = 0; g < vc; ++g) {
        disp[g + 1] = disp[g] + (uint64_t)n[g] * vl;
        nbbs[g] = (uint16_t)((n[g] + block_size - 1) / block_size);   /* sequences.c:687-688 */
    }
    char *b = NULL;
    if (posix_memalign((void **)&b, 64, disp[vc] ? disp[vc] : 64)) { free(n); free(nbbs); free(disp); free(sp); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); }
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t g = 0; g < vc; ++g) fill_group(b + disp[g], n[g], vl, lengths, sp, g * (uint64_t)vl, count);
    free(sp);
    out->vc = vc; out->vD = disp[vc]; out->b = b; out->n = n; out->nbbs = nbbs; out->disp = disp;
    return SWIMM_OK;
}

void swimm_single_chunk_free(swimm_single_chunk *c)
{
    if (!c) return;
    free(c->b); free(c->n); free(c->nbbs); free(c->disp);
    memset(c, 0, sizeof *c);
}

int swimm_assemble_chunks(const uint16_t *lengths, const char *codes, uint64_t count, int vl, uint64_t max_chunk_size,
                          swimm_chunks *out)
{
    memset(out, 0, sizeof *out);
    int rc = check_args(lengths, codes, count, vl);
    if (rc) return rc;
    uint64_t vc = 0;
    uint16_t *n = group_lengths(lengths, count, vl, &vc);
    uint64_t *gdisp = (uint64_t *)malloc((vc + 1) * sizeof(uint64_t));
    const char **sp = sequence_pointers(lengths, codes, count);
    uint32_t *rel = (uint32_t *)malloc(vc * sizeof(uint32_t));
    if (!n || !gdisp || !sp || !rel) { free(n); free(gdisp); free(sp); free(rel); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); }
    gdisp[0] = 0;
    for (uint64_t g = 0; g < vc; ++g) gdisp[g + 1] = gdisp[g] + (uint64_t)n[g] * vl;
    /* greedy split, sequences.c:533-557: keep adding groups while the running size is <= max */
    uint32_t cap = 16, cc = 0;
    uint32_t *cgroups = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint64_t i = 0;
    while (cgroups && i < vc) {
        uint64_t size = 0;
        uint32_t j = 0;
        while (i < vc && size <= max_chunk_size) {
            size += (uint64_t)n[i] * vl + sizeof(uint16_t) + sizeof(uint32_t);
            j++;
            i++;
        }
        if (cc == cap) { cap *= 2; cgroups = (uint32_t *)realloc(cgroups, cap * sizeof(uint32_t)); if (!cgroups) break; }
        cgroups[cc++] = j;
    }
    char *b = NULL;
    int bad = !cgroups || posix_memalign((void **)&b, 64, gdisp[vc] ? gdisp[vc] : 64);
    char **cb = (char **)malloc(cc * sizeof(char *));
    uint16_t **cn = (uint16_t **)malloc(cc * sizeof(uint16_t *));
    uint32_t **cd = (uint32_t **)malloc(cc * sizeof(uint32_t *));
    uint64_t *cvd = (uint64_t *)malloc(cc * sizeof(uint64_t));
    uint64_t *cfirst = (uint64_t *)malloc(cc * sizeof(uint64_t));
    if (bad || !cb || !cn || !cd || !cvd || !cfirst) {
        free(n); free(gdisp); free(sp); free(rel); free(cgroups); free(b); free(cb); free(cn); free(cd); free(cvd); free(cfirst);
        return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory.");
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t g = 0; g < vc; ++g) fill_group(b + gdisp[g], n[g], vl, lengths, sp, g * (uint64_t)vl, count);
    uint64_t g0 = 0;
    for (uint32_t c = 0; c < cc; ++c) {
        uint64_t base = gdisp[g0];
        uint64_t vd = gdisp[g0 + cgroups[c]] - base;
        if (vd > 0xFFFFFFFFull) {
            free(n); free(gdisp); free(sp); free(rel); free(cgroups); free(b); free(cb); free(cn); free(cd); free(cvd); free(cfirst);
            return FAIL(SWIMM_E_ARG, "SWIMM: a chunk of %llu bytes does not fit 32-bit displacements; lower the maximum chunk size.", (unsigned long long)vd);
        }
        cb[c] = b + base;
        cn[c] = n + g0;
        cd[c] = rel + g0;
        for (uint32_t j = 0; j < cgroups[c]; ++j) rel[g0 + j] = (uint32_t)(gdisp[g0 + j] - base);   /* sequences.c:577-579 */
        cvd[c] = vd;
        cfirst[c] = g0;
        g0 += cgroups[c];
    }
    out->vc = vc; out->vD = gdisp[vc]; out->chunk_count = cc; out->b_all = b; out->chunk_b = cb;
    out->chunk_groups = cgroups; out->chunk_n = cn; out->chunk_disp = cd; out->chunk_vD = cvd;
    out->chunk_first_group = cfirst; out->n_all_ = n; out->disp_all_ = rel;
    free(gdisp);
    free(sp);
    return SWIMM_OK;
}

void swimm_chunks_free(swimm_chunks *c)
{
    if (!c) return;
    free(c->b_all); free(c->chunk_b); free(c->chunk_groups); free(c->chunk_n); free(c->chunk_disp);
    free(c->chunk_vD); free(c->chunk_first_group); free(c->n_all_); free(c->disp_all_);
    memset(c, 0, sizeof *c);
}
