/* seqio.c -- FASTA input, preprocessed-database files, query batch (see swimm_host.h). */
#define _GNU_SOURCE
#include "swimm_host.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/time.h>

static __thread char g_err[512];

const char *swimm_host_last_error(void) { return g_err; }

int swimm_host_fail_(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define FAIL swimm_host_fail_

double swimm_wtime(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec / 1000000.0;
}

/* alphabet: A0 B1 C2 D3 E4 F5 G6 H7 I8 K9 L10 M11 N12 P13 Q14 R15 S16 T17 V18 W19 X20 Y21 Z22, J/O/U -> 23
 * (sequences.c:164-175: J,O,U -> 'Z'+1, then letters above J / O / U shift down by 1 / 2 / 3) */
static signed char g_code[256];
static int g_code_ready;
static void init_codes(void)
{
    if (g_code_ready) return;
    for (int c = 0; c < 256; ++c) g_code[c] = SWIMM_DUMMY_CODE;
    int next = 0;
    for (int c = 'A'; c <= 'Z'; ++c) {
        if (c == 'J' || c == 'O' || c == 'U') continue;
        g_code[c] = (signed char)next;
        g_code[tolower(c)] = (signed char)next;
        next++;
    }
    g_code_ready = 1;
}

void swimm_recode(char *s, size_t n)
{
    init_codes();
    for (size_t i = 0; i < n; ++i) s[i] = g_code[(unsigned char)s[i]];
}

/* ---- FASTA ------------------------------------------------------------------------------ */

static int read_whole(const char *path, char **buf, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening input sequence file '%s'.", path);
    if (fseeko(f, 0, SEEK_END) != 0) { fclose(f); return FAIL(SWIMM_E_FILE, "SWIMM: cannot seek in '%s'.", path); }
    off_t sz = ftello(f);
    rewind(f);
    char *b = (char *)malloc((size_t)sz + 2);
    if (!b) { fclose(f); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory for sequences."); }
    size_t got = fread(b, 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) { free(b); return FAIL(SWIMM_E_FILE, "SWIMM: short read on '%s'.", path); }
    b[got] = '\n';   /* sentinel: the last line always ends */
    b[got + 1] = 0;
    *buf = b;
    *len = got + 1;
    return SWIMM_OK;
}

int swimm_fasta_read(const char *path, swimm_fasta *out)
{
    memset(out, 0, sizeof *out);
    char *buf = NULL;
    size_t len = 0;
    int rc = read_whole(path, &buf, &len);
    if (rc) return rc;
    /* pass 1: count records */
    uint64_t count = 0;
    for (size_t i = 0; i < len;) {
        if (buf[i] == '>') count++;
        char *nl = (char *)memchr(buf + i, '\n', len - i);
        i = (size_t)(nl - buf) + 1;
    }
    char *seqbuf = (char *)malloc(len + 1);
    char **titles = (char **)malloc((count + 1) * sizeof(char *));
    char **seqs = (char **)malloc((count + 1) * sizeof(char *));
    uint32_t *lengths = (uint32_t *)malloc((count + 1) * sizeof(uint32_t));
    if (!seqbuf || !titles || !seqs || !lengths) {
        free(buf); free(seqbuf); free(titles); free(seqs); free(lengths);
        return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory for sequences.");
    }
    /* pass 2: titles stay in the file buffer (newline -> NUL), residues are compacted into seqbuf */
    uint64_t k = 0, total = 0;
    size_t w = 0;
    int in_record = 0;
    for (size_t i = 0; i < len;) {
        char *nl = (char *)memchr(buf + i, '\n', len - i);
        size_t e = (size_t)(nl - buf);
        if (buf[i] == '>') {
            size_t te = e;
            while (te > i && buf[te - 1] == '\r') te--;
            buf[te] = 0;
            titles[k] = buf + i;
            seqs[k] = seqbuf + w;
            lengths[k] = 0;
            k++;
            in_record = 1;
        } else if (in_record) {
            size_t le = e;
            while (le > i && (buf[le - 1] == '\r' || buf[le - 1] == ' ' || buf[le - 1] == '\t')) le--;
            const size_t n = le - i;
            if (n && !memchr(buf + i, ' ', n) && !memchr(buf + i, '\t', n)) {   // the usual case: one memcpy per line
                memcpy(seqbuf + w, buf + i, n);
                w += n; lengths[k - 1] += (uint32_t)n; total += n;
            } else {
                for (size_t j = i; j < le; ++j) {
                    unsigned char ch = (unsigned char)buf[j];
                    if (ch == ' ' || ch == '\t' || ch == '\r') continue;
                    seqbuf[w++] = (char)ch;
                    lengths[k - 1]++;
                    total++;
                }
            }
        }
        i = e + 1;
    }
    out->count = count;
    out->residues = total;
    out->titles = titles;
    out->seqs = seqs;
    out->lengths = lengths;
    /* arena_: both big buffers are released through one pointer pair */
    out->arena_ = buf;
    seqs[count] = seqbuf;   /* remember the residue buffer's base for free() */
    return SWIMM_OK;
}

void swimm_fasta_free(swimm_fasta *f)
{
    if (!f) return;
    if (f->seqs) free(f->seqs[f->count]);
    free(f->arena_);
    free(f->titles);
    free(f->seqs);
    free(f->lengths);
    memset(f, 0, sizeof *f);
}

/* stable ascending order by length: counting sort (same order as the reference's merge sort, which
 * takes the left element on <=, sequences.c:780) */
static uint64_t *stable_length_order(const uint32_t *lengths, uint64_t n)
{
    uint64_t *cnt = (uint64_t *)calloc(65537, sizeof(uint64_t));
    uint64_t *order = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!cnt || !order) { free(cnt); free(order); return NULL; }
    for (uint64_t i = 0; i < n; ++i) cnt[lengths[i] + 1]++;
    for (int l = 0; l < 65536; ++l) cnt[l + 1] += cnt[l];
    for (uint64_t i = 0; i < n; ++i) order[cnt[lengths[i]]++] = i;
    free(cnt);
    return order;
}

static int check_lengths(const swimm_fasta *f, const char *what)
{
    for (uint64_t i = 0; i < f->count; ++i)
        if (f->lengths[i] > 65535)
            return FAIL(SWIMM_E_FORMAT, "SWIMM: %s sequence %llu ('%.60s') has %u residues; the format stores lengths in 16 bits (max 65535).",
                        what, (unsigned long long)i, f->titles[i], f->lengths[i]);
    return SWIMM_OK;
}

/* preprocess_db (sequences.c:4-220) at the scale SURVEY 8(f2) names: the reference walks the FASTA three times with fgets
 * and mallocs every sequence and every title (sequences.c:28-50,55-58,64-80,88-91,102-119); here the file is mapped and
 * walked ONCE, front to back: titles go into one arena, residues -- recoded on the way -- into another, both reserved as
 * address space and touched only as far as they fill, and the pages of the file are handed back as the walk leaves them
 * behind.  Peak memory is therefore the size of the output (.seq + .desc) plus 32 bytes per record, whatever the size of
 * the input (round 2 held the file and a compacted copy: twice the FASTA).  Same bytes out: stable counting sort by length. */
typedef struct { uint64_t *title_off, *seq_off; uint32_t *title_len, *len; uint64_t n, cap; } rec_index;

static int rec_push(rec_index *x, uint64_t toff, uint32_t tlen, uint64_t soff)
{
    if (x->n == x->cap) {
        const uint64_t cap = x->cap ? x->cap * 2 : (1u << 16);
        uint64_t *a = (uint64_t *)realloc(x->title_off, cap * sizeof(uint64_t));
        if (a) x->title_off = a;
        uint64_t *b = (uint64_t *)realloc(x->seq_off, cap * sizeof(uint64_t));
        if (b) x->seq_off = b;
        uint32_t *c = (uint32_t *)realloc(x->title_len, cap * sizeof(uint32_t));
        if (c) x->title_len = c;
        uint32_t *d = (uint32_t *)realloc(x->len, cap * sizeof(uint32_t));
        if (d) x->len = d;
        if (!a || !b || !c || !d) return 1;
        x->cap = cap;
    }
    x->title_off[x->n] = toff; x->title_len[x->n] = tlen; x->seq_off[x->n] = soff; x->len[x->n] = 0;
    x->n++;
    return 0;
}

int swimm_preprocess_db(const char *fasta_path, const char *out_prefix, uint64_t *n_sequences, uint64_t *n_residues)
{
    init_codes();
    int fdin = open(fasta_path, O_RDONLY);
    if (fdin < 0) return FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening input sequence file '%s'.", fasta_path);
    struct stat sb;
    if (fstat(fdin, &sb) != 0) { close(fdin); return FAIL(SWIMM_E_FILE, "SWIMM: cannot stat '%s'.", fasta_path); }
    const size_t len = (size_t)sb.st_size;
    if (len == 0) { close(fdin); return FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' holds no FASTA record.", fasta_path); }
    const char *buf = (const char *)mmap(NULL, len, PROT_READ, MAP_PRIVATE, fdin, 0);
    close(fdin);
    if (buf == MAP_FAILED) return FAIL(SWIMM_E_FILE, "SWIMM: cannot map '%s'.", fasta_path);
    (void)madvise((void *)buf, len, MADV_SEQUENTIAL);
    /* the two arenas: address space for the worst case (everything residues / everything titles), memory as they fill */
    char *res = (char *)mmap(NULL, len + 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    char *tit = (char *)mmap(NULL, len + 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    rec_index x = {NULL, NULL, NULL, NULL, 0, 0};
    int rc = SWIMM_OK;
    uint64_t *order = NULL;
    FILE *fd = NULL, *fs = NULL;
    char *out = NULL;
    uint16_t *l16 = NULL;
    if (res == MAP_FAILED || tit == MAP_FAILED) { rc = FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory for sequences."); goto done; }
    {
        uint64_t w = 0, tw = 0;
        int in_record = 0;
        const size_t page = 4096, drop_every = (size_t)64 << 20;
        size_t dropped = 0;
        for (size_t i = 0; i < len;) {
            const char *nl = (const char *)memchr(buf + i, '\n', len - i);
            const size_t e = nl ? (size_t)(nl - buf) : len;         /* (the last line may lack its newline) */
            if (buf[i] == '>') {
                size_t te = e;
                while (te > i && buf[te - 1] == '\r') te--;
                if (rec_push(&x, tw, (uint32_t)(te - i), w)) { rc = FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory for sequences."); goto done; }
                memcpy(tit + tw, buf + i, te - i);
                tw += te - i;
                in_record = 1;
            } else if (in_record) {
                size_t le = e;
                while (le > i && (buf[le - 1] == '\r' || buf[le - 1] == ' ' || buf[le - 1] == '\t')) le--;
                const size_t n = le - i;
                uint64_t w0 = w;
                if (n && !memchr(buf + i, ' ', n) && !memchr(buf + i, '\t', n)) {   /* the usual case: one memcpy per line */
                    memcpy(res + w, buf + i, n);
                    w += n;
                } else {
                    for (size_t j = i; j < le; ++j) {
                        const unsigned char ch = (unsigned char)buf[j];
                        if (ch == ' ' || ch == '\t' || ch == '\r') continue;
                        res[w++] = (char)ch;
                    }
                }
                for (uint64_t k = w0; k < w; ++k) res[k] = g_code[(unsigned char)res[k]];    /* sequences.c:164-175, while the line is in cache */
                const uint64_t L = (uint64_t)x.len[x.n - 1] + (w - w0);
                if (L > 65535) {
                    rc = FAIL(SWIMM_E_FORMAT, "SWIMM: database sequence %llu ('%.60s') has more than 65535 residues; the format stores lengths in 16 bits.",
                              (unsigned long long)(x.n - 1), tit + x.title_off[x.n - 1]);
                    goto done;
                }
                x.len[x.n - 1] = (uint32_t)L;
            }
            i = e + 1;
            if (i - dropped >= drop_every) {                           /* the walk is done with these pages of the file */
                const size_t upto = i / page * page;
                (void)madvise((void *)(buf + dropped), upto - dropped, MADV_DONTNEED);
                dropped = upto;
            }
        }
        if (x.n == 0) { rc = FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' holds no FASTA record.", fasta_path); goto done; }
        (void)munmap((void *)buf, len);
        buf = NULL;
        order = stable_length_order(x.len, x.n);
        if (!order) { rc = FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); goto done; }
        char name[4096];
        size_t max_title = 0;
        /* .desc : title lines (with '>') in sorted order, sequences.c:128-141 */
        snprintf(name, sizeof name, "%s.desc", out_prefix);
        fd = fopen(name, "wb");
        if (!fd) { rc = FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening sequence header file."); goto done; }
        (void)setvbuf(fd, NULL, _IOFBF, (size_t)4 << 20);
        /* ... and beside it <out>.didx, this build's sidecar (not one of the reference's three files, which stay byte-identical;
         * optional on read): where every title line starts, so that a report's r titles are r seeks instead of a walk through
         * the whole file (the reference reads all N titles, sequences.c:757-761).  Header: "SWIMDIDX", N, size of .desc. */
        snprintf(name, sizeof name, "%s.didx", out_prefix);
        FILE *fx = fopen(name, "wb");
        uint64_t at = 0;
        if (fx) {
            (void)setvbuf(fx, NULL, _IOFBF, (size_t)4 << 20);
            const uint64_t hdr[3] = {SWIMM_DIDX_MAGIC, x.n, 0};
            fwrite(hdr, sizeof hdr, 1, fx);
        }
        for (uint64_t i = 0; i < x.n; ++i) {
            const uint64_t s = order[i];
            if (x.title_len[s] > max_title) max_title = x.title_len[s];
            fwrite(tit + x.title_off[s], 1, x.title_len[s], fd);
            fputc('\n', fd);
            if (fx) fwrite(&at, sizeof at, 1, fx);
            at += x.title_len[s] + 1;
        }
        if (fx) {                      /* (a sidecar that could not be written whole is removed: the reader then walks the file) */
            int bad = ferror(fx);
            if (!bad) bad = fseek(fx, 16, SEEK_SET) != 0 || fwrite(&at, sizeof at, 1, fx) != 1;
            bad = fclose(fx) != 0 || bad;
            if (bad) { snprintf(name, sizeof name, "%s.didx", out_prefix); (void)remove(name); }
        }
        if (ferror(fd)) { rc = FAIL(SWIMM_E_FILE, "SWIMM: write error on '%s.desc'.", out_prefix); goto done; }
        fclose(fd); fd = NULL;
        (void)munmap(tit, len + 1);
        tit = (char *)MAP_FAILED;
        /* .info : "%ld %ld %d", no newline; max title = longest line incl. '>' + newline + 1 (sequences.c:36,187) */
        snprintf(name, sizeof name, "%s.info", out_prefix);
        FILE *fi = fopen(name, "wb");
        if (!fi) { rc = FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening info file."); goto done; }
        fprintf(fi, "%ld %ld %d", (long)x.n, (long)w, (int)(max_title + 2));
        fclose(fi);
        /* .seq : uint16 lengths, then recoded residues, both in sorted order (sequences.c:201-205) */
        snprintf(name, sizeof name, "%s.seq", out_prefix);
        fs = fopen(name, "wb");
        if (!fs) { rc = FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening sequence file."); goto done; }
        l16 = (uint16_t *)malloc(x.n * sizeof(uint16_t));
        const size_t blk = (size_t)64 << 20;
        out = (char *)malloc(blk);
        if (!l16 || !out) { rc = FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); goto done; }
        for (uint64_t i = 0; i < x.n; ++i) l16[i] = (uint16_t)x.len[order[i]];
        fwrite(l16, sizeof(uint16_t), x.n, fs);
        size_t fill = 0;                                              /* gather the sorted residues and write them in large blocks */
        for (uint64_t i = 0; i < x.n; ++i) {
            const uint64_t s = order[i];
            size_t L = x.len[s], done_ = 0;
            while (done_ < L) {
                const size_t n = L - done_ < blk - fill ? L - done_ : blk - fill;
                memcpy(out + fill, res + x.seq_off[s] + done_, n);
                fill += n; done_ += n;
                if (fill == blk) { fwrite(out, 1, fill, fs); fill = 0; }
            }
        }
        if (fill) fwrite(out, 1, fill, fs);
        if (ferror(fs)) { rc = FAIL(SWIMM_E_FILE, "SWIMM: write error on '%s.seq'.", out_prefix); goto done; }
        if (n_sequences) *n_sequences = x.n;
        if (n_residues) *n_residues = w;
    }
done:
    if (fd) fclose(fd);
    if (fs) fclose(fs);
    free(out); free(l16); free(order);
    free(x.title_off); free(x.seq_off); free(x.title_len); free(x.len);
    if (buf) (void)munmap((void *)buf, len);
    if (res != MAP_FAILED) (void)munmap(res, len + 1);
    if (tit != MAP_FAILED) (void)munmap(tit, len + 1);
    return rc;
}

/* ---- preprocessed database ------------------------------------------------------------ */

int swimm_db_load(const char *prefix, swimm_db *out)
{
    memset(out, 0, sizeof *out);
    char name[4096];
    snprintf(name, sizeof name, "%s.info", prefix);
    FILE *fi = fopen(name, "r");
    if (!fi) return FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening info file.");
    long cnt = 0, D = 0;
    int mt = 0;
    int got = fscanf(fi, "%ld %ld %d", &cnt, &D, &mt);
    fclose(fi);
    if (got != 3 || cnt <= 0 || D < 0) return FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' is not a valid info file.", name);
    snprintf(name, sizeof name, "%s.seq", prefix);
    /* The file is MAPPED, not read: lengths and codes point into the page cache (the reference reads it into malloc'ed arrays,
     * sequences.c:430-470: 1.4 s of copying for 7 GB), and the first pass over the codes -- the alphabet check below, on all
     * threads -- is what brings the pages in. */
    const int fd = open(name, O_RDONLY);
    if (fd < 0) return FAIL(SWIMM_E_FILE, "SWIMM: An error occurred while opening sequence file.");
    struct stat sb;
    const uint64_t need = (uint64_t)cnt * sizeof(uint16_t) + (uint64_t)D;
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size < need) { close(fd); return FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' is truncated.", name); }
    void *map = mmap(NULL, (size_t)need, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory.");
    (void)madvise(map, (size_t)need, MADV_WILLNEED);
    uint16_t *lengths = (uint16_t *)map;
    char *codes = (char *)map + (size_t)cnt * sizeof(uint16_t);
#define free_db_arrays() munmap(map, (size_t)need)
    uint64_t sum = 0;
    for (long i = 0; i < cnt; ++i) {
        sum += lengths[i];
        if (i && lengths[i] < lengths[i - 1]) { free_db_arrays(); return FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' is not sorted by length.", name); }
    }
    if (sum != (uint64_t)D) { free_db_arrays(); return FAIL(SWIMM_E_FORMAT, "SWIMM: lengths in '%s' do not add up to %ld residues.", name, D); }
    long bad_at = -1;   /* a residue code outside the alphabet, if any (this scan is the slow part of loading 7e9 residues) */
#pragma omp parallel for schedule(static) reduction(max : bad_at)
    for (long i = 0; i < D; ++i)
        if ((unsigned char)codes[i] > SWIMM_DUMMY_CODE && i > bad_at) bad_at = i;
    if (bad_at >= 0) {
        const int bad = codes[bad_at];
        free_db_arrays();
        return FAIL(SWIMM_E_FORMAT, "SWIMM: residue code %d in '%s' is outside 0..23.", bad, name);
    }
    out->count = (uint64_t)cnt;
    out->residues = (uint64_t)D;
    out->max_title_length = mt;
    out->lengths = lengths;
    out->codes = codes;
    out->map_base = map;
    out->map_bytes = need;
    return SWIMM_OK;
#undef free_db_arrays
}

void swimm_db_free(swimm_db *db)
{
    if (!db) return;
    if (db->map_base) munmap(db->map_base, (size_t)db->map_bytes);
    else { free(db->lengths); free(db->codes); }
    memset(db, 0, sizeof *db);
}

typedef struct { int64_t line; uint64_t pos; } want_t;
static int want_cmp(const void *a, const void *b)
{
    int64_t x = ((const want_t *)a)->line, y = ((const want_t *)b)->line;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* title line `line` through the sidecar: [off[line], off[line + 1]) of .desc, '>' and the line end stripped */
static char *title_at(int fd_desc, const uint64_t *off, uint64_t count, uint64_t desc_size, uint64_t line)
{
    const uint64_t b = off[line], e = line + 1 < count ? off[line + 1] : desc_size;
    if (e < b || e > desc_size || e - b > ((uint64_t)1 << 26)) return NULL;
    char *t = (char *)malloc(e - b + 1);
    if (!t) return NULL;
    if (e > b && pread(fd_desc, t, e - b, (off_t)b) != (ssize_t)(e - b)) { free(t); return NULL; }
    size_t n = e - b;
    while (n > 0 && (t[n - 1] == '\n' || t[n - 1] == '\r')) n--;
    t[n] = 0;
    if (n > 0 && t[0] == '>') memmove(t, t + 1, n);
    return t;
}

int swimm_db_titles(const char *prefix, uint64_t count, const int64_t *idx, uint64_t n_idx, char **titles_out)
{
    /* Only the wanted lines are copied: the file is mapped and walked with memchr up to the last wanted line (the
     * reference reads every title into its own malloc, sequences.c:757-761; at 3.5e7 titles that is the slowest part
     * of printing a report). */
    char name[4096];
    snprintf(name, sizeof name, "%s.desc", prefix);
    int fd = open(name, O_RDONLY);
    if (fd < 0) return FAIL(SWIMM_E_DESC, "SWIMM: An error occurred while opening sequence description file.");
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return FAIL(SWIMM_E_DESC, "SWIMM: cannot stat '%s'.", name); }
    want_t *w = (want_t *)malloc((n_idx ? n_idx : 1) * sizeof(want_t));
    if (!w) { close(fd); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); }
    for (uint64_t i = 0; i < n_idx; ++i) {
        if (idx[i] < 0 || (uint64_t)idx[i] >= count) { free(w); close(fd); return FAIL(SWIMM_E_ARG, "SWIMM: title index %lld outside the database.", (long long)idx[i]); }
        w[i].line = idx[i];
        w[i].pos = i;
        titles_out[i] = NULL;
    }
    /* the sidecar <prefix>.didx (swimm_preprocess_db): N line offsets, valid only for a .desc of the size it names.  r titles are
     * r positioned reads of a few dozen bytes -- at 3.5e7 titles the walk below reads the whole 3 GB file for a hit near its end. */
    {
        snprintf(name, sizeof name, "%s.didx", prefix);
        const int fx = open(name, O_RDONLY);
        if (fx >= 0) {
            uint64_t hdr[3] = {0, 0, 0};
            struct stat sx;
            const int ok = fstat(fx, &sx) == 0 && pread(fx, hdr, sizeof hdr, 0) == (ssize_t)sizeof hdr && hdr[0] == SWIMM_DIDX_MAGIC && hdr[1] == count &&
                           hdr[2] == (uint64_t)sb.st_size && (uint64_t)sx.st_size == sizeof hdr + count * sizeof(uint64_t);
            const uint64_t *off = ok ? (const uint64_t *)mmap(NULL, (size_t)sx.st_size, PROT_READ, MAP_PRIVATE, fx, 0) : NULL;
            close(fx);
            if (ok && off != MAP_FAILED) {
                int bad = 0;
                for (uint64_t i = 0; i < n_idx && !bad; ++i) {
                    titles_out[i] = title_at(fd, off + 3, count, (uint64_t)sb.st_size, (uint64_t)idx[i]);
                    bad = titles_out[i] == NULL;
                }
                munmap((void *)off, (size_t)sx.st_size);
                if (!bad) { free(w); close(fd); return SWIMM_OK; }
                for (uint64_t i = 0; i < n_idx; ++i) { free(titles_out[i]); titles_out[i] = NULL; }      /* (damaged sidecar: walk the file) */
            }
        }
        snprintf(name, sizeof name, "%s.desc", prefix);
    }
    qsort(w, n_idx, sizeof(want_t), want_cmp);
    const size_t len = (size_t)sb.st_size;
    const char *base = len ? (const char *)mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0) : NULL;
    close(fd);
    if (len && base == MAP_FAILED) { free(w); return FAIL(SWIMM_E_DESC, "SWIMM: cannot map '%s'.", name); }
    if (len) (void)madvise((void *)base, len, MADV_SEQUENTIAL);
    int64_t ln = 0;
    uint64_t k = 0;
    size_t pos = 0;
    while (k < n_idx && pos < len) {
        const char *nl = (const char *)memchr(base + pos, '\n', len - pos);
        const size_t e = nl ? (size_t)(nl - base) : len;
        if (ln == w[k].line) {
            size_t b0 = pos, e0 = e;
            while (e0 > b0 && base[e0 - 1] == '\r') e0--;
            if (e0 > b0 && base[b0] == '>') b0++;
            char *t = (char *)malloc(e0 - b0 + 1);
            if (!t) { if (len) munmap((void *)base, len); free(w); return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory."); }
            memcpy(t, base + b0, e0 - b0);
            t[e0 - b0] = 0;
            titles_out[w[k].pos] = t;
            k++;
            while (k < n_idx && w[k].line == ln) titles_out[w[k++].pos] = strdup(t);
        }
        ln++;
        pos = e + 1;
    }
    if (len) munmap((void *)base, len);
    free(w);
    if (k < n_idx) return FAIL(SWIMM_E_DESC, "SWIMM: '%s' has fewer than %llu lines.", name, (unsigned long long)count);
    return SWIMM_OK;
}

/* ---- queries -------------------------------------------------------------------------- */

int swimm_queries_load(const char *fasta_path, int pad_even, swimm_queries *out)
{
    memset(out, 0, sizeof *out);
    swimm_fasta f;
    int rc = swimm_fasta_read(fasta_path, &f);
    if (rc) return rc;
    if (f.count == 0) { swimm_fasta_free(&f); return FAIL(SWIMM_E_FORMAT, "SWIMM: '%s' holds no FASTA record.", fasta_path); }
    if ((rc = check_lengths(&f, "query"))) { swimm_fasta_free(&f); return rc; }
    for (uint64_t i = 0; i < f.count; ++i) {
        if (f.lengths[i] == 0) {
            rc = FAIL(SWIMM_E_FORMAT, "SWIMM: query %llu ('%.60s') is empty.", (unsigned long long)i, f.titles[i]);
            swimm_fasta_free(&f);
            return rc;
        }
        if (pad_even && f.lengths[i] == 65535) { swimm_fasta_free(&f); return FAIL(SWIMM_E_FORMAT, "SWIMM: query %llu cannot be even-padded past 65535.", (unsigned long long)i); }
    }
    uint64_t *order = stable_length_order(f.lengths, f.count);
    uint64_t Q = 0, tbytes = 0;
    for (uint64_t i = 0; i < f.count; ++i) {
        Q += f.lengths[i] + (pad_even ? (f.lengths[i] & 1) : 0);
        tbytes += strlen(f.titles[i]) + 1;
    }
    char *a = (char *)malloc(Q + 64);
    uint16_t *m = (uint16_t *)malloc(f.count * sizeof(uint16_t));
    uint16_t *real = (uint16_t *)malloc(f.count * sizeof(uint16_t));
    uint32_t *disp = (uint32_t *)malloc((f.count + 1) * sizeof(uint32_t));
    char **titles = (char **)malloc(f.count * sizeof(char *));
    char *arena = (char *)malloc(tbytes + 1);
    if (!order || !a || !m || !real || !disp || !titles || !arena) {
        free(order); free(a); free(m); free(real); free(disp); free(titles); free(arena);
        swimm_fasta_free(&f);
        return FAIL(SWIMM_E_NOMEM, "SWIMM: An error occurred while allocating memory for query sequences.");
    }
    uint64_t pos = 0, tp = 0;
    for (uint64_t k = 0; k < f.count; ++k) {
        uint64_t s = order[k];
        uint32_t L = f.lengths[s];
        disp[k] = (uint32_t)pos;
        memcpy(a + pos, f.seqs[s], L);
        swimm_recode(a + pos, L);
        real[k] = (uint16_t)L;
        m[k] = (uint16_t)L;
        if (pad_even && (L & 1)) { a[pos + L] = SWIMM_DUMMY_CODE; m[k]++; }   /* sequences.c:382-385 */
        pos += m[k];
        size_t tl = strlen(f.titles[s]) + 1;
        memcpy(arena + tp, f.titles[s], tl);
        titles[k] = arena + tp;
        tp += tl;
    }
    disp[f.count] = (uint32_t)pos;
    out->count = f.count;
    out->Q = Q;
    out->a = a; out->m = m; out->lengths = real; out->disp = disp; out->titles = titles; out->arena_ = arena;
    free(order);
    swimm_fasta_free(&f);
    return SWIMM_OK;
}

void swimm_queries_free(swimm_queries *q)
{
    if (!q) return;
    free(q->a); free(q->m); free(q->lengths); free(q->disp); free(q->titles); free(q->arena_);
    memset(q, 0, sizeof *q);
}
