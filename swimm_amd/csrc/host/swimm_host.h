/*
 * swimm_host.h -- host-side (CPU, plain C) half of the MI355X SWIMM build: file formats, query and
 * database layout, top-r, substitution tables.  Used by the `swimm` program (main.c) and, through
 * ctypes, by the Python tests and bench.py (swimm_amd/host.py).
 *
 * Each function names the reference code whose behaviour it reproduces (file:line under
 * /root/reference); the formats are byte-compatible, the implementation is
 * not: one pass over an mmap'ed FASTA, counting sort by length, heap top-r.
 * Every function returns 0 on success or a non-zero status with swimm_host_last_error() set
 * (the reference prints and exit()s instead; main.c maps the statuses back to its exit codes).
 */
#ifndef SWIMM_HOST_H_INCLUDED
#define SWIMM_HOST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWIMM_VERSION "1.1.3-mi355x"
#define SWIMM_DUMMY_CODE 23   /* J, O, U            (DUMMY_ELEMENT recoded, sequences.h:17) */
#define SWIMM_PAD_CODE 24     /* lane padding        (PREPROCESSED_DUMMY_ELEMENT, sequences.h:18) */
#define SWIMM_SEQ_LEN_MULT 5  /* group length multiple (SEQ_LEN_MULT, sequences.h:19) */
#define SWIMM_DIDX_MAGIC 0x584449444d495753ull   /* "SWIMDIDX": <prefix>.didx, line offsets of <prefix>.desc (this build's sidecar) */

/* statuses (main.c exits with the reference's codes: 1 = memory, 2 = file, 3 = .desc) */
#define SWIMM_OK 0
#define SWIMM_E_NOMEM 1
#define SWIMM_E_FILE 2
#define SWIMM_E_DESC 3
#define SWIMM_E_FORMAT 4
#define SWIMM_E_ARG 5

const char *swimm_host_last_error(void);

/* 'A'..'Z' (any case) -> 0..23, in place; anything else -> 23.  sequences.c:164-175 / 393-402. */
void swimm_recode(char *s, size_t n);

/* ---- FASTA records (one pass, whole file in memory) ---- */
typedef struct {
    uint64_t count;
    uint64_t residues;      /* total residues */
    char **titles;          /* title lines WITH the leading '>' , NUL-terminated, no newline */
    char **seqs;            /* raw letters, not recoded, not NUL-terminated */
    uint32_t *lengths;
    char *arena_;           /* owns everything above */
} swimm_fasta;
int swimm_fasta_read(const char *path, swimm_fasta *out);
void swimm_fasta_free(swimm_fasta *f);

/* ---- preprocess: FASTA -> <out>.seq / <out>.info / <out>.desc   (preprocess_db, sequences.c:4-220)
 * stable ascending length sort; .info = "%ld %ld %d" (count, residues, longest title line + 2);
 * .desc = sorted title lines incl. '>'; .seq = uint16 lengths then recoded residues. */
int swimm_preprocess_db(const char *fasta_path, const char *out_prefix, uint64_t *n_sequences, uint64_t *n_residues);

/* ---- preprocessed database in memory (the read half of assemble_*_db, sequences.c:437-473) ---- */
typedef struct {
    uint64_t count, residues;
    int max_title_length;
    uint16_t *lengths;      /* ascending */
    char *codes;            /* concatenated, 0..23 */
    void *map_base;         /* the .seq file, mapped read-only: lengths and codes point into it (no copy of a 7 GB file; the
                             * alphabet check touches its pages on all threads).  NULL: lengths / codes are malloc'ed */
    uint64_t map_bytes;
} swimm_db;
int swimm_db_load(const char *prefix, swimm_db *out);
void swimm_db_free(swimm_db *db);
/* N title lines of <prefix>.desc (load_database_headers, sequences.c:736-767); only the requested
 * indices are materialised: titles[i] = line idx[i], '>' stripped, newline stripped.  With the sidecar <prefix>.didx that
 * swimm_preprocess_db writes (uint64 offset of every line; ignored unless it names this .desc's size and count) each title is
 * one positioned read; without it the file is walked up to the last wanted line. */
int swimm_db_titles(const char *prefix, uint64_t count, const int64_t *idx, uint64_t n_idx, char **titles_out);

/* ---- queries (load_query_sequences, sequences.c:223-423) ----
 * stable ascending length sort; pad_even != 0 appends one code 23 to odd-length queries and bumps m
 * (modes 0/2, sequences.c:378-387); pad_even == 0 keeps them (mode 1, sequences.c:347-364). */
typedef struct {
    uint64_t count, Q;      /* Q = sum of m */
    char *a;                /* recoded, concatenated */
    uint16_t *m;            /* lengths as stored in a */
    uint16_t *lengths;      /* real lengths */
    uint32_t *disp;         /* count+1 offsets into a */
    char **titles;          /* with leading '>' */
    char *arena_;
} swimm_queries;
int swimm_queries_load(const char *fasta_path, int pad_even, swimm_queries *out);
void swimm_queries_free(swimm_queries *q);

/* ---- lane-interleaved database (assemble_single_chunk_db, sequences.c:618-734) ----
 * group g = sequences [g*vl, (g+1)*vl); n[g] = longest member rounded up to x5; byte of position j,
 * lane k at disp[g] + j*vl + k, code 24 past a sequence's end / past the last sequence. */
typedef struct {
    uint64_t vc, vD;
    char *b;
    uint16_t *n, *nbbs;
    uint64_t *disp;         /* vc+1 */
} swimm_single_chunk;
int swimm_assemble_single_chunk(const uint16_t *lengths, const char *codes, uint64_t count, int vl, int block_size,
                                swimm_single_chunk *out);
void swimm_single_chunk_free(swimm_single_chunk *c);

/* ---- the same layout split into chunks (assemble_multiple_chunks_db, sequences.c:425-616):
 * groups are added to a chunk while its running size (bytes + 6 per group) is <= max_chunk_size,
 * so a chunk may overshoot by one group (sequences.c:533-557). */
typedef struct {
    uint64_t vc, vD;
    uint32_t chunk_count;
    char *b_all;                 /* one buffer; chunk_b[i] point into it */
    char **chunk_b;
    uint32_t *chunk_groups;      /* groups per chunk */
    uint16_t **chunk_n;
    uint32_t **chunk_disp;       /* relative to chunk start */
    uint64_t *chunk_vD;
    uint64_t *chunk_first_group; /* prefix sum of chunk_groups (chunk_accum..., MICsearch.c:46-49) */
    uint16_t *n_all_;
    uint32_t *disp_all_;
} swimm_chunks;
int swimm_assemble_chunks(const uint16_t *lengths, const char *codes, uint64_t count, int vl, uint64_t max_chunk_size,
                          swimm_chunks *out);
void swimm_chunks_free(swimm_chunks *c);

/* ---- top-r rows of the reference's sorted listing (sort_scores, utils.c:71-86 + swimm.c:151-160):
 * score descending, ties by LARGER index first.  O(n log r). */
void swimm_topr(const int32_t *scores, uint64_t n, uint32_t r, int32_t *out_scores, int64_t *out_idx);
/* k-way merge of per-shard top-r lists (scores/idx are [lists][r], -1 index = empty slot) */
void swimm_topr_merge(const int32_t *scores, const int64_t *idx, uint32_t lists, uint32_t r, int32_t *out_scores,
                      int64_t *out_idx);

/* ---- execution mode 0: search on the host CPU, same arguments as cpu_search_avx2_sp (CPUsearch.h:37-39)
 * plus the lane width `vl` the database was assembled with; scores[(q*vc + s)*vl + lane]. ---- */
int swimm_cpu_search(const char *a, const uint16_t *m, uint64_t query_count, const uint32_t *a_disp, const char *b,
                     const uint16_t *n, uint64_t vc, const uint64_t *b_disp, const char *submat, int open_gap,
                     int extend_gap, int n_threads, int block_size, int vl, int32_t *scores, double *work_time);

/* ---- substitution tables (submat.c:4-227): 768 bytes [query_code*32 + db_code], or NULL ---- */
const char *swimm_submat(const char *name);
const char *swimm_submat_label(const char *name);

/* ---- CPU placement of the host threads that serve one GPU (affinity.h: the device's sysfs local_cpulist, shared by whole
 * physical cores among the devices that name the same CPUs; an even share of the allowed CPUs when sysfs says nothing).
 * The reference leaves its per-device host threads to the OpenMP runtime (MICsearch.c:53).  plan: pure (sysfs_root = "/sys"
 * in production), returns the number of CPUs written or -1; allowed: the calling thread's current CPUs; apply:
 * sched_setaffinity of the calling thread (threads created afterwards inherit it). ---- */
int swimm_affinity_plan(const char *sysfs_root, const char *const *pci_bdf, int n_devices, int device, const int *allowed, int n_allowed,
                        int *out_cpus, int cap);
int swimm_affinity_allowed(int *out, int cap);
int swimm_affinity_apply(const int *cpus, int n);

/* wall clock (dwalltime, utils.c:89-97) */
double swimm_wtime(void);

#ifdef __cplusplus
}
#endif
#endif
