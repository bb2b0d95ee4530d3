/*
 * swimm_hip.h -- C-ABI of libswimm_hip.so, the MI355X (gfx950) search back-end.
 *
 * This is the drop-in boundary for the SWIMM search hot path: plain C types only, no HIP or
 * torch types, loadable with dlopen()/dlsym() (swimm_amd/csrc/host/hip_loader.c) or ctypes
 * (swimm_amd/hip_backend.py).  The reference has no plugin layer; its seam is the set of plain
 * C search functions main() calls (swimm.c:66-119).  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference).
 *
 * Conventions that differ from the reference on purpose:
 *   - every function returns an int status (0 = ok) instead of printf + exit(); the message is
 *     available from swimm_hip_last_error() (thread-local);
 *   - the database stays resident in HBM between searches (288 GB per GPU) instead of being
 *     re-sent chunk by chunk on every offload (MICsearch.c:85-91);
 *   - a missing GPU / missing library is an error, never a silent CPU fallback.
 *
 * Data layouts accepted are exactly the reference's own:
 *   queries  : `a` = recoded residues (0..23) of all queries concatenated, `m` = lengths as
 *              stored in `a` (even-padded or not), `a_disp` = offsets, ascending length
 *              (load_query_sequences, sequences.c:223-423);
 *   database : chunks from assemble_multiple_chunks_db (sequences.c:425-616): per chunk the
 *              lane-interleaved bytes b (byte of group g, position j, lane k at
 *              b_disp[g] + j*vl + k, pad code 24), group lengths n[], offsets b_disp[];
 *              `vl` must divide 128.
 *   scores   : int32 scores[q * score_stride + sorted_db_index]  (CPUsearch.c:548,
 *              MICsearch.c:333-334 use score_stride = vect_sequences_db_count * vl).
 */
#ifndef SWIMM_HIP_H_INCLUDED
#define SWIMM_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWIMM_HIP_ABI_VERSION 1
#define SWIMM_HIP_SUBMAT_BYTES 768 /* 24 rows x 32 cols int8, submat.h:4-6 */

typedef struct swimm_hip_ctx swimm_hip_ctx; /* one per GPU; not thread-safe, one host thread per ctx */

/* ABI version of the loaded library. */
int swimm_hip_abi_version(void);

/* Message of the last failing call made by the calling thread ("" if none). */
const char *swimm_hip_last_error(void);

/* Number of visible gfx950 devices (replaces the -x num_mics probe; 0 and an error string when
 * no GPU is usable). */
int swimm_hip_device_count(void);

/* PCI address of a visible device as sysfs spells it ("0000:0c:00.0"), so that a multi-process caller can tell which
 * physical devices its ranks sit on and which CPUs are local to each.  No reference counterpart. */
int swimm_hip_device_pci_bus_id(int device, char *buf, size_t buf_len);

/* Binds the CALLING host thread -- and the threads it creates from here on: the context's uploader thread, an OpenMP team --
 * to the CPUs local to `device`, one of the `num_devices` devices 0..num_devices-1 this process drives: the device's sysfs
 * local_cpulist, shared by whole physical cores with the other devices that name the same CPUs; an even share of the
 * thread's allowed CPUs when sysfs says nothing (swimm_amd/csrc/host/affinity.h).  Call it from the device's host thread
 * before swimm_hip_create.  swimm_hip_search_chunks does so for its per-device threads.  `cpulist_out` (may be NULL) gets
 * the CPUs in "0-7,128-135" form.  SWIMM_HIP_BIND=0 in the environment makes it a no-op (a scheduler already placed the
 * process).  The reference leaves its per-MIC host threads to the OpenMP runtime (MICsearch.c:53). */
int swimm_hip_bind_host_thread(int device, int num_devices, char *cpulist_out, size_t cpulist_len);

/* Per-device state: stream, scratch.  Replaces the per-MIC host thread prologue
 * `#pragma offload_transfer ... ALLOC` (MICsearch.c:53-71). */
int swimm_hip_create(int device, swimm_hip_ctx **out);

/* Replaces the `... FREE` epilogue (MICsearch.c:340-346). */
void swimm_hip_destroy(swimm_hip_ctx *ctx);

/* Queries + substitution matrix + gap penalties, sent once (transfer X1, MICsearch.c:34-36 and
 * 67-71: a, m, a_disp, submat, queryProfiles).  Requires 0 <= open_gap, extend_gap and
 * open_gap + extend_gap <= 127 (the reference stores the sum in an int8 lane, CPUsearch.c:518). */
int swimm_hip_set_queries(swimm_hip_ctx *ctx, const char *a, const uint16_t *m, const uint32_t *a_disp,
                          uint32_t query_count, const char *submat, int open_gap, int extend_gap);

/* One database chunk in the reference's chunk layout (transfer X2 "in" half, MICsearch.c:85-88:
 * ptr_chunk_b / ptr_chunk_n / ptr_chunk_b_disp).  `first_group` is the chunk's offset in
 * vector groups within the whole database (chunk_accum_vect_sequences_db_count,
 * MICsearch.c:46-49), so that sorted index = (first_group + g) * vl + lane.  The chunk is
 * re-tiled on the device and stays resident until swimm_hip_clear_db(). */
int swimm_hip_add_chunk(swimm_hip_ctx *ctx, const char *b, uint64_t vD, const uint16_t *n,
                        const uint32_t *b_disp, uint32_t group_count, uint32_t vl, uint64_t first_group);

/* The same database content without the host-side lane interleave: `n_seq` consecutive sequences of the sorted
 * database exactly as the .seq file stores them (sequences.c:201-205) -- `lengths[i]` residues each, codes
 * concatenated -- starting at sorted index `first_seq`.  Replaces assemble_multiple_chunks_db (sequences.c:425-616)
 * + the copy above for a caller that has the preprocessed database in memory: the device builds its layout itself.
 * A slab must stay below 4 GiB of residues; every slab but the last must hold a multiple of 128 sequences.
 * Both arrays are read when the slab is copied: inside this call by default; with the option "lazy_upload" by the next search,
 * until which BOTH `lengths` and `codes` must stay valid. */
int swimm_hip_add_sequences(swimm_hip_ctx *ctx, const uint16_t *lengths, const char *codes, uint64_t n_seq, uint64_t first_seq);

int swimm_hip_clear_db(swimm_hip_ctx *ctx);

/* The search itself (kernels K2-K5 + score scatter X3, MICsearch.c:91-334; same result as
 * cpu_search_avx2_sp, CPUsearch.c:482-967).  For every query q and every resident sequence
 * writes the exact Gotoh local-alignment score to scores[q*score_stride + sorted_index];
 * entries of sequences that are not resident on this device are left untouched.
 * *work_time (seconds, may be NULL) brackets kernels + D2H like the reference's workTime. */
int swimm_hip_search(swimm_hip_ctx *ctx, int32_t *scores, uint64_t score_stride, double *work_time);

/* Same search, but only the first r rows of the reference's sorted listing per query come back
 * (replaces sort_scores + print loop, utils.c:71-86 / swimm.c:151-160): order is score
 * descending, ties by larger sorted index first.  top_scores / top_index are [query_count][r];
 * rows beyond the number of resident sequences are filled with score -1, index -1.
 * `n_valid` = number of real sequences in the whole database (indices >= n_valid are padding
 * lanes and never reported). */
int swimm_hip_search_topr(swimm_hip_ctx *ctx, uint32_t r, uint64_t n_valid, int32_t *top_scores,
                          int64_t *top_index, double *work_time);

/* Statistics of the last search on this ctx (all optional, pass NULL to skip):
 * kernel_ms = device time of the DP kernels (HIP events), cells = DP cells computed including
 * padding, promoted = (query, sequence) pairs recomputed in int32, launches = DP kernel launches. */
int swimm_hip_last_stats(swimm_hip_ctx *ctx, double *kernel_ms, uint64_t *cells, uint64_t *promoted,
                         uint32_t *launches);

/* Launch plan the last search used for query `q` (after the ascending-length order of set_queries):
 * rows of the query held per wavefront, wavefronts per workgroup, passes over the database. */
int swimm_hip_last_plan(swimm_hip_ctx *ctx, uint32_t q, int *rows_per_wave, int *waves, int *passes);

/* With the option "time_launches" set before the search: the sum of the durations of the search's pipeline-kernel
 * launches, each measured by HIP events on the stream the launch ran on (what a kernel trace reports per dispatch;
 * launches that share the chip on two streams each count with their own, longer, duration), and their number.
 * Measurement aid for `roofline.kernel_ms` in bench.py; no reference counterpart. */
int swimm_hip_last_launch_ms(swimm_hip_ctx *ctx, double *sum_ms, uint32_t *launches);

/* Name of the dominant DP kernel of that plan, as the code object carries it and rocprofv3 lists it (demangled, e.g.
 * "void swimm::sw_pipe_kernel<24, 2, true, false, false>(swimm::PipeParams)"), so that a profile can be matched to a search
 * without guessing.  Measurement aid only (row (d) of SURVEY.md section 8); no reference counterpart. */
int swimm_hip_last_kernel_name(swimm_hip_ctx *ctx, uint32_t q, char *buf, size_t buf_len);

/* Tuning knobs (optional; 20 keys).  What stays is what a parity test uses as a decomposition cut (the result must not depend on it), what
 * the `swimm` program maps a command-line flag to, and the measurement aid; everything else the planner decides.  key:
 *   "rows_per_wave"  0 = launch shape chosen per query (default); 8, 12, ... 36 forces the rows per wavefront
 *                    (the int16 / int32 first tiers have 16, 24, 32 only)
 *   "waves"          0 = chosen per query; 1..16 forces the wavefronts per workgroup;  "max_waves" caps them
 *   "wg_limit"       caps the persistent workgroups of a launch (0 = what the chip holds)
 *   "f16"            1 = default: packed binary16 first tier, exact below 2048 - P * extend with P = 4 * max(1, floor(32 / extend)) columns
 *                    between two renormalisations (1920 for extend 2, 1928 for 3, 1923 for 5; with an extend penalty above 237 that
 *                    limit falls below 1100 and the int16 tier is the first), int16 and int32 re-runs above; 0 = packed int16 first tier
 *   "force_i32"      1 = everything in int32 (one sequence per lane)
 *   "tail_mode"      0 = auto: unusually long groups go through the lane-systolic kernel, 1 = every group, 2 = none
 *   "tail_frac"      a group is "unusually long" above this percentage of a CU's mean load (default 30) ...
 *   "tail_cap"       ... and joins the lane-systolic tail as long as the tail stays below this many per mille of the search's cells
 *                    (default 25; 0 = no cap)
 *   "dynamic"        1 = default: workgroups pull groups from a global queue; 0 = static longest-first partition
 *   "resident"       group-resident batch launches (ONE launch per launch shape for all the queries that share it: each workgroup
 *                    takes a (group, query) item through all its passes back to back, the strip boundary in scratch only it
 *                    touches; no launch boundary).  -1 = default: formed when the call has two or more
 *                    queries (or streams its database in) and the database is small beside the chip; 0 = never: one launch
 *                    per pass of every query ("bnd_mib" applies); 1 = always, every query joins
 *   "sp_threshold"   65536 = default (none): queries of at least this many rows are aligned by the SCORE-PROFILE kernel instead of the
 *                    query-profile pipeline (the reference's query_length_threshold: `-p S` = 0, `-p Q` = none, `-p A -u N` = N,
 *                    MICsearch.c:39-43, swimm.c:81-85).  Exact like every path; slower on gfx950 at every query length
 *   "cut"            35 = default: per device group, the longest pairs leave the group (they run whole through the lane-systolic kernel
 *                    beside the pipeline kernel, which then stops at the longest pair left) when that saves more padded pipeline
 *                    cells than value/10 times the pairs' own cells; 0 = never
 *   "stack"          1 = default: short queries (up to 72 rows) of a batch share workgroups -- two to four of them stacked along the
 *                    strips of one 4-wave workgroup, each with its own score row -- instead of padding each to a launch
 *                    shape of its own; 0 = every query its own workgroups
 *   "time_launches"  1 = bracket every pipeline launch with events (swimm_hip_last_launch_ms); default 0
 *   "lazy_upload"    0 = default: add_chunk / add_sequences copy the caller's buffers before they return; 1 = they only
 *                    record them and the next search streams the chunks in, copying and tiling chunk k+1 while chunk k
 *                    is being aligned (the double-buffered transfer of MICsearch.c:85-91) -- the buffers must then stay
 *                    valid until that search has returned.  swimm_hip_search_chunks always works this way.  The copies
 *                    are made by a thread the context owns (started with the first recorded chunk, parked between
 *                    searches, joined by swimm_hip_destroy); the calls of one context still come from one thread at a time.
 *   "upload_piece_kib"  with lazy_upload, chunks and slabs larger than this are recorded in pieces of about this size (default
 *                    98304 = 96 MiB), so that a database handed over as one buffer still streams in as several ranges
 *   "score_mib"      HBM budget of the score rows (4 B per query and sequence): the query list is walked in batches
 *                    that fit (default 32768; 0 = one query per batch)
 *   "bnd_mib"        HBM budget of the pass-boundary buffer of multi-pass queries (4x the tiled residue bytes of the
 *                    groups in flight): the group list is cut into runs that fit (default 16384)
 * Unknown key -> error.  The environment variable SWIMM_HIP_OPTIONS="key=value,key=value" applies the same knobs to
 * every context at creation (for swimm_hip_search_chunks and the `swimm` program, whose contexts the caller never sees).
 * (Rounds 1-3 carried eleven more -- rotate, alternate, split, lane_rows, lane_room, wgs_per_cu, and the measured-and-rejected
 * tall, batch_order, bulk_streams, lane_acquire, upload_head=0; their measurements are in profiles/NOTES.md.) */
int swimm_hip_set_option(swimm_hip_ctx *ctx, const char *key, int value);

/* Whole-call drop-in with the argument list of mic_search_knc_ap_multiple_chunks
 * (MICsearch.h:35-38, called at swimm.c:88-90): shards the chunks statically over `num_gpus`
 * devices (one host thread each), searches, scatters into `scores`
 * (stride vect_sequences_db_count * vl) and writes *workTime.  `mic_threads` of the original is
 * meaningless on a GPU and dropped; `query_length_threshold` is the option "sp_threshold" (through
 * SWIMM_HIP_OPTIONS for this call; default: no query takes the score profile); `vl` (the lane
 * width the chunks were assembled with) is added.  Returns 0 or an error status. */
int swimm_hip_search_chunks(const char *query_sequences, const uint16_t *query_sequences_lengths,
                            uint32_t query_sequences_count, const uint32_t *query_disp,
                            uint64_t vect_sequences_db_count, char **chunk_b, uint32_t chunk_count,
                            const uint32_t *chunk_vect_sequences_db_count, uint16_t **chunk_n,
                            uint32_t **chunk_b_disp, const uint64_t *chunk_vD, const char *submat,
                            int open_gap, int extend_gap, int num_gpus, uint32_t vl, int32_t *scores,
                            double *workTime);

#ifdef __cplusplus
}
#endif
#endif /* SWIMM_HIP_H_INCLUDED */
