// Internal interface between the library's host side (swimm_hip.cpp, search.cpp, plan.cpp, upload.cpp) and the device code
// (sw_kernels.hip).  Not installed; the public boundary is include/swimm_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swimm {

constexpr int kCodes = 25;        // residue codes 0..24 (24 = lane padding), sequences.h:17-18
// On the device the residues are RENUMBERED (by the two tiling kernels; every table a kernel indexes by a database residue --
// query profiles, the score-profile kernel's matrix -- is built in that order by the host side).  Why: the profile row of code d
// starts in LDS bank group d mod 16, a ds_read_b128 serves 8 lanes per clock, and two of them conflict when their codes are
// equal mod 16 but different.  With the reference's alphabetical codes (A, S), (C, V), (F, Y), (D, W), (I, padding) ... share a
// group: 2.1 % per pair of lanes for Robinson-Robinson frequencies, 45 % of all reads; with this order the sharing pairs are
// (L, X) (A, Z) (G, U) (R, W) (I, C) (N, M) (Q, H) (F, Y) (B, padding): 1.0 %, 25 % of reads.
constexpr uint8_t kDevCode[kCodes] = {1, 8, 20, 9, 10, 7, 2, 22, 4, 11, 0, 21, 5, 12, 6, 3, 13, 14, 15, 19, 16, 23, 17, 18, 24};     // reference code -> device code
constexpr uint8_t kHostCode[kCodes] = {10, 0, 6, 15, 8, 12, 14, 5, 1, 3, 4, 9, 13, 16, 17, 18, 20, 22, 23, 19, 2, 11, 7, 21, 24};    // device code -> reference code
constexpr int kGroupSeqs = 128;   // sequences per device group: 64 lanes x 2 packed int16 halves
constexpr int kChunkCols = 4;     // DB columns per pipeline step (one dword per sequence)
constexpr int kMaxWaves = 16;     // waves per workgroup (1024 threads)

// One device group = 128 consecutive sorted sequences, tiled [chunk][lane][A0..A3 B0..B3].
struct GroupDesc {
    const uint8_t *db;   // tiled residues of this group
    uint32_t ncols;      // padded length, multiple of kChunkCols
    uint32_t seq0;       // local score slot of lane 0's sequence A; lane l holds the slots seq0 + 2l (A) and seq0 + 2l + 1 (B)
};

// One unit of work for a workgroup's pipeline: self-contained (32 bytes, one scalar load), so that starting a group
// does not cost two dependent global loads (item -> group descriptor).
struct Item {
    const uint8_t *db;   // tiled residues of the group
    uint32_t ncols;      // padded length, multiple of kChunkCols
    uint32_t seq0;       // local score slot of lane 0's sequence A; lane l holds the slots seq0 + 2l (A) and seq0 + 2l + 1 (B)
    uint32_t half;       // int32 tier only: 0 = sequences 0..63 of the group, 1 = 64..127
    uint32_t out_slot;   // (unused: the int32 tier writes out[seq0 + 2*lane + half])
    uint64_t bnd_off;    // first column of this item in the pass-boundary buffer
};

// group-resident launches: one query of the batch the launch serves
// A "query" of a launch may also be a STACK of short queries that share one workgroup: query A in the first strips (waves),
// query B in the next ones, ... each padded to whole strips, one pass.  A wave that starts a member takes a zero top
// boundary instead of its upper neighbour's bottom row (seam), and every wave's best goes to the score row of ITS member.
constexpr uint32_t kNoTab = 0xFFFFFFFFu;
struct QDesc {
    uint32_t prof_off;      // first element of the query's (or the stack's concatenated) profile in PipeParams::prof
    uint32_t prof_stride;   // rows allocated per residue code
    uint32_t passes;        // ceil(rows / (W * T)); a stack has one
    uint32_t out_off;       // first element of the query's score row in PipeParams::out
    uint32_t seam_mask;     // stack: bit k = wave k starts a member query
    uint32_t wave_tab;      // stack: PipeParams::wave_out[wave_tab + k] = first element of wave k's member's score row; kNoTab = a plain query
    uint32_t pad_[2];
};

struct PipeParams {
    const Item *items;
    const uint32_t *wg_first;   // static partition: [n_wg + 1] item ranges
    const uint32_t *wg_chunks;  // static partition: [n_wg] total column chunks per workgroup
    uint32_t *queue;            // dynamic queue: cursor into items[] (sorted longest first), zeroed before the launch;
                                // nullptr selects the static partition
    uint32_t n_items;           // dynamic queue: length of items[]
    const uint32_t *avail;      // dynamic queue of a database that is still streaming in (one launch over the whole list, in upload order): how
                                // many items of the list have landed so far, published behind every part's tiling kernel (publish_items_kernel);
                                // wave 0 waits for an item's turn before it hands the item out.  nullptr: everything is resident
    uint32_t max_steps;         // dynamic queue: upper bound of a workgroup's loop (all chunks of the list + waves)
    const int16_t *prof;        // query profile prof[d * prof_stride + row]
    uint32_t prof_stride;       // rows allocated per code (>= passes * W * T)
    uint32_t r0;                // first query row of this pass
    uint2 *bnd;                 // pass boundary rows (H,F per column per lane), updated in place
    int first_pass, last_pass;  // one pass per launch
    const QDesc *qdesc;         // group-resident launches: the batch's queries (items are (group, query) pairs, n_items = groups x n_queries)
    uint32_t n_queries;         // (qdesc != nullptr selects the group-resident kernel: every workgroup takes a group through all
                                // the passes of an item's query back to back)
    uint32_t bnd_wg_cols;       // group-resident launches: bnd holds this many columns per workgroup, touched by it alone
    const uint32_t *wave_out;   // stacks of short queries: per (stack, wave) the score row of the wave's member (group-resident launches: indexed
                                // through QDesc::wave_tab; a per-pass launch: non-null selects the stack below)
    uint32_t seam_mask;         // per-pass launch of ONE stack: its seam bits ...
    uint32_t wave_tab;          // ... and its first entry in wave_out
    int32_t *out;               // packed mode: score row of this query; int32 mode: out32
    int goe, ge;                // open+extend, extend
    uint32_t *err;              // watchdog word shared with the lane kernel
    unsigned long long *stamps; // diagnostic build only (-DSWIMM_STAMPS): per-wave-index cycle sums [16][8]
};

// The pipeline kernel's binary16 tier stores column j with an offset (j mod P) * ge (sw_kernels.hip, cell2_ofs) and takes the
// offsets back every P = 4 * f16_renorm_chunks(ge) columns; the running best carries the COMING column's offset, so up to
// P * ge (at most 128; 4 ge when ge > 32) sits on top of a score.
__host__ __device__ constexpr int f16_renorm_chunks(int ge) { return ge <= 0 ? 32 : (32 / ge < 1 ? 1 : 32 / ge); }
// ... so a first-tier result below this is exact, and anything that left the exact range shows as a result >= this
constexpr int f16_exact_below(int ge) { return 2048 - 4 * f16_renorm_chunks(ge) * (ge > 0 ? ge : 0); }

enum class Mode { PK16, I32, F16 };   // F16: packed binary16 first tier (pipeline kernel only)

// ---- lane-systolic kernel: ONE wave aligns one packed pair (or one sequence in int32 mode) with its 64
// lanes as the query strips; used for the long-sequence tail and for the int32 promotion re-runs, where
// a whole 128-sequence group on one workgroup would be a serial chain.
constexpr int kLaneRows = 8;                  // query rows per lane -> 512 rows per pass
struct LaneItem {
    const uint8_t *db;     // tiled residues of the item's device group
    uint32_t lane;         // which lane's dword pair of that group (0..63)
    uint32_t half;         // int32 mode: 0 = sequence A of the pair, 1 = B
    uint32_t ncols;        // multiple of kChunkCols
    uint32_t slot_a;       // score slot of A (packed) / of the sequence (int32)
    uint32_t slot_b;       // score slot of B (packed only)
    uint32_t bnd_off;      // first column of this item in the pass-boundary buffer
    uint32_t pad_;
};
// one query of a lane-systolic launch (a launch serves a batch of queries: their tail items are independent chains,
// and 20 queries x the 22 ms chain of a 35 000-residue sequence run side by side instead of one after the other)
struct LaneQ {
    uint64_t prof_off;          // first element of the query's profile in LaneParams::prof
    uint64_t out_off;           // first element of the query's score row in LaneParams::out
    uint64_t bnd0;              // first column of the query's boundary rows in bnd[0] / bnd[1] (multi-pass queries)
    uint32_t prof_stride;
    uint32_t m;                 // query rows
    uint32_t passes;            // ceil(m / (64 * rows per lane)); all passes run in ONE launch, chained per item
    uint32_t queue0;            // queue[queue0 + pass]: work cursor of (query, pass), zeroed before the launch
    uint32_t prog0;             // prog[prog0 + pass * n_items + item]: boundary columns published so far (global column + 1), zeroed
    uint32_t items0, n_items;   // the query's items: items[items0 .. items0 + n_items) (the tail: every query the whole list; re-runs: each its own)
    uint32_t pad_;
};
struct LaneParams {
    const LaneItem *items;
    const LaneQ *lq;
    const uint32_t *block_map;  // [grid]: query << 8 | pass of every workgroup, ascending in pass (a producer is dispatched before its consumers)
    uint32_t *queue;
    uint32_t *prog;
    const int16_t *prof;
    unsigned long long *bnd[2]; // boundary rows (H | F << 32 per column): pass p writes bnd[p & 1], reads bnd[(p - 1) & 1]
    int32_t *out;
    int goe, ge;
    uint32_t *err;              // watchdog word
};
size_t lane_lds_bytes(int rows_per_lane);
// rows_per_lane: kLaneRows (any query), or 4 / 2 for a one-pass launch of a query of <= 256 / <= 128 rows
hipError_t launch_lane(Mode mode, int rows_per_lane, int n_wg, const LaneParams &p, hipStream_t s);

// ---- score-profile kernel (the reference's second lookup technique, MICsearch.c:257-313: per block of database columns a table
// [query residue][column][lane] built once and read linearly by every query row).  One wave per workgroup, 32 query rows per pass,
// the table of a chunk of 4 columns in LDS (24 KB); selected per query by the option "sp_threshold" (`swimm -p S`, `-p A -u`).
constexpr int kSpRows = 32;
constexpr int kSpSubStride = 40;  // halfs per database residue in the kernel's matrix copy: 24 query residues + padding (80 B: 16-byte reads, banks spread)
struct SpParams {
    const Item *items;          // every group of the range, longest first
    uint32_t n_items;
    uint32_t *queue;            // work cursor, zeroed before the launch
    const int8_t *qcodes;       // the query's residue codes, padded with the dummy code 23 to a multiple of kSpRows
    uint32_t r0;                // first query row of this pass
    const uint16_t *sub16;      // substitution matrix as binary16 bits, transposed: [25 database residues][kSpSubStride], entry q = S(query residue q, d)
    uint2 *bnd;                 // pass boundary rows (H, F per column and lane), as in PipeParams
    int first_pass, last_pass;
    int32_t *out;               // the query's score row
    int goe, ge;
};
hipError_t launch_sp(int n_wg, const SpParams &p, hipStream_t s);

size_t pipe_lds_bytes(Mode mode, int rows_per_wave, int waves, bool resident);
// which (tier, rows per wave, hand-over scheme) kernels exist
bool pipe_has_variant(Mode mode, int rows_per_wave);
// registers / occupancy of one instantiation (for the host-side launch plan)
hipError_t pipe_kernel_attributes(Mode mode, int rows_per_wave, bool resident, int *num_regs);
hipError_t grow_kernel_attributes(int rows_per_wave, int *num_regs);     // ... of the instantiation that walks a growing item list
// mangled symbol of the instantiation (nullptr when there is none)
const char *pipe_kernel_symbol(Mode mode, int rows_per_wave, bool dynamic, bool resident);
hipError_t launch_pipe(Mode mode, int rows_per_wave, int waves, int n_wg, const PipeParams &p, hipStream_t s);

// Re-tile one reference-layout chunk (sequences.c:506-526 byte interleave) into device groups.
hipError_t launch_retile(const uint8_t *b, const uint16_t *n, const uint32_t *disp, uint32_t vl_groups, uint32_t vl,
                         const uint64_t *goff, const uint32_t *gcols, uint32_t dev_groups, uint32_t max_cols /* longest group */, uint8_t *tiled,
                         uint32_t *seq_len /* [dev_groups*128], zeroed; gets every sequence's true length */, hipStream_t s);

// Tile a slab of sorted sequences (lengths + concatenated codes, as in the .seq file) into device groups directly.
hipError_t launch_tile_sequences(const uint8_t *codes, const uint16_t *lens /* [n_seq] */, const uint32_t *gsrc /* [dev_groups]: offset of every group's first residue in codes */,
                                 uint32_t n_seq, const uint64_t *goff, const uint32_t *gcols, uint32_t dev_groups, uint32_t max_cols /* longest group */, uint8_t *tiled,
                                 hipStream_t s);

// appends the slots whose score is >= thr (tier left its exact range) to list (up to cap) and zeroes them
// items of a streaming search's list that are on the device (PipeParams::avail); kAvailAbort = the upload failed: give up
constexpr uint32_t kAvailAbort = 0xFFFFFFFFu;
// one wave that lasts `microseconds` and does nothing (probe: do two streams share a hardware queue?)
hipError_t launch_spin(uint32_t microseconds, hipStream_t s);
hipError_t launch_publish_items(uint32_t *avail, uint32_t value, hipStream_t s);
hipError_t launch_collect_saturated(int32_t *scores, uint64_t n, int thr, uint32_t *list, uint32_t *count, uint32_t cap, hipStream_t s);
// per-block top-64 candidates of every query's score row (rows n_slots apart): out_keys[(query * n_blocks + block) * 64 + i] =
// ((score<<32 | global index) + 1), 0 = empty; group_base[g] = global sorted index of the group's first sequence,
// group_valid[g] = real sequences in it
hipError_t launch_topk64(const int32_t *scores, uint64_t n_slots, const int64_t *group_base, const uint32_t *group_valid,
                         unsigned long long *out_keys, int n_blocks, uint32_t n_queries, hipStream_t s);

}  // namespace swimm
