// swimm_hip.cpp -- C-ABI shim of libswimm_hip.so (see include/swimm_hip.h).
//
// Host-side orchestration of the gfx950 kernels in sw_kernels.hip: device-resident database, per-query
// launch plan (rows per wave T, waves per workgroup W, passes) from a measured rate table, longest-first work
// lists for the persistent workgroups (dynamic queue, static partition as an option), the long-sequence tail
// on a second stream, the binary16 -> int16 -> int32 promotion ladder on a third, top-r and score scatter.
// Structural template: mic_search_knc_ap_multiple_chunks (MICsearch.c:4-354) -- X1 = set_queries,
// X2-in = add_chunk (kept resident), compute = search, X3 = scatter into the caller's scores.
#include "../../include/swimm_hip.h"
#include "sw_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <cxxabi.h>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

using namespace swimm;

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) return fail("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class T>
struct DevBuf {   // grow-only device scratch
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// the lengths of a run of bulk groups, longest first, with the makespan factors already worked out for it
struct BulkCols {
    std::vector<uint32_t> cols;
    uint64_t total = 0;
    std::map<int, double> cache;    // n_wg -> LPT makespan / mean load
};

struct Plan {      // static partition of a work list over n_wg persistent workgroups
    int n_wg = 0;
    DevBuf<Item> items;          // grouped by workgroup (static partition)
    DevBuf<uint32_t> wg_first, wg_chunks;
    DevBuf<Item> queue_items;    // the same items sorted longest first (dynamic queue); bnd_off = columns before the item in this order
    std::vector<uint32_t> queue_cols;   // their column counts (host copy, for cutting the list into boundary-buffer segments)
    DevBuf<Item> split_items;    // the even-ranked items of that list followed by the odd-ranked ones (two-stream launches)
    uint32_t split_n[2] = {0, 0};
    uint64_t split_cols[2] = {0, 0};
    uint32_t n_items = 0;
    uint64_t bnd_cols = 0;   // columns the pass-boundary buffer must hold
    uint64_t max_wg_chunks = 0, total_chunks = 0;
    void release() { items.release(); wg_first.release(); wg_chunks.release(); queue_items.release(); split_items.release(); }
};

// lane-systolic work list (long-sequence tail, int32 promotion): items sorted longest first, pulled
// dynamically by the waves
struct LaneList {
    DevBuf<LaneItem> items;
    uint32_t n = 0;
    uint64_t cols = 0;       // boundary columns (sum of ncols)
    uint64_t cell_cols = 0;  // sum of ncols (for the cell statistics)
    void release() { items.release(); n = 0; }
};

struct DbPlan {          // per (mode, n_wg): main partition + the groups handed to the lane kernel
    Plan main;
    bool have_main = false;
    LaneList tail;
};

// scratch of one lane-kernel stream: boundary rows of even / odd passes, per-pass queues, progress counters
struct LaneScratch {
    DevBuf<unsigned long long> bnd[2];
    DevBuf<uint32_t> queue, prog;
    void release() { bnd[0].release(); bnd[1].release(); queue.release(); prog.release(); }
};

struct Uploader;      // the thread that copies a lazily uploaded database (below, with upload_chunk)

struct ChunkRec {
    uint8_t *d_tiled = nullptr;
    uint32_t *d_len = nullptr;  // chunk-layout chunks: every slot's true length, written by the re-tile kernel
    uint64_t first_seq = 0;     // global sorted index of the chunk's first sequence
    uint64_t n_seq = 0;         // group_count * vl
    uint32_t group0 = 0, n_groups = 0;
    uint64_t cols = 0;          // padded columns of the chunk's device groups
    // upload source: the caller's buffers.  Eager mode (default) copies inside add_chunk / add_sequences; with the
    // option "lazy_upload" they are only recorded and the first search streams them in (X2 overlapped with compute,
    // MICsearch.c:85-91), so they must stay valid until that search has returned.
    int kind = 0;               // 0 = reference chunk layout (re-tile), 1 = .seq slab (tile)
    const char *h_b = nullptr; uint64_t vD = 0; const uint16_t *h_n = nullptr; const uint32_t *h_disp = nullptr;
    uint32_t group_count = 0, vl = 0;
    const char *h_codes = nullptr; uint64_t code_bytes = 0;
    std::vector<uint32_t> off;  // kind 1: residue offset of every sequence (n_seq + 1)
    std::vector<uint64_t> goff; // byte offset of every device group in d_tiled
    std::vector<uint32_t> gcols;
    bool uploaded = false, lens_known = false;
    hipEvent_t ready = nullptr; // recorded on the upload stream behind the chunk's (re-)tile kernel
};

struct QueryPlan { int T, W, passes; uint32_t mpad; size_t prof_off; Mode mode = Mode::F16; bool dynamic = true, resident = false; };

}  // namespace

struct swimm_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;     // second bulk stream: multi-pass queries run the two halves of the group list side by side
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipStream_t stream2 = nullptr;      // lane-systolic tail runs beside the bulk kernel
    hipEvent_t ev_tail = nullptr;
    hipStream_t stream_up = nullptr;    // uploads: H2D copies, (re-)tile kernels, work lists -- never waits for a DP kernel
    hipEvent_t ev_copied = nullptr;
    DevBuf<uint8_t> up_b; DevBuf<uint16_t> up_n; DevBuf<uint32_t> up_disp, up_gcols, up_off; DevBuf<uint64_t> up_goff;   // upload scratch, reused chunk after chunk
    int opt_lazy_upload = 0;            // 1: add_chunk / add_sequences record the caller's buffers, the first search streams them in
    hipStream_t stream3 = nullptr;      // promotion re-runs
    hipEvent_t ev_ready = nullptr, ev_tail3 = nullptr;
    std::vector<hipEvent_t> ev_query;   // [2q] bulk done, [2q+1] tail done
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 0;
    // options (swimm_hip_set_option)
    int opt_T = 0, opt_maxW = 0, opt_W = 0, opt_wgs_per_cu = 0;   // launch shape: 0 = chosen per query
    int opt_force_i32 = 0;              // 1: everything in int32
    int opt_f16 = 1;                    // 1: packed binary16 first tier (exact below 2048, then int16, then int32)
    int opt_tail_mode = 0;              // 0 auto, 1 every group through the lane kernel, 2 none
    int opt_tail_frac = 50;             // a group goes to the lane kernel when it is longer than this percentage of a CU's mean load
    int opt_dynamic = 1;                // 1: workgroups pull items from a global queue (default); 0: static partition by the host
    int opt_lane_rows = 1;              // 1: one-pass lane launches of short queries use 2 / 4 rows per lane instead of 8
    int opt_resident = -1;              // group-resident batch launches: -1 = when the batch has two or more queries that are not rotated, 0 never, 1 always
    int opt_lane_room = -1;             // launch shapes must leave a lane-systolic wave its registers: -1 = when the database has a long-sequence tail, 0 never, 1 always
    bool batch_now = false;             // the search in progress runs its non-rotated queries as group-resident batch launches
    std::vector<uint8_t> stream_tail;   // streaming search: the tail flags of the whole database (pick_tail)
    bool streaming_now = false;         // the search in progress streams its database in (per-range launches, no group-resident batches)
    DevBuf<QDesc> d_qdesc;              // group-resident launches: per batch, its queries
    int opt_time_launches = 0;          // 1: every pipeline launch is bracketed by events on its own stream (measurement aid, bench.py)
    std::vector<hipEvent_t> launch_ev;  // pairs (before, after), grown on demand
    size_t launch_ev_used = 0;
    double launch_ms_sum = 0;           // sum of the pipeline launches' own durations in the last search
    uint32_t launch_ms_n = 0;
    int opt_rotate = 1;                 // 1: eight or more one-pass queries run whole on three streams in rotation; 0: they join the group-resident batch
    int opt_alternate = 1;              // 1: the passes of consecutive multi-pass queries alternate between two streams
    int opt_split = 1;                  // 1: multi-pass queries run the even- and odd-ranked groups as two kernels on two streams
    int opt_bnd_mib = 16384;            // HBM budget of the pass-boundary buffer (MiB)
    int opt_score_mib = 32768;          // HBM budget of the score rows of one query batch (MiB)
    int opt_lane_acquire = 0;           // 1: chained lane passes take an agent-scope acquire after every progress poll (default: sc1 loads only)
    int opt_wg_limit = 0;               // > 0: at most this many persistent workgroups per pipeline launch (tests: long per-workgroup item sequences on a small database)
    // caches that depend on the resident database / the code objects
    BulkCols bulk;                      // the resident database's bulk groups (built on demand) and their makespan factors
    int regs_cache[2][3][40] = {};      // VGPRs of sw_pipe_kernel<T, tier, dynamic, group-resident or not>, looked up once
    DevBuf<uint32_t> d_queue;           // one cursor per pipeline launch of a search
    uint32_t queue_next = 0;
    // queries (host copies; profiles are built per search because T/W may change)
    std::vector<int8_t> qcodes;
    std::vector<uint16_t> qm;
    std::vector<uint32_t> qdisp;
    int8_t submat[SWIMM_HIP_SUBMAT_BYTES];
    int open_gap = 10, extend_gap = 2, max_pos = 0;
    bool have_queries = false;
    // database
    std::vector<ChunkRec> chunks;
    std::vector<GroupDesc> groups;
    std::vector<uint64_t> group_col_off;
    std::vector<uint32_t> seq_len;      // true length of every local slot (from the re-tile kernel)
    uint64_t total_cols = 0;
    bool groups_dirty = true;
    std::map<int, DbPlan> plans;        // key: n_wg (packed mode), n_wg | 1<<30 (whole-db int32 mode)
    // scratch
    DevBuf<int32_t> d_scores;
    DevBuf<int16_t> d_prof;
    DevBuf<uint2> d_bnd, d_bnd_b;       // pass-boundary rows; the second one for the queries whose passes run on stream_b
    DevBuf<uint2> d_bnd_c;              // ... and a third for the group-resident launches of a database that streams in (three ranges in flight)
    Uploader *up = nullptr;             // the thread that copies a lazily uploaded database (created with the first recorded chunk)
    void *pin = nullptr;                // pinned arena the work lists travel through (list_copy)
    size_t pin_cap = 0, pin_used = 0;
    DevBuf<int64_t> d_gbase;
    DevBuf<uint32_t> d_gvalid;
    DevBuf<unsigned long long> d_keys;
    DevBuf<uint32_t> d_err;             // pipeline watchdog word
    DevBuf<unsigned long long> d_stamps;   // diagnostic build only
    LaneScratch tail_scratch;           // lane kernel on stream 2 (long-sequence tail)
    LaneScratch tail_scratch_a, tail_scratch_b;   // one-pass queries that run whole on the main stream / on stream_b
    // a search whose time is set by the long-sequence chains (a small database with one extreme sequence, many queries)
    // runs up to three queries' tail launches side by side: two more streams (created on first use), scratch and events
    hipStream_t stream_t[2] = {nullptr, nullptr};
    hipEvent_t ev_tail_t[2] = {nullptr, nullptr};
    LaneScratch tail_scratch_t[2];
    LaneScratch rerun_scratch;          // lane kernel on stream 3 (promotion re-runs)
    DevBuf<LaneItem> d_rerun_items;
    DevBuf<uint32_t> d_satlist;
    // stats of the last search
    double kernel_ms = 0;
    uint64_t cells = 0, promoted = 0, promoted16 = 0;
    std::vector<QueryPlan> last_plans;
    uint32_t launches = 0;
};

namespace {

void release_plans(swimm_hip_ctx *c)
{
    for (auto &kv : c->plans) { kv.second.main.release(); kv.second.tail.release(); }
    c->plans.clear();
    c->bulk.cols.clear(); c->bulk.total = 0; c->bulk.cache.clear();
}

int regs_to_waves_per_simd(int regs)
{
    const int alloc = (regs + 7) / 8 * 8;   // MI355X_MICROARCH: 8-register granule, 512 per SIMD lane
    return std::max(1, std::min(8, 512 / std::max(alloc, 8)));
}

// how many workgroups of W waves of the T-row kernel one CU holds (VGPRs: 8-register granule, 512 per SIMD
// lane; LDS: 160 KiB)
int kernel_regs(const swimm_hip_ctx *c, Mode mode, int T, bool resident, int *out)
{
    int &regs = const_cast<swimm_hip_ctx *>(c)->regs_cache[resident ? 1 : 0][(int)mode][T];
    if (regs == 0) HIP_TRY(pipe_kernel_attributes(mode, T, resident, &regs));
    *out = regs;
    return 0;
}

int wgs_per_cu(const swimm_hip_ctx *c, Mode mode, int T, int W, bool resident, int *out)
{
    int regs = 0;
    if (kernel_regs(c, mode, T, resident, &regs)) return 1;
    const int waves_cu = 4 * regs_to_waves_per_simd(regs);
    const size_t lds = pipe_lds_bytes(T, W, resident);
    int n = std::min(waves_cu / W, (int)(163840 / lds));
    if (c->opt_wgs_per_cu > 0) n = c->opt_wgs_per_cu;
    *out = std::max(1, n);
    return 0;
}

// a query of several passes runs the group-resident kernel (one launch) unless that is switched off
bool resident_for(const swimm_hip_ctx *c, int passes) { (void)passes; return c->batch_now; }

// a run of consecutive device groups that is searched as one unit: the whole resident database (work lists cached),
// or one chunk of a database that is still streaming in
// Work lists travel on the upload stream -- except while a search streams its database in: the upload stream then
// belongs to the uploader thread's chunk copies (0.1 GB each), and the lists take the promotion stream, idle until
// the ladder at the end of the search.
hipStream_t list_stream(const swimm_hip_ctx *c) { return c->streaming_now ? c->stream3 : c->stream_up; }

// ... and through a pinned arena: a copy from pageable memory would queue for the runtime's staging buffers behind
// the uploader's chunk copies (measured: 1.2 ms per range's lists instead of 0.3).  list_sync() ends a batch of copies.
int list_copy(swimm_hip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    if (c->pin_used + bytes > c->pin_cap) {
        HIP_TRY(hipStreamSynchronize(list_stream(c)));        // copies in flight still read the arena
        c->pin_used = 0;
        if (bytes > c->pin_cap) {
            if (c->pin) { HIP_TRY(hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
            const size_t cap = std::max<size_t>(2 * bytes, (size_t)4 << 20);
            HIP_TRY(hipHostMalloc(&c->pin, cap, hipHostMallocDefault));
            c->pin_cap = cap;
        }
    }
    char *at = (char *)c->pin + c->pin_used;
    memcpy(at, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, list_stream(c)));
    c->pin_used += (bytes + 255) & ~(size_t)255;
    return 0;
}
int list_sync(swimm_hip_ctx *c)
{
    HIP_TRY(hipStreamSynchronize(list_stream(c)));
    c->pin_used = 0;
    return 0;
}

struct Range { uint32_t g0 = 0, g1 = 0; uint64_t cols = 0; };
Range whole_range(const swimm_hip_ctx *c) { Range r; r.g0 = 0; r.g1 = (uint32_t)c->groups.size(); r.cols = c->total_cols; return r; }

std::vector<uint8_t> pick_tail(const swimm_hip_ctx *c, const Range &rg);

// persistent workgroups of a pipeline launch: what the chip holds, unless the caller caps it
int n_workgroups(const swimm_hip_ctx *c, int per_cu)
{
    const int n = c->num_cu * per_cu;
    return c->opt_wg_limit > 0 ? std::min(n, c->opt_wg_limit) : n;
}

// How evenly the bulk groups of the resident database spread over n_wg workgroups: makespan of the longest-first
// greedy schedule (what the dynamic queue, and the static partition, produce) over the mean load.  1.00x for a
// large database; a small one whose longest group is a sizeable part of a workgroup's share reaches 1.4 - 1.9
// with 3 - 4 workgroups per CU, and then fewer, larger workgroups are the better launch shape.
static double lpt_imbalance(BulkCols &b, int n_wg)
{
    auto it = b.cache.find(n_wg);
    if (it != b.cache.end()) return it->second;
    const std::vector<uint32_t> &cols = b.cols;
    const uint64_t total = b.total;
    double r = 1.0;
    if (!cols.empty() && total > 0) {
        const int n = std::max(1, std::min<int>(n_wg, (int)cols.size()));
        // Longest-first greedy.  With many groups per workgroup the schedule ends within one short group of the mean
        // load; the exact simulation only matters (and is only run) while a workgroup gets fewer than 64 groups.
        if (cols.size() >= (size_t)64 * n) {
            const double mean = (double)total / n_wg;
            r = std::max((double)cols[0], mean + 0.5 * cols[cols.size() - cols.size() / 8 - 1]) / mean;
        } else {
            std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> heap;
            for (int w = 0; w < n; ++w) heap.push(0);
            uint64_t mx = 0;
            for (uint32_t x : cols) { uint64_t l = heap.top() + x; heap.pop(); heap.push(l); mx = std::max(mx, l); }
            r = (double)mx / ((double)total / n_wg);   // fewer groups than workgroups: the idle ones count
        }
    }
    b.cache[n_wg] = r;
    return r;
}

// the bulk groups of a range (those the tail picker leaves to the pipeline kernel), longest first
static void bulk_cols_of(const swimm_hip_ctx *c, const Range &rg, BulkCols &b)
{
    const std::vector<uint8_t> is_tail = pick_tail(c, rg);
    b.cols.clear(); b.total = 0; b.cache.clear();
    for (uint32_t g = rg.g0; g < rg.g1; ++g)
        if (!is_tail[g - rg.g0]) { b.cols.push_back(c->groups[g].ncols); b.total += c->groups[g].ncols; }
    std::sort(b.cols.begin(), b.cols.end(), std::greater<uint32_t>());
}

double plan_imbalance(swimm_hip_ctx *c, int n_wg)
{
    if (c->bulk.cols.empty() && !c->groups.empty()) bulk_cols_of(c, whole_range(c), c->bulk);     // once per database
    return lpt_imbalance(c->bulk, n_wg);
}

// Measured throughput (GCUPS of padded cells) of every launch shape of the f16-tier pipeline kernel: rows per wave
// T = 8, 12, ... 36 (lines) by waves per workgroup W = 1..16 (columns), workgroups per CU by occupancy
// (tools/plan_sweep.py --scale 1.0 on one MI355X, profiles/r02_plan_sweep.txt; round 1's table, before the next-chunk
// prefetch, was 2-5 % lower and had the 8-wave shapes a little further behind the 4-wave ones).  W = 4, 8, 12, 16 put the same number of waves
// on each of the CU's 4 SIMDs; any other W runs like the next multiple of 4 (2 x 6 waves behave like 4+4+2+2).
static const float kShapeGcups[8][16] = {
    {2648, 4406, 5432, 6668, 6283, 6811, 7102, 7832, 6982, 6376, 6976, 7578, 6850, 6875, 5896, 7802},  // T=8
    {3154, 4918, 6308, 7339, 6506, 6819, 7204, 8185, 7142, 6730, 7484, 8118, 7147, 7256, 7739, 8158},  // T=12
    {3700, 5675, 6768, 8168, 6496, 5518, 7106, 8065, 5692, 6337, 6946, 7561, 6327, 6795, 7255, 7723},  // T=16
    {3895, 5893, 6987, 8264, 5406, 6190, 7304, 8334, 5795, 6487, 7105, 7776, 6269, 7035, 7535, 8037},  // T=20
    {4050, 6271, 7188, 8332, 5487, 6435, 7450, 8474, 5890, 6645, 7222, 7935, 6429, 7149, 7621, 8150},  // T=24
    {4161, 6397, 7264, 8398, 6255, 6469, 7553, 8609, 6052, 6740, 7383, 8061, 6722, 7229, 7752, 8263},  // T=28
    {4130, 6470, 6405, 8592, 4870, 5827, 6764, 7683, 6120, 6796, 7444, 8120, 0, 0, 0, 0},                 // T=32
    {4144, 6660, 6398, 8523, 4902, 5900, 6674, 7809, 6154, 6874, 7083, 8217, 0, 0, 0, 0},                 // T=36
};

// Rows per wave T, waves per workgroup W and number of passes for a query of m rows: the shape with the lowest
// predicted time, passes x (padded cells of a pass / measured rate of that shape x makespan factor + launch cost).
// With more than one pass the strip boundaries go through HBM and the first wave waits for its loads: not
// measurable for W >= 8 (every query of 464 ... 5478 rows on a c5-shaped shard runs at 0.97 of its shape's rate,
// like the one-pass ones), 17 % for the 4-wave shapes.
// `room_for_lane_waves`: the database has a long-sequence tail that the lane kernel aligns on a second stream
// while this kernel runs; only shapes that leave the 80 VGPRs per SIMD lane a lane-systolic wave needs are
// admitted (e.g. 3 waves x 144, 4 x 104).
// `overlapped`: the query runs beside two others (one-pass queries in rotation, see search_device), which cover the
// workgroups that finish early: the makespan term is dropped.
// `rg` / `rb`: plan for one range of a database that is streaming in (its columns, its own makespan factors) instead
// of the whole resident database.
int choose_plan(swimm_hip_ctx *c, Mode mode, int m, bool room_for_lane_waves, bool overlapped, QueryPlan *out,
                const Range *rg = nullptr, BulkCols *rb = nullptr)
{
    struct Cand { double base; double pass_base; int T, W, passes, n_wg; };
    std::vector<Cand> cands;
    const double cols = rg ? (double)rg->cols : (double)c->total_cols;
    for (int ti = 7; ti >= 0; --ti) {
        const int T = 8 + 4 * ti;
        if (c->opt_T && T != c->opt_T) continue;
        if (!pipe_has_variant(mode, T)) continue;
        if (T == 28 && !c->opt_T && resident_for(c, 2)) continue;   // the group-resident 28-row kernel does not fit 128 VGPRs (16 spilled)
        int maxW = (T > 28) ? 12 : 16;        // __launch_bounds__ of the instantiations
        if (c->opt_maxW > 0) maxW = std::min(maxW, c->opt_maxW);
        const int strips = std::max(1, (m + T - 1) / T);
        for (int W = 1; W <= maxW; ++W) {
            if (c->opt_W > 0 && W != std::min(c->opt_W, maxW)) continue;
            const int passes = (strips + W - 1) / W;
            if (overlapped && passes != 1) continue;   // only one-pass queries take part in the rotation
            int per_cu = 1;
            if (wgs_per_cu(c, mode, T, W, resident_for(c, passes), &per_cu)) return 1;
            if (room_for_lane_waves && !c->opt_T) {
                int regs = 0;
                if (kernel_regs(c, mode, T, resident_for(c, passes), &regs)) return 1;
                const int alloc = (regs + 7) / 8 * 8;
                if (alloc * ((per_cu * W + 3) / 4) > 512 - 80) continue;
            }
            // seconds: every pass aligns T x W rows against the whole resident database at the shape's rate, and costs
            // a launch (pipeline fill and drain, staging, the last workgroups running alone: ~0.15 ms, which is what
            // makes fewer, taller passes the better plan on a database of 1e8 residues)
            const double pass_base = cols * kGroupSeqs * T * W / ((double)kShapeGcups[ti][W - 1] * 1e9);
            cands.push_back(Cand{passes * (pass_base + 150e-6), pass_base, T, W, passes, n_workgroups(c, per_cu)});
        }
    }
    // The makespan factor (>= 1) of a shape costs a simulated schedule per distinct workgroup count: cheapest shapes
    // first, and stop at the first one that cannot win even with a perfectly even schedule.  (The order of equal
    // costs is the order of the loops above: taller strips first.)
    std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.base < b.base; });
    double best_cost = -1;
    for (const Cand &k : cands) {
        if (best_cost >= 0 && k.base >= best_cost * (1.0 - 1e-9)) break;
        const double imb = overlapped ? 1.0 : (rb ? lpt_imbalance(*rb, k.n_wg) : plan_imbalance(c, k.n_wg));
        const double cost = k.passes * (k.pass_base * imb + 150e-6);
        if (best_cost < 0 || cost < best_cost * (1.0 - 1e-9)) {
            best_cost = cost;
            out->T = k.T; out->W = k.W; out->passes = k.passes; out->mpad = (uint32_t)(k.passes * k.W * k.T);
        }
    }
    // (no admissible shape leaves a lane-systolic wave its registers, e.g. under a max_waves cap: the tail then shares)
    if (best_cost < 0 && room_for_lane_waves) return choose_plan(c, mode, m, false, overlapped, out, rg, rb);
    if (best_cost < 0) { fail("no kernel variant for rows_per_wave=%d waves=%d", c->opt_T, c->opt_W); return 1; }
    return 0;
}

// upper bound of the profile elements of a query batch (25 codes x rows padded to at most 16 x 36 and to the lane kernel's 512)
static uint64_t prof_elems_bound(const uint16_t *qm, uint32_t qn)
{
    uint64_t n = 0;
    for (uint32_t q = 0; q < qn; ++q) n += (uint64_t)kCodes * ((uint64_t)qm[q] + 16 * 36 + 64 * kLaneRows);
    return n;
}

// The launch shapes of a group-resident batch (all queries of a shape run in one launch): the 4-wave shapes only --
// measured (profiles/r02_ab_batch.txt), the group-resident kernel equals the per-pass kernel with 4-wave workgroups and
// loses 9 % with 8.  First the shape that wastes the fewest padded rows over the whole batch at that shape's rate; a
// query leaves it for a shape of its own only when that saves more than 12 % of its time (short queries: 96 instead of 128
// rows), because every further shape is a further launch with an end of its own (60 queries of 1 200-1 400 residues:
// 8 270 GCUPS in one launch, 8 010 in four).
int choose_batch_shapes(swimm_hip_ctx *c, Mode mode, const uint16_t *qm, uint32_t qn, std::vector<QueryPlan> &qps)
{
    double cost[8] = {};
    int Wof[8] = {};
    bool ok[8] = {};
    auto rows_of = [](int m, int T, int W) { return (double)((m + T * W - 1) / (T * W)) * T * W; };
    for (int ti = 7; ti >= 0; --ti) {
        const int T = 8 + 4 * ti;
        if (c->opt_T && T != c->opt_T) continue;
        if (!pipe_has_variant(mode, T)) continue;
        if (T == 28 && !c->opt_T) continue;                       // the group-resident 28-row kernel does not fit 128 VGPRs
        Wof[ti] = c->opt_W > 0 ? std::min(c->opt_W, T > 28 ? 12 : 16) : std::min(4, c->opt_maxW > 0 ? c->opt_maxW : 4);
        ok[ti] = true;
        for (uint32_t q = 0; q < qn; ++q) cost[ti] += rows_of(qm[q], T, Wof[ti]) / kShapeGcups[ti][Wof[ti] - 1];
    }
    int common = -1;
    for (int ti = 7; ti >= 0; --ti)
        if (ok[ti] && (common < 0 || cost[ti] < cost[common])) common = ti;
    if (common < 0) return fail("no group-resident kernel variant for rows_per_wave=%d waves=%d", c->opt_T, c->opt_W);
    for (uint32_t q = 0; q < qn; ++q) {
        int pick = common;
        const double cc = rows_of(qm[q], 8 + 4 * common, Wof[common]) / kShapeGcups[common][Wof[common] - 1];
        double bc = cc;
        for (int ti = 7; ti >= 0; --ti) {
            if (!ok[ti]) continue;
            const double x = rows_of(qm[q], 8 + 4 * ti, Wof[ti]) / kShapeGcups[ti][Wof[ti] - 1];
            if (x < 0.88 * cc && x < bc) { bc = x; pick = ti; }
        }
        QueryPlan &qp = qps[q];
        qp.T = 8 + 4 * pick; qp.W = Wof[pick];
        const int strips = std::max(1, (qm[q] + qp.T - 1) / qp.T);
        qp.passes = (strips + qp.W - 1) / qp.W;
        qp.mpad = (uint32_t)(qp.passes * qp.W * qp.T);
    }
    return 0;
}

struct WorkUnit { uint32_t group, half, out_slot; uint32_t ncols; uint64_t bnd_off; };

// LPT: longest unit first onto the least-loaded workgroup; cost = columns (exact, every column of a
// unit costs the same T*W*128 cells)
int build_plan(swimm_hip_ctx *c, const std::vector<WorkUnit> &units, int n_wg, Plan &pl)
{
    n_wg = std::max(1, std::min<int>(n_wg, (int)units.size()));
    std::vector<uint32_t> order(units.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return units[a].ncols > units[b].ncols; });
    typedef std::pair<uint64_t, int> Load;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int w = 0; w < n_wg; ++w) heap.push(Load(0, w));
    std::vector<std::vector<uint32_t>> bins(n_wg);
    std::vector<uint64_t> load(n_wg, 0);
    for (uint32_t idx : order) {
        Load l = heap.top(); heap.pop();
        bins[l.second].push_back(idx);
        load[l.second] = l.first + units[idx].ncols;
        heap.push(Load(load[l.second], l.second));
    }
    std::vector<Item> items; items.reserve(units.size());
    std::vector<uint32_t> first(n_wg + 1, 0), chunks(n_wg, 0);
    pl.max_wg_chunks = 0; pl.total_chunks = 0;
    for (int w = 0; w < n_wg; ++w) {
        first[w] = (uint32_t)items.size();
        for (uint32_t idx : bins[w]) {
            const WorkUnit &u = units[idx];
            const GroupDesc &gd = c->groups[u.group];
            Item it{}; it.db = gd.db; it.ncols = gd.ncols; it.seq0 = gd.seq0; it.half = u.half; it.out_slot = u.out_slot; it.bnd_off = u.bnd_off;
            items.push_back(it);
        }
        chunks[w] = (uint32_t)(load[w] / kChunkCols);
        pl.max_wg_chunks = std::max<uint64_t>(pl.max_wg_chunks, chunks[w]);
        pl.total_chunks += chunks[w];
    }
    first[n_wg] = (uint32_t)items.size();
    pl.n_wg = n_wg;
    std::vector<Item> sorted; sorted.reserve(units.size());
    pl.queue_cols.clear();
    uint64_t before = 0;
    for (uint32_t idx : order) {
        const WorkUnit &u = units[idx];
        const GroupDesc &gd = c->groups[u.group];
        Item it{}; it.db = gd.db; it.ncols = gd.ncols; it.seq0 = gd.seq0; it.half = u.half; it.out_slot = u.out_slot; it.bnd_off = before;
        before += u.ncols;
        sorted.push_back(it);
        pl.queue_cols.push_back(u.ncols);
    }
    pl.n_items = (uint32_t)sorted.size();
    {   // the same list as two interleaved halves, each sorted longest first, boundary offsets counted along this order
        std::vector<Item> split; split.reserve(sorted.size());
        uint64_t off = 0;
        for (int h = 0; h < 2; ++h) {
            pl.split_n[h] = 0; pl.split_cols[h] = 0;
            for (size_t i = (size_t)h; i < sorted.size(); i += 2) {
                Item it = sorted[i];
                it.bnd_off = off;
                off += pl.queue_cols[i];
                pl.split_cols[h] += pl.queue_cols[i];
                pl.split_n[h]++;
                split.push_back(it);
            }
        }
        HIP_TRY(pl.split_items.reserve(split.size()));
        if (list_copy(c, pl.split_items.p, split.data(), split.size() * sizeof(Item))) return 1;
    }
    HIP_TRY(pl.queue_items.reserve(sorted.size()));
    if (list_copy(c, pl.queue_items.p, sorted.data(), sorted.size() * sizeof(Item))) return 1;
    HIP_TRY(pl.items.reserve(items.size()));
    HIP_TRY(pl.wg_first.reserve(first.size()));
    HIP_TRY(pl.wg_chunks.reserve(chunks.size()));
    if (list_copy(c, pl.items.p, items.data(), items.size() * sizeof(Item)) ||
        list_copy(c, pl.wg_first.p, first.data(), first.size() * sizeof(uint32_t)) ||
        list_copy(c, pl.wg_chunks.p, chunks.data(), chunks.size() * sizeof(uint32_t)) || list_sync(c)) return 1;
    return 0;
}

int upload_lane_items(swimm_hip_ctx *c, std::vector<LaneItem> &v, LaneList &ll)
{
    std::stable_sort(v.begin(), v.end(), [](const LaneItem &a, const LaneItem &b) { return a.ncols > b.ncols; });
    uint64_t cols = 0;
    for (LaneItem &it : v) {
        if (cols + it.ncols > 0xFFFFFFFFull) return fail("lane work list exceeds 2^32 boundary columns");
        it.bnd_off = (uint32_t)cols;
        cols += it.ncols;
    }
    ll.n = (uint32_t)v.size();
    ll.cols = cols;
    ll.cell_cols = cols;
    HIP_TRY(ll.items.reserve(v.size()));
    if (list_copy(c, ll.items.p, v.data(), v.size() * sizeof(LaneItem)) || list_sync(c)) return 1;
    return 0;
}

// Which groups leave the workgroup pipeline for the lane-systolic kernel: a group is one serial chain on
// one workgroup, so any group longer than a fraction of the mean per-workgroup load would set the
// kernel's makespan (Swiss-Prot's 35 000-residue titin against a 360-residue mean).  Longest first, move
// groups while ncols > tail_alpha * (remaining columns / n_wg).
std::vector<uint8_t> pick_tail(const swimm_hip_ctx *c, const Range &rg)   // -> flag per group of the range
{
    const uint32_t n = rg.g1 - rg.g0;
    // (a database that is streaming in: "long" is judged against the whole database, not against the range that
    // happens to hold the group -- a range of nothing but the longest sequences is bulk work like any other)
    if (c->streaming_now && c->stream_tail.size() == c->groups.size() && n != c->groups.size())
        return std::vector<uint8_t>(c->stream_tail.begin() + rg.g0, c->stream_tail.begin() + rg.g1);
    std::vector<uint8_t> is_tail(n, 0);
    if (c->opt_tail_mode == 2) return is_tail;                        // never
    if (c->opt_tail_mode == 1) { std::fill(is_tail.begin(), is_tail.end(), 1); return is_tail; }   // always
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->groups[rg.g0 + a].ncols > c->groups[rg.g0 + b].ncols; });
    uint64_t rest = rg.cols;
    // the yardstick is the load of a CU, however many workgroups share it
    for (uint32_t g : order) {
        const double mean = (double)rest / c->num_cu;
        if ((double)c->groups[rg.g0 + g].ncols <= c->opt_tail_frac * 0.01 * mean) break;
        is_tail[g] = 1;
        rest -= c->groups[rg.g0 + g].ncols;
    }
    return is_tail;
}

// work lists of one range for launches of n_wg workgroups: the pipeline kernel's items and the lane-systolic tail
int make_db_plan(swimm_hip_ctx *c, Mode mode, int n_wg, bool no_tail, const Range &rg, bool exact_lengths, DbPlan &dp)
{
    std::vector<WorkUnit> units;
    std::vector<LaneItem> tail;
    uint64_t bnd_cols = 0;
    const uint64_t col0 = rg.g0 < c->group_col_off.size() ? c->group_col_off[rg.g0] : 0;
    if (mode != Mode::I32) {
        std::vector<uint8_t> is_tail = pick_tail(c, rg);
        if (no_tail) std::fill(is_tail.begin(), is_tail.end(), 0);   // every group through the pipeline kernel
        for (uint32_t g = rg.g0; g < rg.g1; ++g) {
            const GroupDesc &gd = c->groups[g];
            if (!is_tail[g - rg.g0]) { units.push_back(WorkUnit{g, 0, 0, gd.ncols, c->group_col_off[g] - col0}); continue; }
            for (uint32_t l = 0; l < 64; ++l) {
                // a pair only runs to the end of its longer member, not to the end of the group (when the lengths are
                // already known: a chunk that is still streaming in runs to the end of its group -- padding scores 0)
                const uint32_t len = exact_lengths ? std::max(c->seq_len[gd.seq0 + l], c->seq_len[gd.seq0 + 64 + l]) : gd.ncols;
                if (len == 0) continue;                 // empty pair: scores stay 0
                LaneItem li{};
                li.db = gd.db; li.lane = l; li.half = 0; li.ncols = (len + kChunkCols - 1) / kChunkCols * kChunkCols;
                li.slot_a = gd.seq0 + l; li.slot_b = gd.seq0 + 64 + l;
                tail.push_back(li);
            }
        }
        bnd_cols = rg.cols;
    } else {
        for (uint32_t g = rg.g0; g < rg.g1; ++g)
            for (uint32_t h = 0; h < 2; ++h)
                units.push_back(WorkUnit{g, h, c->groups[g].seq0 / 64 + h, c->groups[g].ncols,
                                         2 * (c->group_col_off[g] - col0) + (uint64_t)h * c->groups[g].ncols});
        bnd_cols = 2 * rg.cols;
    }
    if (!units.empty()) {
        if (build_plan(c, units, n_wg, dp.main)) return 1;
        dp.main.bnd_cols = bnd_cols;
        dp.have_main = true;
    }
    if (!tail.empty() && upload_lane_items(c, tail, dp.tail)) return 1;
    return 0;
}

// the resident database's work lists, cached per launch shape
int get_db_plan(swimm_hip_ctx *c, Mode mode, int n_wg, bool whole_db, DbPlan **out)
{
    const int key = n_wg | (mode == Mode::I32 ? (1 << 30) : 0) | (whole_db ? (1 << 29) : 0);   // packed int16 and f16 share plans
    auto it = c->plans.find(key);
    if (it != c->plans.end()) { *out = &it->second; return 0; }
    DbPlan &dp = c->plans[key];
    if (make_db_plan(c, mode, n_wg, whole_db, whole_range(c), true, dp)) return 1;
    *out = &dp;
    return 0;
}

void fill_common(const swimm_hip_ctx *c, const QueryPlan &qp, PipeParams &p, uint2 *bnd)
{
    p.prof = c->d_prof.p + qp.prof_off;
    p.prof_stride = qp.mpad;
    p.bnd = bnd;
    p.goe = c->open_gap + c->extend_gap;
    p.ge = c->extend_gap;
}

// columns of boundary rows (64 lanes x 8 B each) the pass-boundary buffer may hold
static uint64_t bnd_budget_cols(const swimm_hip_ctx *c) { return ((uint64_t)c->opt_bnd_mib << 20) / (64 * sizeof(uint2)); }

// Cuts the longest-first item list into runs whose boundary rows fit the budget (always at least one item).  A
// multi-pass query takes every run through all its passes before the next run starts, so the buffer holds one
// run's columns only: 4x the run's tiled residue bytes instead of 4x the whole database.
static void boundary_segments(const swimm_hip_ctx *c, const Plan &pl, std::vector<std::pair<uint32_t, uint32_t>> &segs, uint64_t *max_cols)
{
    const uint64_t budget = bnd_budget_cols(c);
    segs.clear();
    uint64_t mx = 0, cur = 0;
    uint32_t first = 0;
    for (uint32_t i = 0; i < pl.n_items; ++i) {
        if (i > first && cur + pl.queue_cols[i] > budget) { segs.push_back({first, i}); mx = std::max(mx, cur); first = i; cur = 0; }
        cur += pl.queue_cols[i];
    }
    if (pl.n_items > first) { segs.push_back({first, pl.n_items}); mx = std::max(mx, cur); }
    if (max_cols) *max_cols = mx;
}

// Multi-pass query, whole list in one boundary run: the even- and the odd-ranked groups go through their passes as two
// kernels on two streams.  A pass of one half cannot start before the previous pass of the same half has ended, but
// it can start while the other half is in full swing, so the end of every launch -- the last workgroups finishing
// alone, 4 % of a 4 ms pass on a 2e8-residue database -- and the start of the next are covered by the other kernel.
static bool use_split(const swimm_hip_ctx *c, const QueryPlan &qp, const Plan &pl, size_t n_segs)
{
    // (two passes gain nothing: measured -0.3 % on c2; neither do long passes, whose end is a small part of them: c2 with
    // 3 passes of 9 ms each 27.15 ms split, 26.97 ms not -- the split is for passes of up to ~5 ms at 8 000 GCUPS)
    const double pass_cells = (double)pl.total_chunks * kChunkCols * kGroupSeqs * qp.T * qp.W;
    return c->opt_dynamic && c->opt_split && qp.passes > 2 && n_segs == 1 && pl.n_wg >= 2 && pl.split_n[1] >= (uint32_t)pl.n_wg && pass_cells < 4e10;
}

// measurement aid: the launch's own duration, on the stream it runs on (what a kernel trace reports per dispatch)
static int timed_launch(swimm_hip_ctx *c, Mode mode, int T, int W, int n_wg, const PipeParams &p, hipStream_t st)
{
    if (!c->opt_time_launches) { HIP_TRY(launch_pipe(mode, T, W, n_wg, p, st)); return 0; }
    while (c->launch_ev.size() < c->launch_ev_used + 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->launch_ev.push_back(e);
    }
    const bool dbg = getenv("SWIMM_HIP_DEBUG") != nullptr;
    const double t0 = dbg ? now_s() : 0;
    HIP_TRY(hipEventRecord(c->launch_ev[c->launch_ev_used], st));
    const double t1 = dbg ? now_s() : 0;
    HIP_TRY(launch_pipe(mode, T, W, n_wg, p, st));
    const double t2 = dbg ? now_s() : 0;
    HIP_TRY(hipEventRecord(c->launch_ev[c->launch_ev_used + 1], st));
    if (dbg) fprintf(stderr, "swimm_hip: host time of a timed launch: event %.3f ms, launch %.3f ms, event %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    c->launch_ev_used += 2;
    return 0;
}

// Group-resident passes (sw_pipe_kernel<.., RES = true>): one launch per multi-pass query, no launch boundary between passes
// and no boundary rows shared between workgroups.
// boundary scratch of that mode: per workgroup, the columns of the longest group of the list (64 lanes x 8 B each)
static uint64_t resident_bnd_elems(const Plan &pl) { return pl.n_items ? (uint64_t)pl.n_wg * pl.queue_cols[0] * 64 : 0; }

// One group-resident launch for a batch of queries that share the launch shape: the items are (group, query) pairs, every
// workgroup takes an item through all the passes of its query back to back.  `qd` = the batch's entries in d_qdesc.
int run_resident_batch(swimm_hip_ctx *c, Mode mode, int T, int W, const Plan &pl, const QDesc *qd, uint32_t nq, uint64_t pass_sum, uint32_t max_passes,
                       hipStream_t st, DevBuf<uint2> &bnd)
{
    PipeParams p{};
    p.prof = c->d_prof.p;
    p.prof_stride = 0;
    p.bnd = bnd.p;
    p.goe = c->open_gap + c->extend_gap;
    p.ge = c->extend_gap;
    if (c->queue_next >= c->d_queue.cap) return fail("pipeline launch cursors exhausted");
    if (bnd.cap < resident_bnd_elems(pl) && max_passes > 1) return fail("internal: boundary scratch too small");
    const uint64_t n_virtual = (uint64_t)pl.n_items * nq;
    if (n_virtual > 0xFFFFFFF0ull) return fail("group-resident batch of %u queries x %u groups exceeds 2^32 items", nq, pl.n_items);
    p.items = pl.queue_items.p;
    p.n_items = (uint32_t)n_virtual;
    const int n_wg = (int)std::min<uint64_t>((uint64_t)pl.n_wg, n_virtual);
    // every item-pass takes its chunks, or the pipeline's depth if it is shorter than that
    p.max_steps = (uint32_t)std::min<uint64_t>((pl.total_chunks + (uint64_t)pl.n_items * (kMaxWaves + 1)) * pass_sum + kMaxWaves + 1, 0x3ffffff0u);
    p.queue = c->d_queue.p + c->queue_next++;
    p.qdesc = qd;
    p.n_queries = nq;
    p.bnd_wg_cols = pl.queue_cols[0];
    p.r0 = 0;
    p.first_pass = 1; p.last_pass = 0;
    p.out = c->d_scores.p;
    p.err = c->d_err.p;
#ifdef SWIMM_STAMPS
    HIP_TRY(c->d_stamps.reserve(16 * 8 + 3072));
    HIP_TRY(hipMemsetAsync(c->d_stamps.p, 0, 16 * 8 * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(c->d_stamps.p + 15 * 8 + 2, 0xff, sizeof(unsigned long long), st));   // min slot
    p.stamps = c->d_stamps.p;
#endif
    if (timed_launch(c, mode, T, W, n_wg, p, st)) return 1;
#ifdef SWIMM_STAMPS
    {
        unsigned long long h[16 * 8];
        HIP_TRY(hipMemcpyAsync(h, c->d_stamps.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int w = 0; w < W; ++w)
            fprintf(stderr, "stamps (resident batch of %u) wave %2d: load/wait %8.0f  compute %8.0f  tail %8.0f  barrier %8.0f  cycles per active step (%llu active of %llu steps per wg)\n", nq,
                    w, (double)h[w * 8 + 0] / h[w * 8 + 4], (double)h[w * 8 + 1] / h[w * 8 + 4], (double)h[w * 8 + 2] / h[w * 8 + 4],
                    (double)h[w * 8 + 3] / h[w * 8 + 4], h[w * 8 + 4] / n_wg, h[w * 8 + 5] / n_wg);
        fprintf(stderr, "stamps: workgroup run time mean %.1f us, longest %.1f us; first start to last end %.1f us (%llu workgroups)\n",
                (double)h[15 * 8 + 0] / h[15 * 8 + 4] / 100.0, (double)h[15 * 8 + 1] / 100.0, (double)(h[15 * 8 + 3] - h[15 * 8 + 2]) / 100.0, h[15 * 8 + 4]);
    }
#endif
    c->launches++;
    c->cells += pl.total_chunks * kChunkCols * (uint64_t)(W * T) * pass_sum * (mode == Mode::I32 ? 64 : 128);
    return 0;
}

// `st`: stream of the one-kernel-per-pass path (one-pass queries rotate over three streams)
int run_passes(swimm_hip_ctx *c, Mode mode, const QueryPlan &qp, const Plan &pl, int32_t *out_row, hipStream_t st, bool allow_split, DevBuf<uint2> &bnd)
{
    std::vector<std::pair<uint32_t, uint32_t>> segs;
    uint64_t seg_cols = pl.bnd_cols;
    if (c->opt_dynamic && qp.passes > 1) boundary_segments(c, pl, segs, &seg_cols);
    else segs.push_back({0u, pl.n_items});
    if (qp.passes > 1 && bnd.cap < seg_cols * 64) return fail("internal: boundary buffer too small");
    if (allow_split && use_split(c, qp, pl, segs.size())) {
        HIP_TRY(hipEventRecord(c->ev_a, c->stream));               // stream B joins after everything queued so far
        HIP_TRY(hipStreamWaitEvent(c->stream_b, c->ev_a, 0));
        // every kernel asks for the full complement of workgroups: the two kernels of a pass share the CUs while both
        // have work, and the one that still has groups left takes over the slots the other one frees
        const int n_half = pl.n_wg;
        for (int pass = 0; pass < qp.passes; ++pass)
            for (int h = 0; h < 2; ++h) {
                PipeParams p{};
                fill_common(c, qp, p, bnd.p);
                if (c->queue_next >= c->d_queue.cap) return fail("pipeline launch cursors exhausted");
                p.items = pl.split_items.p + (h ? pl.split_n[0] : 0);
                p.n_items = pl.split_n[h];
                p.max_steps = (uint32_t)std::min<uint64_t>(pl.split_cols[h] / kChunkCols + kMaxWaves + 1, 0x3ffffff0u);
                p.queue = c->d_queue.p + c->queue_next++;
                p.r0 = (uint32_t)(pass * qp.W * qp.T);
                p.first_pass = pass == 0;
                p.last_pass = pass == qp.passes - 1;
                p.out = out_row;
                p.err = c->d_err.p;
                if (timed_launch(c, mode, qp.T, qp.W, (int)std::min<uint32_t>((uint32_t)n_half, pl.split_n[h]), p, h ? c->stream_b : c->stream)) return 1;
                c->launches++;
                c->cells += pl.split_cols[h] * (uint64_t)(qp.W * qp.T) * (mode == Mode::I32 ? 64 : 128);
            }
        HIP_TRY(hipEventRecord(c->ev_b, c->stream_b));              // and the main stream continues after both halves
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_b, 0));
        return 0;
    }
    for (const auto &sg : segs) {
        uint64_t col0 = 0, seg_chunks = pl.total_chunks;
        if (c->opt_dynamic) {
            col0 = 0; seg_chunks = 0;
            for (uint32_t i = 0; i < sg.first; ++i) col0 += pl.queue_cols[i];
            for (uint32_t i = sg.first; i < sg.second; ++i) seg_chunks += pl.queue_cols[i] / kChunkCols;
        }
        const int n_wg = c->opt_dynamic ? (int)std::min<uint32_t>((uint32_t)pl.n_wg, sg.second - sg.first) : pl.n_wg;
        for (int pass = 0; pass < qp.passes; ++pass) {
            PipeParams p{};
            fill_common(c, qp, p, bnd.p);
            p.items = pl.items.p;
            p.wg_first = pl.wg_first.p;
            p.wg_chunks = pl.wg_chunks.p;
            if (c->opt_dynamic) {
                if (c->queue_next >= c->d_queue.cap) return fail("pipeline launch cursors exhausted");
                p.items = pl.queue_items.p + sg.first;
                p.n_items = sg.second - sg.first;
                p.max_steps = (uint32_t)std::min<uint64_t>(seg_chunks + kMaxWaves + 1, 0x3ffffff0u);
                p.queue = c->d_queue.p + c->queue_next++;
                p.bnd = bnd.p - col0 * 64;      // the items' offsets count columns from the start of the whole list
            }
            p.r0 = (uint32_t)(pass * qp.W * qp.T);
            p.first_pass = pass == 0;
            p.last_pass = pass == qp.passes - 1;
            p.out = out_row;
#ifdef SWIMM_STAMPS
            HIP_TRY(c->d_stamps.reserve(16 * 8 + 3072));
            HIP_TRY(hipMemsetAsync(c->d_stamps.p, 0, 16 * 8 * sizeof(unsigned long long), st));
            HIP_TRY(hipMemsetAsync(c->d_stamps.p + 15 * 8 + 2, 0xff, sizeof(unsigned long long), st));   // min slot
            p.stamps = c->d_stamps.p;
#endif
            p.err = c->d_err.p;
            if (timed_launch(c, mode, qp.T, qp.W, n_wg, p, st)) return 1;
#ifdef SWIMM_STAMPS
            {
                unsigned long long h[16 * 8];
                HIP_TRY(hipMemcpyAsync(h, c->d_stamps.p, sizeof h, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                for (int w = 0; w < qp.W; ++w)
                    fprintf(stderr, "stamps wave %2d: load/wait %8.0f  compute %8.0f  tail %8.0f  barrier %8.0f  cycles per active step (%llu active of %llu steps per wg)\n",
                            w, (double)h[w * 8 + 0] / h[w * 8 + 4], (double)h[w * 8 + 1] / h[w * 8 + 4], (double)h[w * 8 + 2] / h[w * 8 + 4],
                            (double)h[w * 8 + 3] / h[w * 8 + 4], h[w * 8 + 4] / pl.n_wg, h[w * 8 + 5] / pl.n_wg);
                fprintf(stderr, "stamps: most steps of any workgroup %llu, longest workgroup %.0f cycles\n", h[6], (double)h[7]);
                if (getenv("SWIMM_STAMPS_DUMP")) {
                    std::vector<unsigned long long> pw(3072);
                    HIP_TRY(hipMemcpy(pw.data(), c->d_stamps.p + 128, pw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    const unsigned long long t0 = h[15 * 8 + 2];
                    for (int b = 0; b < std::min(pl.n_wg, 1024); b += (b < 16 ? 1 : 37))
                        fprintf(stderr, "  wg %4d: start %7.1f us end %7.1f us chunks %llu\n", b, (double)(pw[2048 + b] - t0) / 100.0, (double)(pw[b] - t0) / 100.0, pw[1024 + b]);
                }
                fprintf(stderr, "stamps: workgroup run time mean %.1f us, longest %.1f us; first start to last end %.1f us (%llu workgroups)\n",
                        (double)h[15 * 8 + 0] / h[15 * 8 + 4] / 100.0, (double)h[15 * 8 + 1] / 100.0, (double)(h[15 * 8 + 3] - h[15 * 8 + 2]) / 100.0, h[15 * 8 + 4]);
            }
#endif
            c->launches++;
            c->cells += seg_chunks * kChunkCols * (uint64_t)(qp.W * qp.T) * (mode == Mode::I32 ? 64 : 128);
        }
    }
    return 0;
}

int run_lane_passes(swimm_hip_ctx *c, Mode mode, const QueryPlan &qp, int m, const LaneList &ll, int32_t *out_row, hipStream_t st,
                    LaneScratch &sc)
{
    if (ll.n == 0) return 0;
    const int rows_pass = 64 * kLaneRows;
    const int passes = (m + rows_pass - 1) / rows_pass;
    const size_t need_bnd = passes > 1 ? (size_t)ll.cols + 64 : 0, need_prog = (size_t)passes * ll.n;
    if (sc.bnd[0].cap < need_bnd || sc.bnd[1].cap < need_bnd || sc.queue.cap < (size_t)passes || sc.prog.cap < need_prog)
        return fail("internal: lane scratch too small (%zu/%zu columns, %zu/%zu counters)", sc.bnd[0].cap, need_bnd, sc.prog.cap, need_prog);
    // never more than one workgroup per CU in total: the whole grid becomes resident (a pass waits for the one
    // before it), pass-major so that producers are dispatched first; 4 waves per workgroup
    int per_pass = passes > 1 ? std::max(1, c->num_cu / passes) : c->num_cu * 6;   // a single pass chains nothing: fill the chip
    per_pass = (int)std::min<uint64_t>(per_pass, (ll.n + 3) / 4);
    if (passes > c->num_cu) return fail("query of %d rows needs %d chained passes, more than the %d CUs", m, passes, c->num_cu);
    LaneParams p{};
    p.items = ll.items.p;
    p.n_items = ll.n;
    p.queue = sc.queue.p;
    p.prog = sc.prog.p;
    p.prof = c->d_prof.p + qp.prof_off;
    p.prof_stride = qp.mpad;
    p.m = (uint32_t)m;
    p.passes = (uint32_t)passes;
    p.wg_per_pass = (uint32_t)per_pass;
    p.bnd[0] = sc.bnd[0].p;
    p.bnd[1] = sc.bnd[1].p;
    p.bnd_dummy = (uint32_t)ll.cols;
    p.out = out_row;
    p.goe = c->open_gap + c->extend_gap;
    p.ge = c->extend_gap;
    p.err = c->d_err.p;
    p.agent_acquire = c->opt_lane_acquire;
    HIP_TRY(hipMemsetAsync(sc.queue.p, 0, passes * sizeof(uint32_t), st));
    if (passes > 1) HIP_TRY(hipMemsetAsync(sc.prog.p, 0, need_prog * sizeof(uint32_t), st));
    // a one-pass launch of a short query uses fewer rows per lane: the serial walk down a lane's rows is the step latency
    const int rows_per_lane = (passes == 1 && m <= 128 && c->opt_lane_rows) ? 2 : (passes == 1 && m <= 256 && c->opt_lane_rows) ? 4 : kLaneRows;
    HIP_TRY(launch_lane(mode, rows_per_lane, passes * per_pass, p, st));
    c->launches++;
    c->cells += ll.cell_cols * (uint64_t)rows_pass * passes * (mode == Mode::PK16 ? 2 : 1);
    return 0;
}

int reserve_lane_scratch(LaneScratch &sc, size_t cols, size_t items, int passes)
{
    HIP_TRY(sc.queue.reserve(256));
    if (passes > 1) {
        HIP_TRY(sc.bnd[0].reserve(cols + 64));
        HIP_TRY(sc.bnd[1].reserve(cols + 64));
    }
    HIP_TRY(sc.prog.reserve(std::max<size_t>(1, items * (size_t)passes)));
    return 0;
}

// the work lists are derived from the resident database: rebuild them after it changed
int refresh_plans(swimm_hip_ctx *c)
{
    if (!c->groups_dirty) return 0;
    release_plans(c);
    c->groups_dirty = false;
    return 0;
}

// Registers a chunk's device groups (geometry only: nothing is copied here).
int register_chunk(swimm_hip_ctx *c, ChunkRec &rec, const std::vector<uint32_t> &lens_or_empty)
{
    const uint32_t dev_groups = rec.n_groups;
    uint64_t bytes = 0;
    for (uint32_t g = 0; g < dev_groups; ++g) { rec.goff[g] = bytes; bytes += (uint64_t)rec.gcols[g] * kGroupSeqs; }
    HIP_TRY(hipMalloc((void **)&rec.d_tiled, std::max<uint64_t>(bytes, 16)));
    if (rec.kind == 0) {
        hipError_t e = hipMalloc((void **)&rec.d_len, (size_t)dev_groups * kGroupSeqs * sizeof(uint32_t));
        if (e != hipSuccess) { (void)hipFree(rec.d_tiled); rec.d_tiled = nullptr; return fail("hipMalloc(sequence lengths): %s", hipGetErrorString(e)); }
    }
    if (hipEventCreateWithFlags(&rec.ready, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(rec.d_tiled); (void)hipFree(rec.d_len);
        return fail("hipEventCreate failed");
    }
    rec.group0 = (uint32_t)c->groups.size();
    rec.cols = 0;
    for (uint32_t g = 0; g < dev_groups; ++g) {
        GroupDesc gd;
        gd.db = rec.d_tiled + rec.goff[g];
        gd.ncols = rec.gcols[g];
        gd.seq0 = (uint32_t)((rec.group0 + g) * kGroupSeqs);
        c->groups.push_back(gd);
        c->group_col_off.push_back(c->total_cols);
        c->total_cols += rec.gcols[g];
        rec.cols += rec.gcols[g];
    }
    const size_t base = c->seq_len.size();
    c->seq_len.resize(base + (size_t)dev_groups * kGroupSeqs, 0);
    for (size_t i = 0; i < lens_or_empty.size(); ++i) c->seq_len[base + i] = lens_or_empty[i];
    rec.lens_known = rec.kind == 1;
    c->chunks.push_back(std::move(rec));
    c->groups_dirty = true;
    release_plans(c);
    return 0;
}

// X2 (MICsearch.c:85-88): one chunk's bytes to the device and into device groups, all on the upload stream.  The
// copies come from pageable memory, so every hipMemcpyAsync returns only when its source has been consumed; what
// stays asynchronous is the (re-)tile kernel, whose end `ready` marks.  Scratch is reused chunk after chunk (the
// stream is in order: the next chunk's copy cannot overtake this chunk's kernel).
int upload_chunk(swimm_hip_ctx *c, ChunkRec &r)
{
    if (r.uploaded) return 0;
    hipStream_t s = c->stream_up;
    const uint32_t dev_groups = r.n_groups;
    uint32_t max_cols = 0;
    for (uint32_t x : r.gcols) max_cols = std::max(max_cols, x);
    const double t_up0 = now_s();
    HIP_TRY(c->up_gcols.reserve(dev_groups));
    HIP_TRY(c->up_goff.reserve(dev_groups));
    HIP_TRY(hipMemcpyAsync(c->up_gcols.p, r.gcols.data(), dev_groups * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->up_goff.p, r.goff.data(), dev_groups * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    if (r.kind == 0) {
        HIP_TRY(c->up_b.reserve(std::max<uint64_t>(r.vD, 16)));
        HIP_TRY(c->up_n.reserve(r.group_count));
        HIP_TRY(c->up_disp.reserve(r.group_count));
        HIP_TRY(hipMemsetAsync(r.d_len, 0, (size_t)dev_groups * kGroupSeqs * sizeof(uint32_t), s));
        HIP_TRY(hipMemcpyAsync(c->up_n.p, r.h_n, r.group_count * sizeof(uint16_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->up_disp.p, r.h_disp, r.group_count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->up_b.p, r.h_b, r.vD, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(c->ev_copied, s));
        HIP_TRY(launch_retile(c->up_b.p, c->up_n.p, c->up_disp.p, r.group_count, r.vl, c->up_goff.p, c->up_gcols.p, dev_groups, max_cols, r.d_tiled, r.d_len, s));
    } else {
        HIP_TRY(c->up_b.reserve(std::max<uint64_t>(r.code_bytes, 16)));
        HIP_TRY(c->up_off.reserve(r.off.size()));
        HIP_TRY(hipMemcpyAsync(c->up_off.p, r.off.data(), r.off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->up_b.p, r.h_codes, r.code_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(c->ev_copied, s));
        HIP_TRY(launch_tile_sequences(c->up_b.p, c->up_off.p, (uint32_t)r.n_seq, c->up_goff.p, c->up_gcols.p, dev_groups, max_cols, r.d_tiled, s));
    }
    HIP_TRY(hipEventRecord(r.ready, s));
    HIP_TRY(hipEventSynchronize(c->ev_copied));      // the caller's buffers have been read
    if (getenv("SWIMM_HIP_DEBUG")) {
        const uint64_t bytes = r.kind == 0 ? r.vD : r.code_bytes;
        fprintf(stderr, "swimm_hip: chunk of %.1f MB copied in %.2f ms (%.1f GB/s)\n", bytes / 1e6, (now_s() - t_up0) * 1e3, bytes / 1e9 / (now_s() - t_up0));
    }
    r.uploaded = true;
    r.h_b = nullptr; r.h_n = nullptr; r.h_disp = nullptr; r.h_codes = nullptr;
    std::vector<uint32_t>().swap(r.off);
    return 0;
}

// The uploader of a database that streams in (option "lazy_upload"): a thread of its own, one per context, started when
// the first chunk is recorded and parked between searches.  The copies come from pageable memory, so each one blocks its
// caller for the length of the transfer: on this thread the link is busy back to back (0.6 GB in 13-15 ms) while the
// searching thread plans, builds work lists and launches, and since the link delivers 1.8x faster than the kernels
// consume, the GPU waits for the first range only.  (A thread per search would do, but its first HIP call costs
// up to 5 ms on some runs.)
struct Uploader {
    swimm_hip_ctx *c;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<size_t> order;      // the job: chunk indices in the order they travel
    bool have_job = false, busy = false, quit = false, stop = false;
    size_t issued = 0;              // the first `issued` chunks of `order` have their `ready` event recorded
    bool failed = false;
    std::string err;

    explicit Uploader(swimm_hip_ctx *ctx) : c(ctx) { th = std::thread([this]() { run(); }); }
    ~Uploader()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    void run()
    {
        bool dev_ok = hipSetDevice(c->device) == hipSuccess;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&]() { return have_job || quit; });
            if (quit) return;
            have_job = false;
            const std::vector<size_t> job = order;
            lk.unlock();
            bool ok = dev_ok;
            std::string e = ok ? "" : "uploader: hipSetDevice failed";
            for (size_t i = 0; i < job.size(); ++i) {
                bool skip;
                { std::lock_guard<std::mutex> g(mu); skip = stop; }
                if (ok && !skip && upload_chunk(c, c->chunks[job[i]])) { ok = false; e = g_err; }
                std::lock_guard<std::mutex> g(mu);
                issued = i + 1; failed = !ok; err = e;
                cv.notify_all();
            }
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void post(const std::vector<size_t> &job)
    {
        { std::lock_guard<std::mutex> lk(mu); order = job; issued = 0; failed = false; err.clear(); stop = false; have_job = true; busy = true; }
        cv.notify_all();
    }
    int wait_issued(size_t n, std::string *e)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return issued >= n || failed || !busy; });
        if (failed) { *e = err; return 1; }
        return issued >= n ? 0 : 1;
    }
    void finish(bool abandon)       // the job has been walked to its end (abandon: without copying what is left)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (abandon) stop = true;
        cv.wait(lk, [&]() { return !busy; });
    }
};

int ensure_uploader(swimm_hip_ctx *c)
{
    if (!c->up) c->up = new Uploader(c);
    return 0;
}

// true lengths of the chunk-layout chunks' slots come from the re-tile kernel: fetched when somebody needs them
// (lane-systolic work lists, promotion re-runs), not inside add_chunk
int sync_lengths(swimm_hip_ctx *c)
{
    bool any = false;
    for (ChunkRec &r : c->chunks) {
        if (r.lens_known || !r.uploaded) continue;
        HIP_TRY(hipMemcpyAsync(c->seq_len.data() + (size_t)r.group0 * kGroupSeqs, r.d_len, (size_t)r.n_groups * kGroupSeqs * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, c->stream_up));
        r.lens_known = true;
        any = true;
    }
    if (any) HIP_TRY(hipStreamSynchronize(c->stream_up));
    return 0;
}

// Device part of a search for the queries [qb, qe) (ascending-length order of set_queries): leaves exact scores in
// d_scores[(q - qb) * S + local_slot].  The callers walk the query list in batches whose score rows fit the
// `score_mib` budget.  One object per call; the phases run in the order of run().
struct SearchRun {
    swimm_hip_ctx *c;
    uint32_t qb, qe, qn = 0;
    bool dbg = false;
    double t_begin = 0, t_sized = 0, t_issued = 0;
    const uint16_t *qm = nullptr;           // the batch's query lengths / offsets into qcodes
    const uint32_t *qdisp = nullptr;
    uint64_t S = 0;                         // score slots per query
    // the database: one range (resident, cached work lists) or the ranges a lazily uploaded database streams in as
    bool streaming = false;
    std::vector<Range> ranges;
    std::vector<std::pair<size_t, size_t>> range_chunks;     // streaming: positions [first, last) in `up_order` of every range's chunks
    std::vector<size_t> up_order;                            // streaming: the chunks in the order they travel
    std::vector<std::map<int, DbPlan>> stream_plans;         // streaming: work lists per (range, workgroup count), released when the search has drained
    // the launch plan
    Mode main_mode = Mode::F16;
    bool lane_room = false, many_short = false, alternate = false;
    uint32_t longest_cols = 0;
    std::vector<QueryPlan> qps;
    std::vector<uint8_t> rotated;
    std::vector<std::vector<QueryPlan>> rqps;                // streaming, per-pass launches: a launch shape per (range, query)
    size_t prof_elems = 0;
    int tail_lanes = 1;                     // tail launches in flight at a time (decided with the buffer sizes)

    SearchRun(swimm_hip_ctx *ctx, uint32_t b, uint32_t e) : c(ctx), qb(b), qe(e) {}
    ~SearchRun()
    {
        if (!streaming) return;
        if (c->up) c->up->finish(true);          // (an early return: the chunks not yet copied stay where they are)
        (void)hipDeviceSynchronize();
        release_stream_plans();
    }
    void release_stream_plans()
    {
        for (auto &m : stream_plans) for (auto &kv : m) { kv.second.main.release(); kv.second.tail.release(); }
        stream_plans.clear();
    }
    int wait_uploaded(size_t n)             // until the first n chunks of `up_order` are on their way
    {
        std::string err;
        if (c->up->wait_issued(n, &err)) return fail("%s", err.empty() ? "upload failed" : err.c_str());
        return 0;
    }
    const QueryPlan &qp_of(size_t ri, uint32_t q) const { return rqps.empty() ? qps[q] : rqps[ri][q]; }
    int plan_of(size_t ri, uint32_t q, DbPlan **out);

    int begin(uint64_t *slots_out);
    int layout_ranges();
    int plan_queries();
    int upload_profiles();
    int size_buffers();
    int issue();
    int promotion_ladder();
    int drain();
    int run(uint64_t *slots_out)
    {
        return begin(slots_out) || layout_ranges() || plan_queries() || upload_profiles() || size_buffers() || issue() || promotion_ladder() || drain();
    }
};

// the work lists of a (range, launch shape): cached for the resident database, temporary for a streaming chunk
int SearchRun::plan_of(size_t ri, uint32_t q, DbPlan **out)
{
    int per_cu = 1;
    if (wgs_per_cu(c, main_mode, qp_of(ri, q).T, qp_of(ri, q).W, c->batch_now && !rotated[q], &per_cu)) return 1;
    const int n_wg = n_workgroups(c, per_cu);
    if (!streaming) return get_db_plan(c, main_mode, n_wg, rotated[q] != 0 || qps[q].resident, out);
    auto it = stream_plans[ri].find(n_wg);
    if (it == stream_plans[ri].end()) {
        DbPlan &dp = stream_plans[ri][n_wg];
        bool exact = true;
        for (size_t ci = range_chunks[ri].first; ci < range_chunks[ri].second; ++ci) exact = exact && c->chunks[up_order[ci]].lens_known;
        if (make_db_plan(c, main_mode, n_wg, qps[q].resident, ranges[ri], exact, dp)) return 1;
        *out = &dp;
    } else {
        *out = &it->second;
    }
    return 0;
}

int SearchRun::begin(uint64_t *slots_out)
{
    if (!c->have_queries) return fail("swimm_hip_search: no queries set");
    if (c->groups.empty()) return fail("swimm_hip_search: no database chunk resident");
    HIP_TRY(hipSetDevice(c->device));
    if (refresh_plans(c)) return 1;
    qn = qe - qb;
    dbg = getenv("SWIMM_HIP_DEBUG") != nullptr;
    t_begin = now_s();
    qm = c->qm.data() + qb;
    qdisp = c->qdisp.data() + qb;
    S = (uint64_t)c->groups.size() * kGroupSeqs;
    *slots_out = S;

    return 0;
}

int SearchRun::layout_ranges()
{
    // Chunks whose bytes are still on the host (option "lazy_upload"): this search streams them in -- chunk k+1 is
    // copied and tiled on the upload stream while chunk k is being aligned (X2 overlapped with compute,
    // MICsearch.c:85-91) -- and every chunk is then one range with work lists of its own.  Otherwise the whole resident
    // database is one range with cached work lists.
    streaming = false;
    for (const ChunkRec &r : c->chunks) streaming = streaming || !r.uploaded;
    c->streaming_now = streaming;
    if (streaming) {
        uint64_t mb = 16, mn = 1, mg = 1, mo = 1;
        for (const ChunkRec &r : c->chunks) {
            if (r.uploaded) continue;
            mb = std::max<uint64_t>(mb, r.kind == 0 ? r.vD : r.code_bytes); mn = std::max<uint64_t>(mn, r.group_count);
            mg = std::max<uint64_t>(mg, r.n_groups); mo = std::max<uint64_t>(mo, r.off.size());
        }
        // Consecutive chunks form a range, and every range is as large as it can be without the GPU running dry before
        // it has arrived: the link delivers a chunk in bytes / 40 GB/s, the kernels consume it in (rows of all queries) x
        // residues / 8 000 GCUPS -- 1.9x longer for one 375-row query, so the first range is one chunk, the second one or
        // two, and the rest of the database follows in two or three large launches; a batch of long queries is
        // compute-bound from the first chunk on and runs as two ranges.
        // The end of the database with the LONGER sequences travels first: consecutive ranges run on alternating
        // streams, so the few long chains a range ends with are covered by the next range's workgroups -- and the last
        // range, which nothing covers, is then the one with the short sequences, whose launches end evenly.
        c->stream_tail.clear();
        c->stream_tail = pick_tail(c, whole_range(c));
        const size_t nc = c->chunks.size();
        const bool descending = nc > 1 && (double)c->chunks[nc - 1].cols / std::max<uint32_t>(1, c->chunks[nc - 1].n_groups) >
                                              (double)c->chunks[0].cols / std::max<uint32_t>(1, c->chunks[0].n_groups);
        for (size_t i = 0; i < nc; ++i) up_order.push_back(descending ? nc - 1 - i : i);
        double rows = 0;
        for (uint32_t q = 0; q < qn; ++q) rows += qm[q];
        auto up_s = [&](const ChunkRec &r) { return (double)(r.kind == 0 ? r.vD : r.code_bytes) / 40e9; };
        auto dp_s = [&](const ChunkRec &r) { return 0.85 * rows * (double)r.cols * kGroupSeqs / 8000e9; };   // (rather too short: the GPU must not wait)
        double t_up = 0, t_gpu = 0;
        for (size_t i = 0; i < nc;) {
            Range rg; rg.g0 = c->chunks[up_order[i]].group0; rg.g1 = rg.g0 + c->chunks[up_order[i]].n_groups; rg.cols = 0;
            const size_t first = i;
            double work = 0;
            do {
                const ChunkRec &r = c->chunks[up_order[i]];
                rg.g0 = std::min(rg.g0, r.group0); rg.g1 = std::max(rg.g1, r.group0 + r.n_groups); rg.cols += r.cols;
                t_up += up_s(r); work += dp_s(r);
                ++i;
            } while (i < nc && t_up + up_s(c->chunks[up_order[i]]) <= t_gpu);
            t_gpu = std::max(t_gpu, t_up) + work;
            ranges.push_back(rg);
            range_chunks.push_back({first, i});
        }
        // the upload scratch grows now, not between two chunks (growing frees the old buffer)
        HIP_TRY(c->up_b.reserve(mb)); HIP_TRY(c->up_n.reserve(mn)); HIP_TRY(c->up_disp.reserve(mn));
        HIP_TRY(c->up_gcols.reserve(mg)); HIP_TRY(c->up_goff.reserve(mg)); HIP_TRY(c->up_off.reserve(mo));
    } else {
        if (sync_lengths(c)) return 1;
        ranges.push_back(whole_range(c));
    }
    // The uploader (Uploader, above) walks the chunk list from the first moment of the search while this thread plans,
    // builds work lists and launches.
    if (streaming) {
        if (ensure_uploader(c)) return 1;
        c->up->post(up_order);
    }
    stream_plans.resize(streaming ? ranges.size() : 0);
    return 0;
}

int SearchRun::plan_queries()
{
    // query profiles prof[q][d][row] = submat[query[row]*32 + d] (queryProfiles, MICsearch.c:34-36,
    // transposed so that consecutive query rows are contiguous for one residue code); rows past the
    // query's end are zero, like the reference's dummy row 23
    main_mode = c->opt_force_i32 ? Mode::I32 : (c->opt_f16 ? Mode::F16 : Mode::PK16);
    // a database with a long-sequence tail is searched with launch shapes that leave room for lane-systolic waves
    lane_room = false;
    longest_cols = 0;
    for (const GroupDesc &g : c->groups) longest_cols = std::max(longest_cols, g.ncols);
    if (c->opt_tail_mode != 2 && main_mode != Mode::I32)
        lane_room = c->opt_tail_mode == 1 || (double)longest_cols > c->opt_tail_frac * 0.01 * (double)c->total_cols / c->num_cu;
    if (c->opt_lane_room >= 0) lane_room = c->opt_lane_room != 0 && main_mode != Mode::I32;
    if (dbg) fprintf(stderr, "swimm_hip: ranges laid out, uploader started %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    qps.assign(qn, QueryPlan{});
    rotated.assign(qn, 0);
    uint32_t n_short = 0;
    for (uint32_t q = 0; q < qn; ++q) n_short += qm[q] <= 64 * kLaneRows;
    // (with a handful of short queries the last ones' chains would stick out at the end of the search; and a database
    // with an extreme sequence -- c3's 35 000 residues are 6x a CU's mean load -- keeps the tail kernel, whose chain
    // is 3.6x faster per column than a 4-wave workgroup's)
    many_short = n_short >= 8 && !streaming;
    const bool rotate = many_short && c->opt_rotate && (double)longest_cols <= 2.0 * (double)c->total_cols / c->num_cu;
    prof_elems = 0;
    // Group-resident batch launches (option "resident"): ONE launch per launch shape whose items are (group, query) pairs.
    // Every query gets the 4-wave shape that wastes the fewest padded rows at that shape's rate, and the queries of a shape
    // run together -- short one-pass queries included: a batch is the better home for them than the rotation below (300
    // queries of 80-120 residues against 1e8: 6 290 -> 6 600 GCUPS with one shape for all, more with a shape per query).
    c->batch_now = c->opt_dynamic && (c->opt_resident == 1 || (c->opt_resident < 0 && (qn >= 2 || streaming))) &&
                   !((uint64_t)qn * S > 0xFFFFFFFFull || prof_elems_bound(qm, qn) > 0xFFFFFFFFull);   // (a batch addresses score rows and profiles with 32-bit offsets)
    // (no register room is reserved for the lane-systolic waves here: among the 4-wave shapes only the 8-row one would pass that
    // filter, at 6 000 instead of 8 400 GCUPS)
    if (c->batch_now) {
        // A batch launch is one persistent kernel for the whole batch: lane-systolic tail kernels launched beside it would
        // find no free slot until it ends (measured: c3 -16 %, a 1e8-residue database -25 %).  So a batch takes EVERY group
        // through the pipeline kernel -- which is fine as long as the longest item (longest group x the passes of the
        // longest query; the queue hands it out first) is at most half a workgroup's share of the batch; otherwise (c3: a
        // 35 000-residue sequence is 2.3 shares) the batch is not formed and the queries run one launch per pass beside their
        // tail kernels.
        double share = 0;
        uint32_t max_passes = 1;
        int most_wg = 1;
        if (choose_batch_shapes(c, main_mode, qm, qn, qps)) return 1;
        for (uint32_t q = 0; q < qn; ++q) {
            int per_cu = 1;
            if (wgs_per_cu(c, main_mode, qps[q].T, qps[q].W, true, &per_cu)) return 1;
            share += (double)qps[q].passes * (double)c->total_cols / n_workgroups(c, per_cu);
            max_passes = std::max<uint32_t>(max_passes, (uint32_t)qps[q].passes);
            most_wg = std::max(most_wg, n_workgroups(c, per_cu));
        }
        if (dbg)
            fprintf(stderr, "swimm_hip: batch of %u queries: longest item %u columns x %u passes, a workgroup's share %.0f column-passes, %zu groups for up to %d workgroups\n",
                    qn, longest_cols, max_passes, share, c->groups.size(), most_wg);
        // (a database that streams in as several ranges: consecutive ranges overlap on two streams, the long items travel
        // and start first, and the last range is the one with the short sequences -- the longest item may take 0.9 of
        // the whole search before it sticks out at the end)
        const bool ranges_overlap = streaming && ranges.size() > 1;
        if (c->opt_resident < 0 && (double)longest_cols * max_passes > (ranges_overlap ? 0.9 : 0.5) * share) c->batch_now = false;
        // ... and it pays on a database that is small for the chip: with few groups per workgroup every per-pass launch fills and
        // drains its pipelines for two or three items and ends unbalanced (1e8 residues: +13-15 %); with dozens of groups per
        // workgroup the per-pass launches with per-query shapes are 2-3 % ahead (c5 at 10 %: 8 470 vs 8 250 GCUPS)
        if (c->opt_resident < 0 && !streaming && c->groups.size() >= (size_t)16 * most_wg) c->batch_now = false;
    }
    if (dbg) fprintf(stderr, "swimm_hip: batch decided %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    // No batch, eight or more short queries: those that fit one pass run whole -- every group through the pipeline kernel,
    // no tail kernel -- on three streams in rotation (issue()); a long sequence's serial chain, which bounds a lone short
    // query, is then covered by the neighbours' work.
    if (!c->batch_now && rotate)
        for (uint32_t q = 0; q < qn; ++q)
            if (qm[q] <= 64 * kLaneRows) rotated[q] = choose_plan(c, main_mode, qm[q], false, true, &qps[q]) == 0;   // fails when no one-pass shape exists
    std::vector<BulkCols> rbulk;
    if (streaming && !c->batch_now && ranges.size() > 1) {
        rqps.assign(ranges.size(), std::vector<QueryPlan>(qn));
        rbulk.resize(ranges.size());
        for (size_t ri = 0; ri < ranges.size(); ++ri) bulk_cols_of(c, ranges[ri], rbulk[ri]);
    }
    for (uint32_t q = 0; q < qn; ++q) {
        if (!c->batch_now && !rotated[q] && choose_plan(c, main_mode, qm[q], lane_room, false, &qps[q])) return 1;   // (a batch's shapes are chosen above)
        if (dbg)
            fprintf(stderr, "swimm_hip: query %u m=%u -> T=%d W=%d passes=%d (lane_room=%d)\n", q, qm[q], qps[q].T, qps[q].W, qps[q].passes, (int)lane_room);
        // A database that is streaming in, per-pass launches: every range gets the launch shape that suits ITS groups -- the
        // range with the longest sequences is small beside the chip (a few long chains over hundreds of workgroups), and
        // fewer, taller workgroups finish it sooner than the shape that is best for the database as a whole.
        if (!rqps.empty())
            for (size_t ri = 0; ri < ranges.size(); ++ri) {
                if (choose_plan(c, main_mode, qm[q], lane_room, false, &rqps[ri][q], &ranges[ri], &rbulk[ri])) return 1;
                qps[q].mpad = std::max(qps[q].mpad, rqps[ri][q].mpad);
                if (dbg) fprintf(stderr, "swimm_hip:   range %zu: T=%d W=%d passes=%d\n", ri, rqps[ri][q].T, rqps[ri][q].W, rqps[ri][q].passes);
            }
        const uint32_t lane_rows = (uint32_t)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows) * (64 * kLaneRows));
        qps[q].mpad = std::max(qps[q].mpad, lane_rows);
        qps[q].prof_off = prof_elems;
        prof_elems += (size_t)kCodes * qps[q].mpad;
    }
    // Two or more multi-pass queries: their passes alternate between the two bulk streams (each with a boundary buffer
    // of its own), so that the end of every launch -- the last workgroups finishing alone -- is covered by a kernel of
    // the other query.  (Within ONE such query the even/odd split of run_passes does the same.)
    uint32_t n_multi = 0;
    for (uint32_t q = 0; q < qn; ++q) n_multi += !rotated[q] && qps[q].passes > 1;
    alternate = n_multi >= 2 && c->opt_alternate && !streaming;
    return 0;
}

int SearchRun::upload_profiles()
{
    std::vector<int16_t> prof(prof_elems, 0);
    for (uint32_t q = 0; q < qn; ++q) {
        const int8_t *qa = c->qcodes.data() + qdisp[q];
        for (int d = 0; d < kCodes; ++d) {
            int16_t *row = prof.data() + qps[q].prof_off + (size_t)d * qps[q].mpad;
            for (uint32_t r = 0; r < qm[q]; ++r) row[r] = c->submat[(int)qa[r] * 32 + d];
        }
    }
    c->last_plans.resize(c->qm.size());
    for (uint32_t q = 0; q < qn; ++q) { qps[q].mode = main_mode; qps[q].dynamic = c->opt_dynamic != 0; qps[q].resident = c->batch_now && !rotated[q]; c->last_plans[qb + q] = qps[q]; }
    for (auto &rv : rqps)
        for (uint32_t q = 0; q < qn; ++q) {
            rv[q].mpad = qps[q].mpad; rv[q].prof_off = qps[q].prof_off;      // one profile per query, padded for the tallest plan
            rv[q].mode = main_mode; rv[q].dynamic = qps[q].dynamic; rv[q].resident = false;
        }
    if (dbg) fprintf(stderr, "swimm_hip: launch shapes chosen and profiles built %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    HIP_TRY(c->d_prof.reserve(prof_elems));
    HIP_TRY(hipMemcpyAsync(c->d_prof.p, prof.data(), prof_elems * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c->d_scores.reserve((size_t)qn * S));
    HIP_TRY(hipMemsetAsync(c->d_scores.p, 0, (size_t)qn * S * sizeof(int32_t), c->stream));

    return 0;
}

int SearchRun::size_buffers()
{

    if (dbg) fprintf(stderr, "swimm_hip: profile copy and score reset issued %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    tail_lanes = 1;
    // buffers that later launches grow are sized up front: a reallocation in the middle of the
    // multi-stream phase would free memory a kernel in flight still uses
    {
        uint64_t need_bnd = 0;
        size_t tail_cols = 0, tail_items = 0, launch_total = 16;
        int max_passes = 1;
        for (uint32_t q = 0; q < qn; ++q) max_passes = std::max(max_passes, (int)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows)));
        if (streaming) {
            // a range's work lists are built when its turn comes (the GPU is busy with the range before it by then):
            // size the shared buffers from the geometry alone
            const uint64_t budget = bnd_budget_cols(c);
            for (size_t ri = 0; ri < ranges.size(); ++ri) {
                const std::vector<uint8_t> is_tail = main_mode != Mode::I32 ? pick_tail(c, ranges[ri]) : std::vector<uint8_t>(ranges[ri].g1 - ranges[ri].g0, 0);
                size_t t_items = 0, t_cols = 0;
                uint32_t longest_main = 0, longest_all = 0;
                for (uint32_t g = ranges[ri].g0; g < ranges[ri].g1; ++g) {
                    longest_all = std::max(longest_all, c->groups[g].ncols);
                    if (is_tail[g - ranges[ri].g0]) { t_items += 64; t_cols += (size_t)64 * c->groups[g].ncols; }
                    else longest_main = std::max(longest_main, c->groups[g].ncols);
                }
                tail_items = std::max(tail_items, t_items);
                tail_cols = std::max(tail_cols, t_cols);
                for (uint32_t q = 0; q < qn; ++q) {
                    const QueryPlan &qp = qp_of(ri, q);
                    if (qp.passes <= 1) { launch_total += 2; continue; }
                    int per_cu = 1;
                    if (wgs_per_cu(c, main_mode, qp.T, qp.W, c->batch_now && !rotated[q], &per_cu)) return 1;
                    const uint64_t cols = ranges[ri].cols * (main_mode == Mode::I32 ? 2 : 1);
                    if (qps[q].resident) need_bnd = std::max<uint64_t>(need_bnd, (uint64_t)n_workgroups(c, per_cu) * longest_all * 64);   // (a batch takes every group)
                    else need_bnd = std::max<uint64_t>(need_bnd, std::min<uint64_t>(cols, std::max<uint64_t>(budget, longest_main)) * 64);
                    launch_total += (size_t)qp.passes * (size_t)(cols / std::max<uint64_t>(budget, 1) + 2);
                }
            }
        } else
        for (size_t ri = 0; ri < ranges.size(); ++ri)
            for (uint32_t q = 0; q < qn; ++q) {
                DbPlan *dp = nullptr;
                if (plan_of(ri, q, &dp)) return 1;
                size_t nsegs = 1;
                if (qps[q].resident && dp->have_main) {
                    need_bnd = std::max<uint64_t>(need_bnd, resident_bnd_elems(dp->main));
                } else if (qps[q].passes > 1 && dp->have_main) {
                    uint64_t cols = dp->main.bnd_cols;
                    if (c->opt_dynamic) {
                        std::vector<std::pair<uint32_t, uint32_t>> segs;
                        boundary_segments(c, dp->main, segs, &cols);
                        nsegs = segs.size();
                    }
                    need_bnd = std::max<uint64_t>(need_bnd, cols * 64);
                }
                launch_total += (size_t)qps[q].passes * std::max<size_t>(nsegs, 2);   // two kernels per pass when the list is split over two streams
                tail_cols = std::max<size_t>(tail_cols, dp->tail.cols);
                tail_items = std::max<size_t>(tail_items, dp->tail.n);
            }
        if (dbg) fprintf(stderr, "swimm_hip: buffer sizes known %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
        HIP_TRY(c->d_bnd.reserve(need_bnd));
        if (alternate || c->batch_now || streaming) HIP_TRY(c->d_bnd_b.reserve(need_bnd));
        if (c->batch_now && streaming) HIP_TRY(c->d_bnd_c.reserve(need_bnd));
        HIP_TRY(c->d_queue.reserve(launch_total));           // one zeroed queue cursor per pipeline launch of this search
        HIP_TRY(hipMemsetAsync(c->d_queue.p, 0, launch_total * sizeof(uint32_t), c->stream));
        c->queue_next = 0;
        if (reserve_lane_scratch(c->tail_scratch, tail_cols, tail_items, max_passes)) return 1;
        // How many queries' tail launches run side by side.  One, normally: the chains are a small part of the search
        // and a second launch only takes registers from the bulk kernels (c3: -5 %).  But each query costs at least the
        // longest sequence's chain (0.63 us per column with 8 rows per lane, 22 ms for 35 000 residues), whatever the
        // size of the database: when those chains add up to more than the bulk work, up to three run at a time.
        if (tail_items > 0 && !c->batch_now && !many_short) {
            double chains = 0, rows = 0;
            for (uint32_t q = 0; q < qn; ++q) {
                const int lane_passes = (int)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows));
                const int lr = (lane_passes == 1 && qm[q] <= 128 && c->opt_lane_rows) ? 2 : (lane_passes == 1 && qm[q] <= 256 && c->opt_lane_rows) ? 4 : kLaneRows;
                chains += (double)longest_cols * 0.63e-6 * lr / kLaneRows * (lane_passes > 1 ? 1.1 : 1.0);
                rows += qm[q];
            }
            const double bulk = rows * (double)c->total_cols * kGroupSeqs / 8000e9;
            if (chains > 0.6 * bulk) tail_lanes = (int)std::min(3.0, std::ceil(chains / std::max(0.6 * bulk, 1e-6)));
            if (dbg) fprintf(stderr, "swimm_hip: tail chains %.1f ms against %.1f ms of bulk work: %d tail launches at a time\n", chains * 1e3, bulk * 1e3, tail_lanes);
        }
        for (int i = 0; i + 1 < tail_lanes; ++i) {
            if (!c->stream_t[i]) {
                HIP_TRY(hipStreamCreate(&c->stream_t[i]));
                HIP_TRY(hipEventCreateWithFlags(&c->ev_tail_t[i], hipEventDisableTiming));
            }
            if (reserve_lane_scratch(c->tail_scratch_t[i], tail_cols, tail_items, max_passes)) return 1;
        }
        if (reserve_lane_scratch(c->tail_scratch_a, 0, tail_items, 1) || reserve_lane_scratch(c->tail_scratch_b, 0, tail_items, 1)) return 1;
        if (reserve_lane_scratch(c->rerun_scratch, (size_t)1 << 22, 4096, max_passes)) return 1;
        HIP_TRY(c->d_satlist.reserve((size_t)std::min<uint64_t>(S, 0xFFFFFFFEull) + 1));   // every slot could leave a tier's range
        HIP_TRY(c->d_rerun_items.reserve(4096));
    }
    return 0;
}

int SearchRun::issue()
{
    t_sized = now_s();
    HIP_TRY(c->d_err.reserve(1));
    HIP_TRY(hipMemsetAsync(c->d_err.p, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    HIP_TRY(hipEventRecord(c->ev_ready, c->stream));          // profiles uploaded, scores zeroed
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_ready, 0));
    for (int i = 0; i + 1 < tail_lanes; ++i) HIP_TRY(hipStreamWaitEvent(c->stream_t[i], c->ev_ready, 0));
    HIP_TRY(hipStreamWaitEvent(c->stream_b, c->ev_ready, 0));
    uint32_t one_pass_seen = 0, multi_seen = 0, tail_seen = 0;
    HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_ready, 0));
    while (c->ev_query.size() < 2 * (size_t)qn) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_query.push_back(e);
    }
    std::map<std::pair<int, int>, std::vector<uint32_t>> batches;     // launch shape (T, W) -> queries whose bulk part runs group-resident
    // the query table of the group-resident launches: per launch shape, the queries in ascending length (the kernel takes
    // index nq - 1, the longest, first); the same for every range, so it travels once
    std::map<std::pair<int, int>, size_t> qdesc_off;
    if (c->batch_now) {
        std::map<std::pair<int, int>, std::vector<uint32_t>> by_shape;
        for (uint32_t q = 0; q < qn; ++q)
            if (qps[q].resident) by_shape[std::make_pair(qps[q].T, qps[q].W)].push_back(q);
        std::vector<QDesc> qd_host;
        for (auto &kv : by_shape) {
            qdesc_off[kv.first] = qd_host.size();
            for (uint32_t q : kv.second) {
                if ((uint64_t)q * S > 0xFFFFFFFFull || qps[q].prof_off > 0xFFFFFFFFull) return fail("group-resident batch: score rows beyond 2^32 elements (lower score_mib)");
                qd_host.push_back(QDesc{(uint32_t)qps[q].prof_off, qps[q].mpad, (uint32_t)qps[q].passes, (uint32_t)((uint64_t)q * S)});
            }
        }
        HIP_TRY(c->d_qdesc.reserve(qd_host.size()));
        if (list_copy(c, c->d_qdesc.p, qd_host.data(), qd_host.size() * sizeof(QDesc)) || list_sync(c)) return 1;
    }
    if (streaming)          // the first range's work lists need its geometry only: ready before its bytes are
        for (uint32_t q = 0; q < qn; ++q) { DbPlan *dp = nullptr; if (plan_of(0, q, &dp)) return 1; }
    for (size_t ri = 0; ri < ranges.size(); ++ri) {
        if (streaming) {
            if (wait_uploaded(range_chunks[ri].second)) return 1;
            ChunkRec &last = c->chunks[up_order[range_chunks[ri].second - 1]];   // the upload stream is in order: its last chunk's event covers the range
            if (dbg) fprintf(stderr, "swimm_hip: range %zu (%llu columns): host copies done %.3f ms after the call began\n", ri, (unsigned long long)ranges[ri].cols, (now_s() - t_begin) * 1e3);
            HIP_TRY(hipStreamWaitEvent(c->stream, last.ready, 0));
            HIP_TRY(hipStreamWaitEvent(c->stream_b, last.ready, 0));
            HIP_TRY(hipStreamWaitEvent(c->stream2, last.ready, 0));
            for (int i = 0; i + 1 < tail_lanes; ++i) HIP_TRY(hipStreamWaitEvent(c->stream_t[i], last.ready, 0));
        }
        // Longest query first: its promotion re-runs (a handful of long serial chains on stream 3) then overlap
        // the bulk kernels of the shorter queries instead of running alone at the end.
        for (uint32_t k = 0; k < qn; ++k) {
            const uint32_t q = qn - 1 - k;                 // queries arrive sorted by ascending length
            DbPlan *dp = nullptr;
            if (plan_of(ri, q, &dp)) return 1;
            int32_t *row = c->d_scores.p + (size_t)q * S;
            if (dbg)
                fprintf(stderr, "swimm_hip: range %zu query %u: %d workgroups, %u tail items, main %s\n", ri, q, dp->main.n_wg, dp->tail.n, dp->have_main ? "yes" : "no");
            // The long-sequence tail (a few long serial chains, one wave each) runs beside the bulk kernel: 3 bulk waves
            // (144 VGPRs) + 1 lane wave (80) fill a SIMD's 512 registers exactly.  One tail launch at a time: several were
            // measured 5 % slower on c3, and the chained passes of concurrent launches could wait for each other's workgroups.
            //
            // A batch of short one-pass queries is different (see the plans above): each of them is bound from below by
            // the longest sequence's serial chain (2.4 ms for 5 000 residues, longer than the query's whole bulk work on
            // a database of 1e8 residues), wherever that sequence is aligned.  They run whole on one of three streams in
            // rotation, so that three are in flight and each one's chain is covered by the others' work: 300 queries of
            // 100 residues against 1e8: 3 480 -> 4 750 GCUPS, of 40 residues: 1 570 -> 3 000.  (Three streams, because HIP
            // multiplexes streams onto four hardware queues and the fourth carries the promotion re-runs; with seven
            // streams a tail kernel landed in the bulk stream's queue and held it back: -18 % on c3.)
            hipStream_t tail_stream = c->stream2, bulk_stream = c->stream;
            LaneScratch *tail_scratch = &c->tail_scratch;
            // (with an extreme sequence in the database the short queries keep their tail kernel, but still take turns on
            // the three streams: three 17 ms chains at a time instead of one)
            if (rotated[q] || (many_short && qps[q].passes == 1 && qm[q] <= 64 * kLaneRows)) {
                switch (one_pass_seen++ % 3) {
                case 0: tail_stream = bulk_stream = c->stream; tail_scratch = &c->tail_scratch_a; break;
                case 1: tail_stream = bulk_stream = c->stream_b; tail_scratch = &c->tail_scratch_b; break;
                default: tail_stream = bulk_stream = c->stream2; break;
                }
            }
            if (tail_lanes > 1 && dp->tail.n > 0 && tail_stream == c->stream2) {      // chain-bound search: the tail launches take turns on up to three streams
                const uint32_t ti = tail_seen++ % (uint32_t)tail_lanes;
                if (ti > 0) { tail_stream = c->stream_t[ti - 1]; tail_scratch = &c->tail_scratch_t[ti - 1]; }
            }
            DevBuf<uint2> *bnd = &c->d_bnd;
            // (the one-pass queries of such a batch take their turn as well: they are the shortest, the batch ends with them,
            // and a launch that runs alone ends with a few workgroups holding the chip -- c3's last eight queries: 26 ms at 5 900 GCUPS)
            if (alternate && !rotated[q] && !qps[q].resident && !(many_short && qps[q].passes == 1 && qm[q] <= 64 * kLaneRows) && (multi_seen++ & 1)) { bulk_stream = c->stream_b; bnd = &c->d_bnd_b; }
            if (streaming && (ri & 1)) { bulk_stream = c->stream_b; bnd = &c->d_bnd_b; }     // consecutive ranges overlap
            // (the tail's first tier is the bulk's: binary16 pairs, unless that tier is switched off; the ladder below re-runs
            // what reaches 2048 in int16 -- rare, the chains are long but the scores are not)
            if (!qps[q].resident && run_lane_passes(c, main_mode == Mode::F16 ? Mode::F16 : Mode::PK16, qps[q], qm[q], dp->tail, row, tail_stream, *tail_scratch)) return 1;
            if (qps[q].resident) {      // every group goes into the group-resident launch of its shape, below
                if (dp->have_main) batches[std::make_pair(qps[q].T, qps[q].W)].push_back(q);
                else HIP_TRY(hipEventRecord(c->ev_query[2 * q], bulk_stream));
                HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], tail_stream));
                continue;
            }
            if (dp->have_main && run_passes(c, main_mode, qp_of(ri, q), dp->main, row, bulk_stream, !streaming && !alternate, *bnd)) return 1;
            if (ri + 1 == ranges.size()) {
                HIP_TRY(hipEventRecord(c->ev_query[2 * q], bulk_stream));
                HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], tail_stream));
            }
        }
        // Group-resident launches: ONE launch per launch shape for all the batch's queries of that shape -- the items are
        // (group, query) pairs, so even a small database gives every workgroup hundreds of them, the pipelines fill and
        // drain once per batch, and no pass waits for the slowest workgroup of the one before it.  Shapes alternate
        // between the two bulk streams.
        if (!batches.empty()) {
            uint32_t bi = 0;
            for (auto &kv : batches) {
                std::sort(kv.second.begin(), kv.second.end());          // (the order of the query table)
                const size_t off = qdesc_off[kv.first];
                const int T = kv.first.first, W = kv.first.second;
                const uint32_t nqb = (uint32_t)kv.second.size();
                DbPlan *dp = nullptr;
                if (plan_of(ri, kv.second[0], &dp)) return 1;
                uint64_t pass_sum = 0;
                uint32_t max_p = 1;
                for (uint32_t q : kv.second) { pass_sum += qps[q].passes; max_p = std::max<uint32_t>(max_p, qps[q].passes); }
                // (a database that streams in: three ranges in flight, each on a stream and a boundary scratch of its own --
                // no tail kernels beside group-resident launches, so the tail stream serves as the third)
                const uint32_t si = streaming ? (uint32_t)((bi + ri) % 3) : (bi & 1);
                hipStream_t st = si == 0 ? c->stream : si == 1 ? c->stream_b : c->stream2;
                DevBuf<uint2> &bnd = si == 0 ? c->d_bnd : si == 1 ? c->d_bnd_b : c->d_bnd_c;
                if (run_resident_batch(c, main_mode, T, W, dp->main, c->d_qdesc.p + off, nqb, pass_sum, max_p, st, bnd)) return 1;
                for (uint32_t q : kv.second) HIP_TRY(hipEventRecord(c->ev_query[2 * q], st));
                ++bi;
            }
            batches.clear();
        }
        // the next range's work lists need its geometry only: build them now, while the GPU aligns this range and
        // before the host blocks in the next range's copies
        if (dbg && streaming) fprintf(stderr, "swimm_hip: range %zu: launches issued %.3f ms after the call began\n", ri, (now_s() - t_begin) * 1e3);
        if (streaming && ri + 1 < ranges.size())
            for (uint32_t q = 0; q < qn; ++q) { DbPlan *dp = nullptr; if (plan_of(ri + 1, q, &dp)) return 1; }
        if (dbg && streaming) fprintf(stderr, "swimm_hip: range %zu: next range's work lists built %.3f ms after the call began\n", ri, (now_s() - t_begin) * 1e3);
    }
    if (streaming) {
        // the ranges alternated between the two bulk streams: "query q's bulk kernels are done" = both have drained
        HIP_TRY(hipEventRecord(c->ev_b, c->stream_b));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_b, 0));
        HIP_TRY(hipEventRecord(c->ev_a, c->stream2));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_a, 0));
        for (uint32_t q = 0; q < qn; ++q) HIP_TRY(hipEventRecord(c->ev_query[2 * q], c->stream));
        c->up->finish(false);
        if (sync_lengths(c)) return 1;             // the promotion re-runs stop every alignment at its true length
    }
    return 0;
}

int SearchRun::promotion_ladder()
{
    t_issued = now_s();
    // promotion ladder (the reference's int8 -> int16 -> int32, CPUsearch.c:678-957, one rung higher):
    // f16 results >= 2048 are re-run as packed int16 pairs, int16 results >= 32767 as int32 sequences; each
    // re-run is a lane-systolic item (one wave per alignment), issued on stream 3 as soon as the query's own
    // kernels are done
    if (main_mode != Mode::I32) {
        const uint32_t cap = (uint32_t)std::min<uint64_t>(S, 0xFFFFFFFEull);   // list capacity = all slots: no query can overflow it
        std::vector<uint32_t> list;
        auto collect = [&](uint32_t q, int thr, std::vector<uint32_t> &out) -> int {
            int32_t *row = c->d_scores.p + (size_t)q * S;
            uint32_t *d_count = c->d_satlist.p + cap;
            HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), c->stream3));
            HIP_TRY(launch_collect_saturated(row, S, thr, c->d_satlist.p, d_count, cap, c->stream3));
            uint32_t count = 0;
            HIP_TRY(hipMemcpyAsync(&count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream3));
            HIP_TRY(hipStreamSynchronize(c->stream3));
            if (count > cap) return fail("more than %u alignments of query %u left the %s range: use force_i32", cap, q, thr == 2048 ? "f16" : "int16");
            out.resize(count);
            if (count) {
                HIP_TRY(hipMemcpyAsync(out.data(), c->d_satlist.p, count * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream3));
                HIP_TRY(hipStreamSynchronize(c->stream3));
            }
            return 0;
        };
        auto rerun = [&](uint32_t q, Mode mode, std::vector<LaneItem> &items) -> int {
            if (items.empty()) return 0;
            std::stable_sort(items.begin(), items.end(), [](const LaneItem &a, const LaneItem &b) { return a.ncols > b.ncols; });
            uint64_t cols = 0;
            for (LaneItem &it : items) { it.bnd_off = (uint32_t)cols; cols += it.ncols; }
            const int rpasses = (int)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows));
            if (items.size() > c->d_rerun_items.cap || (rpasses > 1 && cols + 64 > c->rerun_scratch.bnd[0].cap) ||
                items.size() * (size_t)rpasses > c->rerun_scratch.prog.cap) {
                // growing a buffer frees the old one, which waits for the whole device: rare (first big batch)
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(c->d_rerun_items.reserve(items.size() * 2));
                if (reserve_lane_scratch(c->rerun_scratch, cols * 2, items.size() * 2, rpasses)) return 1;
            }
            HIP_TRY(hipMemcpyAsync(c->d_rerun_items.p, items.data(), items.size() * sizeof(LaneItem), hipMemcpyHostToDevice, c->stream3));
            HIP_TRY(hipStreamSynchronize(c->stream3));       // `items` is a host temporary
            LaneList ll;
            ll.items.p = c->d_rerun_items.p; ll.items.cap = c->d_rerun_items.cap;
            ll.n = (uint32_t)items.size(); ll.cols = cols; ll.cell_cols = cols;
            const int rc = run_lane_passes(c, mode, qps[q], qm[q], ll, c->d_scores.p + (size_t)q * S, c->stream3, c->rerun_scratch);
            ll.items.p = nullptr; ll.items.cap = 0;           // borrowed
            return rc;
        };
        for (uint32_t k = 0; k < qn; ++k) {
            const uint32_t q = qn - 1 - k;
            const long bound = (long)qm[q] * c->max_pos;     // no alignment of this query can score more
            if (!((main_mode == Mode::F16 && bound >= 2048) || bound >= 32767)) continue;
            HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_query[2 * q], 0));
            HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_query[2 * q + 1], 0));
            if (main_mode == Mode::F16 && bound >= 2048) {
                if (collect(q, 2048, list)) return 1;
                std::vector<LaneItem> items;
                std::vector<uint8_t> seen;
                for (uint32_t slot : list) {                      // re-run the packed PAIR the slot belongs to
                    const uint32_t g = slot / kGroupSeqs, l = slot % 64;
                    const uint32_t pair = g * 64 + l;
                    if (seen.size() <= pair) seen.resize(pair + 1, 0);
                    if (seen[pair]) continue;
                    seen[pair] = 1;
                    const GroupDesc &gd = c->groups[g];
                    const uint32_t len = std::max(c->seq_len[gd.seq0 + l], c->seq_len[gd.seq0 + 64 + l]);
                    LaneItem li{};
                    li.db = gd.db; li.lane = l; li.half = 0; li.ncols = (len + kChunkCols - 1) / kChunkCols * kChunkCols;
                    li.slot_a = gd.seq0 + l; li.slot_b = gd.seq0 + 64 + l;
                    if (li.ncols) items.push_back(li);
                }
                c->promoted16 += list.size();
                if (rerun(q, Mode::PK16, items)) return 1;
            }
            if (bound < 32767) continue;                         // cannot saturate int16
            if (collect(q, 32767, list)) return 1;
            std::vector<LaneItem> items;
            for (uint32_t slot : list) {
                const uint32_t g = slot / kGroupSeqs, within = slot % kGroupSeqs;
                LaneItem li{};
                li.db = c->groups[g].db; li.lane = within % 64; li.half = within / 64;
                li.ncols = (c->seq_len[slot] + kChunkCols - 1) / kChunkCols * kChunkCols;
                li.slot_a = slot; li.slot_b = 0;
                if (li.ncols) items.push_back(li);
            }
            c->promoted += items.size();
            if (rerun(q, Mode::I32, items)) return 1;
        }
    }
    return 0;
}

int SearchRun::drain()
{
    HIP_TRY(hipEventRecord(c->ev_tail, c->stream2));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail, 0));
    for (int i = 0; i + 1 < tail_lanes; ++i) {
        HIP_TRY(hipEventRecord(c->ev_tail_t[i], c->stream_t[i]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail_t[i], 0));
    }
    HIP_TRY(hipEventRecord(c->ev_b, c->stream_b));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_b, 0));
    HIP_TRY(hipEventRecord(c->ev_tail3, c->stream3));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail3, 0));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->kernel_ms += ms;
    for (size_t i = 0; i + 1 < c->launch_ev_used; i += 2) {
        float lm = 0;
        HIP_TRY(hipEventElapsedTime(&lm, c->launch_ev[i], c->launch_ev[i + 1]));
        c->launch_ms_sum += lm;
        c->launch_ms_n++;
        if (dbg) {
            float at = 0;
            (void)hipEventElapsedTime(&at, c->ev0, c->launch_ev[i]);
            fprintf(stderr, "swimm_hip: pipeline launch %zu: starts %.3f ms after the search's first event, runs %.3f ms\n", i / 2, at, lm);
        }
    }
    c->launch_ev_used = 0;
    if (dbg)
        fprintf(stderr, "swimm_hip: queries %u..%u%s: plans + buffers %.3f s, launches issued %.3f s, ladder + drain %.3f s (device %.3f s)\n", qb, qe,
                streaming ? " (streaming upload)" : "", t_sized - t_begin, t_issued - t_sized, now_s() - t_issued, ms * 1e-3);
    uint32_t werr = 0;
    HIP_TRY(hipMemcpy(&werr, c->d_err.p, sizeof werr, hipMemcpyDeviceToHost));
    if (werr) return fail("pipeline watchdog expired (code %u): results discarded", werr);
    if (streaming) { release_plans(c); c->groups_dirty = true; }   // (the cached lists of the resident database are built on the next search)
    return 0;
}

int search_device(swimm_hip_ctx *c, uint32_t qb, uint32_t qe, uint64_t *slots_out)
{
    SearchRun run(c, qb, qe);
    return run.run(slots_out);
}

}  // namespace

extern "C" {

int swimm_hip_abi_version(void) { return SWIMM_HIP_ABI_VERSION; }

const char *swimm_hip_last_error(void) { return g_err.c_str(); }

// Test hook: SWIMM_HIP_VIRTUAL_GPUS=N presents N devices on a box with fewer; virtual device d runs on physical
// device d % real count.  Lets the multi-GPU host logic (sharding, one thread per device, merging) be exercised on
// the one-GPU test box; results are identical by construction, timing is meaningless.
static int virtual_gpus(int real)
{
    const char *v = getenv("SWIMM_HIP_VIRTUAL_GPUS");
    const int n = v ? atoi(v) : 0;
    return real > 0 && n > real ? n : real;
}

int swimm_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail("hipGetDeviceCount: %s", hipGetErrorString(e)); return 0; }
    if (n == 0) fail("no HIP device visible");
    return virtual_gpus(n);
}

int swimm_hip_create(int device, swimm_hip_ctx **out)
{
    if (!out) return fail("swimm_hip_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= virtual_gpus(n)) return fail("swimm_hip_create: device %d not in [0,%d)", device, virtual_gpus(n));
    device %= std::max(n, 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("swimm_hip_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    swimm_hip_ctx *c = new swimm_hip_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess || hipStreamCreate(&c->stream_b) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreate(&c->stream_up) != hipSuccess || hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreate(&c->stream3) != hipSuccess || hipEventCreateWithFlags(&c->ev_tail3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail("swimm_hip_create: stream/event creation failed");
    }
    // SWIMM_HIP_OPTIONS="key=value,key=value": the same knobs as swimm_hip_set_option, for callers that never see the
    // context (swimm_hip_search_chunks, the `swimm` program)
    if (const char *env = getenv("SWIMM_HIP_OPTIONS")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string kv = all.substr(pos, end - pos);
            pos = end + 1;
            if (kv.empty()) continue;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos || eq == 0 || eq + 1 >= kv.size()) { swimm_hip_destroy(c); return fail("SWIMM_HIP_OPTIONS: '%s' is not key=value", kv.c_str()); }
            char *rest = nullptr;
            const long v = strtol(kv.c_str() + eq + 1, &rest, 10);
            if (*rest != 0) { swimm_hip_destroy(c); return fail("SWIMM_HIP_OPTIONS: value of '%s' is not an integer", kv.c_str()); }
            if (swimm_hip_set_option(c, kv.substr(0, eq).c_str(), (int)v)) { const std::string msg = g_err; swimm_hip_destroy(c); return fail("SWIMM_HIP_OPTIONS: %s", msg.c_str()); }
        }
    }
    *out = c;
    return 0;
}

void swimm_hip_destroy(swimm_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    delete c->up; c->up = nullptr;
    swimm_hip_clear_db(c);
    c->d_scores.release(); c->d_prof.release(); c->d_bnd.release(); c->d_bnd_b.release(); c->d_bnd_c.release(); c->d_qdesc.release();
    if (c->pin) { (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_cap = c->pin_used = 0; }
    c->d_gbase.release(); c->d_gvalid.release(); c->d_keys.release(); c->d_err.release(); c->tail_scratch.release(); c->tail_scratch_a.release(); c->tail_scratch_b.release(); c->tail_scratch_t[0].release(); c->tail_scratch_t[1].release(); c->rerun_scratch.release(); c->d_rerun_items.release(); c->d_satlist.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
    if (c->ev_tail3) (void)hipEventDestroy(c->ev_tail3);
    for (hipEvent_t e : c->ev_query) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->launch_ev) (void)hipEventDestroy(e);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (int i = 0; i < 2; ++i) {
        if (c->stream_t[i]) (void)hipStreamDestroy(c->stream_t[i]);
        if (c->ev_tail_t[i]) (void)hipEventDestroy(c->ev_tail_t[i]);
    }
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    c->up_b.release(); c->up_n.release(); c->up_disp.release(); c->up_gcols.release(); c->up_off.release(); c->up_goff.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int swimm_hip_set_queries(swimm_hip_ctx *c, const char *a, const uint16_t *m, const uint32_t *a_disp,
                          uint32_t query_count, const char *submat, int open_gap, int extend_gap)
{
    if (!c || !a || !m || !a_disp || !submat) return fail("swimm_hip_set_queries: NULL argument");
    if (query_count == 0) return fail("swimm_hip_set_queries: no queries");
    if (open_gap < 0 || extend_gap < 0 || open_gap + extend_gap > 127)
        return fail("swimm_hip_set_queries: need 0 <= open, extend and open+extend <= 127 (got %d, %d)", open_gap, extend_gap);
    size_t total = 0;
    for (uint32_t q = 0; q < query_count; ++q) {
        if (m[q] == 0) return fail("swimm_hip_set_queries: query %u is empty", q);
        total = std::max<size_t>(total, (size_t)a_disp[q] + m[q]);
    }
    for (size_t i = 0; i < total; ++i)
        if ((unsigned char)a[i] > 23) return fail("swimm_hip_set_queries: residue code %d at %zu is outside 0..23", (int)a[i], i);
    c->qcodes.assign((const int8_t *)a, (const int8_t *)a + total);
    c->qm.assign(m, m + query_count);
    c->qdisp.assign(a_disp, a_disp + query_count);
    memcpy(c->submat, submat, SWIMM_HIP_SUBMAT_BYTES);
    c->open_gap = open_gap; c->extend_gap = extend_gap;
    c->max_pos = 0;
    for (int i = 0; i < SWIMM_HIP_SUBMAT_BYTES; ++i) c->max_pos = std::max<int>(c->max_pos, c->submat[i]);
    c->have_queries = true;
    return 0;
}

int swimm_hip_add_chunk(swimm_hip_ctx *c, const char *b, uint64_t vD, const uint16_t *n, const uint32_t *b_disp,
                        uint32_t group_count, uint32_t vl, uint64_t first_group)
{
    if (!c || !b || !n || !b_disp) return fail("swimm_hip_add_chunk: NULL argument");
    if (group_count == 0) return fail("swimm_hip_add_chunk: empty chunk");
    if (vl == 0 || vl > (uint32_t)kGroupSeqs || kGroupSeqs % vl != 0)
        return fail("swimm_hip_add_chunk: lane width %u must divide %d", vl, kGroupSeqs);
    if (vD > 0xFFFFFFFFull) return fail("swimm_hip_add_chunk: chunk larger than 4 GiB");
    for (uint32_t g = 0; g < group_count; ++g)
        if ((uint64_t)b_disp[g] + (uint64_t)n[g] * vl > vD)
            return fail("swimm_hip_add_chunk: group %u (disp %u, n %u) runs past vD=%llu", g, b_disp[g], n[g], (unsigned long long)vD);
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t per = kGroupSeqs / vl;
    ChunkRec rec;
    rec.kind = 0;
    rec.h_b = b; rec.vD = vD; rec.h_n = n; rec.h_disp = b_disp; rec.group_count = group_count; rec.vl = vl;
    rec.n_groups = (group_count + per - 1) / per;
    rec.goff.resize(rec.n_groups);
    rec.gcols.resize(rec.n_groups);
    for (uint32_t g = 0; g < rec.n_groups; ++g) {
        uint32_t mx = 0;
        for (uint32_t v = g * per; v < std::min(group_count, (g + 1) * per); ++v) mx = std::max<uint32_t>(mx, n[v]);
        mx = std::max<uint32_t>(mx, 1);
        rec.gcols[g] = (mx + kChunkCols - 1) / kChunkCols * kChunkCols;
    }
    rec.first_seq = first_group * vl;
    rec.n_seq = (uint64_t)group_count * vl;
    if (register_chunk(c, rec, {})) return 1;
    if (c->opt_lazy_upload ? ensure_uploader(c) : upload_chunk(c, c->chunks.back())) return 1;
    return 0;
}

int swimm_hip_add_sequences(swimm_hip_ctx *c, const uint16_t *lengths, const char *codes, uint64_t n_seq, uint64_t first_seq)
{
    if (!c || !lengths || !codes) return fail("swimm_hip_add_sequences: NULL argument");
    if (n_seq == 0) return fail("swimm_hip_add_sequences: empty slab");
    if (n_seq > 0x7FFFFFFFull) return fail("swimm_hip_add_sequences: more than 2^31 sequences in one slab");
    ChunkRec rec;
    rec.kind = 1;
    rec.off.resize(n_seq + 1);
    uint64_t total = 0;
    for (uint64_t i = 0; i < n_seq; ++i) { rec.off[i] = (uint32_t)total; total += lengths[i]; if (total > 0xFFFFFFF0ull) return fail("swimm_hip_add_sequences: slab larger than 4 GiB"); }
    rec.off[n_seq] = (uint32_t)total;
    HIP_TRY(hipSetDevice(c->device));
    rec.h_codes = codes; rec.code_bytes = total;
    rec.n_groups = (uint32_t)((n_seq + kGroupSeqs - 1) / kGroupSeqs);
    rec.goff.resize(rec.n_groups);
    rec.gcols.resize(rec.n_groups);
    for (uint32_t g = 0; g < rec.n_groups; ++g) {
        uint32_t mx = 1;
        const uint64_t e = std::min<uint64_t>(n_seq, (uint64_t)(g + 1) * kGroupSeqs);
        for (uint64_t i = (uint64_t)g * kGroupSeqs; i < e; ++i) mx = std::max<uint32_t>(mx, lengths[i]);
        rec.gcols[g] = (mx + kChunkCols - 1) / kChunkCols * kChunkCols;
    }
    rec.first_seq = first_seq;
    rec.n_seq = n_seq;
    std::vector<uint32_t> lens(lengths, lengths + n_seq);
    if (register_chunk(c, rec, lens)) return 1;
    if (c->opt_lazy_upload ? ensure_uploader(c) : upload_chunk(c, c->chunks.back())) return 1;
    return 0;
}

int swimm_hip_clear_db(swimm_hip_ctx *c)
{
    if (!c) return fail("swimm_hip_clear_db: NULL ctx");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                 // nothing in flight may still read the chunks
    for (auto &ch : c->chunks) { (void)hipFree(ch.d_tiled); (void)hipFree(ch.d_len); if (ch.ready) (void)hipEventDestroy(ch.ready); }
    c->chunks.clear(); c->groups.clear(); c->group_col_off.clear(); c->seq_len.clear();
    c->total_cols = 0;
    release_plans(c);
    c->groups_dirty = true;
    return 0;
}

// queries per batch: the score rows of a batch (4 B per query and database slot) stay within the budget
static uint32_t query_batch(const swimm_hip_ctx *c)
{
    const uint64_t S = (uint64_t)c->groups.size() * kGroupSeqs;
    const uint64_t budget = (uint64_t)c->opt_score_mib << 20;
    const uint64_t per_query = std::max<uint64_t>(1, S * sizeof(int32_t));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->qm.size(), budget / per_query));
}

static void reset_stats(swimm_hip_ctx *c) { c->kernel_ms = 0; c->cells = 0; c->promoted = 0; c->promoted16 = 0; c->launches = 0; c->launch_ms_sum = 0; c->launch_ms_n = 0; c->launch_ev_used = 0; }

int swimm_hip_search(swimm_hip_ctx *c, int32_t *scores, uint64_t score_stride, double *work_time)
{
    if (!c || !scores) return fail("swimm_hip_search: NULL argument");
    const double t0 = now_s();
    reset_stats(c);
    const uint32_t qtotal = (uint32_t)c->qm.size(), B = query_batch(c);
    if (qtotal == 0) return fail("swimm_hip_search: no queries set");
    for (uint32_t qb = 0; qb < qtotal; qb += B) {
        const uint32_t qe = std::min(qtotal, qb + B), qn = qe - qb;
        uint64_t S = 0;
        if (search_device(c, qb, qe, &S)) return 1;
        // score scatter (X3, MICsearch.c:333-334): each chunk's slice goes to its global offset
        for (const ChunkRec &ch : c->chunks) {
            if (ch.first_seq + ch.n_seq > score_stride)
                return fail("swimm_hip_search: chunk at %llu+%llu exceeds score_stride %llu", (unsigned long long)ch.first_seq,
                            (unsigned long long)ch.n_seq, (unsigned long long)score_stride);
            HIP_TRY(hipMemcpy2DAsync(scores + (size_t)qb * score_stride + ch.first_seq, score_stride * sizeof(int32_t),
                                     c->d_scores.p + (size_t)ch.group0 * kGroupSeqs, S * sizeof(int32_t),
                                     ch.n_seq * sizeof(int32_t), qn, hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (work_time) *work_time = now_s() - t0;
    return 0;
}

int swimm_hip_search_topr(swimm_hip_ctx *c, uint32_t r, uint64_t n_valid, int32_t *top_scores, int64_t *top_index,
                          double *work_time)
{
    if (!c || !top_scores || !top_index) return fail("swimm_hip_search_topr: NULL argument");
    if (r == 0) return fail("swimm_hip_search_topr: r must be > 0");
    const double t0 = now_s();
    reset_stats(c);
    const uint32_t qtotal = (uint32_t)c->qm.size(), B = query_batch(c);
    if (qtotal == 0) return fail("swimm_hip_search_topr: no queries set");
    typedef std::pair<int32_t, int64_t> Hit;   // larger pair first == score desc, then larger index first (utils.c:12,52)
    for (uint32_t qb = 0; qb < qtotal; qb += B) {
        const uint32_t qe = std::min(qtotal, qb + B), qn = qe - qb;
        uint64_t S = 0;
        if (search_device(c, qb, qe, &S)) return 1;
        int32_t *out_s = top_scores + (size_t)qb * r;
        int64_t *out_i = top_index + (size_t)qb * r;
        // the device keys carry the global index in 32 bits (score << 32 | index, + 1): a database part whose indices do not
        // fit takes the host selection below instead
        uint64_t max_index = 0;
        for (const ChunkRec &ch : c->chunks) max_index = std::max<uint64_t>(max_index, ch.first_seq + ch.n_seq + kGroupSeqs);
        if (r <= 64 && max_index < 0xFFFFFFFEull) {
            // device path: per-block top-64 candidate keys, final selection over n_blocks*64 keys on the host
            std::vector<int64_t> gbase(c->groups.size());
            std::vector<uint32_t> gvalid(c->groups.size());
            for (const ChunkRec &ch : c->chunks)
                for (uint32_t i = 0; i < ch.n_groups; ++i) {
                    const uint64_t first = ch.first_seq + (uint64_t)i * kGroupSeqs;
                    uint64_t cnt = (uint64_t)i * kGroupSeqs < ch.n_seq ? std::min<uint64_t>(kGroupSeqs, ch.n_seq - (uint64_t)i * kGroupSeqs) : 0;
                    if (first >= n_valid) cnt = 0; else cnt = std::min<uint64_t>(cnt, n_valid - first);
                    gbase[ch.group0 + i] = (int64_t)first;
                    gvalid[ch.group0 + i] = (uint32_t)cnt;
                }
            const int n_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(256, S / 1024));
            HIP_TRY(c->d_gbase.reserve(gbase.size()));
            HIP_TRY(c->d_gvalid.reserve(gvalid.size()));
            HIP_TRY(c->d_keys.reserve((size_t)qn * n_blocks * 64));
            HIP_TRY(hipMemcpyAsync(c->d_gbase.p, gbase.data(), gbase.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gvalid.p, gvalid.data(), gvalid.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            for (uint32_t q = 0; q < qn; ++q)
                HIP_TRY(launch_topk64(c->d_scores.p + (size_t)q * S, S, c->d_gbase.p, c->d_gvalid.p,
                                      c->d_keys.p + (size_t)q * n_blocks * 64, n_blocks, c->stream));
            std::vector<unsigned long long> keys((size_t)qn * n_blocks * 64);
            HIP_TRY(hipMemcpyAsync(keys.data(), c->d_keys.p, keys.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (uint32_t q = 0; q < qn; ++q) {
                unsigned long long *kq = keys.data() + (size_t)q * n_blocks * 64;
                const size_t nk = (size_t)n_blocks * 64;
                const size_t k = std::min<size_t>(r, nk);
                std::partial_sort(kq, kq + k, kq + nk, std::greater<unsigned long long>());
                for (uint32_t i = 0; i < r; ++i) {
                    const bool have = i < k && kq[i] != 0;
                    const unsigned long long key = have ? kq[i] - 1 : 0;
                    out_s[(size_t)q * r + i] = have ? (int32_t)(key >> 32) : -1;
                    out_i[(size_t)q * r + i] = have ? (int64_t)(key & 0xFFFFFFFFull) : -1;
                }
            }
            continue;
        }
        // r > 64: whole score rows come back and the host selects (O(N log r))
        std::vector<int32_t> host((size_t)qn * S);
        HIP_TRY(hipMemcpyAsync(host.data(), c->d_scores.p, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<Hit> hits;
        for (uint32_t q = 0; q < qn; ++q) {
            hits.clear();
            for (const ChunkRec &ch : c->chunks) {
                const int32_t *row = host.data() + (size_t)q * S + (size_t)ch.group0 * kGroupSeqs;
                for (uint64_t i = 0; i < ch.n_seq; ++i) {
                    const uint64_t gi = ch.first_seq + i;
                    if (gi < n_valid) hits.push_back(Hit(row[i], (int64_t)gi));
                }
            }
            const size_t k = std::min<size_t>(r, hits.size());
            std::partial_sort(hits.begin(), hits.begin() + k, hits.end(), std::greater<Hit>());
            for (uint32_t i = 0; i < r; ++i) {
                out_s[(size_t)q * r + i] = i < k ? hits[i].first : -1;
                out_i[(size_t)q * r + i] = i < k ? hits[i].second : -1;
            }
        }
    }
    if (work_time) *work_time = now_s() - t0;
    return 0;
}

int swimm_hip_last_stats(swimm_hip_ctx *c, double *kernel_ms, uint64_t *cells, uint64_t *promoted, uint32_t *launches)
{
    if (!c) return fail("swimm_hip_last_stats: NULL ctx");
    if (kernel_ms) *kernel_ms = c->kernel_ms;
    if (cells) *cells = c->cells;
    if (promoted) *promoted = c->promoted;
    if (launches) *launches = c->launches;
    return 0;
}

int swimm_hip_last_plan(swimm_hip_ctx *c, uint32_t q, int *rows_per_wave, int *waves, int *passes)
{
    if (!c) return fail("swimm_hip_last_plan: NULL ctx");
    if (q >= c->last_plans.size()) return fail("swimm_hip_last_plan: query %u was not part of the last search", q);
    if (rows_per_wave) *rows_per_wave = c->last_plans[q].T;
    if (waves) *waves = c->last_plans[q].W;
    if (passes) *passes = c->last_plans[q].passes;
    return 0;
}

int swimm_hip_last_launch_ms(swimm_hip_ctx *c, double *sum_ms, uint32_t *launches)
{
    if (!c) return fail("swimm_hip_last_launch_ms: NULL ctx");
    if (!c->opt_time_launches) return fail("swimm_hip_last_launch_ms: set the option \"time_launches\" before the search");
    if (sum_ms) *sum_ms = c->launch_ms_sum;
    if (launches) *launches = c->launch_ms_n;
    return 0;
}

int swimm_hip_last_kernel_name(swimm_hip_ctx *c, uint32_t q, char *buf, size_t buf_len)
{
    if (!c || !buf || buf_len == 0) return fail("swimm_hip_last_kernel_name: NULL argument");
    if (q >= c->last_plans.size()) return fail("swimm_hip_last_kernel_name: query %u was not part of the last search", q);
    const QueryPlan &qp = c->last_plans[q];
    const char *sym = pipe_kernel_symbol(qp.mode, qp.T, qp.dynamic, qp.resident);
    if (!sym) return fail("swimm_hip_last_kernel_name: no kernel for rows_per_wave=%d", qp.T);
    int status = 0;
    char *dem = abi::__cxa_demangle(sym, nullptr, nullptr, &status);
    snprintf(buf, buf_len, "%s", status == 0 && dem ? dem : sym);
    free(dem);
    return 0;
}

int swimm_hip_set_option(swimm_hip_ctx *c, const char *key, int value)
{
    if (!c || !key) return fail("swimm_hip_set_option: NULL argument");
    if (!strcmp(key, "rows_per_wave")) {
        if (value != 0 && (value < 8 || value > 36 || value % 4)) return fail("rows_per_wave must be 0 (auto) or a multiple of 4 in 8..36");
        c->opt_T = value;
    } else if (!strcmp(key, "waves")) {
        if (value < 0 || value > kMaxWaves) return fail("waves must be 0 (auto) .. %d", kMaxWaves);
        c->opt_W = value;
    } else if (!strcmp(key, "max_waves")) {
        if (value < 0 || value > kMaxWaves) return fail("max_waves must be 0..%d", kMaxWaves);
        c->opt_maxW = value;
    } else if (!strcmp(key, "force_i32")) {
        c->opt_force_i32 = value != 0;
    } else if (!strcmp(key, "bnd_mib")) {
        if (value < 1) return fail("bnd_mib must be >= 1");
        c->opt_bnd_mib = value;
    } else if (!strcmp(key, "score_mib")) {
        if (value < 0) return fail("score_mib must be >= 0 (0 = one query per batch)");
        c->opt_score_mib = value;
    } else if (!strcmp(key, "tail_frac")) {
        if (value < 1 || value > 1000) return fail("tail_frac must be 1..1000 (percent of a CU's mean load)");
        c->opt_tail_frac = value;
        release_plans(c);
    } else if (!strcmp(key, "rotate")) {
        c->opt_rotate = value != 0;
    } else if (!strcmp(key, "lane_rows")) {
        c->opt_lane_rows = value != 0;
    } else if (!strcmp(key, "lane_room")) {
        if (value < -1 || value > 1) return fail("lane_room must be -1 (auto), 0 or 1");
        c->opt_lane_room = value;
    } else if (!strcmp(key, "resident")) {
        if (value < -1 || value > 1) return fail("resident must be -1 (auto), 0 or 1");
        c->opt_resident = value;
    } else if (!strcmp(key, "time_launches")) {
        c->opt_time_launches = value != 0;
    } else if (!strcmp(key, "alternate")) {
        c->opt_alternate = value != 0;
    } else if (!strcmp(key, "split")) {
        c->opt_split = value != 0;
    } else if (!strcmp(key, "dynamic")) {
        c->opt_dynamic = value != 0;
    } else if (!strcmp(key, "f16")) {
        c->opt_f16 = value != 0;
    } else if (!strcmp(key, "tail_mode")) {
        if (value < 0 || value > 2) return fail("tail_mode must be 0 (auto), 1 (all groups via the lane kernel) or 2 (none)");
        c->opt_tail_mode = value;
        release_plans(c);
    } else if (!strcmp(key, "lazy_upload")) {
        c->opt_lazy_upload = value != 0;
    } else if (!strcmp(key, "lane_acquire")) {
        c->opt_lane_acquire = value != 0;
    } else if (!strcmp(key, "wg_limit")) {
        if (value < 0) return fail("wg_limit must be >= 0 (0 = as many workgroups as the chip holds)");
        c->opt_wg_limit = value;
        release_plans(c);
    } else if (!strcmp(key, "wgs_per_cu")) {
        if (value < 0 || value > 16) return fail("wgs_per_cu must be 0..16");
        c->opt_wgs_per_cu = value;
        release_plans(c);
    } else {
        return fail("swimm_hip_set_option: unknown key '%s'", key);
    }
    return 0;
}

int swimm_hip_search_chunks(const char *query_sequences, const uint16_t *query_sequences_lengths,
                            uint32_t query_sequences_count, const uint32_t *query_disp,
                            uint64_t vect_sequences_db_count, char **chunk_b, uint32_t chunk_count,
                            const uint32_t *chunk_vect_sequences_db_count, uint16_t **chunk_n,
                            uint32_t **chunk_b_disp, const uint64_t *chunk_vD, const char *submat,
                            int open_gap, int extend_gap, int num_gpus, uint32_t vl, int32_t *scores,
                            double *workTime)
{
    if (!chunk_b || !chunk_vect_sequences_db_count || !chunk_n || !chunk_b_disp || !chunk_vD || !scores)
        return fail("swimm_hip_search_chunks: NULL argument");
    if (chunk_count == 0) return fail("swimm_hip_search_chunks: no chunks");
    const int avail = swimm_hip_device_count();
    if (avail <= 0) return 1;
    if (num_gpus <= 0 || num_gpus > avail) return fail("swimm_hip_search_chunks: %d GPUs requested, %d visible", num_gpus, avail);
    const double t0 = now_s();
    // chunk_accum_vect_sequences_db_count, MICsearch.c:46-49
    std::vector<uint64_t> accum(chunk_count, 0);
    for (uint32_t i = 1; i < chunk_count; ++i) accum[i] = accum[i - 1] + chunk_vect_sequences_db_count[i - 1];
    // static shard (north_star): chunks dealt longest-first onto the least-loaded GPU, cost = padded bytes
    std::vector<uint32_t> order(chunk_count);
    for (uint32_t i = 0; i < chunk_count; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return chunk_vD[a] > chunk_vD[b]; });
    std::vector<std::vector<uint32_t>> shard(num_gpus);
    std::vector<uint64_t> load(num_gpus, 0);
    for (uint32_t ci : order) {
        int best = 0;
        for (int g = 1; g < num_gpus; ++g) if (load[g] < load[best]) best = g;
        shard[best].push_back(ci);
        load[best] += chunk_vD[ci];
    }
    const uint64_t stride = vect_sequences_db_count * vl;
    std::vector<std::string> errs(num_gpus);
    std::vector<std::thread> th;
    for (int g = 0; g < num_gpus; ++g) {
        th.emplace_back([&, g]() {   // one host thread per device, as MICsearch.c:53
            if (shard[g].empty()) return;
            swimm_hip_ctx *ctx = nullptr;
            auto bail = [&]() { errs[g] = swimm_hip_last_error(); if (ctx) swimm_hip_destroy(ctx); };
            if (swimm_hip_create(g, &ctx)) return bail();
            // the caller's chunks outlive this call: stream them in while the search runs (X2 overlapped with compute)
            if (swimm_hip_set_option(ctx, "lazy_upload", 1)) return bail();
            if (swimm_hip_set_queries(ctx, query_sequences, query_sequences_lengths, query_disp, query_sequences_count,
                                      submat, open_gap, extend_gap)) return bail();
            for (uint32_t ci : shard[g])
                if (swimm_hip_add_chunk(ctx, chunk_b[ci], chunk_vD[ci], chunk_n[ci], chunk_b_disp[ci],
                                        chunk_vect_sequences_db_count[ci], vl, accum[ci])) return bail();
            if (swimm_hip_search(ctx, scores, stride, nullptr)) return bail();
            swimm_hip_destroy(ctx);
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < num_gpus; ++g)
        if (!errs[g].empty()) return fail("GPU %d: %s", g, errs[g].c_str());
    if (workTime) *workTime = now_s() - t0;
    return 0;
}

}  // extern "C"
