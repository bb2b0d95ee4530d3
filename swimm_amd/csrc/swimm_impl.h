// swimm_impl.h -- what the translation units of libswimm_hip.so share: the context behind the C-ABI handle, the
// work-list and upload records, and the functions of plan.cpp (launch plans and work lists), upload.cpp (chunks,
// the uploader thread) and search.cpp (one search) that the others call.  Not installed: include/swimm_hip.h is the boundary.
#pragma once
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

namespace swimm_impl {


extern thread_local std::string g_err;     // the calling thread's last error (swimm_hip_last_error)

inline int fail(const char *fmt, ...)
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

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A search that streams its database in builds work lists per range WHILE kernels run -- and a hipMalloc issued then was
// measured to wait for the running launch (c4: 48 ms inside make_db_plan, and the uploader's next copy stuck behind it).  Such
// lists are carved out of one arena the context owns, sized before the search's first launch (SearchRun::layout_ranges);
// the thread that builds them names the arena here and DevBuf::reserve takes from it.
struct DevArena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
    void *take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!base || used + bytes > cap) return nullptr;
        void *q = base + used;
        used += bytes;
        return q;
    }
};
extern thread_local DevArena *g_list_arena;

// (debug aid: device allocations made through DevBuf since the counters were last zeroed -- what a search allocates is on its clock)
struct AllocStats { double seconds = 0; size_t bytes = 0; unsigned calls = 0; };
extern AllocStats g_alloc_stats;
inline double alloc_now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T>
struct DevBuf {   // grow-only device scratch
    T *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;      // p lies in an arena: nothing to free
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (g_list_arena) {
            if (void *q = g_list_arena->take(n * sizeof(T))) {
                if (p && !borrowed) (void)hipFree(p);
                p = (T *)q; cap = n; borrowed = true;
                return hipSuccess;
            }
        }
        const double t0 = alloc_now_s();
        if (p && !borrowed) { hipError_t e = hipFree(p); if (e != hipSuccess) { p = nullptr; cap = 0; return e; } }
        p = nullptr; cap = 0; borrowed = false;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        g_alloc_stats.seconds += alloc_now_s() - t0; g_alloc_stats.bytes += n * sizeof(T); g_alloc_stats.calls++;
        return e;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
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
    DevBuf<LaneQ> lq;                   // the launches' queries ...
    DevBuf<uint32_t> block_map;         // ... and (query, pass) of every workgroup: every launch of a search takes a region of its own
    size_t lq_used = 0, bm_used = 0;    // (a scratch serves one range after the other; the tables of a launch must outlive it)
    void release() { bnd[0].release(); bnd[1].release(); queue.release(); prog.release(); lq.release(); block_map.release(); }
};

// one query of a lane-systolic launch, as the host hands it over
struct LaneQuery {
    uint32_t m; uint64_t prof_off; uint32_t prof_stride; uint64_t out_off;
    uint32_t items0 = 0, n_items = 0;   // its own part of the item list (promotion re-runs); n_items == 0: the whole list (the tail)
    uint64_t cols = 0;                  // ... and that part's boundary columns
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
    std::vector<uint32_t> own_disp;   // a piece of a caller's chunk: its groups' offsets counted from the piece's first byte (h_disp points here)
    uint32_t group_count = 0, vl = 0;
    const char *h_codes = nullptr; uint64_t code_bytes = 0;
    const uint16_t *h_len = nullptr;   // kind 1: the caller's lengths (the tiling kernel turns them into offsets itself)
    std::vector<uint32_t> gsrc; // kind 1: residue offset of every device group's first sequence in h_codes (n_groups + 1)
    std::vector<uint64_t> goff; // byte offset of every device group in d_tiled
    std::vector<uint32_t> gcols;
    bool uploaded = false, lens_known = false;
    uint32_t groups_uploaded = 0;   // device groups on the device so far (a chunk may travel in parts)
    size_t tiled_cap = 0, len_cap = 0;   // sizes of the two device buffers (they go back to the context's pool); tiled_cap 0: d_tiled is a share of another piece's buffer
    hipEvent_t ready = nullptr; // recorded on the upload stream behind the chunk's (re-)tile kernel
};

// what travels in one go: the device groups [g0, g1) of a chunk (the whole chunk, or -- for the chunk a streaming search
// starts with -- a small head part, so that the first launch has its data a few hundred microseconds after the call)
struct UploadPart {
    size_t chunk = 0; uint32_t g0 = 0, g1 = 0; hipEvent_t ready = nullptr;
    uint32_t publish = 0;       // > 0: once the part has landed, this many items of the search's one item list are on the device (PipeParams::avail)
};

struct QueryPlan {
    int T, W, passes; uint32_t mpad; size_t prof_off; Mode mode = Mode::F16; bool dynamic = true, resident = false;
    double est_s = 0;                   // choose_plan: the predicted time of the query's launches (makespan of the work lists included)
    bool sp = false;                    // the query runs through the score-profile kernel (option "sp_threshold")
    bool stack = false;                 // the "query" is a stack of short queries sharing one workgroup (QDesc in sw_kernels.h) ...
    uint32_t seam_mask = 0, wave_tab = 0;   // ... with these seams and this first entry in d_wave_out
};


struct Range { uint32_t g0 = 0, g1 = 0; uint64_t cols = 0; };
struct WorkUnit { uint32_t group, half, out_slot; uint32_t ncols; uint64_t bnd_off; };

}  // namespace swimm_impl

using namespace swimm_impl;

struct swimm_hip_ctx {
    int device = 0;                     // physical device
    int vdevice = 0;                    // the device number the caller asked for (differs under the test hook SWIMM_HIP_VIRTUAL_GPUS)
    hipStream_t stream = nullptr;
    hipStream_t stream_b = nullptr;     // second bulk stream: multi-pass queries run the two halves of the group list side by side
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipStream_t stream2 = nullptr;      // lane-systolic tail runs beside the bulk kernel
    hipEvent_t ev_tail = nullptr;
    hipStream_t stream_up = nullptr;    // uploads: H2D copies, (re-)tile kernels -- never waits for a DP kernel
    hipStream_t stream_list = nullptr;  // work lists (list_copy)
    hipEvent_t ev_copied = nullptr;
    void *up_pin = nullptr; size_t up_pin_cap = 0, up_pin_used = 0;      // pinned staging of a part's small arrays (upload_part)
    DevBuf<uint8_t> up_b; DevBuf<uint16_t> up_n; DevBuf<uint32_t> up_disp, up_gcols, up_off; DevBuf<uint64_t> up_goff;   // upload scratch, reused chunk after chunk
    int opt_upload_piece_kib = 98304;   // lazy_upload: chunks and slabs larger than this are recorded in pieces of about this size (96 MiB, the reference's chunk size)
    int opt_lazy_upload = 0;            // 1: add_chunk / add_sequences record the caller's buffers, the first search streams them in
    hipStream_t stream3 = nullptr;      // promotion re-runs
    hipEvent_t ev_ready = nullptr, ev_tail3 = nullptr;
    std::vector<hipEvent_t> ev_query;   // [2q] bulk done, [2q+1] tail done
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = 0;
    // options (swimm_hip_set_option)
    int opt_T = 0, opt_maxW = 0, opt_W = 0;   // launch shape: 0 = chosen per query
    int opt_force_i32 = 0;              // 1: everything in int32
    int opt_f16 = 1;                    // 1: packed binary16 first tier (exact below 2048, then int16, then int32)
    int opt_tail_mode = 0;              // 0 auto, 1 every group through the lane kernel, 2 none
    int opt_tail_frac = 30;             // a group goes to the lane kernel when it is longer than this percentage of a CU's mean load ...
    int opt_tail_cap = 25;              // ... as long as the tail stays below this many per mille of the search's cells (0 = no cap)
    int opt_dynamic = 1;                // 1: workgroups pull items from a global queue (default); 0: static partition by the host
    int opt_resident = -1;              // group-resident batch launches: -1 = when the batch has two or more queries that are not rotated, 0 never, 1 always
    bool batch_now = false;             // the search in progress runs its non-rotated queries as group-resident batch launches
    std::vector<uint8_t> stream_tail;   // streaming search: the tail flags of the whole database (pick_tail)
    bool streaming_now = false;         // the search in progress streams its database in (per-range launches, no group-resident batches)
    double add_seconds = 0;             // host time spent in add_chunk / add_sequences since the last search (reported under SWIMM_HIP_DEBUG)
    bool tiling_room = false;           // ... and the launch shapes being chosen must leave the tiling waves their registers (plan.cpp, choose_plan)
    DevBuf<QDesc> d_qdesc;              // group-resident launches: per batch, its queries
    DevBuf<uint32_t> d_wave_out;        // stacks of short queries: per (stack, wave) the first element of the wave's member's score row
    int opt_cut = 35;                   // outlier pairs: a group's longest pairs leave it for the lane-systolic kernel when that saves the pipeline kernel more padded cells than opt_cut/10 x the pairs' own (0 = never)
    std::vector<uint32_t> cut_cols;     // per group: the columns the pipeline kernel aligns (<= ncols; the pairs that are longer are lane-systolic items as well)
    std::vector<uint8_t> cut_lane;      // per group: the first lane (pair) that is an outlier (64 = none)
    int opt_sp_threshold = 65536;       // queries of at least this many rows use the score-profile kernel (the reference's query_length_threshold, MICsearch.c:39-43); 65536 = none
    DevBuf<int8_t> d_qcodes;            // score-profile kernel: the residue codes of its queries, each padded with the dummy code to a multiple of 32
    DevBuf<uint16_t> d_sub16;           // ... and the substitution matrix as binary16 bits [24][32]
    int opt_stack = 1;                  // 1: short one-pass queries of a batch share workgroups (several queries stacked along the strips)
    int opt_time_launches = 0;          // 1: every pipeline launch is bracketed by events on its own stream (measurement aid, bench.py)
    std::vector<hipEvent_t> launch_ev;  // pairs (before, after), grown on demand
    size_t launch_ev_used = 0;
    double launch_ms_sum = 0;           // sum of the pipeline launches' own durations in the last search
    uint32_t launch_ms_n = 0;
    int opt_bnd_mib = 16384;            // HBM budget of the pass-boundary buffer (MiB)
    int opt_score_mib = 32768;          // HBM budget of the score rows of one query batch (MiB)
    int opt_wg_limit = 0;               // > 0: at most this many persistent workgroups per pipeline launch (tests: long per-workgroup item sequences on a small database)
    // caches that depend on the resident database / the code objects
    BulkCols bulk;                      // the resident database's bulk groups (built on demand) and their makespan factors
    int regs_cache[2][3][40] = {};      // VGPRs of sw_pipe_kernel<T, tier, dynamic, group-resident or not>, looked up once
    hipEvent_t ev_avail = nullptr;      // ... its count has been reset (on the upload stream)
    DevBuf<uint32_t> d_avail;           // streaming search with one item list: how many of its items have landed (publish_items_kernel)
    DevBuf<Item> d_stream_items;        // ... and that list (every group of the database, in the order the parts travel)
    DevArena list_arena;                // work lists of the ranges of a streaming search (DevArena above)
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
    std::vector<uint16_t> seq_len;      // true length of every local slot (.seq slabs: as handed over; chunk layout: from the re-tile kernel)
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
    std::vector<hipEvent_t> part_ev;    // events of the head parts of a streaming search's first chunk (grown on demand)
    std::vector<std::pair<void *, size_t>> pool;   // device buffers of a cleared database, kept for the chunks registered next (freed by the next search)
    DevBuf<unsigned long long> d_stamps;   // diagnostic build only
    LaneScratch tail_scratch;           // lane kernel on stream 2 (long-sequence tail: the launch of the queries with 8 rows per lane)
    LaneScratch tail_scratch_t[2];      // ... and the launches of the short queries (4 and 2 rows per lane), same stream
    LaneScratch rerun_scratch;          // lane kernel on stream 3 (promotion re-runs)
    DevBuf<LaneItem> d_rerun_items;
    DevBuf<uint32_t> d_satlist;
    DevBuf<uint32_t> d_ladder_counts;   // promotion ladder: where every query's part of the shared list ends
    // stats of the last search
    double kernel_ms = 0;
    uint64_t cells = 0, promoted = 0, promoted16 = 0;
    std::vector<QueryPlan> last_plans;
    uint32_t launches = 0;
};

namespace swimm_impl {

// Device discipline.  A context belongs to one device and every HIP call made for it -- allocations, copies, launches,
// events -- needs that device to be the calling thread's current one.  Every entry point of the C-ABI (and the uploader
// thread) starts with ctx_enter(); the internal functions that issue HIP calls start with CHECK_DEVICE, which compares a
// thread-local record of the last context entered with the context at hand.  With the test hook SWIMM_HIP_VIRTUAL_GPUS all
// "devices" are one physical GPU and a missing hipSetDevice would go unnoticed by HIP itself; the record is kept per
// VIRTUAL device, so it shows even there (tests/test_gpu_parity.py::test_two_contexts_interleaved_on_one_thread).
extern thread_local int g_cur_vdevice;
inline int ctx_enter(swimm_hip_ctx *c)
{
    HIP_TRY(hipSetDevice(c->device));
    g_cur_vdevice = c->vdevice;
    return 0;
}
#define CHECK_DEVICE(c)                                                                                            \
    do {                                                                                                           \
        if (swimm_impl::g_cur_vdevice != (c)->vdevice)                                                             \
            return fail("internal: %s runs for device %d but the calling thread last entered device %d (%s:%d)", __func__, (c)->vdevice, \
                        swimm_impl::g_cur_vdevice, __FILE__, __LINE__);                                            \
    } while (0)

int upload_chunk(swimm_hip_ctx *c, ChunkRec &r);
int upload_part(swimm_hip_ctx *c, ChunkRec &r, uint32_t g0, uint32_t g1, hipEvent_t ready);

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
    std::vector<UploadPart> order;  // the job: the parts in the order they travel
    bool have_job = false, busy = false, quit = false, stop = false;
    bool warm = false;              // the thread has made its first HIP calls (they cost up to 5 ms: not inside a search)
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
        bool dev_ok = ctx_enter(c) == 0;
        if (dev_ok) (void)hipStreamQuery(c->stream_up);      // (the runtime's per-thread set-up happens here, once)
        std::unique_lock<std::mutex> lk(mu);
        warm = true;
        cv.notify_all();
        for (;;) {
            cv.wait(lk, [&]() { return have_job || quit; });
            if (quit) return;
            have_job = false;
            const std::vector<UploadPart> job = order;
            lk.unlock();
            bool ok = dev_ok;
            std::string e = ok ? "" : "uploader: hipSetDevice failed";
            for (size_t i = 0; i < job.size(); ++i) {
                bool skip;
                { std::lock_guard<std::mutex> g(mu); skip = stop; }
                if (ok && !skip && upload_part(c, c->chunks[job[i].chunk], job[i].g0, job[i].g1, job[i].ready)) { ok = false; e = g_err; }
                // (a search that walks ONE item list while the database streams in: tell its launch how far the list has landed --
                // behind the part's tiling kernel on the same stream; also for a part that was on the device already)
                if (ok && !skip && job[i].publish && launch_publish_items(c->d_avail.p, job[i].publish, c->stream_up) != hipSuccess) { ok = false; e = "publish_items launch failed"; }
                std::lock_guard<std::mutex> g(mu);
                issued = i + 1; failed = !ok; err = e;
                cv.notify_all();
            }
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void wait_warm()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return warm; });
    }
    void post(const std::vector<UploadPart> &job)
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

// ---- plan.cpp: occupancy, launch plans, work lists
void release_plans(swimm_hip_ctx *c);
int regs_to_waves_per_simd(int regs);
int kernel_regs(const swimm_hip_ctx *c, Mode mode, int T, bool resident, int *out);
int wgs_per_cu(const swimm_hip_ctx *c, Mode mode, int T, int W, bool resident, int *out);
bool resident_for(const swimm_hip_ctx *c, int passes);
int n_workgroups(const swimm_hip_ctx *c, int per_cu);      // persistent workgroups of a pipeline launch: what the chip holds, unless the caller caps it
hipStream_t list_stream(const swimm_hip_ctx *c);
int list_copy(swimm_hip_ctx *c, void *dst, const void *src, size_t bytes);
int list_sync(swimm_hip_ctx *c);
Range whole_range(const swimm_hip_ctx *c);
std::vector<uint8_t> pick_tail(const swimm_hip_ctx *c, const Range &rg);
void ensure_cuts(swimm_hip_ctx *c);
inline uint32_t bulk_cols(const swimm_hip_ctx *c, uint32_t g) { return g < c->cut_cols.size() ? c->cut_cols[g] : c->groups[g].ncols; }
double lpt_imbalance(BulkCols &b, int n_wg);
void bulk_cols_of(const swimm_hip_ctx *c, const Range &rg, BulkCols &b);
double plan_imbalance(swimm_hip_ctx *c, int n_wg);
int choose_plan(swimm_hip_ctx *c, Mode mode, int m, bool room_for_lane_waves, bool overlapped, QueryPlan *out,
                const Range *rg = nullptr, BulkCols *rb = nullptr);
int choose_one_list_plan(swimm_hip_ctx *c, int m, QueryPlan *out, int *n_wg_out);
double shape_gcups(int T, int W);      // measured rate of a launch shape of the f16-tier pipeline kernel (GCUPS of padded cells)
uint64_t prof_elems_bound(const uint16_t *qm, uint32_t qn);
int choose_batch_shapes(swimm_hip_ctx *c, Mode mode, const uint16_t *qm, uint32_t qn, std::vector<QueryPlan> &qps);
int build_plan(swimm_hip_ctx *c, const std::vector<WorkUnit> &units, int n_wg, Plan &pl);
int upload_lane_items(swimm_hip_ctx *c, std::vector<LaneItem> &v, LaneList &ll);
int make_db_plan(swimm_hip_ctx *c, Mode mode, int n_wg, bool no_tail, const Range &rg, bool exact_lengths, DbPlan &dp);
int get_db_plan(swimm_hip_ctx *c, Mode mode, int n_wg, bool whole_db, DbPlan **out);
void fill_common(const swimm_hip_ctx *c, const QueryPlan &qp, PipeParams &p, uint2 *bnd);
uint64_t bnd_budget_cols(const swimm_hip_ctx *c);
void boundary_segments(const swimm_hip_ctx *c, const Plan &pl, std::vector<std::pair<uint32_t, uint32_t>> &segs, uint64_t *max_cols);
bool use_split(const swimm_hip_ctx *c, const QueryPlan &qp, const Plan &pl, size_t n_segs);

// ---- upload.cpp: chunks (upload_chunk is declared above, before the Uploader)
int refresh_plans(swimm_hip_ctx *c);
int register_chunk(swimm_hip_ctx *c, ChunkRec &rec, const uint16_t *lens_or_null, uint64_t n_lens);
int ensure_uploader(swimm_hip_ctx *c);
int sync_lengths(swimm_hip_ctx *c);
int pool_alloc(swimm_hip_ctx *c, size_t bytes, void **out, size_t *cap);
void pool_trim(swimm_hip_ctx *c);

// ---- search.cpp: the launches of one search
int timed_launch(swimm_hip_ctx *c, Mode mode, int T, int W, int n_wg, const PipeParams &p, hipStream_t st);
uint64_t resident_bnd_elems(const Plan &pl);
int run_resident_batch(swimm_hip_ctx *c, Mode mode, int T, int W, const Plan &pl, const QDesc *qd, uint32_t nq, uint64_t pass_sum, uint32_t max_passes,
                       hipStream_t st, DevBuf<uint2> &bnd);
int run_passes(swimm_hip_ctx *c, Mode mode, const QueryPlan &qp, const Plan &pl, int32_t *out_row, hipStream_t st, bool allow_split, DevBuf<uint2> &bnd);
int run_sp_passes(swimm_hip_ctx *c, const QueryPlan &qp, const Plan &pl, const int8_t *qcodes, int32_t *out_row, hipStream_t st, DevBuf<uint2> &bnd);
int lane_rows_for(const swimm_hip_ctx *c, uint32_t m);
int run_lane_batch(swimm_hip_ctx *c, Mode mode, int rows_per_lane, const std::vector<LaneQuery> &qs, const LaneList &ll, hipStream_t st, LaneScratch &sc);
int reserve_lane_scratch(swimm_hip_ctx *c, LaneScratch &sc, size_t list_cols, size_t items, size_t pass_total, size_t queries, size_t multi_pass_queries, size_t launches = 1);
int search_device(swimm_hip_ctx *c, uint32_t qb, uint32_t qe, uint64_t *slots_out);

}  // namespace swimm_impl
