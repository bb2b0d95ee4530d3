// plan.cpp -- launch plans (rows per wave, waves, passes from the measured rate table and the longest-first
// makespan) and the work lists of the persistent workgroups / the lane-systolic tail.
#include "swimm_impl.h"

namespace swimm_impl {

Range whole_range(const swimm_hip_ctx *c) { Range r; r.g0 = 0; r.g1 = (uint32_t)c->groups.size(); r.cols = c->total_cols; return r; }

void release_plans(swimm_hip_ctx *c)
{
    for (auto &kv : c->plans) { kv.second.main.release(); kv.second.tail.release(); }
    c->plans.clear();
    c->bulk.cols.clear(); c->bulk.total = 0; c->bulk.cache.clear();
    c->cut_cols.clear(); c->cut_lane.clear();
}

// Outlier pairs.  A device group is 128 neighbours of the length-sorted database and the pipeline kernel runs it to its
// longest member: fine in the bulk of the database, where neighbours differ by a residue or two, and a waste at its long
// end -- Swiss-Prot's last group holds 35 000-residue titin beside sequences of 5 000, so 128 lanes run 35 000 columns for
// the sake of a handful.  Per group (true lengths known): the pipeline kernel aligns columns [0, cut) of all 128 sequences,
// and the pairs (lanes) that are longer than cut are ALSO lane-systolic items, whole: a prefix of an alignment never scores
// above the alignment, so the two results merge through the score row's atomicMax, exactly.  The cut is the one that
// minimises  128 x cut  +  rho x (cells of the pairs beyond it),  rho = opt_cut / 10: what a cell costs in the lane-systolic
// kernel (135 instructions per 8-row column against 68, one dependent chain per wave) against a padded cell of the pipeline.
void ensure_cuts(swimm_hip_ctx *c)
{
    if (c->cut_cols.size() == c->groups.size()) return;
    const size_t n = c->groups.size();
    c->cut_cols.resize(n); c->cut_lane.assign(n, 64);
    const double rho = c->opt_cut * 0.1;
    for (size_t g = 0; g < n; ++g) {
        const GroupDesc &gd = c->groups[g];
        c->cut_cols[g] = gd.ncols;
        if (c->opt_cut <= 0 || (size_t)gd.seq0 + kGroupSeqs > c->seq_len.size()) continue;
        // (a cut must save 3 % of the group's padded cells at least: a group whose FIRST pair is within 3 % of its longest member has
        // nothing to cut -- nearly every group of a length-sorted database; two reads instead of 128)
        {
            const uint32_t l0 = std::max(c->seq_len[gd.seq0], c->seq_len[gd.seq0 + 1]);
            if ((double)l0 >= 0.97 * (double)gd.ncols - kChunkCols) continue;
        }
        uint32_t L[64];
        int last = -1;                                 // the last pair that holds a sequence (the database's last group is seldom full)
        bool sorted = true;
        for (int l = 0; l < 64; ++l) {
            L[l] = std::max(c->seq_len[gd.seq0 + 2 * l], c->seq_len[gd.seq0 + 2 * l + 1]);
            if (L[l] > 0) { if (last >= 0 && L[l] < L[last]) sorted = false; last = l; }
        }
        if (last < 1 || !sorted) continue;             // (lengths not reported yet, a single pair, or a caller's database that is not sorted)
        double beyond = 0, best = 128.0 * gd.ncols;    // cost with no pair cut off
        int cut_at = 64;
        for (int l = last; l >= 1; --l) {              // pairs l .. last leave: the pipeline runs to pair l - 1
            beyond += 2.0 * L[l];
            const uint32_t cols = (L[l - 1] + kChunkCols - 1) / kChunkCols * kChunkCols;
            const double cost = 128.0 * std::max<uint32_t>(cols, kChunkCols) + rho * beyond;
            if (cost < best * 0.97) { best = cost; cut_at = l; }       // (3 %: not for a handful of columns)
        }
        if (cut_at < 64) {
            c->cut_lane[g] = (uint8_t)cut_at;
            c->cut_cols[g] = std::max<uint32_t>((L[cut_at - 1] + kChunkCols - 1) / kChunkCols * kChunkCols, kChunkCols);
        }
    }
}

// Registers a launch must leave free on every SIMD while the database is still streaming in.  The tiling kernels need 32; the
// runtime's own copy kernels (pageable host memory travels through them) need more: with 32 free (four waves of 120) a chunk
// copy waited 455 ms for a group-resident launch to end, with 56 free (three waves of 152) it took its 2 ms.
static const int kUploadRoomRegs = 48;

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
    const size_t lds = pipe_lds_bytes(mode, T, W, resident);
    int n = std::min(waves_cu / W, (int)(163840 / lds));
    *out = std::max(1, n);
    return 0;
}

// a query of several passes runs the group-resident kernel (one launch) unless that is switched off
bool resident_for(const swimm_hip_ctx *c, int passes) { (void)passes; return c->batch_now; }

// Work lists travel on a stream of their own (not behind the uploader's chunk copies, not behind the promotion ladder)
hipStream_t list_stream(const swimm_hip_ctx *c) { return c->stream_list; }

// ... and through a pinned arena: a copy from pageable memory would queue for the runtime's staging buffers behind
// the uploader's chunk copies (measured: 1.2 ms per range's lists instead of 0.3).  list_sync() ends a batch of copies.
int list_copy(swimm_hip_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    CHECK_DEVICE(c);
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
AllocStats g_alloc_stats;
double g_list_sync_wait_s = 0;          // (debug aid: seconds the host spent in list_sync since it was last zeroed)
int list_sync(swimm_hip_ctx *c)
{
    struct Tm { double t0; Tm() : t0(now_s()) {} ~Tm() { g_list_sync_wait_s += now_s() - t0; } } tm;
    HIP_TRY(hipStreamSynchronize(list_stream(c)));
    c->pin_used = 0;
    return 0;
}


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
double lpt_imbalance(BulkCols &b, int n_wg)
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
void bulk_cols_of(const swimm_hip_ctx *c, const Range &rg, BulkCols &b)
{
    const std::vector<uint8_t> is_tail = pick_tail(c, rg);
    b.cols.clear(); b.total = 0; b.cache.clear();
    for (uint32_t g = rg.g0; g < rg.g1; ++g)
        if (!is_tail[g - rg.g0]) { b.cols.push_back(bulk_cols(c, g)); b.total += bulk_cols(c, g); }
    std::sort(b.cols.begin(), b.cols.end(), std::greater<uint32_t>());
}

double plan_imbalance(swimm_hip_ctx *c, int n_wg)
{
    if (c->bulk.cols.empty() && !c->groups.empty()) bulk_cols_of(c, whole_range(c), c->bulk);     // once per database
    return lpt_imbalance(c->bulk, n_wg);
}

// Measured throughput (GCUPS of padded cells) of every launch shape of the f16-tier pipeline kernel: rows per wave
// T = 8, 12, ... 36 (lines) by waves per workgroup W = 1..16 (columns), workgroups per CU by occupancy
// (tools/plan_sweep.py --scale 1.0 on one MI355X, profiles/r03_plan_sweep.txt: the binary16 tier in column-offset form with the fused pair score,
// 15-25 % above round 2's table (profiles/r02_plan_sweep.txt) for the shapes of four waves and more).  W = 4, 8, 12, 16 put the same number of waves
// on each of the CU's 4 SIMDs; any other W runs like the next multiple of 4 (2 x 6 waves behave like 4+4+2+2).
static const float kShapeGcups[8][16] = {
    {2505, 4154, 5428, 6628, 6680, 8128, 7191, 8828, 6873, 7535, 8095, 8618, 6252, 6722, 7150, 7597},  // T=8
    {3236, 5153, 6724, 8362, 7541, 6328, 8171, 9212, 6042, 6698, 7350, 7991, 7057, 7561, 7996, 8497},  // T=12
    {3529, 5516, 7361, 9401, 6683, 7457, 8624, 9612, 6442, 7155, 7766, 8403, 7441, 8006, 8516, 9040},  // T=16
    {3707, 5834, 7623, 9695, 7053, 7696, 8874, 9906, 6697, 7415, 8073, 8727, 7682, 8238, 8717, 9240},  // T=20
    {3792, 6093, 7964, 10162, 7261, 7895, 9033, 10084, 6949, 7704, 8364, 9025, 7906, 8512, 9031, 9606},  // T=24
    {3692, 5881, 8136, 10201, 7197, 7887, 9147, 10298, 6920, 7672, 8406, 9140, 8050, 8654, 9242, 9856},  // T=28
    {3777, 6005, 8292, 10334, 7298, 7984, 9240, 10422, 7031, 7772, 8536, 9270, 0, 0, 0, 0},  // T=32
    {3954, 6498, 8776, 10253, 5255, 6252, 7264, 8243, 7253, 8016, 8804, 9554, 0, 0, 0, 0},  // T=36
};

double shape_gcups(int T, int W) { return (T >= 8 && T <= 36 && T % 4 == 0 && W >= 1 && W <= 16) ? (double)kShapeGcups[(T - 8) / 4][W - 1] : 0.0; }

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
                const Range *rg, BulkCols *rb)
{
    struct Cand { double base; double pass_base; int T, W, passes, n_wg; };
    std::vector<Cand> cands;
    const double cols = rg ? (double)rg->cols : (double)c->total_cols;
    for (int ti = 7; ti >= 0; --ti) {
        const int T = 8 + 4 * ti;
        if (c->opt_T && T != c->opt_T) continue;
        if (!pipe_has_variant(mode, T)) continue;
        int maxW = T > 28 ? 12 : 16;        // __launch_bounds__ of the instantiations
        if (c->opt_maxW > 0) maxW = std::min(maxW, c->opt_maxW);
        const int strips = std::max(1, (m + T - 1) / T);
        for (int W = 1; W <= maxW; ++W) {
            if (c->opt_W > 0 && W != std::min(c->opt_W, maxW)) continue;
            const int passes = (strips + W - 1) / W;
            if (overlapped && passes != 1) continue;   // only one-pass queries take part in the rotation
            int per_cu = 1;
            if (wgs_per_cu(c, mode, T, W, resident_for(c, passes), &per_cu)) return 1;
            // registers that must stay free on every SIMD: 80 for a lane-systolic wave; 32 for the tiling waves of a database
            // that is still streaming in (c->tiling_room: a launch whose workgroups fill the register file holds the upload
            // stream's tiling kernels -- and the copies queued behind them -- back for as long as it runs)
            const int keep = std::max(room_for_lane_waves ? 80 : 0, c->tiling_room ? kUploadRoomRegs : 0);
            if (keep && !c->opt_T) {
                int regs = 0;
                if (kernel_regs(c, mode, T, resident_for(c, passes), &regs)) return 1;
                const int alloc = (regs + 7) / 8 * 8;
                if (alloc * ((per_cu * W + 3) / 4) > 512 - keep) continue;
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
    const bool dbg = getenv("SWIMM_HIP_DEBUG_PLAN") != nullptr;
    for (const Cand &k : cands) {
        if (best_cost >= 0 && k.base >= best_cost * (1.0 - 1e-9)) break;
        const double imb = overlapped ? 1.0 : (rb ? lpt_imbalance(*rb, k.n_wg) : plan_imbalance(c, k.n_wg));
        const double cost = k.passes * (k.pass_base * imb + 150e-6);
        if (dbg) fprintf(stderr, "swimm_hip: plan candidate m=%d: %d x %d rows, %d passes, %d workgroups: %.3f ms x makespan %.3f -> %.3f ms%s\n", m, k.W, k.T, k.passes, k.n_wg,
                         k.base * 1e3, imb, cost * 1e3, room_for_lane_waves ? " (room for lane waves)" : "");
        if (best_cost < 0 || cost < best_cost * (1.0 - 1e-9)) {
            best_cost = cost;
            out->T = k.T; out->W = k.W; out->passes = k.passes; out->mpad = (uint32_t)(k.passes * k.W * k.T); out->est_s = cost;
        }
    }
    // (no admissible shape leaves a lane-systolic wave its registers, e.g. under a max_waves cap: the tail then shares)
    if (best_cost < 0 && room_for_lane_waves) return choose_plan(c, mode, m, false, overlapped, out, rg, rb);
    if (best_cost < 0 && c->tiling_room) { c->tiling_room = false; const int rc = choose_plan(c, mode, m, false, overlapped, out, rg, rb); c->tiling_room = true; return rc; }
    if (best_cost < 0) { fail("no kernel variant for rows_per_wave=%d waves=%d", c->opt_T, c->opt_W); return 1; }
    return 0;
}

// The launch shape of a streaming search's ONE launch over the growing item list (search.cpp, layout_ranges): one pass must hold
// the query, and the workgroups -- resident for the whole search, waiting with all their registers whenever the link is behind --
// must leave a tiling wave its registers on EVERY SIMD: the upload stream's tiling kernels have to run beside them on the same
// CUs (workgroups are dealt to shader engines before a free CU is looked for; tools/microbench/spin_probe: beside 248
// register-full workgroups on 256 CUs a newcomer waits for the launch to end, with registers to spare on every CU it runs at
// once).  Registers: read from the code object of the growing-list instantiation.  Cost: padded rows over the shape's measured rate.
int choose_one_list_plan(swimm_hip_ctx *c, int m, QueryPlan *out, int *n_wg_out)
{
    const int kTilingRegs = kUploadRoomRegs;
    double best = -1;
    for (int ti = 7; ti >= 0; --ti) {
        const int T = 8 + 4 * ti;
        if ((c->opt_T && T != c->opt_T) || !pipe_has_variant(Mode::F16, T)) continue;
        int regs = 0;
        if (grow_kernel_attributes(T, &regs) != hipSuccess) continue;
        const int alloc = (regs + 7) / 8 * 8;
        int maxW = T > 28 ? 12 : 16;
        if (c->opt_maxW > 0) maxW = std::min(maxW, c->opt_maxW);
        const int strips = std::max(1, (m + T - 1) / T);
        for (int W = strips; W <= maxW; ++W) {
            if (c->opt_W > 0 && W != std::min(c->opt_W, maxW)) continue;
            const int nominal = std::max(1, std::min(4 * regs_to_waves_per_simd(regs) / W, (int)(163840 / pipe_lds_bytes(Mode::F16, T, W, false))));
            int per_cu = nominal;
            while (per_cu > 0 && alloc * ((per_cu * W + 3) / 4) > 512 - kTilingRegs) --per_cu;
            if (per_cu == 0) continue;
            const double cost = (double)(T * W) / (shape_gcups(T, W) * per_cu / nominal);
            if (best < 0 || cost < best * (1.0 - 1e-9)) {
                best = cost;
                out->T = T; out->W = W; out->passes = 1; out->mpad = (uint32_t)(T * W); out->est_s = 0;
                *n_wg_out = n_workgroups(c, per_cu);
            }
        }
    }
    return best < 0 ? 1 : 0;
}

// upper bound of the profile elements of a query batch (25 codes x rows padded to at most 16 x 36 and to the lane kernel's 512)
uint64_t prof_elems_bound(const uint16_t *qm, uint32_t qn)
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
    // the group-resident instantiations against the per-pass ones the rate table was measured with (profiles/
    // r03_resident_vs_per_pass_by_shape.txt, the fused-pair-score kernels): 3-4 % for every shape but the 32-row one, whose
    // group-resident instantiation needs 135 registers (three waves per SIMD) where the per-pass one fits four in 125
    static const double kResidentFactor[8] = {1.0, 1.0, 0.962, 0.967, 0.969, 0.973, 0.894, 0.960};
    auto rate_of = [&](int ti, int W) { return (double)kShapeGcups[ti][W - 1] * kResidentFactor[ti]; };
    for (int ti = 7; ti >= 0; --ti) {
        const int T = 8 + 4 * ti;
        if (c->opt_T && T != c->opt_T) continue;
        if (!pipe_has_variant(mode, T)) continue;
        Wof[ti] = c->opt_W > 0 ? std::min(c->opt_W, T > 28 ? 12 : 16) : std::min(4, c->opt_maxW > 0 ? c->opt_maxW : 4);
        if (c->tiling_room && !c->opt_T) {          // (a database that is still streaming in: see choose_plan)
            int regs = 0, per_cu = 1;
            if (kernel_regs(c, mode, T, true, &regs) || wgs_per_cu(c, mode, T, Wof[ti], true, &per_cu)) return 1;
            if ((regs + 7) / 8 * 8 * ((per_cu * Wof[ti] + 3) / 4) > 512 - kUploadRoomRegs) continue;
        }
        ok[ti] = true;
        for (uint32_t q = 0; q < qn; ++q) cost[ti] += rows_of(qm[q], T, Wof[ti]) / rate_of(ti, Wof[ti]);
    }
    int common = -1;
    for (int ti = 7; ti >= 0; --ti)
        if (ok[ti] && (common < 0 || cost[ti] < cost[common])) common = ti;
    if (common < 0) return fail("no group-resident kernel variant for rows_per_wave=%d waves=%d", c->opt_T, c->opt_W);
    for (uint32_t q = 0; q < qn; ++q) {
        int pick = common;
        const double cc = rows_of(qm[q], 8 + 4 * common, Wof[common]) / rate_of(common, Wof[common]);
        double bc = cc;
        for (int ti = 7; ti >= 0; --ti) {
            if (!ok[ti]) continue;
            const double x = rows_of(qm[q], 8 + 4 * ti, Wof[ti]) / rate_of(ti, Wof[ti]);
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
            Item it{}; it.db = gd.db; it.ncols = u.ncols; it.seq0 = gd.seq0; it.half = u.half; it.out_slot = u.out_slot; it.bnd_off = u.bnd_off;
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
        Item it{}; it.db = gd.db; it.ncols = u.ncols; it.seq0 = gd.seq0; it.half = u.half; it.out_slot = u.out_slot; it.bnd_off = before;
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
    {   // no group is longer than the threshold for the WHOLE range's mean load (the loop below only ever compares with a smaller
        // mean): no tail, and no need to sort -- every length-sorted database of ordinary proteins leaves here
        uint32_t longest = 0;
        for (uint32_t i = 0; i < n; ++i) longest = std::max(longest, bulk_cols(c, rg.g0 + i));
        if ((double)longest <= c->opt_tail_frac * 0.01 * (double)rg.cols / c->num_cu) return is_tail;
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    // (a group counts with the columns the pipeline kernel would align: its outlier pairs are lane-systolic items anyway)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return bulk_cols(c, rg.g0 + a) > bulk_cols(c, rg.g0 + b); });
    uint64_t rest = rg.cols;
    // The yardstick is the load of a CU, however many workgroups share it -- and the lane-systolic kernel must stay a side
    // show: it aligns a cell at 1.5x the pipeline kernel's instructions and, beside the bulk waves, at a tenth of its rate,
    // so the tail takes at most tail_cap per mille of the search's cells (pairs run to their longer member).  Measured
    // (profiles/r02_tail_fraction.txt): c3 at full size is best at 30 % / 2.5 % of the cells (8 180 GCUPS; 7 910 at 50 % /
    // 1.1 %, 7 750 at 20 % / 9 %), c3 at 30 % of its size at 2.5-7 % (6 170-6 190; 4 900 at 29 %, 2 300 at 93 %).
    // (no cap for a database of fewer groups than CUs: the pipeline kernel could not even give every CU a group, the lane-systolic
    // kernel spreads every alignment over a wave -- 1.2e7 residues: 2 250 GCUPS all through the lane kernel, 1 680 under the cap)
    const bool capped = c->opt_tail_cap > 0 && n >= (uint32_t)c->num_cu;
    const double cap = capped ? (double)rg.cols * kGroupSeqs * (c->opt_tail_cap + 0.5) * 1e-3 : 1e300;
    double tail_cells = 0;
    for (uint32_t g : order) {
        const GroupDesc &gd = c->groups[rg.g0 + g];
        const uint32_t gcols = bulk_cols(c, rg.g0 + g);
        const double mean = (double)rest / c->num_cu;
        if ((double)gcols <= c->opt_tail_frac * 0.01 * mean) break;
        double cells = 0;
        for (uint32_t l = 0; l < 64; ++l) cells += 2.0 * std::max(c->seq_len[gd.seq0 + 2 * l], c->seq_len[gd.seq0 + 2 * l + 1]);
        if (cells == 0) cells = 128.0 * gd.ncols;          // (a chunk whose true lengths the re-tile kernel has yet to report: the group's)
        // (a group longer than a whole CU's mean load would hold up every launch it is part of: for those the cap is 8 % --
        // c3 at 10 % of its size, 423 groups of which most are that long: 2 970 GCUPS under the 2.5 % cap, 3 680 under 8 %)
        if (capped && tail_cells + cells > ((double)gcols > mean ? std::max(cap, (double)rg.cols * kGroupSeqs * 0.08) : cap)) break;
        tail_cells += cells;
        is_tail[g] = 1;
        rest -= gd.ncols;
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
        const bool cuts = !no_tail && exact_lengths && c->opt_tail_mode != 2;       // (a launch that takes every group through the pipeline kernel has no lane-systolic kernel beside it)
        if (cuts) ensure_cuts(c);
        std::vector<uint8_t> is_tail = pick_tail(c, rg);
        if (no_tail) std::fill(is_tail.begin(), is_tail.end(), 0);   // every group through the pipeline kernel
        for (uint32_t g = rg.g0; g < rg.g1; ++g) {
            const GroupDesc &gd = c->groups[g];
            if (!is_tail[g - rg.g0]) {
                units.push_back(WorkUnit{g, 0, 0, cuts ? bulk_cols(c, g) : gd.ncols, c->group_col_off[g] - col0});
                if (!cuts || c->cut_lane[g] >= 64) continue;
            }
            for (uint32_t l = is_tail[g - rg.g0] ? 0 : c->cut_lane[g]; l < 64; ++l) {
                // a pair only runs to the end of its longer member, not to the end of the group (when the lengths are
                // already known: a chunk that is still streaming in runs to the end of its group -- padding scores 0)
                const uint32_t len = exact_lengths ? std::max(c->seq_len[gd.seq0 + 2 * l], c->seq_len[gd.seq0 + 2 * l + 1]) : gd.ncols;
                if (len == 0) continue;                 // empty pair: scores stay 0
                LaneItem li{};
                li.db = gd.db; li.lane = l; li.half = 0; li.ncols = (len + kChunkCols - 1) / kChunkCols * kChunkCols;
                li.slot_a = gd.seq0 + 2 * l; li.slot_b = gd.seq0 + 2 * l + 1;
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
    p.wave_out = qp.stack ? c->d_wave_out.p : nullptr;
    p.seam_mask = qp.seam_mask;
    p.wave_tab = qp.wave_tab;
}

// columns of boundary rows (64 lanes x 8 B each) the pass-boundary buffer may hold
uint64_t bnd_budget_cols(const swimm_hip_ctx *c) { return ((uint64_t)c->opt_bnd_mib << 20) / (64 * sizeof(uint2)); }

// Cuts the longest-first item list into runs whose boundary rows fit the budget (always at least one item).  A
// multi-pass query takes every run through all its passes before the next run starts, so the buffer holds one
// run's columns only: 4x the run's tiled residue bytes instead of 4x the whole database.
void boundary_segments(const swimm_hip_ctx *c, const Plan &pl, std::vector<std::pair<uint32_t, uint32_t>> &segs, uint64_t *max_cols)
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
bool use_split(const swimm_hip_ctx *c, const QueryPlan &qp, const Plan &pl, size_t n_segs)
{
    // (two passes gain nothing: measured -0.3 % on c2; neither do long passes, whose end is a small part of them: c2 with
    // 3 passes of 9 ms each 27.15 ms split, 26.97 ms not -- the split is for passes of up to ~5 ms at 8 000 GCUPS)
    const double pass_cells = (double)pl.total_chunks * kChunkCols * kGroupSeqs * qp.T * qp.W;
    return c->opt_dynamic && qp.passes > 2 && n_segs == 1 && pl.n_wg >= 2 && pl.split_n[1] >= (uint32_t)pl.n_wg && pass_cells < 4e10;
}


}  // namespace swimm_impl
