// search.cpp -- one search on one device: the launches of a pass / a group-resident batch / the lane-systolic
// passes, and the phases of SearchRun (ranges, launch plans, profiles, buffers, issue, promotion ladder, drain).
#include "swimm_impl.h"

namespace swimm_impl {

// measurement aid: the launch's own duration, on the stream it runs on (what a kernel trace reports per dispatch)
int timed_launch(swimm_hip_ctx *c, Mode mode, int T, int W, int n_wg, const PipeParams &p, hipStream_t st)
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
uint64_t resident_bnd_elems(const Plan &pl) { return pl.n_items ? (uint64_t)pl.n_wg * pl.queue_cols[0] * 64 : 0; }

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
    p.wave_out = c->d_wave_out.p;
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
            fprintf(stderr, "stamps (resident batch of %u) wave %2d: load/wait %8.0f (item start %6.0f | residues, boundary, ring %6.0f | next-chunk request %6.0f)  compute %8.0f  tail %8.0f  barrier %8.0f  cycles per active step (%llu active steps per wg)\n", nq,
                    w, (double)h[w * 8 + 0] / h[w * 8 + 4], (double)h[w * 8 + 5] / h[w * 8 + 4], (double)h[w * 8 + 6] / h[w * 8 + 4], (double)h[w * 8 + 7] / h[w * 8 + 4],
                    (double)h[w * 8 + 1] / h[w * 8 + 4], (double)h[w * 8 + 2] / h[w * 8 + 4], (double)h[w * 8 + 3] / h[w * 8 + 4], h[w * 8 + 4] / n_wg);
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
                    fprintf(stderr, "stamps wave %2d: load/wait %8.0f (item start %6.0f | residues, boundary, ring %6.0f | next-chunk request %6.0f)  compute %8.0f  tail %8.0f  barrier %8.0f  cycles per active step (%llu active steps per wg)\n",
                            w, (double)h[w * 8 + 0] / h[w * 8 + 4], (double)h[w * 8 + 5] / h[w * 8 + 4], (double)h[w * 8 + 6] / h[w * 8 + 4], (double)h[w * 8 + 7] / h[w * 8 + 4],
                            (double)h[w * 8 + 1] / h[w * 8 + 4], (double)h[w * 8 + 2] / h[w * 8 + 4], (double)h[w * 8 + 3] / h[w * 8 + 4], h[w * 8 + 4] / pl.n_wg);
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

// The score-profile path of a query (option "sp_threshold"): one launch per pass of 32 rows over every group of the range.
int run_sp_passes(swimm_hip_ctx *c, const QueryPlan &qp, const Plan &pl, const int8_t *qcodes, int32_t *out_row, hipStream_t st, DevBuf<uint2> &bnd)
{
    if (qp.passes > 1 && bnd.cap < pl.bnd_cols * 64) return fail("internal: boundary buffer too small for the score-profile passes");
    for (int pass = 0; pass < qp.passes; ++pass) {
        if (c->queue_next >= c->d_queue.cap) return fail("pipeline launch cursors exhausted");
        SpParams p{};
        p.items = pl.queue_items.p;
        p.n_items = pl.n_items;
        p.queue = c->d_queue.p + c->queue_next++;
        p.qcodes = qcodes;
        p.r0 = (uint32_t)(pass * kSpRows);
        p.sub16 = c->d_sub16.p;
        p.bnd = bnd.p;
        p.first_pass = pass == 0;
        p.last_pass = pass == qp.passes - 1;
        p.out = out_row;
        p.goe = c->open_gap + c->extend_gap;
        p.ge = c->extend_gap;
        HIP_TRY(launch_sp((int)std::min<uint32_t>(pl.n_items, (uint32_t)c->num_cu * 6), p, st));
        c->launches++;
        c->cells += pl.total_chunks * kChunkCols * (uint64_t)kSpRows * 128;
    }
    return 0;
}

// rows per lane of a query's lane-systolic launch: a query of one pass with up to 128 / 256 rows takes 2 / 4 (the serial walk
// down a lane's rows is the step latency), everything else 8
int lane_rows_for(const swimm_hip_ctx *c, uint32_t m)
{
    (void)c;
    return m <= 128 ? 2 : m <= 256 ? 4 : kLaneRows;
}

// at most this many boundary columns (8 B each, two buffers: 1 GiB + 1 GiB) per lane-systolic launch: a batch that needs more
// is cut into launches that follow each other on the stream (the chains of a few dozen queries side by side already hide
// each other's latency; a single query whose list is longer than this still gets what it needs)
static const uint64_t kLaneBndColsMax = (uint64_t)1 << 27;

// ONE lane-systolic launch for a batch of queries over the same item list (the long-sequence tail of a range, or the
// promotion re-runs of one query): every (query, pass) is a set of workgroups with a work cursor of its own, the passes
// of a query chained item by item through bnd / prog (sw_lane_kernel).  The items are independent chains -- a 35 000-residue
// sequence is 22 ms per query on one wave, whatever else the chip does -- so the queries of a batch go side by side:
// c3's 20 queries spend 22 ms + the skew of the chained passes on that chain, not 20 x 22 ms in turn.
// Never more than one workgroup per CU in total while any query has chained passes: the whole grid becomes resident (a
// pass waits for the one before it), ascending in pass so that producers are dispatched first; 4 waves per workgroup.
int run_lane_batch(swimm_hip_ctx *c, Mode mode, int rows_per_lane, const std::vector<LaneQuery> &qs, const LaneList &ll, hipStream_t st, LaneScratch &sc)
{
    if (qs.empty() || (ll.n == 0 && qs[0].n_items == 0)) return 0;
    const int rows_pass = 64 * rows_per_lane;
    size_t at = 0;
    while (at < qs.size()) {
        // the next run of queries that fits the chip (one workgroup per (query, pass) at least) and the boundary budget
        std::vector<LaneQ> lq;
        uint32_t pass_total = 0, max_passes = 1;
        uint64_t bnd_cols = 0;
        size_t end = at;
        size_t prog_total = 0;
        uint64_t cells = 0;
        for (; end < qs.size(); ++end) {
            const uint32_t passes = (qs[end].m + rows_pass - 1) / rows_pass;
            const uint32_t n_q = qs[end].n_items ? qs[end].n_items : ll.n;
            const uint64_t cols_q = qs[end].n_items ? qs[end].cols : ll.cols;
            if (passes > 255) return fail("query of %u rows needs %u chained lane passes (255 at most)", qs[end].m, passes);
            if ((int)passes > c->num_cu) return fail("query of %u rows needs %u chained passes, more than the %d CUs", qs[end].m, passes, c->num_cu);
            if (passes > 1 && rows_per_lane != kLaneRows) return fail("internal: a multi-pass query in a short-lane launch");
            const uint64_t need = passes > 1 ? cols_q + 64 : 0;
            // (launches with chained passes must be resident as a whole: one workgroup per (query, pass) at least, one per CU at most)
            const uint32_t limit = rows_per_lane == kLaneRows ? (uint32_t)c->num_cu : 4096u;
            if (end > at && (pass_total + passes > limit || bnd_cols + need > kLaneBndColsMax)) break;
            LaneQ d{};
            d.prof_off = qs[end].prof_off; d.out_off = qs[end].out_off; d.bnd0 = bnd_cols; d.prof_stride = qs[end].prof_stride; d.m = qs[end].m;
            d.passes = passes; d.queue0 = pass_total; d.prog0 = (uint32_t)prog_total; d.items0 = qs[end].items0; d.n_items = n_q;
            if (prog_total + (size_t)passes * n_q > 0xFFFFFFF0ull) return fail("lane-systolic launch: more than 2^32 progress counters");
            lq.push_back(d);
            pass_total += passes; bnd_cols += need; max_passes = std::max(max_passes, passes);
            prog_total += (size_t)passes * n_q;
            cells += cols_q * (uint64_t)rows_pass * passes;
        }
        const size_t need_prog = max_passes > 1 ? prog_total : 0;
        if (sc.bnd[0].cap < bnd_cols || sc.bnd[1].cap < bnd_cols || sc.queue.cap < pass_total || sc.prog.cap < need_prog)
            return fail("internal: lane scratch too small (%zu/%llu columns, %zu/%zu counters, %zu/%zu queries)", sc.bnd[0].cap, (unsigned long long)bnd_cols, sc.prog.cap,
                        need_prog, sc.lq.cap, lq.size());
        // workgroups per (query, pass): the chip's budget dealt evenly (every (query, pass) walks the same items); a launch
        // without chained passes may ask for more than the chip holds at once
        const uint32_t budget = max_passes > 1 ? (uint32_t)c->num_cu : (uint32_t)c->num_cu * 6;
        const uint32_t per = std::max<uint32_t>(1, budget / pass_total);
        std::vector<uint32_t> block_map;
        block_map.reserve((size_t)per * pass_total);
        for (uint32_t ps = 0; ps < max_passes; ++ps)
            for (size_t i = 0; i < lq.size(); ++i)
                if (ps < lq[i].passes)
                    for (uint32_t k = 0; k < std::min<uint32_t>(per, (lq[i].n_items + 3) / 4); ++k) block_map.push_back((uint32_t)(i << 8) | ps);
        // (every launch of the search has its own region of the two tables: a scratch serves one range after the other of a
        // database that streams in, and the next range's tables must not overwrite what a launch in flight still reads)
        if (sc.lq.cap < sc.lq_used + lq.size() || sc.block_map.cap < sc.bm_used + block_map.size())
            return fail("internal: lane scratch too small (%zu/%zu queries, %zu/%zu workgroups of all launches)", sc.lq.cap, sc.lq_used + lq.size(), sc.block_map.cap, sc.bm_used + block_map.size());
        LaneQ *const d_lq = sc.lq.p + sc.lq_used;
        uint32_t *const d_bm = sc.block_map.p + sc.bm_used;
        sc.lq_used += lq.size(); sc.bm_used += block_map.size();
        if (list_copy(c, d_lq, lq.data(), lq.size() * sizeof(LaneQ)) || list_copy(c, d_bm, block_map.data(), block_map.size() * sizeof(uint32_t)) || list_sync(c)) return 1;
        LaneParams p{};
        p.items = ll.items.p;
        p.lq = d_lq;
        p.block_map = d_bm;
        p.queue = sc.queue.p;
        p.prog = sc.prog.p;
        p.prof = c->d_prof.p;
        p.bnd[0] = sc.bnd[0].p;
        p.bnd[1] = sc.bnd[1].p;
        p.out = c->d_scores.p;
        p.goe = c->open_gap + c->extend_gap;
        p.ge = c->extend_gap;
        p.err = c->d_err.p;
        HIP_TRY(hipMemsetAsync(sc.queue.p, 0, pass_total * sizeof(uint32_t), st));
        if (max_passes > 1) HIP_TRY(hipMemsetAsync(sc.prog.p, 0, need_prog * sizeof(uint32_t), st));
        HIP_TRY(launch_lane(mode, rows_per_lane, (int)block_map.size(), p, st));
        c->launches++;
        c->cells += cells * (mode == Mode::PK16 ? 2 : 1);
        at = end;

    }
    return 0;
}

// scratch of a lane-systolic launch of `queries` queries with `pass_total` passes in all over `items` items; bnd_cols:
// boundary columns of its multi-pass queries together (each: the list's columns + 64)
int reserve_lane_scratch(swimm_hip_ctx *c, LaneScratch &sc, size_t list_cols, size_t items, size_t pass_total, size_t queries, size_t multi_pass_queries, size_t launches)
{
    // (`launches`: how often the scratch is used in one search -- once per range; a batch that exceeds the chip is cut further)
    launches = std::max<size_t>(1, launches) + pass_total / std::max(1, c->num_cu) + queries / 4096 + 1;
    sc.lq_used = sc.bm_used = 0;
    size_t bnd_cols = multi_pass_queries * (list_cols + 64);
    bnd_cols = std::min<size_t>(bnd_cols, std::max<size_t>(kLaneBndColsMax, list_cols + 64));
    pass_total = std::min<size_t>(pass_total, 4096 + 255);
    queries = std::min<size_t>(queries, 4096);
    HIP_TRY(sc.queue.reserve(std::max<size_t>(256, pass_total)));
    if (bnd_cols) {
        HIP_TRY(sc.bnd[0].reserve(bnd_cols));
        HIP_TRY(sc.bnd[1].reserve(bnd_cols));
    }
    HIP_TRY(sc.prog.reserve(std::max<size_t>(1, multi_pass_queries ? items * std::min<size_t>(pass_total, (size_t)c->num_cu + 255) : 1)));
    HIP_TRY(sc.lq.reserve(std::max<size_t>(1, queries) * launches));
    HIP_TRY(sc.block_map.reserve(((size_t)c->num_cu * 6 + 4096) * launches));
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
    std::vector<std::pair<size_t, size_t>> range_chunks;     // streaming: positions [first, last) in `up_order` of every range's parts
    std::vector<UploadPart> up_order;                        // streaming: the chunks (the first one in parts) in the order they travel
    std::vector<std::map<int, DbPlan>> stream_plans;         // streaming: work lists per (range, workgroup count), released when the search has drained
    // the launch plan
    Mode main_mode = Mode::F16;
    int f16_thr = 2048;                     // binary16 first tier: results >= this are re-run as packed int16 (f16_exact_below)
    bool lane_room = false, many_short = false, alternate = false;
    bool cut_room = false;                  // outlier pairs exist: the queries of FEW passes leave the lane-systolic waves their registers too
    std::vector<uint8_t> use_sp;            // queries that run through the score-profile kernel (option "sp_threshold")
    std::vector<size_t> qcode_off;          // ... and where their padded residue codes start in d_qcodes
    uint32_t longest_cols = 0;
    std::vector<QueryPlan> qps;
    std::vector<uint8_t> rotated;                            // one-pass queries that run whole (no tail kernel) on three streams in rotation
    std::vector<uint8_t> in_batch;                           // queries of the group-resident batch launches
    std::vector<QueryPlan> one_plan = std::vector<QueryPlan>(1);
    // Stacks: short one-pass queries of a batch that share workgroups -- the members' strips one after the other along the
    // waves of one 4-wave workgroup, every member padded to whole strips, a zero boundary at every seam, one score row per
    // member (sw_kernels.h, QDesc).  A 40-row query alone fills 40 of a 4 x 12-row workgroup's 48 rows at that shape's 7 340
    // GCUPS, or 40 of 2 x 20 at 5 890; two of them stacked fill 80 of 4 x 20 rows at 8 260.
    struct Stack { int T = 0, W = 4; bool resident = false; size_t prof_off = 0; uint32_t rows = 0, seam_mask = 0, wave_tab = 0;
                   std::vector<uint32_t> q, strip0, nstrips; };
    std::vector<Stack> stacks;
    std::vector<int> stack_of;                               // per query: its stack, or -1
    struct Unit { int stack; uint32_t q; };                  // an entry of a group-resident launch's query table: a stack, or a query
    std::map<std::pair<int, int>, std::vector<Unit>> by_shape;
    std::vector<std::vector<QueryPlan>> rqps;                // streaming, per-pass launches: a launch shape per (range, query)
    // Streaming, ONE query that fits one pass: ONE pipeline launch over the whole database, started before the first byte has
    // arrived.  Its item list holds every group in the order the parts travel, and the launch's workgroups pull from it as far
    // as it has landed (sw_pipe_kernel, PipeParams::avail; the uploader publishes the count behind every part).  No ranges, no
    // launch per range: a range of the longest sequences is a handful of 6 ms chains that the workgroups of the NEXT range's
    // launch could only cover where a whole workgroup had ended -- here the same workgroups simply go on with what arrives.
    bool one_list = false, one_list_launched = false;
    bool stalled = false;                   // ... its workgroups gave up waiting for the database (drain): the caller searches the resident copy
    QueryPlan one_list_qp;
    int one_list_wg = 0;
    uint32_t one_list_items = 0;
    uint64_t one_list_chunks = 0;
    std::vector<uint8_t> range_pp;          // streaming: ranges that run like a resident database (launch shapes per query, one launch per pass, tail kernels)
    bool alternate_pp = false;              // ... whose multi-pass queries take turns on the two bulk streams
    size_t prof_elems = 0;
    int tail_lanes = 1;                     // lane-systolic tail launches per range: one per rows-per-lane class in use (informational)
    double stream_free[3] = {0, 0, 0};      // streaming, group-resident ranges: when each of the three streams is expected to have drained (s after t_begin)

    SearchRun(swimm_hip_ctx *ctx, uint32_t b, uint32_t e) : c(ctx), qb(b), qe(e) {}
    ~SearchRun()
    {
        if (wait_before_issue && c->up) { c->up->finish(true); (void)hipDeviceSynchronize(); }      // (an early return before the wait)
        if (!streaming) return;
        if (c->up) c->up->finish(true);          // (an early return: the chunks not yet copied stay where they are)
        if (one_list_launched && c->up && (c->up->failed || c->up->issued < up_order.size()))
            (void)launch_publish_items(c->d_avail.p, kAvailAbort, c->stream_up);   // (its launch must not wait for what will not come)
        (void)hipDeviceSynchronize();
        release_stream_plans();
    }
    void release_stream_plans()
    {
        for (auto &m : stream_plans) for (auto &kv : m) { kv.second.main.release(); kv.second.tail.release(); }
        stream_plans.clear();
    }
    bool wait_before_issue = false;         // a database that is waited for as a whole (layout_ranges): the wait is still to come
    int wait_for_the_database()             // every part handed to the device, the launch streams behind the last part's tiling kernel
    {
        if (wait_uploaded(up_order.size())) return 1;
        const UploadPart &last = up_order.back();            // (the upload stream is in order)
        for (hipStream_t st : {c->stream, c->stream_b, c->stream2, c->stream3}) HIP_TRY(hipStreamWaitEvent(st, last.ready, 0));
        if (dbg) fprintf(stderr, "swimm_hip: host copies done %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
        c->up->finish(false);
        wait_before_issue = false;
        return 0;
    }
    int wait_uploaded(size_t n)             // until the first n chunks of `up_order` are on their way
    {
        std::string err;
        if (c->up->wait_issued(n, &err)) return fail("%s", err.empty() ? "upload failed" : err.c_str());
        return 0;
    }
    const QueryPlan &qp_of(size_t ri, uint32_t q) const { return ri < rqps.size() && !rqps[ri].empty() ? rqps[ri][q] : qps[q]; }
    bool pp_range(size_t ri) const { return ri < range_pp.size() && range_pp[ri] != 0; }
    bool res_of(size_t ri, uint32_t q) const { return in_batch[q] != 0 && !pp_range(ri); }      // the query runs in range ri's group-resident batch launch
    int issue_one_list();
    int plan_of(size_t ri, uint32_t q, DbPlan **out);
    int plan_for(size_t ri, int T, int W, bool resident, bool whole_db, DbPlan **out);
    int build_stacks();

    int begin(uint64_t *slots_out);
    int layout_ranges();
    int plan_queries();
    int upload_profiles();
    int size_buffers();
    int issue();
    int promotion_ladder();
    int ladder_rung(const std::vector<uint32_t> &batch, int thr, Mode mode);
    int drain();
    int run(uint64_t *slots_out)
    {
        return begin(slots_out) || layout_ranges() || plan_queries() || upload_profiles() || size_buffers() || issue() || promotion_ladder() || drain();
    }
};

// the work lists of a (range, launch shape): cached for the resident database, temporary for a streaming chunk
int SearchRun::plan_for(size_t ri, int T, int W, bool resident, bool whole_db, DbPlan **out)
{
    int per_cu = 1;
    if (wgs_per_cu(c, main_mode, T, W, resident, &per_cu)) return 1;
    const int n_wg = n_workgroups(c, per_cu);
    if (!streaming) return get_db_plan(c, main_mode, n_wg, whole_db, out);
    const bool no_tail = resident || whole_db;         // (every group through the launch: no lane-systolic tail beside it)
    const int key = n_wg * 2 + (no_tail ? 1 : 0);
    auto it = stream_plans[ri].find(key);
    if (it == stream_plans[ri].end()) {
        DbPlan &dp = stream_plans[ri][key];
        bool exact = true;
        for (size_t ci = range_chunks[ri].first; ci < range_chunks[ri].second; ++ci) exact = exact && c->chunks[up_order[ci].chunk].lens_known;
        g_list_arena = &c->list_arena;       // (no hipMalloc while kernels run: the lists come out of the arena sized in layout_ranges)
        const int rc = make_db_plan(c, main_mode, n_wg, no_tail, ranges[ri], exact, dp);
        g_list_arena = nullptr;
        if (rc) return 1;
        *out = &dp;
    } else {
        *out = &it->second;
    }
    return 0;
}

int SearchRun::plan_of(size_t ri, uint32_t q, DbPlan **out)
{
    return plan_for(ri, qp_of(ri, q).T, qp_of(ri, q).W, res_of(ri, q), rotated[q] != 0 || use_sp[q] != 0 || res_of(ri, q), out);
}

// Which short queries share workgroups.  Candidates: the one-pass queries of up to 72 rows that run without a tail kernel
// anyway (members of the group-resident batch, or in rotation).  Every candidate takes s = 1 or 2 strips of the height T that
// costs it least (s x T rows at the measured rate of the 4 x T shape); the candidates of one height fill bins of four strips,
// largest first; a bin that does not fill up is dissolved (its queries run as before).
int SearchRun::build_stacks()
{
    stack_of.assign(qn, -1);
    stacks.clear();
    if (!c->opt_stack || !c->opt_dynamic || c->opt_T || c->opt_W || c->opt_maxW) return 0;   // (a forced launch shape is honoured query by query)
    const int W = 4;
    for (int pass = 0; pass < 2; ++pass) {               // the batch's members, then the rotating ones
        std::map<int, std::vector<std::pair<int, uint32_t>>> cls;     // T -> (strips, query)
        for (uint32_t q = 0; q < qn; ++q) {
            if (!(pass == 0 ? in_batch[q] != 0 : rotated[q] != 0) || qm[q] > 72 || use_sp[q]) continue;
            int bt = 0, bs = 0;
            double bc = 0;
            for (int s = 1; s <= 2; ++s)
                for (int T = 8; T <= 36; T += 4) {
                    if (!pipe_has_variant(main_mode, T) || s * T < (int)qm[q]) continue;
                    const double cost = (double)(s * T) / shape_gcups(T, W);
                    if (!bt || cost < bc) { bt = T; bs = s; bc = cost; }
                    break;                                   // (the lowest T that holds the query with s strips; taller only wastes rows)
                }
            if (bt) cls[bt].push_back({bs, q});
        }
        for (auto &kv : cls) {
            auto &v = kv.second;
            std::stable_sort(v.begin(), v.end(), [](const std::pair<int, uint32_t> &a, const std::pair<int, uint32_t> &b) { return a.first > b.first; });
            std::vector<Stack> bins;
            std::vector<int> used;
            for (auto &e : v) {
                size_t b = 0;
                while (b < bins.size() && used[b] + e.first > W) ++b;
                if (b == bins.size()) { bins.emplace_back(); used.push_back(0); bins[b].T = kv.first; bins[b].W = W; bins[b].resident = pass == 0; }
                bins[b].q.push_back(e.second); bins[b].strip0.push_back((uint32_t)used[b]); bins[b].nstrips.push_back((uint32_t)e.first);
                bins[b].seam_mask |= 1u << used[b];
                used[b] += e.first;
            }
            for (size_t b = 0; b < bins.size(); ++b) {
                if (used[b] < W || bins[b].q.size() < 2) continue;      // not full, or one query alone: as before
                bins[b].rows = (uint32_t)(W * bins[b].T);
                for (uint32_t q : bins[b].q) stack_of[q] = (int)stacks.size();
                stacks.push_back(std::move(bins[b]));
            }
        }
    }
    if (dbg) {
        size_t members = 0;
        for (const Stack &st : stacks) members += st.q.size();
        fprintf(stderr, "swimm_hip: %zu short queries stacked into %zu workgroup shapes' worth of strips\n", members, stacks.size());
    }
    return 0;
}

int SearchRun::begin(uint64_t *slots_out)
{
    if (!c->have_queries) return fail("swimm_hip_search: no queries set");
    if (c->groups.empty()) return fail("swimm_hip_search: no database chunk resident");
    CHECK_DEVICE(c);
    if (refresh_plans(c)) return 1;
    qn = qe - qb;
    dbg = getenv("SWIMM_HIP_DEBUG") != nullptr;
    t_begin = now_s();
    qm = c->qm.data() + qb;
    qdisp = c->qdisp.data() + qb;
    S = (uint64_t)c->groups.size() * kGroupSeqs;
    *slots_out = S;
    c->tiling_room = false;
    // (binary16 first tier: results below f16_exact_below(extend) are exact -- the pipeline kernel's column offsets take up to 127 of the
    // 2048; with an extend penalty beyond 237 the tier would be exact below 1 100 only, and the int16 tier is the first)
    main_mode = c->opt_force_i32 ? Mode::I32 : (c->opt_f16 && f16_exact_below(c->extend_gap) >= 1100 ? Mode::F16 : Mode::PK16);
    f16_thr = f16_exact_below(c->extend_gap);
    return 0;
}

int SearchRun::layout_ranges()
{
    // Chunks whose bytes are still on the host (option "lazy_upload"): this search streams them in -- chunk k+1 is
    // copied and tiled on the upload stream while chunk k is being aligned (X2 overlapped with compute,
    // MICsearch.c:85-91).  Otherwise the whole resident database is one range with cached work lists.
    streaming = false;
    for (const ChunkRec &r : c->chunks) streaming = streaming || !r.uploaded;
    c->streaming_now = streaming;
    c->batch_now = false;
    if (!streaming) {
        if (sync_lengths(c)) return 1;
        ranges.push_back(whole_range(c));
        return 0;
    }
    // FIRST the order in which the database travels, and the uploader (a thread of its own, Uploader) on its way: everything
    // else this function and the next ones work out -- which parts form a range, launch shapes, profiles, buffers -- happens
    // while the link is already busy.  (Round 3 laid the ranges out first: 10 ms on a 6.6 M-sequence database, 14 ms on 8.9 M,
    // before the first byte moved.)
    {
        uint64_t mb = 16, mn = 1, mg = 1, mo = 1;
        for (const ChunkRec &r : c->chunks) {
            if (r.uploaded) continue;
            mb = std::max<uint64_t>(mb, r.kind == 0 ? r.vD : r.code_bytes); mn = std::max<uint64_t>(mn, r.group_count);
            mg = std::max<uint64_t>(mg, r.n_groups); mo = std::max<uint64_t>(mo, r.gsrc.size());
            if (r.kind == 1) mn = std::max<uint64_t>(mn, r.n_seq);
        }
        // the upload scratch grows now, not between two chunks (growing frees the old buffer)
        HIP_TRY(c->up_b.reserve(mb)); HIP_TRY(c->up_n.reserve(mn)); HIP_TRY(c->up_disp.reserve(mn));
        HIP_TRY(c->up_gcols.reserve(mg)); HIP_TRY(c->up_goff.reserve(mg)); HIP_TRY(c->up_off.reserve(mo));
    }
    // The end of the database with the LONGER sequences travels first: the few long chains it holds start early and are
    // covered by everything that follows, and what arrives last -- what nothing covers -- is short sequences, whose work ends evenly.
    const size_t nc = c->chunks.size();
    const bool descending = nc > 1 && (double)c->chunks[nc - 1].cols / std::max<uint32_t>(1, c->chunks[nc - 1].n_groups) >
                                          (double)c->chunks[0].cols / std::max<uint32_t>(1, c->chunks[0].n_groups);
    // The chunk that travels first goes in parts -- 16 MiB, 32 MiB, the rest -- so that the first launch has its data
    // after 0.4 ms instead of the 2 ms a whole 96 MiB chunk takes on the link.  And which chunk is that?  Not the one with the
    // longest sequences: its groups are few (a 96 MiB chunk of 5 000-residue sequences is 26 groups per 16 MiB) and each is a
    // chain of many milliseconds (measured in round 3: 33.5 ms against 31.0 without parts).  The chunk with the SHORTEST
    // sequences goes first -- thousands of groups per part: the chip is full 0.4 ms after the call -- then the database in
    // descending order: the long chains start with the second chunk, 4 ms in.
    std::vector<size_t> chunk_order;
    for (size_t i = 0; i < nc; ++i) chunk_order.push_back(descending ? nc - 1 - i : i);
    const bool short_first = descending && nc >= 3 && qn <= 4;
    if (short_first) { chunk_order.insert(chunk_order.begin(), chunk_order.back()); chunk_order.pop_back(); }
    size_t n_part_ev = 0;
    for (size_t i = 0; i < nc; ++i) {
        const size_t ci = chunk_order[i];
        const ChunkRec &r = c->chunks[ci];
        const uint64_t bytes = r.kind == 0 ? r.vD : r.code_bytes;
        bool ascending = true;
        if (r.kind == 0 && !r.uploaded)
            for (uint32_t v = 1; v < r.group_count; ++v) ascending = ascending && r.h_disp[v] >= r.h_disp[v - 1];
        if (i == 0 && short_first && !r.uploaded && r.groups_uploaded == 0 && ascending && bytes >= ((uint64_t)56 << 20) && r.n_groups >= 16) {
            const uint64_t heads[2] = {(uint64_t)16 << 20, (uint64_t)32 << 20};
            uint32_t edge = descending ? r.n_groups : 0;      // the groups still to be dealt: [0, edge) or [edge, n)
            for (int h = 0; h < 2; ++h) {
                uint64_t acc = 0;
                uint32_t cut = edge;
                if (descending) { while (cut > 8 && acc < heads[h]) { --cut; acc += (uint64_t)r.gcols[cut] * kGroupSeqs; } }
                else { while (cut + 8 < r.n_groups && acc < heads[h]) { acc += (uint64_t)r.gcols[cut] * kGroupSeqs; ++cut; } }
                while (c->part_ev.size() <= n_part_ev) {
                    hipEvent_t e;
                    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->part_ev.push_back(e);
                }
                UploadPart pt; pt.chunk = ci; pt.g0 = descending ? cut : edge; pt.g1 = descending ? edge : cut; pt.ready = c->part_ev[n_part_ev++];
                up_order.push_back(pt);
                edge = cut;
            }
            UploadPart rest; rest.chunk = ci; rest.g0 = descending ? 0 : edge; rest.g1 = descending ? edge : r.n_groups; rest.ready = r.ready;
            up_order.push_back(rest);
        } else {
            UploadPart pt; pt.chunk = ci; pt.g0 = 0; pt.g1 = r.n_groups; pt.ready = r.ready;
            up_order.push_back(pt);
        }
    }
    // ONE query that one pass of one workgroup shape holds: one launch over one item list that grows as the parts land.
    one_list = false;
    QueryPlan one_qp{};
    if (qn == 1 && c->opt_dynamic && main_mode == Mode::F16 && c->opt_resident != 1 && (int)qm[0] < c->opt_sp_threshold && c->groups.size() < 0x7FFFFFF0ull &&
        choose_one_list_plan(c, qm[0], &one_qp, &one_list_wg) == 0) {
        one_list = true;
        one_list_qp = one_qp;
        // A part costs that launch nothing but a publication (no launch per range any more), so the parts are FINE: the first
        // ones 4, 8, 16 MiB -- the workgroups have their first items 0.2 ms after the call -- then at most 32 MiB each: what has
        // crossed the link is aligned a millisecond later, not when the rest of a 96 MiB chunk has followed.
        std::vector<UploadPart> fine;
        uint64_t next_cap = (uint64_t)4 << 20;
        for (const UploadPart &pt : up_order) {
            const ChunkRec &r = c->chunks[pt.chunk];
            bool ascending = !r.uploaded && r.groups_uploaded == 0;
            if (r.kind == 0 && ascending)
                for (uint32_t v = 1; v < r.group_count; ++v) ascending = ascending && r.h_disp[v] >= r.h_disp[v - 1];
            if (!ascending) { fine.push_back(pt); continue; }          // (a caller's layout whose groups do not lie in order travels whole)
            // in the direction the database travels: a chunk's own groups longest first when it descends
            uint32_t lo = pt.g0, hi = pt.g1;
            while (lo < hi) {
                uint64_t acc = 0;
                uint32_t a = lo, b = hi;
                if (descending) { a = hi; while (a > lo && (acc < next_cap || a - lo < 8)) { --a; acc += (uint64_t)r.gcols[a] * kGroupSeqs; } b = hi; hi = a; }
                else { b = lo; while (b < hi && (acc < next_cap || hi - b < 8)) { acc += (uint64_t)r.gcols[b] * kGroupSeqs; ++b; } a = lo; lo = b; }
                UploadPart sub = pt;
                sub.g0 = a; sub.g1 = b;
                sub.ready = (descending ? a == pt.g0 : b == pt.g1) ? pt.ready : nullptr;      // (the chunk's / head part's event marks its LAST sub-part)
                if (!sub.ready) {
                    while (c->part_ev.size() <= n_part_ev) {
                        hipEvent_t e;
                        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                        c->part_ev.push_back(e);
                    }
                    sub.ready = c->part_ev[n_part_ev++];
                }
                fine.push_back(sub);
                next_cap = std::min<uint64_t>(next_cap * 2, (uint64_t)32 << 20);
            }
        }
        up_order.swap(fine);
        std::vector<Item> items;
        items.reserve(c->groups.size());
        std::vector<uint32_t> order;
        one_list_chunks = 0;
        for (UploadPart &pt : up_order) {
            const ChunkRec &r = c->chunks[pt.chunk];
            order.resize(pt.g1 - pt.g0);
            for (uint32_t g = pt.g0; g < pt.g1; ++g) order[g - pt.g0] = r.group0 + g;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c->groups[a].ncols > c->groups[b].ncols; });    // longest first within what arrives together
            for (uint32_t g : order) {
                const GroupDesc &gd = c->groups[g];
                Item it{}; it.db = gd.db; it.ncols = gd.ncols; it.seq0 = gd.seq0; it.half = 0; it.out_slot = 0; it.bnd_off = 0;
                items.push_back(it);
                one_list_chunks += gd.ncols / kChunkCols;
            }
            pt.publish = (uint32_t)items.size();
        }
        one_list_items = (uint32_t)items.size();
        HIP_TRY(c->d_stream_items.reserve(items.size()));
        HIP_TRY(c->d_avail.reserve(4));
        // the count starts at zero -- on the UPLOAD stream, ahead of everything the uploader will put there
        HIP_TRY(hipMemsetAsync(c->d_avail.p, 0, 4 * sizeof(uint32_t), c->stream_up));
        if (!c->ev_avail) HIP_TRY(hipEventCreateWithFlags(&c->ev_avail, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_avail, c->stream_up));          // (the launch must not read the count the previous search left there)
        if (list_copy(c, c->d_stream_items.p, items.data(), items.size() * sizeof(Item))) return 1;     // (waited for before the launch, issue())
    } else {
        (void)hipGetLastError();
        g_err.clear();
    }
    const size_t np = up_order.size();
    if (ensure_uploader(c)) return 1;
    c->up->post(up_order);
    if (dbg) fprintf(stderr, "swimm_hip: upload order laid out (%zu parts%s), uploader started %.3f ms after the call began; the add calls before it took %.3f ms\n", np, one_list ? ", one item list" : "", (now_s() - t_begin) * 1e3, c->add_seconds * 1e3);
    c->add_seconds = 0;
    if (one_list) {
        ranges.push_back(whole_range(c));
        range_chunks.push_back({0, np});
        stream_plans.resize(1);
        return 0;
    }
    // Consecutive parts form a range, and every range is as large as it can be without the GPU running dry before
    // it has arrived: the link delivers a chunk in bytes / 40 GB/s, the kernels consume it in (rows of all queries) x
    // residues / 8 000 GCUPS; a batch of long queries is compute-bound from the first chunk on and runs as two ranges.
    c->stream_tail.clear();
    c->stream_tail = pick_tail(c, whole_range(c));
    double rows = 0;
    for (uint32_t q = 0; q < qn; ++q) rows += qm[q];
    auto part_cols = [&](const UploadPart &pt) { uint64_t x = 0; const ChunkRec &r = c->chunks[pt.chunk]; for (uint32_t g = pt.g0; g < pt.g1; ++g) x += r.gcols[g]; return x; };
    std::vector<uint64_t> pcols(np);
    for (size_t i = 0; i < np; ++i) pcols[i] = part_cols(up_order[i]);
    auto up_s = [&](size_t i) { return (double)pcols[i] * kGroupSeqs / 40e9; };
    auto dp_s = [&](size_t i) { return 0.85 * rows * (double)pcols[i] * kGroupSeqs / 8000e9; };   // (rather too short: the GPU must not wait)
    // ... and no larger than it must be: the ranges that run while the database lands take launch shapes that leave the tiling
    // waves their registers (plan_queries), and the last range -- everything that is left once the GPU has work until the link has
    // delivered the last byte -- runs like a resident database.  So a range closes as soon as the ranges so far keep the GPU busy
    // for 1.1 of the whole upload, and what follows is the last range: a 5 478-row query needs 4 % of a 7e9-residue database before
    // the other 96 % are there (first built without this rule: a third of the database in a register-lean early range, +4.9 %).
    double up_total = 0, dp_total = 0;
    for (size_t i = 0; i < np; ++i) { up_total += up_s(i); dp_total += dp_s(i); }
    // A database that lands in under 1.5 % of the time its alignment takes is not streamed at all: the search waits for the
    // uploader and runs as on a resident database -- the same launches, work lists and streams (c3's 20 queries: 5 ms of upload
    // before 840 ms of kernels; as two ranges the first -- an eighth of the database in 108 small launches of register-lean
    // shapes -- cost 6 %; c5: 40 ms before 7.2 s).
    if (up_total <= 0.015 * dp_total) {
        if (dbg) fprintf(stderr, "swimm_hip: the database lands in %.1f ms, its alignment takes %.0f ms: waiting for it, then searching it as a resident one\n", up_total * 1e3, dp_total * 1e3);
        // (.seq slabs carry their sequences' lengths: launch shapes, profiles and work lists are made while the database lands and the
        // wait comes right before the first launch, issue(); the chunk layout's true lengths come from the re-tile kernel: wait now)
        wait_before_issue = true;
        for (const ChunkRec &r : c->chunks) wait_before_issue = wait_before_issue && (r.uploaded || r.lens_known);
        if (!wait_before_issue && wait_for_the_database()) return 1;
        streaming = false;
        c->streaming_now = false;
        c->stream_tail.clear();
        if (sync_lengths(c)) return 1;
        ranges.push_back(whole_range(c));
        return 0;
    }
    double t_up = 0, t_gpu = 0;
    for (size_t i = 0; i < np;) {
        Range rg; rg.g0 = c->chunks[up_order[i].chunk].group0 + up_order[i].g0; rg.g1 = c->chunks[up_order[i].chunk].group0 + up_order[i].g1; rg.cols = 0;
        const size_t first = i;
        double work = 0;
        do {
            const UploadPart &pt = up_order[i];
            const ChunkRec &r = c->chunks[pt.chunk];
            rg.g0 = std::min(rg.g0, r.group0 + pt.g0); rg.g1 = std::max(rg.g1, r.group0 + pt.g1); rg.cols += pcols[i];
            t_up += up_s(i); work += dp_s(i);
            ++i;
            if (i < np) {                              // (a range is a run of consecutive device groups)
                const uint32_t n0 = c->chunks[up_order[i].chunk].group0 + up_order[i].g0, n1 = c->chunks[up_order[i].chunk].group0 + up_order[i].g1;
                if (n0 != rg.g1 && n1 != rg.g0) break;
            }
        } while (i < np && t_up + up_s(i) <= t_gpu && (t_gpu >= 1.1 * up_total || std::max(t_gpu, t_up) + work < 1.1 * up_total));
        t_gpu = std::max(t_gpu, t_up) + work;
        ranges.push_back(rg);
        range_chunks.push_back({first, i});
    }
    stream_plans.resize(ranges.size());
    {   // the arena the ranges' work lists are carved from: per range and launch-shape class three item lists (32 B per group each)
        // and the workgroup tables; a handful of classes per range (distinct workgroup counts x with / without tail)
        const size_t classes = std::min<size_t>(qn, 6) + 2;
        size_t need = (size_t)c->groups.size() * 3 * sizeof(Item) * classes + ranges.size() * classes * ((size_t)c->num_cu * 4 * 8 + 4096) + ((size_t)4 << 20);
        need += (size_t)c->groups.size() / 8 * 64 * sizeof(LaneItem);           // lane-systolic tails: a minority of groups, 64 pairs each
        if (c->list_arena.cap < need) {
            if (c->list_arena.base) HIP_TRY(hipFree(c->list_arena.base));
            c->list_arena = DevArena{};
            HIP_TRY(hipMalloc((void **)&c->list_arena.base, need));
            c->list_arena.cap = need;
        }
        c->list_arena.used = 0;
        // ... and the pinned buffer the lists travel through (list_copy) holds the largest single list, now, not by growing mid-search
        const size_t pin_need = (size_t)c->groups.size() * sizeof(Item) + ((size_t)1 << 20);
        if (c->pin_cap < pin_need) {
            HIP_TRY(hipStreamSynchronize(list_stream(c)));
            if (c->pin) { HIP_TRY(hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
            HIP_TRY(hipHostMalloc(&c->pin, pin_need, hipHostMallocDefault));
            c->pin_cap = pin_need;
            c->pin_used = 0;
        }
    }
    return 0;
}

int SearchRun::plan_queries()
{
    // query profiles prof[q][d][row] = submat[query[row]*32 + d] (queryProfiles, MICsearch.c:34-36,
    // transposed so that consecutive query rows are contiguous for one residue code); rows past the
    // query's end are zero, like the reference's dummy row 23
    range_pp.assign(ranges.size(), 0);
    alternate_pp = false;
    const bool landing = streaming;
    c->tiling_room = landing;              // launch shapes of a database that is still landing leave the tiling waves their registers
    if (one_list) {          // one query, one pass, one launch over the whole database as it lands (layout_ranges)
        qps.assign(1, one_list_qp);
        rotated.assign(1, 0); use_sp.assign(1, 0); in_batch.assign(1, 0); stack_of.assign(1, -1);
        stacks.clear();
        lane_room = cut_room = many_short = alternate = false;
        c->batch_now = false;
        c->tiling_room = false;
        const uint32_t lane_rows = (uint32_t)((qm[0] + 64 * kLaneRows - 1) / (64 * kLaneRows) * (64 * kLaneRows));     // (the promotion re-runs read the same profile)
        qps[0].mpad = std::max(qps[0].mpad, lane_rows);
        qps[0].prof_off = 0;
        prof_elems = (size_t)kCodes * qps[0].mpad;
        if (dbg) fprintf(stderr, "swimm_hip: query 0 m=%u -> one launch of %d x %d rows over the %u groups as they land\n", qm[0], qps[0].W, qps[0].T, one_list_items);
        return 0;
    }
    // a database with a long-sequence tail is searched with launch shapes that leave room for lane-systolic waves
    lane_room = false;
    longest_cols = 0;
    for (const GroupDesc &g : c->groups) longest_cols = std::max(longest_cols, g.ncols);
    if (c->opt_tail_mode != 2 && main_mode != Mode::I32)
        lane_room = c->opt_tail_mode == 1 || (double)longest_cols > c->opt_tail_frac * 0.01 * (double)c->total_cols / c->num_cu;
    if (c->opt_tail_mode != 2 && main_mode != Mode::I32) {          // outlier pairs run through the lane-systolic kernel as well
        ensure_cuts(c);
        // Outlier pairs are a handful of lane-systolic items.  A query of many passes needs no room kept for them: its launches end
        // pass after pass and the lane waves move in at the first boundary, their chain running beside the passes that follow
        // (c4 at 19 % of its size: 9 990 GCUPS with the 4 x 36 shape that keeps the room, 10 540 with 4 x 32 that does not).  A
        // query of one or two passes would see that chain only start when its own kernels end: those keep the room.
        cut_room = false;
        for (size_t g = 0; g < c->cut_lane.size() && !cut_room; ++g) cut_room = c->cut_lane[g] < 64;
    }
    if (dbg) fprintf(stderr, "swimm_hip: ranges laid out, uploader started %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    qps.assign(qn, QueryPlan{});
    rotated.assign(qn, 0);
    // the reference's adaptive profile (MICsearch.c:39-43): queries of at least query_length_threshold rows take the score profile
    use_sp.assign(qn, 0);
    for (uint32_t q = 0; q < qn; ++q) use_sp[q] = main_mode == Mode::F16 && c->opt_dynamic && (int)qm[q] >= c->opt_sp_threshold;
    uint32_t n_short = 0;
    for (uint32_t q = 0; q < qn; ++q) n_short += qm[q] <= 64 * kLaneRows;
    // (with a handful of short queries the last ones' chains would stick out at the end of the search; and a database
    // with an extreme sequence -- c3's 35 000 residues are 6x a CU's mean load -- keeps the tail kernel, whose chain
    // is 3.6x faster per column than a 4-wave workgroup's)
    many_short = n_short >= 8 && !streaming;
    const bool rotate = many_short && (double)longest_cols <= 2.0 * (double)c->total_cols / c->num_cu;
    prof_elems = 0;
    // Group-resident batch launches (option "resident"): ONE launch per launch shape whose items are (group, query) pairs.
    // The batch gets the 4-wave shape that wastes the fewest padded rows at that shape's rate (a query of its own shape
    // when that saves 12 %), and the queries of a shape run together.  Which queries join depends on how small the
    // database is beside the chip (groups per workgroup of the query's launch), measured on c2-shaped databases of 1.0e8 /
    // 2.4e8 / 6.0e8 residues (profiles/r02_short_query_sets.txt): a query that takes several passes gains from the batch up
    // to about 6 groups per workgroup (100 queries of 600-700 residues: 8 010 vs 7 540, 8 105 vs 7 900, 8 150 vs 8 430), a
    // one-pass query only up to about 2 (300 of 80-120: 7 160 vs 6 290 at 1.3, 7 175 vs 7 930 at 3) -- beyond that it runs whole
    // on one of three streams in rotation, beside the batch of the others.
    in_batch.assign(qn, 0);
    c->batch_now = c->opt_dynamic && (c->opt_resident == 1 || (c->opt_resident < 0 && (qn >= 2 || streaming))) &&
                   !((uint64_t)qn * S > 0xFFFFFFFFull || prof_elems_bound(qm, qn) > 0xFFFFFFFFull);   // (a batch addresses score rows and profiles with 32-bit offsets)
    // (no register room is reserved for the lane-systolic waves here: among the 4-wave shapes only the 8-row one would pass that
    // filter, at 6 000 instead of 8 400 GCUPS)
    if (c->batch_now) {
        if (choose_batch_shapes(c, main_mode, qm, qn, qps)) return 1;
        const bool pick = c->opt_resident < 0 && !streaming;         // (forced, or a database that streams in: every query joins)
        uint32_t joined = 0;
        int multi_wg = 1;                                            // (the multi-pass queries join or stay out together: outside, they would bring tail kernels)
        for (uint32_t q = 0; q < qn; ++q) {
            int per_cu = 1;
            if (wgs_per_cu(c, main_mode, qps[q].T, qps[q].W, true, &per_cu)) return 1;
            if (qps[q].passes > 1) multi_wg = std::max(multi_wg, n_workgroups(c, per_cu));
        }
        // (round 3, profiles/r03_short_query_sets.txt: a query that would run ALONE in one pass of a 12- or 16-wave workgroup
        // but takes several of the batch's 4-wave passes gains from the batch at every database size tried -- 100 queries of
        // 280-320 residues: 8 040 vs 7 560 against 1.0e8 residues, 8 100 vs 7 700 against 6.0e8, 8 140 vs 7 730 against 1.2e9;
        // of 400-440: 8 370 vs 8 010 and 8 350 vs 8 030 -- so when every multi-pass member of the batch is of that kind,
        // they join whatever the size; with longer queries among them (450-560 at 6.0e8: 7 930 vs 8 160) the limit of 6 stays.
        // One-pass queries are judged by groups per CU, whatever their shape's occupancy: 1 000 queries of 20-60 residues
        // 7 670 in the batch at 5.2 groups per CU, but 7 600 vs 7 830 at 8.3 and 7 650 vs 8 070 at 12.2.)
        bool all_medium = true;
        for (uint32_t q = 0; q < qn && all_medium; ++q) {
            if (qps[q].passes == 1 || use_sp[q]) continue;
            QueryPlan alone{};
            c->batch_now = false;
            all_medium = choose_plan(c, main_mode, qm[q], false, true, &alone) == 0;     // (fails when no one-pass shape holds the query)
            c->batch_now = true;
        }
        const double per_cu_groups = (double)c->groups.size() / std::max(1, c->num_cu);
        for (uint32_t q = 0; q < qn; ++q) {
            const double per_wg = (double)c->groups.size() / multi_wg;
            in_batch[q] = !use_sp[q] && (!pick || (qps[q].passes == 1 ? per_cu_groups < 6.0 : all_medium || per_wg < 6.0));
            joined += in_batch[q];
        }
        // A batch launch is one persistent kernel for the whole batch: lane-systolic tail kernels launched beside it would
        // find no free slot until it ends (measured: c3 -16 %, a 1e8-residue database -25 %).  So a batch takes EVERY group
        // through the pipeline kernel -- which is fine as long as the longest item (longest group x the passes of the
        // longest query; the queue hands it out first) is at most half a workgroup's share of the batch; otherwise (c3: a
        // 35 000-residue sequence is 2.3 shares) the batch is not formed and the queries run one launch per pass beside their
        // tail kernels.
        double share = 0;
        uint32_t max_passes = 1;
        for (uint32_t q = 0; q < qn; ++q) {
            if (!in_batch[q]) continue;
            int per_cu = 1;
            if (wgs_per_cu(c, main_mode, qps[q].T, qps[q].W, true, &per_cu)) return 1;
            share += (double)qps[q].passes * (double)c->total_cols / n_workgroups(c, per_cu);
            max_passes = std::max<uint32_t>(max_passes, (uint32_t)qps[q].passes);
        }
        if (dbg)
            fprintf(stderr, "swimm_hip: batch of %u of %u queries: longest item %u columns x %u passes, a workgroup's share %.0f column-passes, %zu groups\n",
                    joined, qn, longest_cols, max_passes, share, c->groups.size());
        // (a database that streams in as several ranges: consecutive ranges overlap on two streams, the long items travel
        // and start first, and the last range is the one with the short sequences -- the longest item may take 0.9 of
        // the whole search before it sticks out at the end)
        const bool ranges_overlap = streaming && ranges.size() > 1;
        if (c->opt_resident < 0 && (double)longest_cols * max_passes > (ranges_overlap ? 0.9 : 0.5) * share) joined = 0;
        if (pick && joined < 2) joined = 0;                          // (one query alone gains nothing from a batch launch)
        if (joined == 0) { c->batch_now = false; in_batch.assign(qn, 0); }
    }
    if (dbg) fprintf(stderr, "swimm_hip: batch decided %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    // The one-pass queries outside the batch run whole -- every group through the pipeline kernel, no tail kernel -- on
    // three streams in rotation (issue()) when there are eight or more of them (a long sequence's serial chain, which
    // bounds a lone short query, is then covered by the neighbours' work), and always beside a batch (tail kernels would
    // find no room).  They are planned with the per-pass kernel's registers.
    const bool batch_formed = c->batch_now;
    c->batch_now = false;
    if (rotate || batch_formed)
        for (uint32_t q = 0; q < qn; ++q)
            if (!in_batch[q] && !use_sp[q] && qm[q] <= 64 * kLaneRows) {
                rotated[q] = choose_plan(c, main_mode, qm[q], false, true, &qps[q]) == 0;   // fails when no one-pass shape exists
                if (!rotated[q] && batch_formed) { in_batch[q] = 1; if (choose_batch_shapes(c, main_mode, qm + q, 1, one_plan)) return 1; qps[q] = one_plan[0]; }   // ... then it stays in the batch
            }
    c->batch_now = batch_formed;
    if (build_stacks()) return 1;
    c->batch_now = false;
    const bool any_batch = batch_formed;
    // A search that is long beside its upload (many queries, or a long one): the link has delivered everything while the first
    // small ranges were aligned, and the LAST range is most of the database.  That range runs as the resident database would --
    // a launch shape per query, one launch per pass, tail kernels, two streams in alternation -- instead of joining the
    // group-resident batch launches of the ranges before it: on a database of dozens of groups per workgroup the per-pass
    // launches are the faster ones (c5 at a quarter of its size: 7 195 ms resident, 7 420 ms as one batch launch per range).
    if (streaming && batch_formed && ranges.size() >= 2) {
        const Range &last = ranges.back();
        if (2 * last.cols >= c->total_cols && (uint64_t)(last.g1 - last.g0) >= 8ull * (uint64_t)n_workgroups(c, 4)) range_pp.back() = 1;
    }
    std::vector<BulkCols> rbulk;
    if (streaming && ranges.size() > 1 && (!batch_formed || range_pp.back())) {
        rqps.assign(ranges.size(), std::vector<QueryPlan>());
        rbulk.resize(ranges.size());
        for (size_t ri = 0; ri < ranges.size(); ++ri) {
            if (batch_formed && !range_pp[ri]) continue;
            rqps[ri].resize(qn);
            bulk_cols_of(c, ranges[ri], rbulk[ri]);
        }
    }
    for (uint32_t q = 0; q < qn; ++q) {
        const bool room = lane_room || (cut_room && qm[q] <= 1024);
        if (stack_of[q] >= 0) {                            // the query runs as a member of a stack: the stack's shape, one pass
            const Stack &st = stacks[stack_of[q]];
            qps[q].T = st.T; qps[q].W = st.W; qps[q].passes = 1; qps[q].mpad = 0;
        }
        if (use_sp[q]) {                                   // the score-profile kernel: one wave of 32 rows per workgroup
            qps[q].T = kSpRows; qps[q].W = 1; qps[q].passes = (int)((qm[q] + kSpRows - 1) / kSpRows); qps[q].mpad = 0; qps[q].sp = true;
        } else
        if (!in_batch[q] && !rotated[q] && choose_plan(c, main_mode, qm[q], room, false, &qps[q])) return 1;   // (a batch's shapes are chosen above)
        if (dbg)
            fprintf(stderr, "swimm_hip: query %u m=%u -> T=%d W=%d passes=%d (lane_room=%d)\n", q, qm[q], qps[q].T, qps[q].W, qps[q].passes, (int)room);
        // A database that is streaming in, per-pass launches: every range gets the launch shape that suits ITS groups -- the
        // range with the longest sequences is small beside the chip (a few long chains over hundreds of workgroups), and
        // fewer, taller workgroups finish it sooner than the shape that is best for the database as a whole.
        if (!rqps.empty())
            for (size_t ri = 0; ri < ranges.size(); ++ri) {
                if (rqps[ri].empty()) continue;
                if (use_sp[q]) { rqps[ri][q] = qps[q]; continue; }
                c->tiling_room = streaming && ri + 1 < ranges.size();       // (the last range's launches start when everything has landed)
                const int rc_plan = choose_plan(c, main_mode, qm[q], room, false, &rqps[ri][q], &ranges[ri], &rbulk[ri]);
                c->tiling_room = landing;
                if (rc_plan) return 1;
                qps[q].mpad = std::max(qps[q].mpad, rqps[ri][q].mpad);
                if (dbg) fprintf(stderr, "swimm_hip:   range %zu: T=%d W=%d passes=%d\n", ri, rqps[ri][q].T, rqps[ri][q].W, rqps[ri][q].passes);
            }
        const uint32_t lane_rows = (uint32_t)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows) * (64 * kLaneRows));
        // (a query's own profile serves the lane-systolic kernel -- tail, promotion re-runs -- and its own pipeline launches; a
        // member of a stack has neither when no alignment of it can leave the first tier's range)
        const bool own_profile = stack_of[q] < 0 || (long)qm[q] * c->max_pos >= (main_mode == Mode::F16 ? f16_thr : 32767);
        qps[q].mpad = own_profile ? std::max(qps[q].mpad, lane_rows) : 0;
        qps[q].prof_off = prof_elems;
        prof_elems += (size_t)kCodes * qps[q].mpad;
    }
    for (Stack &st : stacks) {                             // a stack's own profile: its members' rows, strip after strip
        st.prof_off = prof_elems;
        prof_elems += (size_t)kCodes * st.rows;
    }
    // Two or more multi-pass queries: their passes alternate between the two bulk streams (each with a boundary buffer
    // of its own), so that the end of every launch -- the last workgroups finishing alone -- is covered by a kernel of
    // the other query.  (Within ONE such query the even/odd split of run_passes does the same.)
    uint32_t n_multi = 0;
    for (uint32_t q = 0; q < qn; ++q) n_multi += !rotated[q] && !in_batch[q] && !use_sp[q] && qps[q].passes > 1;
    alternate = n_multi >= 2 && !streaming;
    if (!range_pp.empty() && range_pp.back()) {
        uint32_t n_multi_pp = 0;
        for (uint32_t q = 0; q < qn; ++q) n_multi_pp += !use_sp[q] && stack_of[q] < 0 && rqps.back()[q].passes > 1;
        alternate_pp = n_multi_pp >= 2;
        if (dbg) fprintf(stderr, "swimm_hip: the last range (%u groups, %llu columns) runs like a resident database: one launch per pass%s\n", ranges.back().g1 - ranges.back().g0,
                         (unsigned long long)ranges.back().cols, alternate_pp ? ", queries alternating on two streams" : "");
    }
    c->batch_now = any_batch;
    c->tiling_room = false;
    return 0;
}

int SearchRun::upload_profiles()
{
    std::vector<int16_t> prof(prof_elems, 0);
    for (uint32_t q = 0; q < qn; ++q) {
        if (qps[q].mpad == 0) continue;
        const int8_t *qa = c->qcodes.data() + qdisp[q];
        for (int d = 0; d < kCodes; ++d) {
            int16_t *row = prof.data() + qps[q].prof_off + (size_t)d * qps[q].mpad;
            for (uint32_t r = 0; r < qm[q]; ++r) row[r] = c->submat[(int)qa[r] * 32 + kHostCode[d]];     // (d: the device's residue numbering, sw_kernels.h)
        }
    }
    for (const Stack &st : stacks)
        for (size_t k = 0; k < st.q.size(); ++k) {
            const uint32_t q = st.q[k];
            const int8_t *qa = c->qcodes.data() + qdisp[q];
            for (int d = 0; d < kCodes; ++d) {
                int16_t *row = prof.data() + st.prof_off + (size_t)d * st.rows + (size_t)st.strip0[k] * st.T;
                for (uint32_t r = 0; r < qm[q]; ++r) row[r] = c->submat[(int)qa[r] * 32 + kHostCode[d]];
            }
        }
    c->last_plans.resize(c->qm.size());
    for (uint32_t q = 0; q < qn; ++q) { qps[q].mode = main_mode; qps[q].dynamic = c->opt_dynamic != 0; qps[q].resident = in_batch[q] != 0; c->last_plans[qb + q] = qps[q]; }
    for (auto &rv : rqps)
        for (uint32_t q = 0; q < qn && !rv.empty(); ++q) {
            rv[q].mpad = qps[q].mpad; rv[q].prof_off = qps[q].prof_off;      // one profile per query, padded for the tallest plan
            rv[q].mode = main_mode; rv[q].dynamic = qps[q].dynamic; rv[q].resident = false;
        }
    {   // the score-profile kernel's inputs: its queries' residue codes (padded with the dummy residue) and the matrix as binary16
        std::vector<int8_t> codes;
        qcode_off.assign(qn, 0);
        for (uint32_t q = 0; q < qn; ++q) {
            if (!use_sp[q]) continue;
            qcode_off[q] = codes.size();
            const int8_t *qa = c->qcodes.data() + qdisp[q];
            codes.insert(codes.end(), qa, qa + qm[q]);
            codes.resize((codes.size() + kSpRows - 1) / kSpRows * kSpRows, (int8_t)23);
        }
        if (!codes.empty()) {
            uint16_t sub16[kCodes * kSpSubStride] = {};      // [database residue d][query residue q]; d = 24 (lane padding) scores 0
            for (int d = 0; d < kCodes; ++d)
                for (int qr = 0; qr < 24 && kHostCode[d] != 24; ++qr) {
                    const _Float16 h = (_Float16)(float)c->submat[qr * 32 + kHostCode[d]];
                    memcpy(&sub16[d * kSpSubStride + qr], &h, sizeof(uint16_t));
                }
            HIP_TRY(c->d_qcodes.reserve(codes.size()));
            HIP_TRY(c->d_sub16.reserve(kCodes * kSpSubStride));
            HIP_TRY(hipMemcpyAsync(c->d_qcodes.p, codes.data(), codes.size(), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_sub16.p, sub16, sizeof sub16, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));          // (host temporaries)
        }
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
        uint64_t need_bnd = 0;          // the per-pass launches' boundary rows (and, resident database, a batch's)
        uint64_t need_res = 0;          // streaming: the group-resident range launches' boundary scratch, on three buffers in rotation
        bool pp_on_b = false;           // streaming: some per-pass launches run on stream B with the second buffer
        size_t tail_cols = 0, tail_items = 0, launch_total = 16;
        int max_passes = 1;
        for (uint32_t q = 0; q < qn; ++q) max_passes = std::max(max_passes, (int)((qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows)));
        if (one_list) {
            // (one launch, no boundary rows, no tail: nothing to size but the launch cursor and the promotion ladder's scratch below)
        } else if (streaming) {
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
                    else {
                        longest_main = std::max(longest_main, c->groups[g].ncols);
                        if (g < c->cut_lane.size() && c->cut_lane[g] < 64) { t_items += 64 - c->cut_lane[g]; t_cols += (size_t)(64 - c->cut_lane[g]) * c->groups[g].ncols; }   // outlier pairs
                    }
                }
                tail_items = std::max(tail_items, t_items);
                tail_cols = std::max(tail_cols, t_cols);
                for (uint32_t q = 0; q < qn; ++q) {
                    const QueryPlan &qp = qp_of(ri, q);
                    if (qp.passes <= 1) { launch_total += 2; continue; }
                    int per_cu = 1;
                    if (wgs_per_cu(c, main_mode, qp.T, qp.W, res_of(ri, q), &per_cu)) return 1;
                    const uint64_t cols = ranges[ri].cols * (main_mode == Mode::I32 ? 2 : 1);
                    if (res_of(ri, q)) need_res = std::max<uint64_t>(need_res, (uint64_t)n_workgroups(c, per_cu) * longest_all * 64);   // (a batch takes every group)
                    // (only the dynamic queue's list is cut into runs that fit the budget: the static partition takes the range whole)
                    else {
                        need_bnd = std::max<uint64_t>(need_bnd, (c->opt_dynamic ? std::min<uint64_t>(cols, std::max<uint64_t>(budget, longest_all)) : cols) * 64);
                        pp_on_b = pp_on_b || (pp_range(ri) ? alternate_pp : (ri & 1) != 0);      // (issue(): which stream a per-pass launch of this range takes)
                    }
                    // (runs of the boundary buffer: the greedy cut closes a run when the next item would overflow it, so two
                    // consecutive runs together exceed the budget -- at most 2 cols / budget + 1 of them, and never more than items)
                    const uint64_t items = (uint64_t)(ranges[ri].g1 - ranges[ri].g0) * (main_mode == Mode::I32 ? 2 : 1);
                    launch_total += (size_t)qp.passes * (size_t)(std::min<uint64_t>(items, 2 * cols / std::max<uint64_t>(budget, 1) + 1) + 2);
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
                    // ... rounded up to what a search that STREAMS the same database in reserves from the geometry alone (below: all the
                    // groups' columns, or the budget when the list is cut into runs): the database that replaces this one finds the
                    // buffer large enough (a run is up to one item short of the budget: the 17 GB buffer of c4 was freed and allocated
                    // again for 0.2 % more, 1 s inside the first cold search)
                    const uint32_t longest = longest_cols;          // (plan_queries: the longest group of the database)
                    const uint64_t all = c->total_cols * (main_mode == Mode::I32 ? 2 : 1);
                    const uint64_t bound = c->opt_dynamic ? std::min<uint64_t>(all, std::max<uint64_t>(bnd_budget_cols(c), longest)) : all;
                    need_bnd = std::max<uint64_t>(need_bnd, std::max<uint64_t>(cols, bound) * 64);
                }
                launch_total += (size_t)qps[q].passes * std::max<size_t>(nsegs, 2);   // two kernels per pass when the list is split over two streams
                tail_cols = std::max<size_t>(tail_cols, dp->tail.cols);
                tail_items = std::max<size_t>(tail_items, dp->tail.n);
            }
        for (uint32_t q = 0; q < qn; ++q) {                 // the score-profile queries: a launch per pass of 32 rows and range, the boundary of a whole range
            if (!use_sp[q]) continue;
            launch_total += ranges.size() * (size_t)qps[q].passes;
            if (qps[q].passes > 1)
                for (const Range &rg : ranges) need_bnd = std::max<uint64_t>(need_bnd, rg.cols * 64);
        }
        if (dbg) fprintf(stderr, "swimm_hip: buffer sizes known %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
        // (round 4, first version: all three buffers at the per-pass size -- 3 x 17 GB for c4's last range, 0.97 s of hipMalloc inside
        // the first cold search -- where the group-resident ranges that take turns on them need 0.3 GB each)
        HIP_TRY(c->d_bnd.reserve(std::max(need_bnd, need_res)));
        if (alternate || c->batch_now || (streaming && !one_list)) HIP_TRY(c->d_bnd_b.reserve(std::max(streaming && !pp_on_b ? (uint64_t)0 : need_bnd, need_res)));
        if (c->batch_now && streaming) HIP_TRY(c->d_bnd_c.reserve(std::max<uint64_t>(need_res, 1)));
        HIP_TRY(c->d_queue.reserve(launch_total));           // one zeroed queue cursor per pipeline launch of this search
        HIP_TRY(hipMemsetAsync(c->d_queue.p, 0, launch_total * sizeof(uint32_t), c->stream));
        c->queue_next = 0;
        // The long-sequence tail of a range: ONE lane-systolic launch per rows-per-lane class (8 / 4 / 2) for all the queries
        // that have a tail, each class on a stream and a scratch of its own.
        {
            size_t passes_of[3] = {0, 0, 0}, queries_of[3] = {0, 0, 0}, multi = 0;
            for (uint32_t q = 0; q < qn; ++q) {
                const int lr = lane_rows_for(c, qm[q]);
                const int cls = lr == kLaneRows ? 0 : lr == 4 ? 1 : 2;
                const size_t ps = (qm[q] + 64 * lr - 1) / (64 * lr);
                passes_of[cls] += ps; queries_of[cls]++; multi += ps > 1;
            }
            tail_lanes = tail_items > 0 ? 1 + (queries_of[1] > 0) + (queries_of[2] > 0) : 1;
            if (tail_items > 0 && reserve_lane_scratch(c, c->tail_scratch, tail_cols, tail_items, passes_of[0], queries_of[0], multi, ranges.size())) return 1;
            for (int i = 0; i < 2; ++i) {
                if (tail_items == 0 || queries_of[i + 1] == 0) continue;
                if (reserve_lane_scratch(c, c->tail_scratch_t[i], tail_cols, tail_items, passes_of[i + 1], queries_of[i + 1], 0, ranges.size())) return 1;
            }
            if (dbg && tail_items) fprintf(stderr, "swimm_hip: tail of %zu items: %zu / %zu / %zu queries with 8 / 4 / 2 rows per lane, %zu chained passes in all\n", tail_items, queries_of[0], queries_of[1], queries_of[2], passes_of[0]);
        }
        if (reserve_lane_scratch(c, c->rerun_scratch, (size_t)1 << 22, 4096, max_passes, 1, max_passes > 1 ? 1 : 0)) return 1;
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
    HIP_TRY(hipStreamWaitEvent(c->stream_b, c->ev_ready, 0));
    uint32_t one_pass_seen = 0;
    double alt_rows[2] = {0, 0};
    HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_ready, 0));
    while (c->ev_query.size() < 2 * (size_t)qn) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->ev_query.push_back(e);
    }
    // the query table of the group-resident launches: per launch shape, the stacks of short queries, then the queries in
    // ascending length (the kernel takes index nq - 1, the longest, first); the same for every range, so it travels once
    std::map<std::pair<int, int>, size_t> qdesc_off;
    by_shape.clear();
    {
        for (size_t si = 0; si < stacks.size(); ++si)
            if (stacks[si].resident) by_shape[std::make_pair(stacks[si].T, stacks[si].W)].push_back(Unit{(int)si, 0});
        for (uint32_t q = 0; q < qn; ++q)
            if (qps[q].resident && stack_of[q] < 0) by_shape[std::make_pair(qps[q].T, qps[q].W)].push_back(Unit{-1, q});
        std::vector<QDesc> qd_host;
        std::vector<uint32_t> wave_host;
        for (Stack &st : stacks) {                          // every stack's waves: the score row of the member the wave belongs to
            st.wave_tab = (uint32_t)wave_host.size();
            for (int k = 0, mi = 0; k < st.W; ++k) {
                while (mi + 1 < (int)st.q.size() && (uint32_t)k >= st.strip0[mi + 1]) ++mi;
                if ((uint64_t)st.q[mi] * S > 0xFFFFFFFFull) return fail("stacked queries: score rows beyond 2^32 elements (lower score_mib)");
                wave_host.push_back((uint32_t)((uint64_t)st.q[mi] * S));
            }
        }
        for (auto &kv : by_shape) {
            qdesc_off[kv.first] = qd_host.size();
            for (const Unit &u : kv.second) {
                QDesc d{};
                if (u.stack >= 0) {
                    const Stack &st = stacks[u.stack];
                    if (st.prof_off > 0xFFFFFFFFull) return fail("group-resident batch: profiles beyond 2^32 elements");
                    d.prof_off = (uint32_t)st.prof_off; d.prof_stride = st.rows; d.passes = 1; d.out_off = 0; d.seam_mask = st.seam_mask; d.wave_tab = st.wave_tab;
                } else {
                    const uint32_t q = u.q;
                    if ((uint64_t)q * S > 0xFFFFFFFFull || qps[q].prof_off > 0xFFFFFFFFull) return fail("group-resident batch: score rows beyond 2^32 elements (lower score_mib)");
                    d.prof_off = (uint32_t)qps[q].prof_off; d.prof_stride = qps[q].mpad; d.passes = (uint32_t)qps[q].passes; d.out_off = (uint32_t)((uint64_t)q * S);
                    d.seam_mask = 0; d.wave_tab = kNoTab;
                }
                qd_host.push_back(d);
            }
        }
        HIP_TRY(c->d_qdesc.reserve(qd_host.size()));
        HIP_TRY(c->d_wave_out.reserve(wave_host.size()));
        if (list_copy(c, c->d_qdesc.p, qd_host.data(), qd_host.size() * sizeof(QDesc)) ||
            list_copy(c, c->d_wave_out.p, wave_host.data(), wave_host.size() * sizeof(uint32_t)) || list_sync(c)) return 1;
    }
    if (wait_before_issue && wait_for_the_database()) return 1;
    if (streaming && !one_list)          // the first range's work lists need its geometry only: ready before its bytes are
        for (uint32_t q = 0; q < qn; ++q) { DbPlan *dp = nullptr; if (plan_of(0, q, &dp)) return 1; }
    for (size_t ri = 0; ri < ranges.size(); ++ri) {
        if (one_list) {                  // ONE launch over the whole list, now -- it follows the upload by itself
            if (issue_one_list()) return 1;
            break;
        }
        if (streaming) {
            if (wait_uploaded(range_chunks[ri].second)) return 1;
            const UploadPart &last = up_order[range_chunks[ri].second - 1];   // the upload stream is in order: its last part's event covers the range
            if (dbg) fprintf(stderr, "swimm_hip: range %zu (%llu columns): host copies done %.3f ms after the call began\n", ri, (unsigned long long)ranges[ri].cols, (now_s() - t_begin) * 1e3);
            HIP_TRY(hipStreamWaitEvent(c->stream, last.ready, 0));
            HIP_TRY(hipStreamWaitEvent(c->stream_b, last.ready, 0));
            HIP_TRY(hipStreamWaitEvent(c->stream2, last.ready, 0));
        }
        // The long-sequence tail of this range first (a few long serial chains, one wave each, beside the bulk kernels: 3 bulk
        // waves of 144 VGPRs + 1 lane wave of 80 fill a SIMD's 512 registers exactly): every query that has one joins ONE
        // lane-systolic launch per rows-per-lane class, so the chains of all the queries run side by side.
        {
            std::vector<LaneQuery> cls[3];
            std::vector<uint32_t> cls_q[3];
            const LaneList *ll = nullptr;
            for (uint32_t k = 0; k < qn; ++k) {
                const uint32_t q = qn - 1 - k;
                if ((stack_of[q] >= 0 && !pp_range(ri)) || res_of(ri, q) || rotated[q] || use_sp[q]) continue;      // (every group through the pipeline kernel)
                if (stack_of[q] >= 0) continue;            // (a stack's members run whole, below, also in a range that runs per pass)
                DbPlan *dp = nullptr;
                if (plan_of(ri, q, &dp)) return 1;
                if (dp->tail.n == 0) continue;
                if (!ll) ll = &dp->tail;                  // (the same items whatever the launch shape: the first list serves all)
                const int lr = lane_rows_for(c, qm[q]);
                const int ci = lr == kLaneRows ? 0 : lr == 4 ? 1 : 2;
                cls[ci].push_back(LaneQuery{qm[q], qps[q].prof_off, qps[q].mpad, (uint64_t)q * S});
                cls_q[ci].push_back(q);
            }
            for (int ci = 0; ci < 3; ++ci) {
                if (cls[ci].empty()) continue;
                // (one stream for all classes, the long queries' launch -- the longest chains -- first: a stream of its own per class
                // measured 2-3 % slower on c3 at 30 % and 100 % of its size; HIP multiplexes streams onto four hardware queues)
                hipStream_t st = c->stream2;
                LaneScratch &sc = ci == 0 ? c->tail_scratch : c->tail_scratch_t[ci - 1];
                if (run_lane_batch(c, main_mode == Mode::F16 ? Mode::F16 : Mode::PK16, ci == 0 ? kLaneRows : ci == 1 ? 4 : 2, cls[ci], *ll, st, sc)) return 1;
                if (dbg) fprintf(stderr, "swimm_hip: range %zu: tail of %u items for %zu queries in one launch (%d rows per lane)\n", ri, ll->n, cls[ci].size(), ci == 0 ? kLaneRows : ci == 1 ? 4 : 2);
                for (uint32_t q : cls_q[ci]) HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], st));
            }
        }
        // Longest query first: its promotion re-runs (a handful of long serial chains on stream 3) then overlap
        // the bulk kernels of the shorter queries instead of running alone at the end.
        for (uint32_t k = 0; k < qn; ++k) {
            const uint32_t q = qn - 1 - k;                 // queries arrive sorted by ascending length
            if (stack_of[q] >= 0 || res_of(ri, q)) continue;     // its stack / its batch launch takes it, below
            DbPlan *dp = nullptr;
            if (plan_of(ri, q, &dp)) return 1;
            int32_t *row = c->d_scores.p + (size_t)q * S;
            if (use_sp[q]) {                               // the score-profile kernel takes every group of the range, pass by pass
                if (dp->have_main && run_sp_passes(c, qps[q], dp->main, c->d_qcodes.p + qcode_off[q], row, c->stream, c->d_bnd)) return 1;
                HIP_TRY(hipEventRecord(c->ev_query[2 * q], c->stream));
                HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], c->stream));
                continue;
            }
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
            hipStream_t bulk_stream = c->stream;
            // (with an extreme sequence in the database the short queries keep their tail kernel -- in the launches above -- but
            // their bulk kernels still take turns on the three streams)
            if (rotated[q] || (many_short && qps[q].passes == 1 && qm[q] <= 64 * kLaneRows)) {
                switch (one_pass_seen++ % 3) {
                case 0: bulk_stream = c->stream; break;
                case 1: bulk_stream = c->stream_b; break;
                default: bulk_stream = c->stream2; break;
                }
            }
            DevBuf<uint2> *bnd = &c->d_bnd;
            // (the one-pass queries of such a batch take their turn as well: they are the shortest, the batch ends with them,
            // and a launch that runs alone ends with a few workgroups holding the chip -- c3's last eight queries: 26 ms at 5 900 GCUPS)
            // Each query goes to the stream with less work so far (padded rows), longest first: strict turns left one stream
            // 6 % more rows on c3 and the other idle for the last 60 ms.
            if ((alternate || (alternate_pp && pp_range(ri))) && !rotated[q] && !res_of(ri, q) && !(many_short && qps[q].passes == 1 && qm[q] <= 64 * kLaneRows)) {
                const int pick = alt_rows[1] < alt_rows[0] ? 1 : 0;
                alt_rows[pick] += (double)qp_of(ri, q).passes * qp_of(ri, q).W * qp_of(ri, q).T;
                if (pick == 1) { bulk_stream = c->stream_b; bnd = &c->d_bnd_b; }
            }
            if (streaming && !pp_range(ri) && (ri & 1)) { bulk_stream = c->stream_b; bnd = &c->d_bnd_b; }     // consecutive ranges overlap
            if (dbg)
                fprintf(stderr, "swimm_hip: query %u (%u rows, %d passes of %d x %d): bulk on %s\n", q, qm[q], qp_of(ri, q).passes, qp_of(ri, q).W, qp_of(ri, q).T,
                        bulk_stream == c->stream ? "stream A" : bulk_stream == c->stream_b ? "stream B" : "the tail stream");
            if (dp->have_main && run_passes(c, main_mode, qp_of(ri, q), dp->main, row, bulk_stream, pp_range(ri) ? !alternate_pp : !streaming && !alternate, *bnd)) return 1;
            HIP_TRY(hipEventRecord(c->ev_query[2 * q], bulk_stream));
            if (dp->tail.n == 0 || rotated[q]) HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], bulk_stream));      // (no tail launch recorded it above)
        }
        // The stacks of short queries that run in rotation (no group-resident batch): one per-pass launch per stack, on
        // the three streams in turn like the single short queries above.
        for (size_t si = 0; si < stacks.size(); ++si) {
            const Stack &stk = stacks[si];
            if (stk.resident && !pp_range(ri)) continue;
            QueryPlan sp{};
            sp.T = stk.T; sp.W = stk.W; sp.passes = 1; sp.mpad = stk.rows; sp.prof_off = stk.prof_off; sp.mode = main_mode; sp.dynamic = true; sp.resident = false;
            sp.stack = true; sp.seam_mask = stk.seam_mask; sp.wave_tab = stk.wave_tab;
            DbPlan *dp = nullptr;
            if (plan_for(ri, stk.T, stk.W, false, true, &dp)) return 1;
            hipStream_t st = c->stream;
            switch (one_pass_seen++ % 3) {
            case 0: st = c->stream; break;
            case 1: st = c->stream_b; break;
            default: st = c->stream2; break;
            }
            if (dp->have_main && run_passes(c, main_mode, sp, dp->main, c->d_scores.p, st, false, c->d_bnd)) return 1;
            if (ri + 1 == ranges.size())
                for (uint32_t q : stk.q) { HIP_TRY(hipEventRecord(c->ev_query[2 * q], st)); HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], st)); }
        }
        // Group-resident launches: ONE launch per launch shape for all the batch's queries of that shape -- the items are
        // (group, query) pairs, so even a small database gives every workgroup hundreds of them, the pipelines fill and
        // drain once per batch, and no pass waits for the slowest workgroup of the one before it.  Shapes alternate
        // between the two bulk streams.
        {
            uint32_t bi = 0;
            for (auto &kv : by_shape) {
                if (pp_range(ri)) break;                   // (this range runs one launch per pass, above)
                const size_t off = qdesc_off[kv.first];
                const int T = kv.first.first, W = kv.first.second;
                const uint32_t nqb = (uint32_t)kv.second.size();
                DbPlan *dp = nullptr;
                if (plan_for(ri, T, W, true, true, &dp)) return 1;
                uint64_t pass_sum = 0;
                uint32_t max_p = 1;
                for (const Unit &u : kv.second) {
                    const uint32_t ps = u.stack >= 0 ? 1u : (uint32_t)qps[u.q].passes;
                    pass_sum += ps; max_p = std::max<uint32_t>(max_p, ps);
                }
                // (a database that streams in: three ranges in flight, each on a stream and a boundary scratch of its own --
                // no tail kernels beside group-resident launches, so the tail stream serves as the third)
                // Which of the three: the one expected to be free first.  A launch lasts as long as its longest item's chain
                // (the longest group x the passes of the longest query at 1.2 us per column-pass: 18 ms for 5 000 columns x 3)
                // or its share of the chip's time, whichever is longer; a range that queued behind the range with the long
                // chains would wait for them with its data long there.
                uint32_t si = bi & 1;
                if (streaming) {
                    const double now = now_s() - t_begin;
                    uint32_t longest = 0;
                    for (uint32_t g = ranges[ri].g0; g < ranges[ri].g1; ++g) longest = std::max(longest, c->groups[g].ncols);
                    const double lasts = std::max(1.2e-6 * longest * max_p, (double)ranges[ri].cols * kGroupSeqs * (double)(pass_sum * T * W) / 8000e9);
                    si = 0;
                    for (uint32_t k = 1; k < 3; ++k)
                        if (std::max(now, stream_free[k]) < std::max(now, stream_free[si])) si = k;
                    stream_free[si] = std::max(now, stream_free[si]) + lasts;
                }
                hipStream_t st = si == 0 ? c->stream : si == 1 ? c->stream_b : c->stream2;
                DevBuf<uint2> &bnd = si == 0 ? c->d_bnd : si == 1 ? c->d_bnd_b : c->d_bnd_c;
                if (dp->have_main && run_resident_batch(c, main_mode, T, W, dp->main, c->d_qdesc.p + off, nqb, pass_sum, max_p, st, bnd)) return 1;
                for (const Unit &u : kv.second) {
                    if (u.stack >= 0) {
                        for (uint32_t q : stacks[u.stack].q) { HIP_TRY(hipEventRecord(c->ev_query[2 * q], st)); HIP_TRY(hipEventRecord(c->ev_query[2 * q + 1], st)); }
                    } else {
                        HIP_TRY(hipEventRecord(c->ev_query[2 * u.q], st));
                        HIP_TRY(hipEventRecord(c->ev_query[2 * u.q + 1], st));
                    }
                }
                ++bi;
            }
        }
        // the next range's work lists need its geometry only: build them now, while the GPU aligns this range and
        // before the host blocks in the next range's copies
        if (dbg && streaming) fprintf(stderr, "swimm_hip: range %zu: launches issued %.3f ms after the call began\n", ri, (now_s() - t_begin) * 1e3);
        extern double g_list_sync_wait_s;
        g_list_sync_wait_s = 0;
        if (streaming && ri + 1 < ranges.size())
            for (uint32_t q = 0; q < qn; ++q) { DbPlan *dp = nullptr; if (plan_of(ri + 1, q, &dp)) return 1; }
        if (dbg && streaming) fprintf(stderr, "swimm_hip: range %zu: next range's work lists built %.3f ms after the call began (%.3f ms of it waiting for the lists' copies)\n", ri, (now_s() - t_begin) * 1e3, g_list_sync_wait_s * 1e3);
    }
    if (streaming) {
        // the ranges took turns on the bulk streams (and the tail stream, for group-resident launches): "query q's bulk
        // kernels are done" = all of them have drained
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

// The one launch of a streaming search for one query (layout_ranges): every group of the database through the pipeline kernel,
// in the order the parts travel, the workgroups pulling from the list as far as the uploader has published it.
int SearchRun::issue_one_list()
{
    const QueryPlan &qp = qps[0];
    // (the shape and the workgroup count come from choose_one_list_plan: what the chip holds of the growing-list instantiation
    // while every SIMD keeps a tiling wave's registers free)
    const int n_wg = (int)std::min<uint32_t>((uint32_t)one_list_wg, one_list_items);
    if (list_sync(c)) return 1;                                   // the item list has arrived (copied while the plans were made)
    if (c->queue_next >= c->d_queue.cap) return fail("pipeline launch cursors exhausted");
    PipeParams p{};
    fill_common(c, qp, p, nullptr);
    p.items = c->d_stream_items.p;
    p.n_items = one_list_items;
    p.avail = c->d_avail.p;
    p.max_steps = (uint32_t)std::min<uint64_t>(one_list_chunks + kMaxWaves + 1, 0x3ffffff0u);
    p.queue = c->d_queue.p + c->queue_next++;
    p.r0 = 0;
    p.first_pass = 1; p.last_pass = 1;
    p.out = c->d_scores.p;
    p.err = c->d_err.p;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_avail, 0));
    if (timed_launch(c, main_mode, qp.T, qp.W, n_wg, p, c->stream)) return 1;
    one_list_launched = true;
    c->launches++;
    c->cells += one_list_chunks * kChunkCols * (uint64_t)(qp.W * qp.T) * 128;
    HIP_TRY(hipEventRecord(c->ev_query[0], c->stream));
    HIP_TRY(hipEventRecord(c->ev_query[1], c->stream));
    if (dbg) fprintf(stderr, "swimm_hip: one launch of %d workgroups (%d x %d rows) over %u items issued %.3f ms after the call began\n", n_wg, qp.W, qp.T, one_list_items, (now_s() - t_begin) * 1e3);
    // the launch is on its way; now the host waits for the uploader (a failed upload must not leave the launch waiting)
    if (wait_uploaded(up_order.size())) {
        const std::string msg = g_err;
        (void)launch_publish_items(c->d_avail.p, kAvailAbort, c->stream_up);
        return fail("%s", msg.c_str());
    }
    if (dbg) fprintf(stderr, "swimm_hip: host copies done %.3f ms after the call began\n", (now_s() - t_begin) * 1e3);
    return 0;
}

// Promotion ladder (the reference's int8 -> int16 -> int32, CPUsearch.c:678-957, one rung higher): f16 results >= f16_thr (2048
// less the pipeline kernel's largest column offset) are
// re-run as packed int16 pairs, int16 results >= 32767 as int32 sequences; every re-run is a lane-systolic item (one wave per
// alignment).  The queries climb the ladder in two batches -- the longer half (issued first: done while the shorter half's
// kernels still run), then the rest -- and a batch climbs together: one scan per query into a shared list, ONE copy back,
// ONE lane-systolic launch for all the batch's re-runs.  (One query at a time, round 2's way, was a chain of its own: c3 at
// a tenth of its size spent 17 x 2.2 ms on the int16 re-runs of 17 queries, a handful of 3 000-column alignments each, behind
// a tail launch that now ends for all queries at once.)
int SearchRun::promotion_ladder()
{
    t_issued = now_s();
    if (main_mode == Mode::I32) return 0;
    std::vector<uint32_t> climbers;                          // longest first
    for (uint32_t k = 0; k < qn; ++k) {
        const uint32_t q = qn - 1 - k;
        const long bound = (long)qm[q] * c->max_pos;         // no alignment of this query can score more
        if ((main_mode == Mode::F16 && bound >= f16_thr) || bound >= 32767) climbers.push_back(q);
    }
    const size_t half = climbers.size() >= 4 ? (climbers.size() + 1) / 2 : climbers.size();
    for (size_t b0 = 0; b0 < climbers.size(); b0 += std::max<size_t>(half, 1)) {
        const std::vector<uint32_t> batch(climbers.begin() + b0, climbers.begin() + std::min(climbers.size(), b0 + half));
        for (uint32_t q : batch) {
            HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_query[2 * q], 0));
            HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_query[2 * q + 1], 0));
        }
        if (main_mode == Mode::F16 && ladder_rung(batch, f16_thr, Mode::PK16)) return 1;
        if (ladder_rung(batch, 32767, Mode::I32)) return 1;
    }
    return 0;
}

// one rung for a batch of queries: scan (slots whose score reached `thr` are listed and zeroed), re-run in `mode`
int SearchRun::ladder_rung(const std::vector<uint32_t> &batch, int thr, Mode mode)
{
    std::vector<uint32_t> qs;
    for (uint32_t q : batch)
        if ((long)qm[q] * c->max_pos >= thr) qs.push_back(q);
    if (qs.empty()) return 0;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(S, 0xFFFFFFFEull);   // the shared list holds one score row's worth of slots
    uint32_t *d_count = c->d_satlist.p + cap;
    HIP_TRY(c->d_ladder_counts.reserve(qs.size()));
    std::vector<std::vector<uint32_t>> slots(qs.size());     // per query: the slots that left the tier's range
    for (int round = 0;; ++round) {                          // (again while the shared list overflows: what did not fit is still in the score rows)
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), c->stream3));
        for (size_t i = 0; i < qs.size(); ++i) {
            HIP_TRY(launch_collect_saturated(c->d_scores.p + (size_t)qs[i] * S, S, thr, c->d_satlist.p, d_count, cap, c->stream3));
            HIP_TRY(hipMemcpyAsync(c->d_ladder_counts.p + i, d_count, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream3));   // where query i's part of the list ends
        }
        std::vector<uint32_t> counts(qs.size());
        HIP_TRY(hipMemcpyAsync(counts.data(), c->d_ladder_counts.p, qs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream3));
        HIP_TRY(hipStreamSynchronize(c->stream3));
        const bool overflow = counts.back() > cap;
        if (overflow && round > 1000) return fail("promotion ladder: the list of alignments that left the %s range does not drain", thr != 32767 ? "f16" : "int16");
        const uint32_t total = std::min(counts.back(), cap);
        if (total) {
            std::vector<uint32_t> list(total);
            HIP_TRY(hipMemcpyAsync(list.data(), c->d_satlist.p, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream3));
            HIP_TRY(hipStreamSynchronize(c->stream3));
            for (size_t i = 0; i < qs.size(); ++i) {
                const uint32_t l0 = std::min(i ? counts[i - 1] : 0u, cap), l1 = std::min(counts[i], cap);
                slots[i].insert(slots[i].end(), list.begin() + l0, list.begin() + l1);
            }
        }
        if (!overflow) break;
    }
    {
        // every query's re-run items: its own part of one list, longest first, boundary columns counted from the part's start
        std::vector<LaneItem> items;
        std::vector<LaneQuery> lqs;
        uint64_t bnd_cols = 0, pass_total = 0;
        size_t multi = 0;
        std::vector<uint8_t> seen;
        for (size_t i = 0; i < qs.size(); ++i) {
            const uint32_t q = qs[i];
            const std::vector<uint32_t> &list = slots[i];
            const uint32_t l0 = 0, l1 = (uint32_t)list.size();
            const size_t first = items.size();
            if (mode == Mode::PK16) {
                seen.assign(seen.size(), 0);
                for (uint32_t k = l0; k < l1; ++k) {                    // re-run the packed PAIR the slot belongs to
                    const uint32_t slot = list[k], g = slot / kGroupSeqs, l = (slot % kGroupSeqs) / 2;      // lane l holds the group's sequences 2l and 2l + 1
                    const uint32_t pair = g * 64 + l;
                    if (seen.size() <= pair) seen.resize(pair + 1, 0);
                    if (seen[pair]) continue;
                    seen[pair] = 1;
                    const GroupDesc &gd = c->groups[g];
                    const uint32_t len = std::max(c->seq_len[gd.seq0 + 2 * l], c->seq_len[gd.seq0 + 2 * l + 1]);
                    LaneItem li{};
                    li.db = gd.db; li.lane = l; li.half = 0; li.ncols = (len + kChunkCols - 1) / kChunkCols * kChunkCols;
                    li.slot_a = gd.seq0 + 2 * l; li.slot_b = gd.seq0 + 2 * l + 1;
                    if (li.ncols) items.push_back(li);
                }
                c->promoted16 += l1 - l0;
            } else {
                for (uint32_t k = l0; k < l1; ++k) {
                    const uint32_t slot = list[k], g = slot / kGroupSeqs, within = slot % kGroupSeqs;
                    LaneItem li{};
                    li.db = c->groups[g].db; li.lane = within / 2; li.half = within % 2;
                    li.ncols = (c->seq_len[slot] + kChunkCols - 1) / kChunkCols * kChunkCols;
                    li.slot_a = slot; li.slot_b = 0;
                    if (li.ncols) items.push_back(li);
                }
                c->promoted += items.size() - first;
            }
            if (items.size() == first) continue;
            std::stable_sort(items.begin() + first, items.end(), [](const LaneItem &a, const LaneItem &b) { return a.ncols > b.ncols; });
            uint64_t cols = 0;
            for (size_t k = first; k < items.size(); ++k) {
                if (cols + items[k].ncols > 0xFFFFFFFFull) return fail("promotion re-runs of query %u exceed 2^32 boundary columns", q);
                items[k].bnd_off = (uint32_t)cols; cols += items[k].ncols;
            }
            LaneQuery lq{qm[q], qps[q].prof_off, qps[q].mpad, (uint64_t)q * S};
            lq.items0 = (uint32_t)first; lq.n_items = (uint32_t)(items.size() - first); lq.cols = cols;
            lqs.push_back(lq);
            const uint64_t rp = (qm[q] + 64 * kLaneRows - 1) / (64 * kLaneRows);
            pass_total += rp;
            if (rp > 1) { bnd_cols += cols + 64; ++multi; }
        }
        if (!items.empty()) {
            size_t prog_need = 0;
            for (const LaneQuery &lq : lqs) prog_need += (size_t)((lq.m + 64 * kLaneRows - 1) / (64 * kLaneRows)) * lq.n_items;
            LaneScratch &sc = c->rerun_scratch;
            if (items.size() > c->d_rerun_items.cap || bnd_cols > sc.bnd[0].cap || (multi && prog_need > sc.prog.cap) || pass_total > sc.queue.cap || lqs.size() > sc.lq.cap) {
                // growing a buffer frees the old one, which waits for the whole device: rare (first big batch)
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(c->d_rerun_items.reserve(items.size() * 2));
                HIP_TRY(sc.queue.reserve(std::max<size_t>(256, pass_total * 2)));
                if (bnd_cols) { HIP_TRY(sc.bnd[0].reserve(bnd_cols * 2)); HIP_TRY(sc.bnd[1].reserve(bnd_cols * 2)); }
                HIP_TRY(sc.prog.reserve(std::max<size_t>(1, prog_need * 2)));
                HIP_TRY(sc.lq.reserve(lqs.size() * 2 + 8));
                HIP_TRY(sc.block_map.reserve((size_t)c->num_cu * 6 + 4096 + 2 * pass_total));
            }
            HIP_TRY(hipMemcpyAsync(c->d_rerun_items.p, items.data(), items.size() * sizeof(LaneItem), hipMemcpyHostToDevice, c->stream3));
            HIP_TRY(hipStreamSynchronize(c->stream3));       // `items` is a host temporary; stream 3 has drained: the previous rung's tables are free
            sc.lq_used = sc.bm_used = 0;
            LaneList ll;
            ll.items.p = c->d_rerun_items.p; ll.items.cap = c->d_rerun_items.cap;
            ll.n = 0; ll.cols = 0; ll.cell_cols = 0;
            const int rc = run_lane_batch(c, mode, kLaneRows, lqs, ll, c->stream3, sc);
            ll.items.p = nullptr; ll.items.cap = 0;           // borrowed
            if (rc) return rc;
        }
    }
    return 0;
}

int SearchRun::drain()
{
    HIP_TRY(hipEventRecord(c->ev_tail, c->stream2));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail, 0));
    HIP_TRY(hipEventRecord(c->ev_b, c->stream_b));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_b, 0));
    HIP_TRY(hipEventRecord(c->ev_tail3, c->stream3));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail3, 0));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    const double t_drain = now_s();
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dbg) fprintf(stderr, "swimm_hip: drain entered %.3f ms after the call began, every stream drained at %.3f ms\n", (t_drain - t_begin) * 1e3, (now_s() - t_begin) * 1e3);
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
    if (dbg && one_list_launched) {
        uint32_t av[4] = {0, 0, 0, 0};
        (void)hipMemcpy(av, c->d_avail.p, sizeof av, hipMemcpyDeviceToHost);
        fprintf(stderr, "swimm_hip: the item list's published count at the end: %u of %u; watchdog word %u\n", av[0], one_list_items, werr);
    }
    if (werr == 16u && one_list_launched) {     // (the one launch over the landing list ran out of patience: not wrong scores, missing ones)
        stalled = true;
        return fail("the streaming launch stopped waiting for the database");
    }
    if (werr) return fail("pipeline watchdog expired (code %u): results discarded", werr);
    if (streaming) { release_plans(c); c->groups_dirty = true; }   // (the cached lists of the resident database are built on the next search)
    pool_trim(c);                      // (buffers of a cleared database that the new chunks did not take: freed now, not at the start of a search)
    if (dbg) fprintf(stderr, "swimm_hip: drain left %.3f ms after the call began; device allocations since the last search: %u calls, %.1f MB, %.3f ms\n", (now_s() - t_begin) * 1e3,
                     g_alloc_stats.calls, g_alloc_stats.bytes / 1e6, g_alloc_stats.seconds * 1e3);
    g_alloc_stats = AllocStats{};
    return 0;
}

int search_device(swimm_hip_ctx *c, uint32_t qb, uint32_t qe, uint64_t *slots_out)
{
    {
        SearchRun run(c, qb, qe);
        if (run.run(slots_out) == 0) return 0;
        if (!run.stalled) return 1;
    }
    // The one launch of a streaming search gave up waiting for its data (sw_kernels.hip, wait_landed) and has ended; the
    // uploader has handed every part to the device meanwhile.  Once more, on the database that is resident now.
    if (getenv("SWIMM_HIP_DEBUG")) fprintf(stderr, "swimm_hip: the streaming launch stalled; searching the resident database instead\n");
    HIP_TRY(hipDeviceSynchronize());
    for (const ChunkRec &r : c->chunks)
        if (!r.uploaded) return fail("the streaming launch stopped waiting for the database and the upload did not complete");
    SearchRun again(c, qb, qe);
    return again.run(slots_out);
}


}  // namespace swimm_impl
