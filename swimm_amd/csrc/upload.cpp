// upload.cpp -- database chunks: registration, the copy + (re-)tile of one chunk, the uploader thread's entry, lengths.
#include "swimm_impl.h"

namespace swimm_impl {

// the work lists are derived from the resident database: rebuild them after it changed
int refresh_plans(swimm_hip_ctx *c)
{
    if (!c->groups_dirty) return 0;
    release_plans(c);
    c->groups_dirty = false;
    return 0;
}

// Device buffers of the chunks: a caller that replaces its database (clear_db, then add_chunk / add_sequences) gets the old
// database's buffers back instead of paying a hipFree and a hipMalloc per chunk (round 1 measured seven such pairs at most
// of 48 ms).  The pool only bridges that gap: whatever the newly registered chunks did not take is freed by the next search.
int pool_alloc(swimm_hip_ctx *c, size_t bytes, void **out, size_t *cap)
{
    size_t best = c->pool.size();
    for (size_t i = 0; i < c->pool.size(); ++i)
        if (c->pool[i].second >= bytes && c->pool[i].second <= bytes + bytes / 4 + (1u << 20) && (best == c->pool.size() || c->pool[i].second < c->pool[best].second)) best = i;
    if (best < c->pool.size()) {
        *out = c->pool[best].first; *cap = c->pool[best].second;
        c->pool.erase(c->pool.begin() + best);
        return 0;
    }
    const double t0 = now_s();
    struct Tm { double t0; size_t b; ~Tm() { g_alloc_stats.seconds += now_s() - t0; g_alloc_stats.bytes += b; g_alloc_stats.calls++; } } tm{t0, bytes};
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && !c->pool.empty()) {      // the pool still holds what the old database left: hand it back and try again
        (void)hipGetLastError();
        pool_trim(c);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) return fail("hipMalloc of %zu bytes for a database chunk: %s", bytes, hipGetErrorString(e));
    *cap = bytes;
    return 0;
}

void pool_trim(swimm_hip_ctx *c)
{
    if (c->pool.empty()) return;
    const double t0 = now_s();
    for (auto &b : c->pool) (void)hipFree(b.first);
    g_alloc_stats.seconds += now_s() - t0; g_alloc_stats.calls += (unsigned)c->pool.size();
    c->pool.clear();
}

// Registers a chunk's device groups (geometry only: nothing is copied here).
int register_chunk(swimm_hip_ctx *c, ChunkRec &rec, const uint16_t *lens_or_null, uint64_t n_lens)
{
    CHECK_DEVICE(c);
    const uint32_t dev_groups = rec.n_groups;
    uint64_t bytes = 0;
    for (uint32_t g = 0; g < dev_groups; ++g) { rec.goff[g] = bytes; bytes += (uint64_t)rec.gcols[g] * kGroupSeqs; }
    const bool own = rec.d_tiled == nullptr;               // (else: the caller's share of a slab's buffer, swimm_hip_add_sequences)
    if (own && pool_alloc(c, std::max<uint64_t>(bytes, 16), (void **)&rec.d_tiled, &rec.tiled_cap)) return 1;
    if (rec.kind == 0 && pool_alloc(c, (size_t)dev_groups * kGroupSeqs * sizeof(uint32_t), (void **)&rec.d_len, &rec.len_cap)) {
        if (own) (void)hipFree(rec.d_tiled);
        rec.d_tiled = nullptr;
        return 1;
    }
    if (hipEventCreateWithFlags(&rec.ready, hipEventDisableTiming) != hipSuccess) {
        if (own) (void)hipFree(rec.d_tiled);
        (void)hipFree(rec.d_len);
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
    if (lens_or_null) c->seq_len.insert(c->seq_len.end(), lens_or_null, lens_or_null + n_lens);      // (one pass, no zero fill first)
    c->seq_len.resize((size_t)(rec.group0 + dev_groups) * kGroupSeqs, 0);
    rec.lens_known = rec.kind == 1;
    c->chunks.push_back(std::move(rec));
    c->groups_dirty = true;
    release_plans(c);
    return 0;
}

// X2 (MICsearch.c:85-88): the device groups [g0, g1) of a chunk -- normally all of it -- to the device and into the tiled
// layout, all on the upload stream.  The copies come from pageable memory, so every hipMemcpyAsync returns only when its
// source has been consumed; what stays asynchronous is the (re-)tile kernel, whose end `ready` marks.  Scratch is reused
// part after part (the stream is in order: the next copy cannot overtake this part's kernel).  The kernels index groups,
// lane groups and sequences from 0: a part hands them its own slices, and the byte / residue offsets inside those slices
// stay absolute, so the scratch pointer is moved back by the part's first byte.
// A part's small arrays (group geometry, lengths, offsets) go through pinned staging: from pageable memory each of them is a
// blocking copy of 30-50 us, four or five per part and 21 parts for c2 -- 3.5 ms of a 14 ms upload (r04: the chunk layout's
// host copies ended 17 ms after the call, the slabs' 13.3 ms).  The staging is reused part after part: upload_part ends with a
// wait for ev_copied, behind these copies on the stream.
static int stage_small(swimm_hip_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return 0;
    const size_t at = (c->up_pin_used + 63) & ~(size_t)63;
    if (at + bytes > c->up_pin_cap) return fail("internal: upload staging of %zu bytes too small for %zu more", c->up_pin_cap, bytes);
    memcpy((char *)c->up_pin + at, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, (char *)c->up_pin + at, bytes, hipMemcpyHostToDevice, s));
    c->up_pin_used = at + bytes;
    return 0;
}

static int reserve_staging(swimm_hip_ctx *c, size_t need, hipStream_t s)
{
    c->up_pin_used = 0;
    if (need <= c->up_pin_cap) return 0;
    HIP_TRY(hipStreamSynchronize(s));
    if (c->up_pin) { HIP_TRY(hipHostFree(c->up_pin)); c->up_pin = nullptr; c->up_pin_cap = 0; }
    const size_t cap = std::max<size_t>(need + need / 2, (size_t)1 << 20);
    HIP_TRY(hipHostMalloc(&c->up_pin, cap, hipHostMallocDefault));
    c->up_pin_cap = cap;
    return 0;
}

int upload_part(swimm_hip_ctx *c, ChunkRec &r, uint32_t g0, uint32_t g1, hipEvent_t ready)
{
    if (r.uploaded || g0 >= g1) return 0;
    CHECK_DEVICE(c);
    hipStream_t s = c->stream_up;
    const uint32_t dev_groups = g1 - g0;
    uint32_t max_cols = 0;
    for (uint32_t g = g0; g < g1; ++g) max_cols = std::max(max_cols, r.gcols[g]);
    const double t_up0 = now_s();
    const bool dbg_t = getenv("SWIMM_HIP_DEBUG_UPLOAD") != nullptr;
    double t_a = 0, t_b = 0, t_c = 0;
    uint64_t bytes = 0;
    HIP_TRY(c->up_gcols.reserve(dev_groups));
    HIP_TRY(c->up_goff.reserve(dev_groups));
    {
        const uint32_t per = r.kind == 0 ? kGroupSeqs / r.vl : 0;
        const size_t small = r.kind == 0 ? (size_t)(std::min(r.group_count, g1 * per) - g0 * per) * 6 : (size_t)dev_groups * 4 + (size_t)dev_groups * kGroupSeqs * 2;
        if (reserve_staging(c, (size_t)dev_groups * 12 + small + 1024, s)) return 1;
    }
    if (stage_small(c, c->up_gcols.p, r.gcols.data() + g0, dev_groups * sizeof(uint32_t), s) ||
        stage_small(c, c->up_goff.p, r.goff.data() + g0, dev_groups * sizeof(uint64_t), s)) return 1;
    if (r.kind == 0) {
        const uint32_t per = kGroupSeqs / r.vl;
        const uint32_t v0 = g0 * per, v1 = std::min(r.group_count, g1 * per);
        uint64_t b0 = r.h_disp[v0], b1 = b0;
        for (uint32_t v = v0; v < v1; ++v) { b0 = std::min<uint64_t>(b0, r.h_disp[v]); b1 = std::max<uint64_t>(b1, (uint64_t)r.h_disp[v] + (uint64_t)r.h_n[v] * r.vl); }
        bytes = b1 - b0;
        HIP_TRY(c->up_b.reserve(std::max<uint64_t>(bytes, 16)));
        HIP_TRY(c->up_n.reserve(v1 - v0));
        HIP_TRY(c->up_disp.reserve(v1 - v0));
        HIP_TRY(hipMemsetAsync(r.d_len + (size_t)g0 * kGroupSeqs, 0, (size_t)dev_groups * kGroupSeqs * sizeof(uint32_t), s));
        if (stage_small(c, c->up_n.p, r.h_n + v0, (v1 - v0) * sizeof(uint16_t), s) || stage_small(c, c->up_disp.p, r.h_disp + v0, (v1 - v0) * sizeof(uint32_t), s)) return 1;
        t_a = now_s();
        HIP_TRY(hipMemcpyAsync(c->up_b.p, r.h_b + b0, bytes, hipMemcpyHostToDevice, s));
        t_b = now_s();
        HIP_TRY(hipEventRecord(c->ev_copied, s));
        HIP_TRY(launch_retile(c->up_b.p - b0, c->up_n.p, c->up_disp.p, v1 - v0, r.vl, c->up_goff.p, c->up_gcols.p, dev_groups, max_cols, r.d_tiled,
                              r.d_len + (size_t)g0 * kGroupSeqs, s));
    } else {
        const uint64_t s0 = (uint64_t)g0 * kGroupSeqs, s1 = std::min<uint64_t>(r.n_seq, (uint64_t)g1 * kGroupSeqs);
        const uint32_t o0 = r.gsrc[g0], o1 = r.gsrc[g1];
        bytes = o1 - o0;
        HIP_TRY(c->up_b.reserve(std::max<uint64_t>(bytes, 16)));
        HIP_TRY(c->up_off.reserve(dev_groups));
        HIP_TRY(c->up_n.reserve(s1 - s0));
        if (stage_small(c, c->up_off.p, r.gsrc.data() + g0, dev_groups * sizeof(uint32_t), s) || stage_small(c, c->up_n.p, r.h_len + s0, (s1 - s0) * sizeof(uint16_t), s)) return 1;
        t_a = now_s();
        HIP_TRY(hipMemcpyAsync(c->up_b.p, r.h_codes + o0, bytes, hipMemcpyHostToDevice, s));
        t_b = now_s();
        HIP_TRY(hipEventRecord(c->ev_copied, s));
        HIP_TRY(launch_tile_sequences(c->up_b.p - o0, c->up_n.p, c->up_off.p, (uint32_t)(s1 - s0), c->up_goff.p, c->up_gcols.p, dev_groups, max_cols, r.d_tiled, s));
    }
    HIP_TRY(hipEventRecord(ready ? ready : r.ready, s));
    t_c = now_s();
    HIP_TRY(hipEventSynchronize(c->ev_copied));      // the caller's buffers have been read
    if (dbg_t)
        fprintf(stderr, "swimm_hip: upload part: small copies returned after %.3f ms, the %.1f MB copy after %.3f, tiling kernel launched after %.3f, copy event reached after %.3f\n",
                (t_a - t_up0) * 1e3, bytes / 1e6, (t_b - t_up0) * 1e3, (t_c - t_up0) * 1e3, (now_s() - t_up0) * 1e3);
    if (getenv("SWIMM_HIP_DEBUG"))
        fprintf(stderr, "swimm_hip: %s of %.1f MB copied in %.2f ms (%.1f GB/s)\n", dev_groups == r.n_groups ? "chunk" : "part of a chunk", bytes / 1e6, (now_s() - t_up0) * 1e3,
                bytes / 1e9 / (now_s() - t_up0));
    r.groups_uploaded += dev_groups;
    if (r.groups_uploaded >= r.n_groups) {
        r.uploaded = true;
        r.h_b = nullptr; r.h_n = nullptr; r.h_disp = nullptr; r.h_codes = nullptr; r.h_len = nullptr;
        std::vector<uint32_t>().swap(r.gsrc);
    }
    return 0;
}

int upload_chunk(swimm_hip_ctx *c, ChunkRec &r) { return upload_part(c, r, 0, r.n_groups, r.ready); }


int ensure_uploader(swimm_hip_ctx *c)
{
    if (!c->up) {
        c->up = new Uploader(c);
        c->up->wait_warm();          // (once per context, when lazy uploads are switched on or the first chunk is recorded)
    }
    return 0;
}

// true lengths of the chunk-layout chunks' slots come from the re-tile kernel: fetched when somebody needs them
// (lane-systolic work lists, promotion re-runs), not inside add_chunk
int sync_lengths(swimm_hip_ctx *c)
{
    CHECK_DEVICE(c);
    for (ChunkRec &r : c->chunks) {
        if (r.lens_known || !r.uploaded) continue;
        const size_t n = (size_t)r.n_groups * kGroupSeqs;
        std::vector<uint32_t> wide(n);
        HIP_TRY(hipMemcpyAsync(wide.data(), r.d_len, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream_up));
        HIP_TRY(hipStreamSynchronize(c->stream_up));
        for (size_t i = 0; i < n; ++i) c->seq_len[(size_t)r.group0 * kGroupSeqs + i] = (uint16_t)wide[i];
        r.lens_known = true;
    }
    return 0;
}


}  // namespace swimm_impl
