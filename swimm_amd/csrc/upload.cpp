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

// Registers a chunk's device groups (geometry only: nothing is copied here).
int register_chunk(swimm_hip_ctx *c, ChunkRec &rec, const std::vector<uint32_t> &lens_or_empty)
{
    CHECK_DEVICE(c);
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
    CHECK_DEVICE(c);
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


int ensure_uploader(swimm_hip_ctx *c)
{
    if (!c->up) c->up = new Uploader(c);
    return 0;
}

// true lengths of the chunk-layout chunks' slots come from the re-tile kernel: fetched when somebody needs them
// (lane-systolic work lists, promotion re-runs), not inside add_chunk
int sync_lengths(swimm_hip_ctx *c)
{
    CHECK_DEVICE(c);
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


}  // namespace swimm_impl
