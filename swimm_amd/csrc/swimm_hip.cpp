// swimm_hip.cpp -- the C-ABI of libswimm_hip.so (see include/swimm_hip.h): contexts, queries, chunks, search,
// top-r, options.
//
// The host side of the gfx950 kernels in sw_kernels.hip: device-resident database (upload.cpp), per-query launch plan
// (rows per wave T, waves per workgroup W, passes) from a measured rate table and longest-first work lists for the
// persistent workgroups (plan.cpp), and the search itself -- pipeline launches, the long-sequence tail on a second
// stream, the binary16 -> int16 -> int32 promotion ladder on a third (search.cpp); here: top-r and score scatter.
// Structural template: mic_search_knc_ap_multiple_chunks (MICsearch.c:4-354) -- X1 = set_queries,
// X2-in = add_chunk (kept resident), compute = search, X3 = scatter into the caller's scores.
#include "swimm_impl.h"
#include "host/affinity.h"

thread_local std::string swimm_impl::g_err;
thread_local int swimm_impl::g_cur_vdevice = -1;
thread_local swimm_impl::DevArena *swimm_impl::g_list_arena = nullptr;

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

int swimm_hip_device_pci_bus_id(int device, char *buf, size_t buf_len)
{
    if (!buf || buf_len < 13) return fail("swimm_hip_device_pci_bus_id: buffer of at least 13 bytes needed");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= virtual_gpus(n)) return fail("swimm_hip_device_pci_bus_id: device %d not in [0,%d)", device, virtual_gpus(n));
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buf_len, device % std::max(n, 1)));
    for (char *p = buf; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells the address in lower case
    return 0;
}

int swimm_hip_bind_host_thread(int device, int num_devices, char *cpulist_out, size_t cpulist_len)
{
    if (cpulist_out && cpulist_len) cpulist_out[0] = 0;
    if (num_devices <= 0 || device < 0 || device >= num_devices) return fail("swimm_hip_bind_host_thread: device %d of %d", device, num_devices);
    const char *off = getenv("SWIMM_HIP_BIND");
    if (off && !strcmp(off, "0")) return 0;
    std::vector<std::string> bdf(num_devices);
    std::vector<const char *> bdf_p(num_devices, nullptr);
    for (int d = 0; d < num_devices; ++d) {
        char b[64] = "";
        if (swimm_hip_device_pci_bus_id(d, b, sizeof b) == 0) bdf[d] = b;       // (a device the runtime cannot name takes the even share)
        bdf_p[d] = bdf[d].c_str();
    }
    std::vector<int> allowed(SWIMM_AFF_MAX_CPUS), mine(SWIMM_AFF_MAX_CPUS);
    const int na = swimm_affinity_allowed_impl(allowed.data(), (int)allowed.size());
    if (na <= 0) return fail("swimm_hip_bind_host_thread: sched_getaffinity failed");
    const int nm = swimm_affinity_plan_impl("/sys", bdf_p.data(), num_devices, device, allowed.data(), na, mine.data(), (int)mine.size());
    if (nm <= 0) return fail("swimm_hip_bind_host_thread: no CPU for device %d", device);
    if (swimm_affinity_apply_impl(mine.data(), nm) != 0) return fail("swimm_hip_bind_host_thread: sched_setaffinity failed");
    if (cpulist_out && cpulist_len) swimm_affinity_format(mine.data(), nm, cpulist_out, cpulist_len);
    if (getenv("SWIMM_HIP_DEBUG")) {
        char txt[512];
        swimm_affinity_format(mine.data(), nm, txt, sizeof txt);
        fprintf(stderr, "swimm_hip: host thread of device %d (%s) of %d bound to CPUs %s\n", device, bdf_p[device], num_devices, txt);
    }
    return 0;
}

// The context's streams and the hardware queues behind them.  The runtime multiplexes the streams of one priority class onto a
// pool of four hardware queues, and a packet waits for everything in front of it in ITS hardware queue: a work-list copy on a
// stream that shared a queue with a group-resident range launch waited 60-400 ms for that launch to end, and in the second
// context of a process the upload stream did -- the 7 GB of c4 landed after 447 ms instead of 153 (r04; profiles/NOTES.md).  So
//   - the work-list stream is of the HIGH priority class: a pool of its own, never behind a launch stream;
//   - the upload stream stays in the normal class (in a class of its own every copy -> tile -> publish step of the one-launch
//     search cost more: c2 cold 27.5 -> 28.7 ms), and which queue it got is MEASURED: a wave that only lasts 400 us goes to each of
//     the three launch streams in turn and a 64-byte fill to the candidate; a fill that takes as long as the wave shares its queue.
//     A candidate that shares is set aside (alive, so that the next one is dealt another queue) and the next one tried.  Which
//     stream lands on which queue depends on what else the process has created and released; the probe does not.
static hipError_t make_streams(swimm_hip_ctx *c)
{
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess) return e;
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess || (e = hipStreamCreate(&c->stream_b)) != hipSuccess ||
        (e = hipStreamCreate(&c->stream2)) != hipSuccess || (e = hipStreamCreate(&c->stream3)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&c->stream_list, hipStreamDefault, hi)) != hipSuccess) return e;
    void *tmp = nullptr;
    if ((e = hipMalloc(&tmp, 256)) != hipSuccess) return e;
    auto touch = [&](hipStream_t st) { hipError_t r = hipMemsetAsync(tmp, 0, 64, st); return r != hipSuccess ? r : hipStreamSynchronize(st); };
    for (hipStream_t st : {c->stream, c->stream_b, c->stream2, c->stream3, c->stream_list})
        if ((e = touch(st)) != hipSuccess) { (void)hipFree(tmp); return e; }
    const bool dbg = getenv("SWIMM_HIP_DEBUG") != nullptr || getenv("SWIMM_HIP_DEBUG_STREAMS") != nullptr;
    std::vector<hipStream_t> aside;
    hipStream_t best = nullptr;
    for (int attempt = 0; attempt < 8 && e == hipSuccess; ++attempt) {
        hipStream_t cand = nullptr;
        if ((e = hipStreamCreate(&cand)) != hipSuccess || (e = touch(cand)) != hipSuccess) { if (cand) aside.push_back(cand); break; }
        bool shares = false;
        for (hipStream_t launch : {c->stream, c->stream_b, c->stream2}) {
            if ((e = launch_spin(400, launch)) != hipSuccess) break;
            const double t0 = now_s();
            if ((e = touch(cand)) != hipSuccess) break;
            const double dt = now_s() - t0;
            if ((e = hipStreamSynchronize(launch)) != hipSuccess) break;
            if (dt > 250e-6) { shares = true; break; }
        }
        if (e != hipSuccess) { aside.push_back(cand); break; }
        if (dbg) fprintf(stderr, "swimm_hip: upload stream candidate %d %s a hardware queue with a launch stream\n", attempt, shares ? "shares" : "does not share");
        if (!shares) { best = cand; break; }
        aside.push_back(cand);
    }
    if (e == hipSuccess && !best) {          // (every normal-class queue is behind a launch stream: a class of its own after all)
        e = hipStreamCreateWithPriority(&best, hipStreamDefault, lo);
        if (e == hipSuccess) e = touch(best);
    }
    for (hipStream_t st : aside) (void)hipStreamDestroy(st);
    (void)hipFree(tmp);
    c->stream_up = best;
    return e;
}

int swimm_hip_create(int device, swimm_hip_ctx **out)
{
    if (!out) return fail("swimm_hip_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= virtual_gpus(n)) return fail("swimm_hip_create: device %d not in [0,%d)", device, virtual_gpus(n));
    const int vdevice = device;
    device %= std::max(n, 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("swimm_hip_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    swimm_hip_ctx *c = new swimm_hip_ctx();
    c->device = device;
    c->vdevice = vdevice;
    g_cur_vdevice = vdevice;
    c->num_cu = prop.multiProcessorCount;
    if (make_streams(c) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail3, hipEventDisableTiming) != hipSuccess ||
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
    (void)ctx_enter(c);
    delete c->up; c->up = nullptr;
    swimm_hip_clear_db(c);
    pool_trim(c);
    for (hipEvent_t e : c->part_ev) (void)hipEventDestroy(e);
    c->d_scores.release(); c->d_prof.release(); c->d_bnd.release(); c->d_bnd_b.release(); c->d_bnd_c.release(); c->d_avail.release(); c->d_stream_items.release();
    if (c->list_arena.base) { (void)hipFree(c->list_arena.base); c->list_arena = DevArena{}; } c->d_qcodes.release(); c->d_sub16.release(); c->d_qdesc.release(); c->d_wave_out.release();
    if (c->pin) { (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_cap = c->pin_used = 0; }
    if (c->up_pin) { (void)hipHostFree(c->up_pin); c->up_pin = nullptr; c->up_pin_cap = c->up_pin_used = 0; }
    c->d_gbase.release(); c->d_gvalid.release(); c->d_keys.release(); c->d_err.release(); c->tail_scratch.release(); c->tail_scratch_t[0].release(); c->tail_scratch_t[1].release(); c->rerun_scratch.release(); c->d_rerun_items.release(); c->d_satlist.release(); c->d_ladder_counts.release();
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
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
    if (c->stream_list) (void)hipStreamDestroy(c->stream_list);
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    if (c->ev_avail) (void)hipEventDestroy(c->ev_avail);
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

// one chunk, or one piece of a caller's chunk (own_disp: the piece's offsets, counted from its first byte)
static int add_chunk_piece(swimm_hip_ctx *c, const char *b, uint64_t vD, const uint16_t *n, const uint32_t *b_disp, std::vector<uint32_t> own_disp,
                           uint32_t group_count, uint32_t vl, uint64_t first_group)
{
    const uint32_t per = kGroupSeqs / vl;
    ChunkRec rec;
    rec.kind = 0;
    rec.own_disp = std::move(own_disp);
    if (!rec.own_disp.empty()) b_disp = rec.own_disp.data();      // (a vector's storage survives the moves of the record)
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
    if (register_chunk(c, rec, nullptr, 0)) return 1;
    if (c->opt_lazy_upload ? ensure_uploader(c) : upload_chunk(c, c->chunks.back())) return 1;
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
    if (ctx_enter(c)) return 1;
    // A chunk that will stream in (lazy_upload) is recorded in pieces of about upload_piece_kib (default 96 MiB, the
    // reference's chunk size): a caller that hands over its database as one 0.6 GB buffer still gets ranges that overlap
    // copy and alignment.  (Finer pieces do not pay: with 32 MiB the first kernel starts 1.5 ms sooner, but the search
    // takes seven ranges instead of four and ends no earlier -- 31.5 vs 31.0 ms through chunks, 36.4 vs 32.4 through slabs.)
    // Pieces are runs of whole device groups whose bytes are contiguous in the caller's buffer.
    const uint64_t piece = (uint64_t)c->opt_upload_piece_kib << 10;
    const uint32_t per = kGroupSeqs / vl;
    bool ascending = true;
    for (uint32_t g = 1; g < group_count; ++g) ascending = ascending && b_disp[g] >= b_disp[g - 1];
    if (!c->opt_lazy_upload || vD <= piece + piece / 2 || !ascending)
        return add_chunk_piece(c, b, vD, n, b_disp, {}, group_count, vl, first_group);
    for (uint32_t g0 = 0; g0 < group_count;) {
        uint32_t g1 = g0;
        uint64_t end = b_disp[g0];
        do {
            g1 = std::min(group_count, g1 + per);
            for (uint32_t g = g1 - std::min(per, g1 - g0); g < g1; ++g) end = std::max<uint64_t>(end, (uint64_t)b_disp[g] + (uint64_t)n[g] * vl);
        } while (g1 < group_count && end - b_disp[g0] < piece);
        std::vector<uint32_t> disp(g1 - g0);
        for (uint32_t g = g0; g < g1; ++g) disp[g - g0] = b_disp[g] - b_disp[g0];
        if (add_chunk_piece(c, b + b_disp[g0], end - b_disp[g0], n + g0, nullptr, std::move(disp), g1 - g0, vl, first_group + g0)) return 1;
        g0 = g1;
    }
    return 0;
}

// One walk over a slab's lengths, and a light one: per device group of 128 sequences the residues it holds and its longest
// member (two reductions over 128 uint16 the compiler vectorises; no per-sequence offsets -- the tiling kernel makes its own from
// the lengths -- and no second array of the slab's size: this runs inside the caller's "first search after a cold upload", 35 M
// sequences at the full Env-NR size, where round 4's first version -- offsets of every sequence, lengths widened to 32 bits -- took
// 33 ms, most of it first touches of 280 MB of fresh memory).
static void group_sums(const uint16_t *lengths, uint64_t n_seq, std::vector<uint32_t> &gsum, std::vector<uint32_t> &gmax)
{
    const uint64_t ng = (n_seq + kGroupSeqs - 1) / kGroupSeqs;
    gsum.resize(ng); gmax.resize(ng);
    for (uint64_t g = 0; g < ng; ++g) {
        const uint16_t *l = lengths + g * kGroupSeqs;
        const uint32_t n = (uint32_t)std::min<uint64_t>(kGroupSeqs, n_seq - g * kGroupSeqs);
        uint32_t sum = 0;
        uint16_t mx = 1;
        for (uint32_t i = 0; i < n; ++i) { sum += l[i]; mx = std::max(mx, l[i]); }
        gsum[g] = sum; gmax[g] = mx;
    }
}

static int add_sequences_piece(swimm_hip_ctx *c, const uint16_t *lengths, const char *codes, uint64_t n_seq, uint64_t first_seq, const uint32_t *gsum,
                               const uint32_t *gmax, uint8_t *tiled = nullptr, size_t tiled_cap = 0)
{
    ChunkRec rec;
    rec.kind = 1;
    rec.d_tiled = tiled; rec.tiled_cap = tiled_cap;      // (a piece of a slab: its share of the slab's one device buffer)
    rec.n_groups = (uint32_t)((n_seq + kGroupSeqs - 1) / kGroupSeqs);
    rec.goff.resize(rec.n_groups);
    rec.gcols.resize(rec.n_groups);
    rec.gsrc.resize((size_t)rec.n_groups + 1);
    uint64_t total = 0;
    for (uint32_t g = 0; g < rec.n_groups; ++g) {
        rec.gsrc[g] = (uint32_t)total;
        total += gsum[g];
        if (total > 0xFFFFFFF0ull) return fail("swimm_hip_add_sequences: slab larger than 4 GiB");
        rec.gcols[g] = (gmax[g] + kChunkCols - 1) / kChunkCols * kChunkCols;
    }
    rec.gsrc[rec.n_groups] = (uint32_t)total;
    if (ctx_enter(c)) return 1;
    rec.h_codes = codes; rec.code_bytes = total;
    rec.h_len = lengths;
    rec.first_seq = first_seq;
    rec.n_seq = n_seq;
    if (register_chunk(c, rec, lengths, n_seq)) return 1;
    if (c->opt_lazy_upload ? ensure_uploader(c) : upload_chunk(c, c->chunks.back())) return 1;
    return 0;
}

int swimm_hip_add_sequences(swimm_hip_ctx *c, const uint16_t *lengths, const char *codes, uint64_t n_seq, uint64_t first_seq)
{
    if (!c || !lengths || !codes) return fail("swimm_hip_add_sequences: NULL argument");
    if (n_seq == 0) return fail("swimm_hip_add_sequences: empty slab");
    if (n_seq > 0x7FFFFFFFull) return fail("swimm_hip_add_sequences: more than 2^31 sequences in one slab");
    struct Tm { swimm_hip_ctx *c; double t0; ~Tm() { c->add_seconds += now_s() - t0; } } tm{c, now_s()};
    std::vector<uint32_t> gsum, gmax;
    group_sums(lengths, n_seq, gsum, gmax);
    if (!c->opt_lazy_upload) return add_sequences_piece(c, lengths, codes, n_seq, first_seq, gsum.data(), gmax.data());
    // (lazy_upload: pieces of about upload_piece_kib, whole device groups each -- see swimm_hip_add_chunk)
    // ONE device buffer for the slab, shared by its pieces (the first piece owns it): a cold search of a 7e9-residue database is
    // 8 allocations instead of 74 -- and the same 8 sizes as an eager upload of the same slabs, so that a database that replaces
    // another finds its buffers in the pool (the 74 pieces of c4 did not: up to 1 s of hipMalloc / hipFree inside the first search)
    const uint64_t piece = (uint64_t)c->opt_upload_piece_kib << 10;
    const uint64_t ng = gsum.size();
    uint64_t tiled_total = 0;
    for (uint64_t g = 0; g < ng; ++g) tiled_total += (uint64_t)((gmax[g] + kChunkCols - 1) / kChunkCols * kChunkCols) * kGroupSeqs;
    if (ctx_enter(c)) return 1;
    uint8_t *slab = nullptr;
    size_t slab_cap = 0;
    if (pool_alloc(c, std::max<uint64_t>(tiled_total, 16), (void **)&slab, &slab_cap)) return 1;
    uint64_t off = 0, tiled_off = 0;
    for (uint64_t g0 = 0; g0 < ng;) {
        uint64_t g1 = g0, bytes = 0;
        do { bytes += gsum[g1]; ++g1; } while (g1 < ng && bytes < piece);
        if (ng - g1 < 4 && bytes < piece + piece / 2)          // (no sliver at the end)
            for (; g1 < ng; ++g1) bytes += gsum[g1];
        const uint64_t s0 = g0 * kGroupSeqs, s1 = std::min<uint64_t>(n_seq, g1 * kGroupSeqs);
        if (add_sequences_piece(c, lengths + s0, codes + off, s1 - s0, first_seq + s0, gsum.data() + g0, gmax.data() + g0, slab + tiled_off, g0 == 0 ? slab_cap : 0)) {
            if (g0 == 0) (void)hipFree(slab);          // (nobody owns it yet)
            return 1;
        }
        for (uint64_t g = g0; g < g1; ++g) tiled_off += (uint64_t)((gmax[g] + kChunkCols - 1) / kChunkCols * kChunkCols) * kGroupSeqs;
        off += bytes;
        g0 = g1;
    }
    return 0;
}

int swimm_hip_clear_db(swimm_hip_ctx *c)
{
    if (!c) return fail("swimm_hip_clear_db: NULL ctx");
    if (ctx_enter(c)) return 1;
    (void)hipDeviceSynchronize();                 // nothing in flight may still read the chunks
    pool_trim(c);                                  // (whatever an earlier database left and nobody took)
    for (auto &ch : c->chunks) {
        if (ch.d_tiled && ch.tiled_cap) c->pool.push_back({(void *)ch.d_tiled, ch.tiled_cap});      // (cap 0: a share of another piece's buffer)
        if (ch.d_len) c->pool.push_back({(void *)ch.d_len, ch.len_cap});
        if (ch.ready) (void)hipEventDestroy(ch.ready);
    }
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
    if (ctx_enter(c)) return 1;
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
    if (ctx_enter(c)) return 1;
    const double t0 = now_s();
    reset_stats(c);
    const uint32_t qtotal = (uint32_t)c->qm.size(), B = query_batch(c);
    if (qtotal == 0) return fail("swimm_hip_search_topr: no queries set");
    typedef std::pair<int32_t, int64_t> Hit;   // larger pair first == score desc, then larger index first (utils.c:12,52)
    for (uint32_t qb = 0; qb < qtotal; qb += B) {
        const uint32_t qe = std::min(qtotal, qb + B), qn = qe - qb;
        uint64_t S = 0;
        if (search_device(c, qb, qe, &S)) return 1;
        if (getenv("SWIMM_HIP_DEBUG")) fprintf(stderr, "swimm_hip: search_topr: scores on the device %.3f ms after the call\n", (now_s() - t0) * 1e3);
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
            // (blocks per query: enough to fill the chip over the whole batch; every block's 64 candidates come back to the host,
            // so a batch of 1 000 queries takes 8 per query, not 256)
            const int n_blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(8, 4096 / qn), std::min<uint64_t>(256, S / 1024)));
            HIP_TRY(c->d_gbase.reserve(gbase.size()));
            HIP_TRY(c->d_gvalid.reserve(gvalid.size()));
            HIP_TRY(c->d_keys.reserve((size_t)qn * n_blocks * 64));
            HIP_TRY(hipMemcpyAsync(c->d_gbase.p, gbase.data(), gbase.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gvalid.p, gvalid.data(), gvalid.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(launch_topk64(c->d_scores.p, S, c->d_gbase.p, c->d_gvalid.p, c->d_keys.p, n_blocks, qn, c->stream));
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
    if (getenv("SWIMM_HIP_DEBUG")) fprintf(stderr, "swimm_hip: search_topr: lists selected %.3f ms after the call\n", (now_s() - t0) * 1e3);
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
    if (qp.sp) { snprintf(buf, buf_len, "swimm::sw_sp_kernel(swimm::SpParams)"); return 0; }
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
    if (ctx_enter(c)) return 1;          // (some options release the cached work lists)
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
    } else if (!strcmp(key, "tail_cap")) {
        if (value < 0 || value > 1000) return fail("tail_cap must be 0..1000 (per mille of the cells; 0 = no cap)");
        c->opt_tail_cap = value;
        release_plans(c);
    } else if (!strcmp(key, "sp_threshold")) {
        if (value < 0 || value > 65536) return fail("sp_threshold must be 0 (every query through the score-profile kernel) .. 65536 (none)");
        c->opt_sp_threshold = value;
    } else if (!strcmp(key, "cut")) {
        if (value < 0 || value > 1000) return fail("cut must be 0 (never) .. 1000 (tenths: cost of a lane-systolic cell against a padded pipeline cell)");
        c->opt_cut = value;
        release_plans(c);
    } else if (!strcmp(key, "stack")) {
        c->opt_stack = value != 0;
    } else if (!strcmp(key, "resident")) {
        if (value < -1 || value > 1) return fail("resident must be -1 (auto), 0 or 1");
        c->opt_resident = value;
    } else if (!strcmp(key, "time_launches")) {
        c->opt_time_launches = value != 0;
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
        if (c->opt_lazy_upload && ensure_uploader(c)) return 1;      // (the uploader thread starts -- and warms up -- now, not inside the first search)
    } else if (!strcmp(key, "upload_piece_kib")) {
        if (value < 16) return fail("upload_piece_kib must be >= 16");
        c->opt_upload_piece_kib = value;
    } else if (!strcmp(key, "wg_limit")) {
        if (value < 0) return fail("wg_limit must be >= 0 (0 = as many workgroups as the chip holds)");
        c->opt_wg_limit = value;
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
    std::vector<swimm_hip_ctx *> done(num_gpus, nullptr);      // torn down after the clock has stopped (returning tens of GB to the driver takes about a second)
    std::vector<std::thread> th;
    for (int g = 0; g < num_gpus; ++g) {
        th.emplace_back([&, g]() {   // one host thread per device, as MICsearch.c:53
            if (shard[g].empty()) return;
            swimm_hip_ctx *ctx = nullptr;
            auto bail = [&]() { errs[g] = swimm_hip_last_error(); if (ctx) swimm_hip_destroy(ctx); };
            // this thread, and the uploader thread its context starts, on the CPUs local to the device (best effort: a host
            // without the sysfs files gets the even share, a failure to bind is not a failure to search)
            (void)swimm_hip_bind_host_thread(g, num_gpus, nullptr, 0);
            if (swimm_hip_create(g, &ctx)) return bail();
            // the caller's chunks outlive this call: stream them in while the search runs (X2 overlapped with compute)
            if (swimm_hip_set_option(ctx, "lazy_upload", 1)) return bail();
            if (swimm_hip_set_queries(ctx, query_sequences, query_sequences_lengths, query_disp, query_sequences_count,
                                      submat, open_gap, extend_gap)) return bail();
            for (uint32_t ci : shard[g])
                if (swimm_hip_add_chunk(ctx, chunk_b[ci], chunk_vD[ci], chunk_n[ci], chunk_b_disp[ci],
                                        chunk_vect_sequences_db_count[ci], vl, accum[ci])) return bail();
            if (swimm_hip_search(ctx, scores, stride, nullptr)) return bail();
            done[g] = ctx;
        });
    }
    for (auto &t : th) t.join();
    if (workTime) *workTime = now_s() - t0;              // transfers + kernels + the scores' way back, like MICsearch.c:51,350
    for (swimm_hip_ctx *ctx : done) if (ctx) swimm_hip_destroy(ctx);
    for (int g = 0; g < num_gpus; ++g)
        if (!errs[g].empty()) return fail("GPU %d: %s", g, errs[g].c_str());
    return 0;
}

}  // extern "C"
