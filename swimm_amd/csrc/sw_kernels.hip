// sw_kernels.hip -- gfx950 (MI355X / CDNA4) device code of the SWIMM search hot path.
//
// What the reference computes (CPUsearch.c:605-668, MICsearch.c:163-210): for every
// (query, database sequence) pair the Gotoh affine-gap local-alignment score
//     H = max(0, Hdiag + S(q_i, d_j), E, F);  E = max(E - ge, H - goe);  F = max(F - ge, H - goe)
// keeping only max H.  Inter-task SIMD: one vector lane owns one database sequence.
//
// How it is mapped here (not a translation of the SSE/AVX2/KNC loops):
//   * one wavefront lane owns TWO database sequences, packed 2 x 16 bit in every VGPR, so one wave aligns 128
//     sequences.  First tier: packed binary16 integers (v_pk_fma_f16 / v_pk_add_f16 / v_pk_maximum3_f16; exact below
//     f16_exact_below(extend), 1920 for extend 2: the pipeline kernel stores values with column offsets, cell2_ofs);
//     alignments that reach that limit are re-run as packed int16 (v_pk_add_i16 clamp / v_pk_max_i16 /
//     v_pk_sub_u16 clamp), those that reach 32767 in int32 -- the reference's int8 -> int16 -> int32 ladder
//     (CPUsearch.c:678-957) one rung higher: CDNA4 has no packed int8 VALU;
//   * sw_pipe_kernel: a workgroup is a systolic pipeline of W waves over the QUERY: wave k owns query rows
//     [k*T, (k+1)*T) in registers (H and E per row), walks the database columns in chunks of 4, and hands the
//     bottom row (H, F per column) to wave k+1 through an LDS ring, one chunk behind.  Only when the query is
//     longer than W*T rows does a boundary row go through HBM, once per pass (the "strip" traffic of SURVEY.md 8d
//     with T_eff = W*T);
//   * the substitution lookup is a query profile staged in LDS, prof[d][row]: one ds_read_b128 fetches the scores
//     of 8 consecutive query rows for a lane's residue d and the two sequences' halves are combined by v_perm_b32
//     (integer tiers), or -- binary16 tier -- the (score, 1.0) dwords of 4 rows, the pair being formed inside the
//     v_pk_fma_f16 that adds the diagonal (pair_score_plus).  The 25 code rows sit an odd number of 16-byte units
//     apart, so two residues share LDS banks only if their (device) codes are equal mod 16;
//   * workgroups are persistent and align device groups back to back as ONE continuous column stream (the pipeline
//     fills and drains once per launch); which group comes next is decided by a longest-first queue (first two
//     rounds dealt, then one global cursor) -- see the kernel;
//   * sw_lane_kernel: the other axis of parallelism, one wave per alignment with the lanes along the query, for
//     the few very long sequences and for the promotion re-runs.
//
// In the packed-int16 tier E and F are kept clamped at >= 0.  That is exact: H already has a 0 floor, so replacing
// E by max(E, 0) (and F likewise) never changes any H, and max(0, max(E,0) - ge, H - goe) equals
// max(0, E - ge, H - goe) for ge >= 0.  It buys unsigned-saturating subtracts (no separate max with 0).
#include "sw_kernels.h"

namespace swimm {

// diagnostic build (make stamps): s_memtime sums per pipeline segment; never part of the product library
#ifdef SWIMM_STAMPS
#define STAMP(var)                                                                       \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u32(v2s x) { return __builtin_bit_cast(uint32_t, x); }

// ---- cell arithmetic, packed 2 x int16 ---------------------------------------------------
struct OpsPK {
    typedef v2s V;
    static __device__ __forceinline__ V zero() { return (V)(0); }
    static __device__ __forceinline__ V splat(int x) { return (V)((short)x); }
    static __device__ __forceinline__ V from_bits(uint32_t x) { return as_v2s(x); }
    static __device__ __forceinline__ uint32_t bits(V x) { return as_u32(x); }
    static __device__ __forceinline__ V adds(V a, V b) { return __builtin_elementwise_add_sat(a, b); }
    static __device__ __forceinline__ V vmax(V a, V b) { return __builtin_elementwise_max(a, b); }
    // unsigned saturating subtract: operands are >= 0, result clamps at 0
    static __device__ __forceinline__ V subz(V a, V b) { return (V)__builtin_elementwise_sub_sat((v2u)a, (v2u)b); }
};

// ---- cell arithmetic, int32 (promotion tier) ----------------------------------------------
struct OpsI32 {
    typedef int V;
    static __device__ __forceinline__ V zero() { return 0; }
    static __device__ __forceinline__ V splat(int x) { return x; }
    static __device__ __forceinline__ V from_bits(uint32_t x) { return (int)x; }
    static __device__ __forceinline__ uint32_t bits(V x) { return (uint32_t)x; }
    static __device__ __forceinline__ V adds(V a, V b) { return a + b; }
    static __device__ __forceinline__ V vmax(V a, V b) { return a > b ? a : b; }
    static __device__ __forceinline__ V subz(V a, V b) { int d = a - b; return d > 0 ? d : 0; }
};

// ---- cell arithmetic, packed 2 x f16: exact while every H <= 2047 (integers up to 2048 are exact in
// binary16; a sum that would leave that range makes some H >= 2048, which the host detects on the final
// score and re-runs in int16).  Buys v_pk_maximum3_f16 (gfx950): 8.5 instead of 10 VALU ops per 2 cells.  This plain form
// serves the lane-systolic kernel; the pipeline kernel's binary16 tier uses cell2_ofs below (6.5).
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
struct OpsF16 {
    typedef v2h V;
    static __device__ __forceinline__ V zero() { return (V)((_Float16)0.0f); }
    static __device__ __forceinline__ V splat(int x) { return (V)((_Float16)(float)x); }
    static __device__ __forceinline__ V from_bits(uint32_t x) { return __builtin_bit_cast(V, x); }
    static __device__ __forceinline__ uint32_t bits(V x) { return __builtin_bit_cast(uint32_t, x); }
    static __device__ __forceinline__ V vmax(V a, V b) { return __builtin_elementwise_max(a, b); }
    static __device__ __forceinline__ V max3(V a, V b, V c)
    {
        V d;
        asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
};

template <class Ops>
__device__ __forceinline__ void cell(typename Ops::V &hd, typename Ops::V &Hr, typename Ops::V &Er,
                                     typename Ops::V &F, typename Ops::V &best, typename Ops::V S,
                                     typename Ops::V goe, typename Ops::V ge)
{
    typedef typename Ops::V V;
    V t = Ops::adds(hd, S);           // Hdiag + S            (CPUsearch.c:622)
    hd = Hr;                          // next row's diagonal = this row's previous column
    V h = Ops::vmax(Ops::vmax(t, Er), F);   // max with E and F (>= 0 already: CPUsearch.c:624-626)
    Hr = h;
    best = Ops::vmax(best, h);        // CPUsearch.c:636
    // opaque to the optimiser: otherwise it re-associates the running max into one big reduction
    // at the end of the chunk and keeps all 4*T h values alive (60+ VGPR spills)
    asm("" : "+v"(best));
    V u = Ops::subz(h, goe);          // H - (open+extend)    (CPUsearch.c:630)
    Er = Ops::vmax(Ops::subz(Er, ge), u);   // CPUsearch.c:628,631
    F = Ops::vmax(Ops::subz(F, ge), u);     // CPUsearch.c:629,632
}

// two consecutive query rows of one column.  goe / ge are the gap penalties in the form the tier wants
// (f16: already negated).
template <class Ops>
__device__ __forceinline__ void cell2(typename Ops::V &hd, typename Ops::V &H0, typename Ops::V &E0, typename Ops::V &H1,
                                      typename Ops::V &E1, typename Ops::V &F, typename Ops::V &best, typename Ops::V S0,
                                      typename Ops::V S1, typename Ops::V goe, typename Ops::V ge)
{
    cell<Ops>(hd, H0, E0, F, best, S0, goe, ge);
    cell<Ops>(hd, H1, E1, F, best, S1, goe, ge);
}

template <>
__device__ __forceinline__ void cell2<OpsF16>(v2h &hd, v2h &H0, v2h &E0, v2h &H1, v2h &E1, v2h &F, v2h &best, v2h S0, v2h S1,
                                              v2h ngoe, v2h nge)
{
    const v2h z = OpsF16::zero();
    v2h t0 = hd + S0;                         // Hdiag + S
    hd = H0;
    v2h h0 = OpsF16::max3(t0, E0, F);         // E, F >= 0, so this is max(0, ...) too
    H0 = h0;
    v2h u0 = h0 + ngoe;                       // H - (open+extend)
    E0 = OpsF16::max3(E0 + nge, u0, z);
    F = OpsF16::max3(F + nge, u0, z);
    v2h t1 = hd + S1;
    hd = H1;
    v2h h1 = OpsF16::max3(t1, E1, F);
    H1 = h1;
    v2h u1 = h1 + ngoe;
    E1 = OpsF16::max3(E1 + nge, u1, z);
    F = OpsF16::max3(F + nge, u1, z);
    best = OpsF16::max3(best, h0, h1);        // one instruction for two rows
    asm("" : "+v"(best));
}

// The pipeline kernel's binary16 tier in COLUMN-OFFSET form: 6.5 packed ops per 2 cells instead of 7.5 + 1 v_perm_b32.
// Every value of column j is stored with o_j = (j mod P) * ge added (H* = H + o_j, E* = E + o_j on entering column j,
// F' = F + o_j + ge), and the profile is staged as S' = S + ge.  Then
//     t* = Hdiag* + S'                 (the diagonal came from column j-1: o_{j-1} + ge = o_j; one v_pk_fma_f16, pair_score_plus)
//     a  = F' - ge                     (= F + o_j)
//     h* = max3(t*, E*, a)             (a >= o_j because F' >= fl = o_j + ge: the max with 0)
//     u' = h* - (goe - ge)
//     E* <- max(E*, u')                (E + o_{j+1}: the decay by ge is the offset's growth -- ONE op instead of two; E may
//                                       fall below the floor, h* takes its floor from a)
//     F' <- max3(a, u', fl)
// and the running best lives in offset space too (best* = best + o_j: max3 over two rows as before, + ge once per column).
// Every P columns (f16_renorm_chunks) the registers H*, E* and the diagonal are taken back by P * ge.  Exactness: all
// stored values are integers <= max(H) + P * ge (the running best, which carries the coming column's offset; t* never exceeds
// the new h*), so a result below f16_exact_below(ge) = 2048 - P * ge is exact, and the first value that leaves the exact
// range makes a result >= that threshold -- the host re-runs those as packed int16, as before.
__device__ __forceinline__ v2h pk_max_f16(v2h a, v2h b)
{
    v2h d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));       // (asm: the builtin first canonicalises operands that came out of asm)
    return d;
}
// t* for a lane's pair of sequences in ONE instruction.  The LDS profile of this tier holds a dword (S' , 1.0) per row and code;
// xa / xb are the dwords of the pair's two residues:  lo = xa.lo * xb.hi + hd.lo = S'_a + hd.lo,  hi = xa.hi * xb.lo + hd.hi =
// S'_b + hd.hi  (products by 1.0 and sums of small integers: exact).  It replaces the v_perm_b32 that interleaved two lookups
// of a two-rows-per-dword profile, and the add: 6.5 ops per two cells instead of 7.5, for twice the LDS read traffic.
__device__ __forceinline__ v2h pair_score_plus(uint32_t xa, uint32_t xb, v2h hd)
{
    v2h t;
    asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(xa), "v"(xb), "v"(hd));
    return t;
}
__device__ __forceinline__ void cell2_ofs(v2h &hd, v2h &H0, v2h &E0, v2h &H1, v2h &E1, v2h &Fp, v2h &best, uint32_t xa0, uint32_t xb0,
                                          uint32_t xa1, uint32_t xb1, v2h ngo, v2h nge, v2h fl)
{
    v2h a0 = Fp + nge;
    v2h t0 = pair_score_plus(xa0, xb0, hd);
    hd = H0;
    v2h h0 = OpsF16::max3(t0, E0, a0);
    H0 = h0;
    v2h u0 = h0 + ngo;
    E0 = pk_max_f16(E0, u0);
    Fp = OpsF16::max3(a0, u0, fl);
    v2h a1 = Fp + nge;
    v2h t1 = pair_score_plus(xa1, xb1, hd);
    hd = H1;
    v2h h1 = OpsF16::max3(t1, E1, a1);
    H1 = h1;
    v2h u1 = h1 + ngo;
    E1 = pk_max_f16(E1, u1);
    Fp = OpsF16::max3(a1, u1, fl);
    best = OpsF16::max3(best, h0, h1);
    asm("" : "+v"(best));
}

__host__ __device__ constexpr size_t round16(size_t x) { return (x + 15) & ~(size_t)15; }
// +16: the 25 code rows start 16 bytes (mod 256) apart, so codes d and d' share LDS banks for
// ds_read_b128 only when d == d' (mod 16)
__host__ __device__ constexpr int prof_row_bytes(int rows) { return rows * 2 + 16; }
// the pipeline kernel's binary16 tier: a dword (score, 1.0) per row (pair_score_plus), four rows per ds_read_b128; the code rows
// start an ODD number of 16-byte units apart, so that two codes share banks only when they are equal mod 16
// (ds_read_b64 lookups -- 32 bank pairs, no two codes on the same one, no bank conflicts at all -- were measured: c2 8 540 against
// 9 680 GCUPS, every shape 8-15 % slower; only ds_read_b128 reaches the LDS's full rate)
__host__ __device__ constexpr int prof_row_bytes_f16(int rows) { return rows * 4 + ((rows / 4) % 2 ? 32 : 16); }

// a wave's strip of T rows occupies round8(T) rows of the LDS profile, so that its ds_read_b128 stay 16-byte aligned
__host__ __device__ constexpr int strip_lds_rows(int T) { return (T + 7) & ~7; }
// LDS of one workgroup: query profile | hand-over ring (2 chunk slots per wave) | control words
constexpr int kSeqRing = 32;      // item ids of the workgroup's sequence, published by wave 0 (dynamic mode)
// group-resident passes: per wave, a lane-linear landing area for the next pass's strip of the profile (filled by
// global_load_lds, i.e. without registers)
__host__ __device__ constexpr int strip_stage_dwords(int T) { return (kCodes * (T / 2) + 63) / 64 * 64; }
size_t pipe_lds_bytes(Mode mode, int T, int W, bool resident)
{
    const size_t prof = mode == Mode::F16 ? (size_t)kCodes * prof_row_bytes_f16(T * W) : (size_t)kCodes * prof_row_bytes(strip_lds_rows(T) * W);
    return round16(prof) + (size_t)W * 2 * kChunkCols * 64 * sizeof(uint2) + (kSeqRing + 4) * 4 + (resident ? (size_t)W * strip_stage_dwords(T) * 4 : 0);
}

constexpr uint32_t kNoItem = 0xFFFFFFFFu;

// an item descriptor through the constant address space: the index is wave-uniform and the list is read-only for the
// whole launch, so this is one s_load_dwordx8 into scalar registers
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
static_assert(sizeof(Item) == 32, "Item is loaded as eight dwords");
__device__ __forceinline__ Item load_item(const Item *items, uint32_t idx)
{
    const u32x8 r = *(const __attribute__((address_space(4))) u32x8 *)(uintptr_t)(items + idx);
    Item it;
    it.db = (const uint8_t *)(uintptr_t)(((uint64_t)r[1] << 32) | r[0]);
    it.ncols = r[2]; it.seq0 = r[3]; it.half = r[4]; it.out_slot = r[5];
    it.bnd_off = ((uint64_t)r[7] << 32) | r[6];
    return it;
}

static_assert(sizeof(QDesc) == 32, "QDesc is loaded as eight dwords");
__device__ __forceinline__ QDesc load_qdesc(const QDesc *q, uint32_t idx)
{
    const u32x8 r = *(const __attribute__((address_space(4))) u32x8 *)(uintptr_t)(q + idx);
    QDesc d;
    d.prof_off = r[0]; d.prof_stride = r[1]; d.passes = r[2]; d.out_off = r[3]; d.seam_mask = r[4]; d.wave_tab = r[5];
    return d;
}
// one entry of a read-only table, wave-uniform index: a scalar load
__device__ __forceinline__ uint32_t load_u32_uniform(const uint32_t *t, uint32_t idx)
{
    return *(const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(t + idx);
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t x)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ const uint8_t *uniform_ptr(const uint8_t *q) { return (const uint8_t *)(uintptr_t)uniform_u64((uint64_t)(uintptr_t)q); }

// A search that streams its database in for one query: the launch walks the whole item list in upload order while the parts
// are still travelling.  Wave 0 calls this before it hands out item g: it polls the count the upload stream publishes behind
// every part's tiling kernel (agent-scope loads, a sleep between polls) and takes an agent-scope acquire -- this CU's L1 is
// invalidated -- when the count has grown; the other waves load the item's bytes after the step barrier that follows the
// hand-out (MI355X_MICROARCH.md, inter-workgroup visibility: "consumer: poll, ONE agent acquire, barrier, plain loads"; the
// producer side is a kernel boundary).  The wait is bounded (about a second without the count moving past the item): an
// upload that fails publishes kAvailAbort (bit 8 of *err: the host discards the results); a wait that runs out sets bit 16
// and the workgroup takes no further items -- the host then aligns the database once it is resident (search_device).  That
// second case is not expected to happen: it would mean the tiling kernels cannot run beside this launch (the host picks a
// launch shape that leaves them registers on every CU, issue_one_list), and must not become a hang.
// `seen` caches the last count read (wave-uniform).
__device__ __forceinline__ bool wait_landed(const uint32_t *avail, uint32_t g, uint32_t &seen, uint32_t *err)
{
    if (seen == kAvailAbort) return false;           // (gave up before: no second wait)
    if (g < seen) return true;
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v == kAvailAbort) { atomicOr(err, 8u); seen = kAvailAbort; return false; }
        seen = v;
        if (g < v) break;
        if (spins > (1u << 20)) { atomicOr(err, 16u); seen = kAvailAbort; return false; }
        __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// (text shared by the two places where the pipeline kernel requests its next chunk; uses the kernel's local state)
#define SWIMM_REQUEST_NEXT_CHUNK() \
{ \
                const uint8_t *ndb = dbp; \
                uint32_t ncc = cc + 1, nhalf = half; \
                have_next = true; \
                if (RES && ncc == nch && pass + 1 < passes) { \
                    ncc = 0; \
                } else if (ncc == nch) { \
 \
                    if (DYN) { \
                        next_it = __builtin_amdgcn_readfirstlane(seq[(n + 1) & (kSeqRing - 1)]); \
                        if (k == 0 && next_it == kNoItem && lane == 0) *total_lds = c + 1; \
                    } else { \
                        next_it = it + 1 < it_end ? it + 1 : kNoItem; \
                    } \
                    if (next_it != kNoItem) { \
                        const Item niv = load_item(p.items, RES ? next_it / nq : next_it); \
                        ndb = niv.db; ncc = 0; nhalf = niv.half; \
                    } else { \
                        have_next = false; \
                    } \
                } \
                if (have_next) { \
                    if (PK) { \
                        const unsigned long long w = *(const __attribute__((address_space(1))) unsigned long long *)(uintptr_t)(ndb + ((size_t)ncc * 64 + lane) * 8); \
                        nwa = (uint32_t)w; nwb = (uint32_t)(w >> 32); \
                    } else { \
                        nwa = *(const __attribute__((address_space(1))) uint32_t *)(uintptr_t)(ndb + ((size_t)ncc * 64 + lane) * 8 + nhalf * 4); \
                    } \
                } \
            }

// M: 0 = packed int16, 1 = int32 (one sequence per lane), 2 = packed f16
// DYN: false = the workgroup walks the item range the host gave it (static partition, longest first onto the
//      least-loaded workgroup); true = wave 0 pulls the next item from a global cursor over the list sorted longest
//      first and publishes its id to the other waves through LDS.  Same schedule when every CU runs at the same
//      speed; when some CUs are slowed down (lane-systolic waves of the tail kernel share their SIMDs) the dynamic
//      queue keeps all workgroups busy to the end.
// The waves of a workgroup hand chunks over through an LDS ring with one workgroup barrier per chunk.  (Counters
// between neighbouring waves instead of the barrier were tried and lost 1.5 %, see DESIGN.md.)
// RES: group-resident passes.  A query longer than W*T rows needs several passes over every group; with RES the
//      workgroup takes a group through ALL its passes back to back, as one continuous stream of item-passes, before it
//      moves on: the strip boundary of a pass (the last wave's bottom row per column) goes to a scratch area that only
//      this workgroup touches and comes back, to wave 0, one group length later -- while it is still cached -- and no
//      pass waits for the slowest workgroup of the one before it (no launch boundary).  Every wave re-stages the
//      profile rows of its own strip when the pass changes, so the pipeline never drains.  A group shorter than the
//      pipeline (fewer chunks than waves) idles to the pipeline's depth between two of its passes: wave 0 must not
//      start pass p+1 of a column before the last wave has finished pass p of it.
// GROW: the item list is still landing (PipeParams::avail, wait_landed above): wave 0 waits for an item's turn before it hands
//      the item out.  An instantiation of its own (binary16 tier, dynamic queue, one pass), so that the kernels a resident
//      database runs are exactly the ones their register budgets were tuned for.
template <int T, int M, bool DYN, bool RES, bool GROW = false>
__global__ void __launch_bounds__(T > 28 ? 768 : 1024) sw_pipe_kernel(const PipeParams p)
{
    static_assert(T % 4 == 0 && T >= 8, "strips are multiples of 4 rows");
    static_assert(!GROW || (DYN && !RES && M == 2), "a growing item list: dynamic queue, one launch per pass, binary16 tier");
    constexpr bool OFS = M == 2;          // binary16 tier: column-offset form, one profile dword per row (cell2_ofs)
    constexpr int TP = OFS ? T : strip_lds_rows(T);
    constexpr int RB = OFS ? 4 : 2;       // bytes per row of the LDS profile
    constexpr bool PK = M != 1;
    typedef typename std::conditional<M == 0, OpsPK, typename std::conditional<M == 1, OpsI32, OpsF16>::type>::type Ops;
    typedef typename Ops::V V;
    constexpr int C = kChunkCols;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int W = blockDim.x >> 6;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index = query strip
    const int lane = threadIdx.x & 63;
    const int RW = W * TP;
    const int PS = OFS ? prof_row_bytes_f16(RW) : prof_row_bytes(RW);
    unsigned char *prof_lds = smem;
    uint2 *ring = (uint2 *)(smem + round16((size_t)kCodes * PS));
    uint32_t *seq = (uint32_t *)(ring + (size_t)W * 2 * C * 64);     // seq[n % kSeqRing]: id of the workgroup's n-th item
    int *total_lds = (int *)(seq + kSeqRing);                          // chunks in the workgroup's whole sequence, once known

    // stage this pass's window of the query profile: rows [r0, r0 + W*T) of all 25 codes, strip k at LDS row k*TP
    // (group-resident passes: every wave stages its own strip, pass by pass, below)
    if (!RES) {
        const int dw_per_code = RW >> 1;
        for (int idx = threadIdx.x; idx < kCodes * dw_per_code; idx += blockDim.x) {
            const int d = idx / dw_per_code, x = idx - d * dw_per_code;
            const uint32_t *src = (const uint32_t *)(p.prof + (size_t)d * p.prof_stride + p.r0);
            int sx = x;
            if (TP != T) {
                const int strip = (2 * x) / TP, r = 2 * x - strip * TP;
                if (r >= T) continue;                // alignment rows, never read
                sx = (strip * T + r) >> 1;
            }
            uint32_t v = src[sx];
            if (M == 2) {                           // int16 scores -> (binary16 of score + ge, 1.0) per row (cell2_ofs, pair_score_plus)
                const v2s sv = as_v2s(v);
                *(uint2 *)(prof_lds + d * PS + x * 8) = make_uint2(__builtin_bit_cast(uint32_t, (v2h){(_Float16)(float)(sv.x + p.ge), (_Float16)1.0f}),
                                                                   __builtin_bit_cast(uint32_t, (v2h){(_Float16)(float)(sv.y + p.ge), (_Float16)1.0f}));
                continue;
            }
            *(uint32_t *)(prof_lds + d * PS + x * 4) = v;
        }
    }
    uint32_t it, it_end = 0;      // current item id; static mode: end of this workgroup's range
    // dynamic mode, wave 0: every workgroup knows its next item while it aligns the current one (the first two are
    // dealt below).  When it finishes an item it pulls the one after the next from the global cursor: the atomic is
    // issued at the end of that step and its result is published to seq[] at the top of the next step, so the
    // returned value is only live across the barrier, not across the column loop.  (Pulling at the START of an item
    // dealt a third round at time 0, in the wrong order: the workgroup with the longest first group got the longest
    // third one -- 35 % on a 1e8-residue database.)
    uint32_t pending = 0, pub = 2;
    bool pending_valid = false;
    uint32_t landed = 0;          // streaming list (p.avail): items known to be on the device (wave 0)
    if (DYN) {
        // the first two rounds are dealt like LPT deals them when all workgroups start together: item b to
        // workgroup b, then the next gridDim.x items in reverse (the longest first item gets the shortest second
        // one); everything after that comes from the cursor.  (Letting every workgroup pull its first two items
        // from the cursor paired the two longest items on workgroup 0: -10 % on a small database.)
        if (threadIdx.x < 64) {       // (wave 0, uniform: lane 0 writes)
            uint32_t g0 = blockIdx.x, g1 = 2 * gridDim.x - 1 - blockIdx.x;
            if (g0 >= p.n_items) g0 = kNoItem;
            if (g1 >= p.n_items) g1 = kNoItem;
            if (GROW) {                                // the dealt items may still be on their way
                if (g0 != kNoItem && !wait_landed(p.avail, g0, landed, p.err)) g0 = g1 = kNoItem;
                if (g1 != kNoItem && !wait_landed(p.avail, g1, landed, p.err)) g1 = kNoItem;
            }
            if (threadIdx.x == 0) {
                seq[0] = g0;
                seq[1] = g1;
                *total_lds = g0 != kNoItem ? 0x3fffffff : 0;
            }
        }
        __syncthreads();
        it = seq[0];
    } else {
        __syncthreads();
        it = p.wg_first[blockIdx.x];
        it_end = p.wg_first[blockIdx.x + 1];
        if (it >= it_end) it = kNoItem;
    }
    it = __builtin_amdgcn_readfirstlane(it);
    int total = DYN ? 0x3fffffff : (int)p.wg_chunks[blockIdx.x];

    const V goe = Ops::splat(M == 2 ? -p.goe : p.goe), ge = Ops::splat(M == 2 ? -p.ge : p.ge);
    const V ngo = Ops::splat(-(p.goe - p.ge)), pge = Ops::splat(p.ge);          // OFS: -(open), +extend
    const uint32_t renorm_chunks = OFS ? (uint32_t)f16_renorm_chunks(p.ge) : 0u;
    const V nren = Ops::splat(-(int)(renorm_chunks * kChunkCols) * p.ge);          // OFS: what a renormalisation takes back
    V off = Ops::zero(), fl = pge;                                                // OFS: o_j and o_j + ge of the coming column
    uint32_t since = 0;                                                           // OFS: chunks of this item since the last renormalisation
    const V left = OFS ? ge : Ops::zero();                                        // column -1: H = 0, stored with o_{-1} = -ge
    const unsigned char *my_prof = prof_lds + k * TP * RB;

    V H[T], E[T];
    V best = Ops::zero(), diag_top = left;
#pragma unroll
    for (int r = 0; r < T; ++r) { H[r] = left; E[r] = Ops::zero(); }

    uint32_t cc = 0, nch = 0, seq0 = 0, half = 0, n = 0, next_it = kNoItem;
    uint32_t pass = 0, len = 0;            // RES: pass of the current item; steps the item-pass occupies (>= nch)
    // RES: an item is a (group, query) pair -- item id v = group rank * n_queries + (n_queries - 1 - query), queries longest
    // first within a group -- so one launch takes a whole batch of queries that share the launch shape; a single multi-pass
    // query is the batch of one.  Per query: where its profile lies, how many passes it needs, where its scores go.
    uint32_t passes = 1, cur_q = 0, q_stride = RES ? 0 : p.prof_stride;
    const int16_t *q_prof = p.prof;
    int32_t *q_out = p.out;
    // a stack of short queries shares the workgroup: this wave starts a member (zero top boundary), its best goes to its member's row
    bool seam = false;
    if (!RES && p.wave_out != nullptr) {
        seam = (p.seam_mask >> k) & 1u;
        q_out = p.out + load_u32_uniform(p.wave_out, p.wave_tab + (uint32_t)k);
    }
    int staged_win = -1;                   // RES: which (query << 16 | pass) window this wave's strip of the LDS profile holds
    // RES: the strip's profile rows of the NEXT window are requested during the last chunk of the current item-pass, straight
    // into a per-wave landing area in LDS (global_load_lds: no registers), so that the switch itself is a handful of LDS
    // reads and writes.  (Fetching them at the switch stalled the wave for an L2 round trip and, through the step barrier,
    // the whole workgroup: W stalls per item-pass, -3 % on c2, -10 % with 16 waves.)
    constexpr int NPF = strip_stage_dwords(T) / 64;
    uint32_t *const stage = (uint32_t *)(total_lds + 4) + (size_t)k * strip_stage_dwords(T);
    int pf_win = -1;
    auto fetch_strip = [&](const int16_t *prof, uint32_t stride, uint32_t ps, int win) {
        const uint32_t row0 = ps * (uint32_t)(W * T) + (uint32_t)(k * T);
        int ln = lane;
        asm volatile("" : "+v"(ln));     // opaque: otherwise the per-lane addresses are hoisted out of the main loop and held in 2 x NPF registers
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            int idx = ln + 64 * i;
            if (idx >= kCodes * (T / 2)) idx = 0;        // (landing slots past the strip's last dword are never read)
            const int d = idx / (T / 2), x = idx - d * (T / 2);
            // An LDS-DMA the compiler does not see: with the builtin, hipcc (ROCm 7.2) drains the vector-memory counter in
            // front of every later LDS read and every barrier while the DMA might be in flight -- the step's database
            // prefetch with it, every step (-9 % with 8-wave workgroups).  Hidden, the DMA is waited for where it is
            // consumed (the explicit s_waitcnt vmcnt(0) at the window switch); an operation the compiler does not count
            // can only make its own counted waits stricter, never weaker (the counter retires loads in order).
            const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(stage + 64 * i);
            uint32_t m0_saved;      // (M0 is a reserved register for the compiler: put back what it held)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(prof + (size_t)d * stride + row0 + 2 * x), "s"(lds_addr) : "memory");
        }
        pf_win = win;
    };
    const uint32_t nq = RES ? p.n_queries : 1u;
    uint64_t bnd_off = 0;
    const uint8_t *dbp = nullptr;
    uint32_t nwa = 0, nwb = 0;     // residues of the wave's next chunk, loaded one step ahead
    bool have_next = false;

#ifdef SWIMM_STAMPS
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, sumA = 0, sumB = 0, sumC = 0, sumD = 0, nact = 0;
    unsigned long long tA1 = 0, tA2 = 0, sumA1 = 0, sumA2 = 0, sumA3 = 0;      // the prologue in three parts: item start | residues, boundary, ring | next-chunk request
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
    for (int s = 0;; ++s) {
        if (DYN) total = __builtin_amdgcn_readfirstlane(*(volatile int *)total_lds);   // same value in every wave: written before the last barrier
        if (s >= total + W - 1) break;
        if (DYN && (uint32_t)s > p.max_steps) {   // cannot happen (one workgroup can at most take every chunk of the list): never spin forever
            if (threadIdx.x == 0) atomicOr(p.err, 4u);
            break;
        }
        if (DYN && k == 0 && pending_valid) {
            uint32_t g = __builtin_amdgcn_readfirstlane(pending) + 2 * gridDim.x;   // the cursor starts behind the two dealt rounds
            if (g >= p.n_items) g = kNoItem;
            if (GROW && g != kNoItem && !wait_landed(p.avail, g, landed, p.err)) g = kNoItem;
            if (lane == 0) seq[pub & (kSeqRing - 1)] = g;
            ++pub;
            pending_valid = false;
        }
        bool pull = false;
        // (the step's prologue -- item bookkeeping, boundary and ring reads -- runs at the priority the last column left, 0.  Raising it to 3
        // here was measured in round 3, A/B in one process: no effect on any shape -- c2 27.17 vs 27.11 ms, the group-resident kernel with
        // 8-wave workgroups 7 590 vs 7 574 GCUPS)
        const int c = s - k;                      // chunk index of this wave in the workgroup's sequence
        STAMP(tA);
        if (c >= 0 && it != kNoItem) {            // wave-uniform
            if (cc == 0) {                        // first chunk of a new item (RES: item-pass): reset the DP state
                // the item is the same for the whole wave: one scalar load, descriptor in scalar registers
                const uint32_t vi = __builtin_amdgcn_readfirstlane(it);
                const uint32_t gi = RES ? vi / nq : vi;
                const Item iv = load_item(p.items, gi);
                nch = iv.ncols / C; dbp = iv.db; seq0 = iv.seq0;
                half = iv.half;
                bnd_off = RES ? (uint64_t)blockIdx.x * p.bnd_wg_cols : iv.bnd_off;
                if (RES && pass == 0) {               // a new (group, query) item: the query's parameters, one scalar load
                    cur_q = nq - 1 - (vi - gi * nq);
                    const QDesc qd = load_qdesc(p.qdesc, cur_q);
                    passes = qd.passes; q_prof = p.prof + qd.prof_off; q_stride = qd.prof_stride; q_out = p.out + qd.out_off;
                    seam = false;
                    if (qd.wave_tab != kNoTab) {
                        seam = (qd.seam_mask >> k) & 1u;
                        q_out = p.out + load_u32_uniform(p.wave_out, qd.wave_tab + (uint32_t)k);
                    }
                }
                // (wave 0 may read a column's boundary two steps after the last wave stored it at the earliest: the storing wave
                // drains its stores at the top of its next step, below)
                len = (RES && pass + 1 < passes && nch < (uint32_t)W + 1) ? (uint32_t)W + 1 : nch;
                best = Ops::zero(); diag_top = left;
#pragma unroll
                for (int r = 0; r < T; ++r) { H[r] = left; E[r] = Ops::zero(); }
                off = Ops::zero(); fl = pge; since = 0;
                const int win = (int)((cur_q << 16) | pass);
                if (RES && staged_win != win) {
                    // this wave's strip of the profile for this window: rows pass*W*T + k*T .. + T of the query's 25 codes.
                    // Only this wave reads that part of the LDS profile, so nobody has to be waited for.
                    if (pf_win != win) fetch_strip(q_prof, q_stride, pass, win);   // (first item of the workgroup: nothing was requested ahead)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows have landed (an LDS-DMA counts as a vector-memory operation)
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
#pragma unroll
                    for (int i = 0; i < NPF; ++i) {
                        const int idx = ln + 64 * i;
                        const int d = idx / (T / 2), x = idx - d * (T / 2);
                        uint32_t v = *(volatile uint32_t *)(stage + idx);
                        if (M == 2) {
                            const v2s sv = as_v2s(v);
                            if (idx < kCodes * (T / 2))
                                *(uint2 *)(prof_lds + d * PS + (k * TP + 2 * x) * 4) =
                                    make_uint2(__builtin_bit_cast(uint32_t, (v2h){(_Float16)(float)(sv.x + p.ge), (_Float16)1.0f}),
                                               __builtin_bit_cast(uint32_t, (v2h){(_Float16)(float)(sv.y + p.ge), (_Float16)1.0f}));
                        } else
                        if (idx < kCodes * (T / 2)) *(uint32_t *)(prof_lds + d * PS + (k * TP + 2 * x) * 2) = v;
                    }
                    staged_win = win;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            STAMP(tA1);
            if (!RES || cc < nch) {               // (RES: a short group idles here to the pipeline's depth between two passes)
            // database residues of this chunk: 4 columns of the lane's sequence(s).  They were requested one
            // step ago (below): right after the barrier every wave would otherwise stall on this global load
            // before it can form its first LDS address, with nothing else on the SIMD to cover it.
            uint32_t wa, wb = 0;
            if (!have_next) {
                if (PK) {
                    // dbp came out of a descriptor in memory: tell the compiler it is global memory (no flat_load)
                    const unsigned long long w = *(const __attribute__((address_space(1))) unsigned long long *)(uintptr_t)(dbp + ((size_t)cc * 64 + lane) * 8);
                    wa = (uint32_t)w; wb = (uint32_t)(w >> 32);
                } else {
                    wa = *(const __attribute__((address_space(1))) uint32_t *)(uintptr_t)(dbp + ((size_t)cc * 64 + lane) * 8 + half * 4);
                }
                asm volatile("" : "+v"(wa), "+v"(wb));      // (first chunk of a workgroup's sequence only: wait for it here, not in the column loop)
            } else {
                wa = nwa; wb = nwb;
            }
            // top boundary of the strip for these columns: H of the row above, F entering row 0
            uint2 bin[C];
            const bool first_pass = GROW ? true : RES ? pass == 0 : (bool)p.first_pass, last_pass = GROW ? true : RES ? pass + 1 == passes : (bool)p.last_pass;     // (GROW: one pass)
            const bool zero_top = (k == 0 || seam) && (first_pass || seam);      // no row above: H = F = 0 (OFS: o_j and o_j + ge, set column by column)
            if (k == 0 || seam) {
                if (first_pass || seam) {
#pragma unroll
                    for (int jj = 0; jj < C; ++jj) bin[jj] = make_uint2(0u, 0u);
                } else {
                    if (RES) {
                        // written by this workgroup's last wave one group length ago: the step barriers order the two,
                        // and the load goes to L2 (sc1), past whatever this CU's L1 still holds of the previous pass
#pragma unroll
                        for (int jj = 0; jj < C; ++jj) {
                            const unsigned long long v = __hip_atomic_load((const unsigned long long *)(p.bnd + (bnd_off + (uint64_t)cc * C + jj) * 64 + lane),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bin[jj] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                        }
                    } else {
#pragma unroll
                        for (int jj = 0; jj < C; ++jj) bin[jj] = p.bnd[(bnd_off + (uint64_t)cc * C + jj) * 64 + lane];
                    }
                }
            } else {
                const uint2 *src = ring + (size_t)(((k - 1) * 2 + (c & 1)) * C) * 64 + lane;
#pragma unroll
                for (int jj = 0; jj < C; ++jj) bin[jj] = src[jj * 64];
            }
            if (RES && k == W - 1 && !last_pass) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous step's boundary stores have retired
            STAMP(tA2);
            // Request the next chunk of this wave's stream (same item, or the first chunk of the next item).  Which item follows?
            // static: the next of the range; dynamic: the id wave 0 published in seq[] (a chunk that finds none is the last of the
            // workgroup's sequence).  AFTER the boundary loads, so that waiting for those (older, and retired in order) leaves this
            // one in flight; it is waited for at the END of the step (below), when it has long arrived, so that nothing in the
            // column loop depends on a load: the next chunk's HBM latency is covered by this chunk's arithmetic.  (Round 1 issued
            // the request first and used its register at once in the next step: the compiler put s_waitcnt vmcnt(0) in front of
            // the column loop and every wave sat out its own prefetch, every step.)
            SWIMM_REQUEST_NEXT_CHUNK();
            STAMP(tB);
            if constexpr (OFS) {
                // the offsets have grown by ge per column for renorm_chunks chunks: take them back (2 T + 1 ops every 4 * renorm_chunks columns)
                if (since == renorm_chunks) {
#pragma unroll
                    for (int r = 0; r < T; ++r) { H[r] = H[r] + nren; E[r] = E[r] + nren; }
                    diag_top = diag_top + nren; best = best + nren;
                    off = Ops::zero(); fl = pge; since = 0;
                }
                ++since;
            }
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                // The SIMD issues its OLDEST ready wave first, so waves that start a chunk together finish it one
                // after the other and the last one runs alone (5.7 instead of 4.4 cycles per instruction).  A wave
                // that is ahead lowers its own priority column by column, which keeps the SIMD's waves abreast.
                if (jj == 0) __builtin_amdgcn_s_setprio(3); else if (jj == 1) __builtin_amdgcn_s_setprio(2);
                else if (jj == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                const uint32_t da = (wa >> (8 * jj)) & 0xffu;
                const unsigned char *pa = my_prof + da * PS;
                const unsigned char *pb = pa;
                if (PK) pb = my_prof + ((wb >> (8 * jj)) & 0xffu) * PS;
                V hd = diag_top;
                diag_top = Ops::from_bits(bin[jj].x);
                V F = Ops::from_bits(bin[jj].y);
                if constexpr (OFS) {
                    // (a real branch, wave-uniform: as a select it is two v_cndmask_b32 per column for every wave; setting the four
                    // columns' values up ahead of the loop costs eight registers, which the 28-row kernel does not have)
                    if (zero_top) { asm volatile(""); diag_top = off; F = fl; }
                }
                if constexpr (OFS) {
#pragma unroll
                    for (int g = 0; g < T / 4; ++g) {
                        // one ds_read_b128 = the (score, 1.0) dwords of 4 consecutive query rows for one of the lane's two residues
                        const uint4 a = *(const uint4 *)(pa + g * 16);
                        const uint4 b = *(const uint4 *)(pb + g * 16);
                        cell2_ofs(hd, H[4 * g], E[4 * g], H[4 * g + 1], E[4 * g + 1], F, best, a.x, b.x, a.y, b.y, ngo, ge, fl);
                        cell2_ofs(hd, H[4 * g + 2], E[4 * g + 2], H[4 * g + 3], E[4 * g + 3], F, best, a.z, b.z, a.w, b.w, ngo, ge, fl);
                        // How far ahead the scheduler may hoist a column's lookups (each is 4 registers until it is used), measured per strip
                        // height (profiles/r03_plan_sweep.txt): unfenced up to 24 rows; the taller strips gain from a fence after every group of 4
                        // rows (28 rows: 8 620 -> 10 220 GCUPS -- unfenced it spills; 32 rows: 10 100 -> 10 450 and 16 fewer registers, a fourth wave
                        // per SIMD), 36 rows from one after every other group.
                        constexpr int kFence = (T == 28 || T == 32 || (GROW && T == 24)) ? 1 : T == 36 ? 2 : 0;      // (GROW, 24 rows: fenced to fit 120 registers, see launch_grow)
                        if (kFence && (g + 1) % (kFence ? kFence : 1) == 0 && g + 1 < T / 4) __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                for (int r8 = 0; r8 < T / 8; ++r8) {
                    // one ds_read_b128 = the scores of 8 consecutive query rows for this lane's residue
                    const uint4 a = *(const uint4 *)(pa + r8 * 16);
                    const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
                    if (PK) {
                        const uint4 b = *(const uint4 *)(pb + r8 * 16);
                        const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // (A_r, B_r) pairs: low / high int16 of the two lookups
                            const int r = r8 * 8 + q * 2;
                            cell2<Ops>(hd, H[r], E[r], H[r + 1], E[r + 1], F, best,
                                       Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x05040100u)),
                                       Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x07060302u)), goe, ge);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = r8 * 8 + q * 2;
                            cell2<Ops>(hd, H[r], E[r], H[r + 1], E[r + 1], F, best, Ops::from_bits((uint32_t)(int)(short)(aw[q] & 0xffffu)),
                                       Ops::from_bits((uint32_t)((int)aw[q] >> 16)), goe, ge);
                        }
                    }
                }
                if (T % 8 == 4) {                    // last 4 rows of the strip: one ds_read_b64 per residue
                    constexpr int rb = T - 4;
                    const uint2 a = *(const uint2 *)(pa + rb * 2);
                    const uint32_t aw[2] = {a.x, a.y};
                    if (PK) {
                        const uint2 b = *(const uint2 *)(pb + rb * 2);
                        const uint32_t bw[2] = {b.x, b.y};
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r = rb + q * 2;
                            cell2<Ops>(hd, H[r], E[r], H[r + 1], E[r + 1], F, best,
                                       Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x05040100u)),
                                       Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x07060302u)), goe, ge);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r = rb + q * 2;
                            cell2<Ops>(hd, H[r], E[r], H[r + 1], E[r + 1], F, best, Ops::from_bits((uint32_t)(int)(short)(aw[q] & 0xffffu)),
                                       Ops::from_bits((uint32_t)((int)aw[q] >> 16)), goe, ge);
                        }
                    }
                }
                }   // (two-rows-per-dword profile of the integer tiers)
                // bottom boundary of this column: to the next wave through LDS, or (last wave, more passes) to HBM
                const uint2 bout = make_uint2(Ops::bits(H[T - 1]), Ops::bits(F));
                if (k < W - 1) ring[(size_t)((k * 2 + (c & 1)) * C + jj) * 64 + lane] = bout;
                else if (!last_pass) p.bnd[(bnd_off + (uint64_t)cc * C + jj) * 64 + lane] = bout;
                if constexpr (OFS) {
                    best = best + pge;                           // the best so far, in the next column's offset space
                    off = fl; fl = fl + pge;                     // o_{j+1} = o_j + ge
                }
                // keep one column's lookups in flight at a time: without this fence the scheduler hoists
                // all four columns' LDS reads and the kernel needs ~230 VGPRs (spills at 3 waves/SIMD)
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP(tC);
            if (have_next) asm volatile("" : "+v"(nwa), "+v"(nwb));   // the next chunk's residues have arrived (requested before the column loop)
            }   // active step
            const bool scored = !RES ? cc + 1 == nch : cc + 1 == nch;
            if (scored) {   // item(-pass) finished: every strip contributes its best (CPUsearch.c:670-676)
                if (M == 2) {
                    const v2h b2 = __builtin_bit_cast(v2h, Ops::bits(best - off));      // (best* carries the coming column's offset)
                    atomicMax(q_out + seq0 + 2 * lane, (int)(float)b2.x);          // a lane's pair = two neighbours of the sorted database
                    atomicMax(q_out + seq0 + 2 * lane + 1, (int)(float)b2.y);
                } else if (PK) {
                    const v2s b2 = __builtin_bit_cast(v2s, Ops::bits(best));
                    atomicMax(q_out + seq0 + 2 * lane, (int)b2.x);
                    atomicMax(q_out + seq0 + 2 * lane + 1, (int)b2.y);
                } else {
                    atomicMax(q_out + seq0 + 2 * lane + half, (int)Ops::bits(best));
                }
            }
            if (++cc >= len) {
                cc = 0;
                if (RES && pass + 1 < passes) {
                    ++pass;                       // same group, next pass
                } else {
                    pass = 0;
                    ++n;
                    it = next_it;
                    pull = true;
                }
            }
            if (RES && cc != 0 && cc + 1 == nch) {
                // The coming step is the last chunk of this item-pass: request the profile rows of the window that follows -- the
                // next pass of this item, or pass 0 of the next item's query (wave 0 published its id long ago) -- by LDS-DMA.
                // At the END of the step: the wait hipcc puts in front of the next LDS read then finds nothing else young in flight.
                if (pass + 1 < passes) {
                    const int nwin = (int)((cur_q << 16) | (pass + 1));
                    if (nwin != staged_win) fetch_strip(q_prof, q_stride, pass + 1, nwin);
                } else {
                    const uint32_t nv = DYN ? __builtin_amdgcn_readfirstlane(seq[(n + 1) & (kSeqRing - 1)]) : (it + 1 < it_end ? it + 1 : kNoItem);
                    if (nv != kNoItem) {
                        const uint32_t nqi = nq - 1 - (nv - (nv / nq) * nq);
                        const int nwin = (int)(nqi << 16);
                        if (nwin != staged_win) {
                            const QDesc nqd = load_qdesc(p.qdesc, nqi);
                            fetch_strip(p.prof + nqd.prof_off, nqd.prof_stride, 0, nwin);
                        }
                    }
                }
            }
#ifdef SWIMM_STAMPS
            STAMP(tD);
            sumA += tB - tA; sumB += tC - tB; sumC += tD - tC; nact++;
            if (tA2 >= tA1 && tA1 >= tA && tB >= tA2) { sumA1 += tA1 - tA; sumA2 += tA2 - tA1; sumA3 += tB - tA2; }
#endif
        }
        if (DYN && k == 0 && pull) {             // one pull per item finished: the workgroup that finishes first gets the longest group left
            if (lane == 0) pending = atomicAdd(p.queue, 1u);
            pending_valid = true;
        }
#ifdef SWIMM_STAMPS
        { unsigned long long t0, t1; STAMP(t0); __syncthreads(); STAMP(t1); sumD += t1 - t0; }
#else
        __syncthreads();
#endif
    }
#ifdef SWIMM_STAMPS
    if (p.stamps && lane == 0) {
        atomicAdd(p.stamps + k * 8 + 0, sumA); atomicAdd(p.stamps + k * 8 + 1, sumB); atomicAdd(p.stamps + k * 8 + 2, sumC);
        atomicAdd(p.stamps + k * 8 + 3, sumD); atomicAdd(p.stamps + k * 8 + 4, nact);
        atomicAdd(p.stamps + k * 8 + 5, sumA1); atomicAdd(p.stamps + k * 8 + 6, sumA2); atomicAdd(p.stamps + k * 8 + 7, sumA3);
        if (k == 0) {   // when do workgroups end?  (100 MHz wall clock, same on every CU)
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            atomicAdd(p.stamps + 15 * 8 + 0, t - t_begin); atomicMax(p.stamps + 15 * 8 + 1, t - t_begin);
            atomicMin(p.stamps + 15 * 8 + 2, t_begin); atomicMax(p.stamps + 15 * 8 + 3, t); atomicAdd(p.stamps + 15 * 8 + 4, 1ull);
            if (blockIdx.x < 1024) { p.stamps[128 + blockIdx.x] = t; p.stamps[128 + 1024 + blockIdx.x] = (unsigned long long)total; p.stamps[128 + 2048 + blockIdx.x] = t_begin; }
        }
    }
#endif
}

template <int T, int M, bool DYN, bool RES>
static hipError_t launch_one(int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    const size_t lds = pipe_lds_bytes(M == 0 ? Mode::PK16 : M == 1 ? Mode::I32 : Mode::F16, T, W, RES);
    hipError_t e = hipFuncSetAttribute((const void *)sw_pipe_kernel<T, M, DYN, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_pipe_kernel<T, M, DYN, RES>), dim3(n_wg), dim3(W * 64), lds, s, p);
    return hipGetLastError();
}

template <int T, bool DYN, bool RES>
static hipError_t launch_mode(Mode mode, int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    if (mode == Mode::PK16) return launch_one<T, 0, DYN, RES>(W, n_wg, p, s);
    if (mode == Mode::I32) return launch_one<T, 1, DYN, RES>(W, n_wg, p, s);
    return launch_one<T, 2, DYN, RES>(W, n_wg, p, s);
}

// Instantiations: T = 16 / 24 / 32 for every tier; the f16 tier (the default path) also has every other multiple
// of 4 from 8 to 36, so that the launch plan can give a query W = 4, 8, 12 or 16 waves (an equal number on each of
// the CU's 4 SIMDs) with at most 3 padding rows per wave.  Group-resident passes exist for the dynamic queue only.
#define SWIMM_EXTRA_T(X) X(8) X(12) X(20) X(28) X(36)
bool pipe_has_variant(Mode mode, int T)
{
    if (T == 16 || T == 24 || T == 32) return true;
    if (mode != Mode::F16) return false;
#define X(t) if (T == t) return true;
    SWIMM_EXTRA_T(X)
#undef X
    return false;
}

template <bool DYN, bool RES>
static hipError_t launch_any(Mode mode, int T, int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    if (T == 32) return launch_mode<32, DYN, RES>(mode, W, n_wg, p, s);
    if (T == 24) return launch_mode<24, DYN, RES>(mode, W, n_wg, p, s);
    if (T == 16) return launch_mode<16, DYN, RES>(mode, W, n_wg, p, s);
#define X(t) if (T == t) return launch_one<t, 2, DYN, RES>(W, n_wg, p, s);
    SWIMM_EXTRA_T(X)
#undef X
    return hipErrorInvalidValue;
}

template <int T>
static hipError_t launch_grow_one(int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    const size_t lds = pipe_lds_bytes(Mode::F16, T, W, false);
    hipError_t e = hipFuncSetAttribute((const void *)sw_pipe_kernel<T, 2, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sw_pipe_kernel<T, 2, true, false, true>), dim3(n_wg), dim3(W * 64), lds, s, p);
    return hipGetLastError();
}

// the launch of a streaming search's one item list (PipeParams::avail): binary16 tier, dynamic queue, one pass
static hipError_t launch_grow(int T, int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    if (T == 32) return launch_grow_one<32>(W, n_wg, p, s);
    if (T == 24) return launch_grow_one<24>(W, n_wg, p, s);
    if (T == 16) return launch_grow_one<16>(W, n_wg, p, s);
#define X(t) if (T == t) return launch_grow_one<t>(W, n_wg, p, s);
    SWIMM_EXTRA_T(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_pipe(Mode mode, int T, int W, int n_wg, const PipeParams &p, hipStream_t s)
{
    if (W < 1 || W > kMaxWaves || n_wg < 1 || !pipe_has_variant(mode, T)) return hipErrorInvalidValue;
    if (T > 28 && W > 12) return hipErrorInvalidValue;    // __launch_bounds__ of those instantiations
    const bool dyn = p.queue != nullptr;
    if (p.avail != nullptr) return dyn && p.qdesc == nullptr && mode == Mode::F16 ? launch_grow(T, W, n_wg, p, s) : hipErrorInvalidValue;
    if (p.qdesc != nullptr) return dyn && p.n_queries > 0 ? launch_any<true, true>(mode, T, W, n_wg, p, s) : hipErrorInvalidValue;
    return dyn ? launch_any<true, false>(mode, T, W, n_wg, p, s) : launch_any<false, false>(mode, T, W, n_wg, p, s);
}

template <int T, bool DYN, bool RES>
static const void *kernel_ptr(Mode mode)
{
    if (mode == Mode::PK16) return (const void *)sw_pipe_kernel<T, 0, DYN, RES>;
    if (mode == Mode::I32) return (const void *)sw_pipe_kernel<T, 1, DYN, RES>;
    return (const void *)sw_pipe_kernel<T, 2, DYN, RES>;
}

template <bool DYN, bool RES>
static const void *kernel_ptr_t(Mode mode, int T)
{
    if (!pipe_has_variant(mode, T)) return nullptr;
    if (T == 32) return kernel_ptr<32, DYN, RES>(mode);
    if (T == 24) return kernel_ptr<24, DYN, RES>(mode);
    if (T == 16) return kernel_ptr<16, DYN, RES>(mode);
#define X(t) if (T == t) return (const void *)sw_pipe_kernel<t, 2, DYN, RES>;
    SWIMM_EXTRA_T(X)
#undef X
    return nullptr;
}

hipError_t pipe_kernel_attributes(Mode mode, int T, bool resident, int *num_regs)
{
    hipFuncAttributes a;
    const void *f = resident ? kernel_ptr_t<true, true>(mode, T) : kernel_ptr_t<true, false>(mode, T);
    if (!f) return hipErrorInvalidValue;
    hipError_t e = hipFuncGetAttributes(&a, f);
    if (e == hipSuccess) *num_regs = a.numRegs;
    return e;
}

// registers of the instantiation that walks a growing item list (launch_grow)
hipError_t grow_kernel_attributes(int T, int *num_regs)
{
    const void *f = nullptr;
    if (T == 32) f = (const void *)sw_pipe_kernel<32, 2, true, false, true>;
    if (T == 24) f = (const void *)sw_pipe_kernel<24, 2, true, false, true>;
    if (T == 16) f = (const void *)sw_pipe_kernel<16, 2, true, false, true>;
#define X(t) if (T == t) f = (const void *)sw_pipe_kernel<t, 2, true, false, true>;
    SWIMM_EXTRA_T(X)
#undef X
    if (!f) return hipErrorInvalidValue;
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, f);
    if (e == hipSuccess) *num_regs = a.numRegs;
    return e;
}

// the code object's own (mangled) name of the instantiation a launch plan uses: what rocprofv3 lists, demangled
const char *pipe_kernel_symbol(Mode mode, int T, bool dynamic, bool resident)
{
    const void *f = resident ? kernel_ptr_t<true, true>(mode, T) : dynamic ? kernel_ptr_t<true, false>(mode, T) : kernel_ptr_t<false, false>(mode, T);
    return f ? hipKernelNameRefByPtr(f, nullptr) : nullptr;
}

// ---- score-profile kernel ------------------------------------------------------------------------
// The reference's second lookup technique (K7, MICsearch.c:257-313; `-p S`, and the long queries of `-p A -u N`, swimm.c:81-85):
// instead of looking every substitution score up by (query row, database residue), a table sp[query residue q][column][lane] =
// (S(q, residue of sequence A), S(q, residue of sequence B)) is built once per chunk of database columns, and every query row
// reads ITS residue's line of it, lane-linear, no permute.  Here: one wave = one workgroup aligns a device group (128 sequences,
// packed binary16 pairs) against kSpRows query rows per pass, the strip boundary between passes through HBM exactly as in the
// pipeline kernel; the chunk's table is 24 x 4 x 64 dwords = 24 KB of LDS.  On gfx950 the technique cannot pay (DESIGN.md
// section 6b.4: building the table costs ~96 instructions per column and workgroup, reading it saves one v_perm_b32 per row
// pair, and a workgroup has too few rows in flight to amortise the difference); it exists because the reference's interface has
// it, selected by the option "sp_threshold", and it is exact like every other path (tests/test_gpu_parity.py).
__global__ void __launch_bounds__(64) sw_sp_kernel(const SpParams p)
{
    constexpr int T = kSpRows, C = kChunkCols;
    __shared__ uint32_t sp[24 * C * 64];
    __shared__ __attribute__((aligned(16))) uint16_t sub[kCodes * kSpSubStride];     // [database residue][query residue]: a residue's 24 scores are three 16-byte reads
    const int lane = threadIdx.x;
    for (int i = lane; i < kCodes * kSpSubStride / 2; i += 64) ((uint32_t *)sub)[i] = ((const uint32_t *)p.sub16)[i];
    uint32_t qoff[T];                     // the pass's query rows: where each row's residue starts in the table (wave-uniform)
#pragma unroll
    for (int r = 0; r < T; ++r) qoff[r] = __builtin_amdgcn_readfirstlane((uint32_t)(uint8_t)p.qcodes[p.r0 + r]) * (uint32_t)(C * 64);
    __syncthreads();
    const v2h ngoe = OpsF16::splat(-p.goe), nge = OpsF16::splat(-p.ge);
    for (;;) {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(p.queue, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= p.n_items) break;
        const Item iv = load_item(p.items, it);
        const uint32_t nch = iv.ncols / C;
        v2h H[T], E[T];
#pragma unroll
        for (int r = 0; r < T; ++r) { H[r] = OpsF16::zero(); E[r] = OpsF16::zero(); }
        v2h best = OpsF16::zero(), diag_top = OpsF16::zero();
        for (uint32_t cc = 0; cc < nch; ++cc) {
            const unsigned long long w = *(const __attribute__((address_space(1))) unsigned long long *)(uintptr_t)(iv.db + ((size_t)cc * 64 + lane) * 8);
            const uint32_t wa = (uint32_t)w, wb = (uint32_t)(w >> 32);
            uint2 bin[C];
#pragma unroll
            for (int jj = 0; jj < C; ++jj) bin[jj] = p.first_pass ? make_uint2(0u, 0u) : p.bnd[(iv.bnd_off + (uint64_t)cc * C + jj) * 64 + lane];
            // the chunk's score profile: for each of the 24 query residues the pair of scores against this lane's two residues
            // (a residue's 24 scores: three ds_read_b128 per sequence; the pairs are formed two query residues at a time)
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                const uint32_t da = (wa >> (8 * jj)) & 0xffu, db = (wb >> (8 * jj)) & 0xffu;
                const uint4 *ra = (const uint4 *)(sub + da * kSpSubStride), *rb = (const uint4 *)(sub + db * kSpSubStride);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint4 a = ra[k], b = rb[k];
                    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int q = (k * 4 + w) * 2;
                        sp[(q * C + jj) * 64 + lane] = __builtin_amdgcn_perm(bw[w], aw[w], 0x05040100u);
                        sp[((q + 1) * C + jj) * 64 + lane] = __builtin_amdgcn_perm(bw[w], aw[w], 0x07060302u);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                v2h hd = diag_top;
                diag_top = OpsF16::from_bits(bin[jj].x);
                v2h F = OpsF16::from_bits(bin[jj].y);
#pragma unroll
                for (int r = 0; r < T; r += 2)
                    cell2<OpsF16>(hd, H[r], E[r], H[r + 1], E[r + 1], F, best, OpsF16::from_bits(sp[qoff[r] + jj * 64 + lane]),
                                  OpsF16::from_bits(sp[qoff[r + 1] + jj * 64 + lane]), ngoe, nge);
                if (!p.last_pass) p.bnd[(iv.bnd_off + (uint64_t)cc * C + jj) * 64 + lane] = make_uint2(OpsF16::bits(H[T - 1]), OpsF16::bits(F));
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_wave_barrier();          // (the next chunk's table overwrites this one)
        }
        const v2h b2 = best;
        atomicMax(p.out + iv.seq0 + 2 * lane, (int)(float)b2.x);
        atomicMax(p.out + iv.seq0 + 2 * lane + 1, (int)(float)b2.y);
    }
}

hipError_t launch_sp(int n_wg, const SpParams &p, hipStream_t s)
{
    if (n_wg < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sw_sp_kernel, dim3(n_wg), dim3(64), 0, s, p);
    return hipGetLastError();
}

// ---- lane-systolic kernel ------------------------------------------------------------------------
// Same recurrence, other axis of parallelism: the 64 lanes of ONE wave are 64 consecutive strips of
// kLaneRows query rows of the SAME alignment (two database sequences packed per register in int16 mode,
// one in int32 mode).  Lane l works on column j - l; after every column the strip's bottom row
// (H, F), the column's residues + start/end flags, the running best and the item / column ids move
// one lane to the right with v_mov_b32_dpp wave_shr:1 (lane 0 takes the next column instead).
// Items stream through the lanes back to back, so the 64-step skew is paid once per wave, not per
// sequence.  One alignment is spread over a whole wave instead of sharing a wave with 127 others:
// a 35 000-residue sequence is a chain of 35 000 short steps on its own wave, and the few dozen
// such chains of a Swiss-Prot-shaped database run side by side instead of serially on one CU.
__device__ __forceinline__ uint32_t dpp_shr1(uint32_t prev, uint32_t lane0_value)
{
    // lanes 1..63 <- prev of lane-1 ; lane 0 keeps `lane0_value` (no source lane, bound_ctrl off)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)prev, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

constexpr uint32_t kFlagStart = 1u << 16, kFlagEnd = 1u << 17, kFlagReal = 1u << 18;   // kFlagReal: a column of an item (not pipeline fill/drain)

size_t lane_lds_bytes(int rows_per_lane) { return round16((size_t)kCodes * prof_row_bytes(64 * rows_per_lane)); }

// what lane 0 feeds into the pipeline for one chunk of 4 columns (wave-uniform)
struct LaneFeed {
    uint32_t wa, wb;        // residues of the 4 columns (sequence A / B)
    uint2 b[kChunkCols];    // top boundary (H, F) of the 4 columns, zero in the first pass (the chunk being computed: scalar registers)
    uint2 bv;               // ... as it is loaded: lane jj (0..3) holds column jj -- one register pair for the chunk in flight, not four
    uint32_t item, col0;    // item index, first column's index in the boundary buffer
    uint32_t half;          // int32 mode: which of wa / wb is the sequence
    uint32_t fbits;         // 1 = a real chunk of an item (not pipeline fill / drain), 2 = its first column starts the item, 4 = its last column ends it
};

// Passes of a long query (512 rows each) are CHAINED inside one launch: the workgroups of pass p take the
// items in the same order as those of pass p-1 and follow them through global memory.  The wave that runs
// (item, p) publishes, per column, the bottom row of its last lane into bnd[p & 1] with write-through agent-scope
// stores and, one chunk later and behind a full drain of its vector-memory counter, a per-item progress counter (form
// R1 of the guide's inter-workgroup hand-off: sc1 payload, s_waitcnt vmcnt(0), then the flag); the wave that runs
// (item, p+1) polls that counter and reads the rows with sc1 loads before it feeds the columns to its lane 0.  A
// 5 478-row query against a 35 000-residue sequence is then 11 waves a few hundred columns apart instead of 11
// launches one after the other.
// No deadlock: the grid never exceeds one workgroup per CU (so all of it becomes resident), blocks are
// pass-major (producers are dispatched first), every pass consumes the items in the same order and pass 0
// never waits; all spins are bounded and report through p.err.
// the scores of a lane's TR = 2 NW query rows for one residue: one ds_read_b128 / b64 / b32
template <int NW>
__device__ __forceinline__ void lane_prof_load(const unsigned char *q, uint32_t (&w)[NW])
{
    if (NW == 4) { const uint4 v = *(const uint4 *)q; w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else if (NW == 2) { const uint2 v = *(const uint2 *)q; w[0] = v.x; w[1] = v.y; }
    else { w[0] = *(const uint32_t *)q; }
}

// TR = query rows per lane: 8 (512 rows per pass, chained passes for longer queries); 4 and 2 for queries of up to 256 /
// 128 rows, whose step -- a serial walk down the lane's rows -- is then that much shorter, and with it the time a
// 35 000-residue sequence holds up a short query.
// M: 0 = packed int16 pairs, 1 = int32 (one sequence), 2 = packed binary16 pairs (the first tier of the long-sequence
// tail: max3 makes a row 8.5 instead of 10 packed operations and its serial F chain three instead of four long;
// alignments that reach 2048 are re-run in int16 by the promotion ladder, like the pipeline kernel's).
template <int M, int TR>
__global__ void __launch_bounds__(256, 6) sw_lane_kernel(const LaneParams p)
{
    constexpr bool PK = M != 1;
    typedef typename std::conditional<M == 0, OpsPK, typename std::conditional<M == 1, OpsI32, OpsF16>::type>::type Ops;
    typedef typename Ops::V V;
    constexpr int C = kChunkCols, RP = 64 * TR, NW = TR / 2;   // NW dwords of profile per lane and residue
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int PS = prof_row_bytes(RP);
    const uint32_t bm = p.block_map[blockIdx.x];
    const uint32_t pass = bm & 0xffu;
    const LaneQ lq = p.lq[bm >> 8];
    const uint32_t r0 = pass * RP;
    const uint32_t rows = lq.m - r0 < (uint32_t)RP ? lq.m - r0 : (uint32_t)RP;
    const bool first_pass = pass == 0, last_pass = pass + 1 == lq.passes;
    uint32_t *const queue = p.queue + lq.queue0 + pass;
    const LaneItem *const items = p.items + lq.items0;
    const uint32_t n_items = lq.n_items;
    const uint32_t *const prog_in = p.prog + lq.prog0 + (size_t)(first_pass ? 0 : pass - 1) * n_items;   // only read when pass > 0
    uint32_t *const prog_out = p.prog + lq.prog0 + (size_t)pass * n_items;
    const unsigned long long *const bnd_in = p.bnd[(pass + 1) & 1] + lq.bnd0;
    unsigned long long *const bnd_out = p.bnd[pass & 1] + lq.bnd0;
    int32_t *const out = p.out + lq.out_off;
    {
        const int dw_per_code = RP >> 1;
        for (int idx = threadIdx.x; idx < kCodes * dw_per_code; idx += blockDim.x) {
            const int d = idx / dw_per_code, x = idx - d * dw_per_code;
            const uint32_t *src = (const uint32_t *)(p.prof + lq.prof_off + (size_t)d * lq.prof_stride + r0);
            uint32_t v = src[x];
            if (M == 2) {                           // int16 scores -> binary16
                const v2s sv = as_v2s(v);
                v = __builtin_bit_cast(uint32_t, (v2h){(_Float16)(float)sv.x, (_Float16)(float)sv.y});
            }
            *(uint32_t *)(smem + d * PS + x * 4) = v;
        }
    }
    __syncthreads();
    // these waves are long serial chains (the longest alignments, or re-runs a query is waiting for) beside bulk waves
    // that step through priorities 3..0: stay near the top (priority 0 or 3 measured the same on c3)
    __builtin_amdgcn_s_setprio(2);
    const unsigned char *my_prof = smem + lane * TR * 2;
    const int last_lane = (int)((rows + TR - 1) / TR) - 1;      // lane holding the query's last rows in this pass
    const V goe = Ops::splat(M == 2 ? -p.goe : p.goe), ge = Ops::splat(M == 2 ? -p.ge : p.ge);

    V H[TR], E[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) { H[r] = Ops::zero(); E[r] = Ops::zero(); }
    V best = Ops::zero(), diag = Ops::zero();
    // pipeline registers.  The residue stream (D) runs ONE step ahead of the boundary stream so that a lane
    // can issue the profile reads of its next column before it computes the current one.
    uint32_t oDn = 0x1818u;                  // residues this lane will hand to lane+1 (already one step ahead)
    uint32_t Dcur = 0x1818u;                 // residues + flags of the column this lane computes in this step
    uint32_t acur[NW] = {}, bcur[NW] = {};
    uint32_t oH = 0, oF = 0, oT = 0, oS = 0, oC = 0;
    uint2 pb = make_uint2(0u, 0u);           // boundary-side values of the column lane 0 fed one step ago
    uint32_t pitem = 0, pcol = 0;
    // producer state (wave-uniform): next chunk to feed
    bool feeding = true, dead = false;
    uint32_t cc = 0, nch = 0, it_lane = 0, it_half = 0, it_bnd = 0, it_idx = 0, seen = 0;
    const uint8_t *it_db = nullptr;
    int drained = 0;

    auto produce = [&](LaneFeed &f) -> bool {   // false: nothing left and the pipeline has drained
        if (feeding && cc == nch) {
            uint32_t idx = 0;
            if (lane == 0) idx = atomicAdd(queue, 1u);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= n_items) {
                feeding = false;
            } else {
                const LaneItem iv = items[idx];
                it_idx = idx; cc = 0; nch = iv.ncols / C; it_db = iv.db; it_lane = iv.lane; it_half = iv.half; it_bnd = iv.bnd_off;
                seen = 0;
            }
        }
        f.bv = make_uint2(0u, 0u); f.fbits = 0;
        f.wa = f.wb = 0x18181818u;   // pad residues (code 24) while draining
        f.item = it_idx; f.col0 = 0; f.half = it_half;
        if (feeding) {
            // uniform address; the value is NOT looked at here, so the load stays in flight while the
            // previous chunk is computed
            const unsigned long long w = *(const __attribute__((address_space(1))) unsigned long long *)(uintptr_t)(it_db + ((size_t)cc * 64 + it_lane) * 8);
            f.wa = (uint32_t)w;
            f.wb = (uint32_t)(w >> 32);
            f.col0 = it_bnd + cc * C;
            if (!first_pass) {
                const uint32_t need = f.col0 + C;            // the previous pass must have published these columns
                if (seen < need) {
                    uint32_t spins = 0;
                    for (;;) {
                        seen = __hip_atomic_load(prog_in + it_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen >= need) break;
                        if (++spins > (1u << 24)) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(16);
                    }
                    if (dead) { if (lane == 0) atomicOr(p.err, 4u); feeding = false; return false; }
                    // The boundary values below are read with agent-scope atomic loads (sc1: they bypass this CU's L1
                    // and are served by memory the producer's sc1 stores went through to), so nothing stale can be
                    // hit and the fence only has to keep the compiler from moving those loads above the poll.  (A full
                    // agent-scope acquire -- buffer_inv sc1 -- after every poll was measured in round 2: it adds nothing the
                    // sc1 loads do not give and costs 3.7 % on c3, profiles/NOTES.md.)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                {   // lane jj fetches column jj (the lanes above repeat the last column: same cache line, nothing extra)
                    const unsigned long long v = __hip_atomic_load(bnd_in + (size_t)f.col0 + (lane < C ? lane : C - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    f.bv = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                }
            }
            f.fbits = 1u | (cc == 0 ? 2u : 0u) | (cc + 1 == nch ? 4u : 0u);
            ++cc;
            return true;
        }
        drained += C;
        return drained <= 64 + 2 * C;   // the last real column needs 1 + 63 more steps to leave lane 63
    };

    // wave-uniform history of the chunks fed (bit i = the chunk fed i iterations ago): lane 63 computes, chunk
    // aligned, the chunk that was fed 16 iterations earlier
    uint32_t hist_real = 0;
    // Publication of the boundary rows to the next pass (cdna_hip_programming.md section 6, Guideline 16, form R1): the
    // payload is stored write-through (8-byte agent-scope atomic stores = global_store_dwordx2 sc1), the storing wave
    // drains its vector-memory counter completely -- s_waitcnt vmcnt(0), no reliance on the order in which loads and
    // stores retire -- and only then stores the progress word (agent-scope atomic store).  The drain sits at the top
    // of the next chunk, where the wave has to wait for its feed loads anyway, so a chunk's columns become visible one
    // chunk late and the wait is never for a store that was just issued.
    bool pub_pending = false;
    uint32_t pub_item = 0, pub_cols = 0;           // meaningful in lane 63
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pub_pending && lane == 63) __hip_atomic_store(prog_out + pub_item, pub_cols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub_pending = false;
    };
    LaneFeed nxt;
    bool more = produce(nxt);
    while (more) {
        // the feed is wave-uniform: keep the chunk being computed in scalar registers (the loads above land in
        // vector registers), so that only the chunk in flight costs VGPRs
        LaneFeed cur;
        cur.wa = __builtin_amdgcn_readfirstlane(nxt.wa); cur.wb = __builtin_amdgcn_readfirstlane(nxt.wb);
        cur.fbits = __builtin_amdgcn_readfirstlane(nxt.fbits);
        uint32_t cflags[C];
#pragma unroll
        for (int jj = 0; jj < C; ++jj) {
            cur.b[jj].x = __builtin_amdgcn_readlane(nxt.bv.x, jj); cur.b[jj].y = __builtin_amdgcn_readlane(nxt.bv.y, jj);
            cflags[jj] = (cur.fbits & 1u ? kFlagReal : 0u) | (jj == 0 && (cur.fbits & 2u) ? kFlagStart : 0u) | (jj == C - 1 && (cur.fbits & 4u) ? kFlagEnd : 0u);
        }
        cur.item = __builtin_amdgcn_readfirstlane(nxt.item); cur.col0 = __builtin_amdgcn_readfirstlane(nxt.col0);
        cur.half = __builtin_amdgcn_readfirstlane(nxt.half);
        if (!last_pass) publish();   // the previous chunk's boundary stores have retired (the feed loads were waited for just above)
        more = produce(nxt);         // loads of the next chunk are in flight while this one is computed
        if (!PK && cur.half) cur.wa = cur.wb;
        hist_real = (hist_real << 1) | (cur.fbits & 1u);
#pragma unroll
        for (int jj = 0; jj < C; ++jj) {
            uint32_t d0 = ((cur.wa >> (8 * jj)) & 0xffu) | cflags[jj];
            if (PK) d0 |= ((cur.wb >> (8 * jj)) & 0xffu) << 8;
            // residue stream, one step ahead: fetch the scores of the NEXT column now
            const uint32_t Dn = dpp_shr1(oDn, d0);
            uint32_t an[NW], bn[NW] = {};
            lane_prof_load<NW>(my_prof + (Dn & 0xffu) * PS, an);
            if (PK) lane_prof_load<NW>(my_prof + ((Dn >> 8) & 0xffu) * PS, bn);
            // boundary stream: every lane takes its left neighbour's bottom row, lane 0 the stored top boundary
            const uint32_t Hin = dpp_shr1(oH, pb.x), Fin = dpp_shr1(oF, pb.y);
            const uint32_t Tin = dpp_shr1(oT, 0u), Sin = dpp_shr1(oS, pitem), Cin = dpp_shr1(oC, pcol);
            const uint32_t D = Dcur;
            if (D & kFlagStart) {                 // first column of an alignment reaches this lane
#pragma unroll
                for (int r = 0; r < TR; ++r) { H[r] = Ops::zero(); E[r] = Ops::zero(); }
                best = Ops::zero(); diag = Ops::zero();
            }
            V hd = diag;
            diag = Ops::from_bits(Hin);
            V F = Ops::from_bits(Fin);
            const uint32_t *aw = acur;
            if constexpr (PK) {
                const uint32_t *bw = bcur;
#pragma unroll
                for (int q = 0; q < NW; ++q)
                    cell2<Ops>(hd, H[2 * q], E[2 * q], H[2 * q + 1], E[2 * q + 1], F, best, Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x05040100u)),
                               Ops::from_bits(__builtin_amdgcn_perm(bw[q], aw[q], 0x07060302u)), goe, ge);
            } else {
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    cell<Ops>(hd, H[2 * q], E[2 * q], F, best, Ops::from_bits((uint32_t)(int)(short)(aw[q] & 0xffffu)), goe, ge);
                    cell<Ops>(hd, H[2 * q + 1], E[2 * q + 1], F, best, Ops::from_bits((uint32_t)((int)aw[q] >> 16)), goe, ge);
                }
            }
            oH = Ops::bits(H[TR - 1]); oF = Ops::bits(F); oS = Sin; oC = Cin; oT = Tin;
            if (D & kFlagEnd) oT = Ops::bits(Ops::vmax(Ops::from_bits(Tin), best));   // running best of the alignment
            const bool mine = lane == last_lane && (D & kFlagReal);   // fill / drain columns must never reach real memory
            if (!last_pass && mine)   // one lane, four stores per real chunk (lane 63's chunks are chunk aligned)
                __hip_atomic_store(bnd_out + Cin, ((unsigned long long)oF << 32) | oH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine && (D & kFlagEnd)) {   // every pass contributes the best of its own rows
                const LaneItem *iv = items + Sin;
                if (M == 2) {
                    const v2h b2 = __builtin_bit_cast(v2h, oT);
                    atomicMax(out + iv->slot_a, (int)(float)b2.x);
                    atomicMax(out + iv->slot_b, (int)(float)b2.y);
                } else if (PK) {
                    const v2s b2 = __builtin_bit_cast(v2s, oT);
                    atomicMax(out + iv->slot_a, (int)b2.x);
                    atomicMax(out + iv->slot_b, (int)b2.y);
                } else {
                    atomicMax(out + iv->slot_a, (int)oT);
                }
            }
            // advance both streams
            oDn = Dn; Dcur = Dn;
#pragma unroll
            for (int q = 0; q < NW; ++q) { acur[q] = an[q]; bcur[q] = bn[q]; }
            pb = cur.b[jj]; pitem = cur.item; pcol = cur.col0 + jj;
        }
        if (!last_pass && ((hist_real >> 16) & 1u)) {
            // Lane 63 has just stored the boundary of the chunk fed 16 iterations ago, which ends at column oC of item
            // oS; it is published behind the drain at the top of the next chunk (or after the loop).
            pub_pending = true; pub_item = oS; pub_cols = oC + 1;
        }
    }
    if (!last_pass) publish();
}

template <int TR>
static hipError_t launch_lane_tr(Mode mode, int n_wg, const LaneParams &p, hipStream_t s)
{
    const size_t lds = lane_lds_bytes(TR);
    if (mode == Mode::PK16) hipLaunchKernelGGL((sw_lane_kernel<0, TR>), dim3(n_wg), dim3(256), lds, s, p);
    else if (mode == Mode::F16) hipLaunchKernelGGL((sw_lane_kernel<2, TR>), dim3(n_wg), dim3(256), lds, s, p);
    else hipLaunchKernelGGL((sw_lane_kernel<1, TR>), dim3(n_wg), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_lane(Mode mode, int rows_per_lane, int n_wg, const LaneParams &p, hipStream_t s)
{
    if (n_wg < 1) return hipErrorInvalidValue;
    if (rows_per_lane == 8) return launch_lane_tr<8>(mode, n_wg, p, s);
    // (the short-lane variants are for one-pass queries: the host only puts queries of <= 256 / <= 128 rows into such a launch)
    if (rows_per_lane == 4) return launch_lane_tr<4>(mode, n_wg, p, s);
    if (rows_per_lane == 2) return launch_lane_tr<2>(mode, n_wg, p, s);
    return hipErrorInvalidValue;
}

// ---- re-tile: reference chunk layout -> device groups --------------------------------------
// reference byte of VL-group v, position j, lane kk: b[disp[v] + j*vl + kk]   (sequences.c:508-513)
// reference residue code (0..24) -> device code (kDevCode), out of four 64-bit constants
__device__ __forceinline__ uint32_t dev_code(uint32_t h)
{
    constexpr auto pack = [](int i0) { unsigned long long v = 0; for (int i = 0; i < 8 && i0 + i < kCodes; ++i) v |= (unsigned long long)kDevCode[i0 + i] << (8 * i); return v; };
    constexpr unsigned long long t0 = pack(0), t1 = pack(8), t2 = pack(16), t3 = pack(24);
    const unsigned long long t = h < 8 ? t0 : h < 16 ? t1 : h < 24 ? t2 : t3;
    return (uint32_t)(t >> ((h & 7u) * 8)) & 0xffu;
}

// device dword of group g, chunk c, lane l: tiled[goff[g] + (c*64 + l)*8 + {0: seq 2l, 4: seq 2l+1}] -- a lane's pair are
// NEIGHBOURS of the length-sorted database, so the lane-systolic kernel (one wave per pair, run to the longer member's
// end) wastes nothing on the pair's shorter member: with (l, 64+l) the pairs of c3's three longest groups were 1.76 M
// columns, with neighbours 1.07 M.
__global__ void retile_kernel(const uint8_t *__restrict__ b, const uint16_t *__restrict__ n,
                              const uint32_t *__restrict__ disp, uint32_t vl_groups, uint32_t vl,
                              const uint64_t *__restrict__ goff, const uint32_t *__restrict__ gcols,
                              uint8_t *__restrict__ tiled, uint32_t *__restrict__ seq_len)
{
    // (a database that streams in is tiled while the pipeline kernel's waves, priority 1-3, fill the chip: at the default
    // priority 0 these few memory-bound waves would wait for issue slots and hold up the next chunk's copy)
    __builtin_amdgcn_s_setprio(3);
    const uint32_t g = blockIdx.x;
    const uint32_t nch = gcols[g] / kChunkCols;
    const uint32_t per = kGroupSeqs / vl;            // VL-groups per device group
    // (blockIdx.y: a long group is shared by several blocks -- a chunk of 5 000-residue sequences has fewer groups than the chip has CUs)
    // One (chunk, sequence) dword per thread and iteration: the kernel must fit the 32 registers a CU full of pipeline waves
    // leaves free (4 waves x 120 of a SIMD's 512), or a search that streams its database in would wait for a workgroup to end
    // before the next chunk could be tiled (tests/test_codegen.py holds the count).
    for (uint32_t idx = blockIdx.y * blockDim.x + threadIdx.x; idx < nch * 128; idx += gridDim.y * blockDim.x) {
        const uint32_t c = idx >> 7, sl = idx & 127;      // sequence within the device group: lane sl / 2, half sl % 2
        const uint32_t v = g * per + sl / vl, kk = sl % vl;
        uint32_t word = 0, real_end = 0;
#pragma unroll
        for (int jj = 0; jj < kChunkCols; ++jj) {
            const uint32_t col = c * kChunkCols + jj;
            uint32_t code = 24;                   // PREPROCESSED_DUMMY_ELEMENT, sequences.h:18
            if (v < vl_groups && col < n[v]) {
                code = b[(size_t)disp[v] + (size_t)col * vl + kk];
                if (code > 24) code = 24;         // out-of-alphabet bytes score like padding
            }
            if (code != 24) real_end = col + 1;
            word |= dev_code(code) << (8 * jj);
        }
        // true length of every sequence = 1 + its last non-padding column (the reference layout only
        // carries group lengths); the lane-systolic kernel stops each alignment there
        if (real_end) atomicMax(seq_len + (size_t)g * kGroupSeqs + sl, real_end);
        *(uint32_t *)(tiled + goff[g] + (size_t)idx * 4) = word;      // [chunk][lane][A0..A3 | B0..B3]: dword idx = (c * 64 + lane) * 2 + half
    }
}

// The same for lane widths that are multiples of 4 (every width the reference assembles): a thread takes FOUR neighbouring sequences of
// a chunk -- four dword loads (one per column: the four sequences' residues lie side by side in the reference's layout), a byte
// transpose (v_perm_b32), one 16-byte store -- instead of sixteen byte loads and four dword stores by four threads.  (Round 4: the
// byte-load version took 173 us per 32 MiB part of c2 beside the pipeline kernel's waves, a fifth of the part's copy time, and the
// next part's copy queues behind it on the upload stream.)
__global__ void retile4_kernel(const uint8_t *__restrict__ b, const uint16_t *__restrict__ n, const uint32_t *__restrict__ disp, uint32_t vl_groups,
                               uint32_t vl, const uint64_t *__restrict__ goff, const uint32_t *__restrict__ gcols, uint8_t *__restrict__ tiled,
                               uint32_t *__restrict__ seq_len)
{
    __builtin_amdgcn_s_setprio(3);
    const uint32_t g = blockIdx.x;
    const uint32_t nch = gcols[g] / kChunkCols;
    const uint32_t per = kGroupSeqs / vl;
    for (uint32_t idx = blockIdx.y * blockDim.x + threadIdx.x; idx < nch * 32; idx += gridDim.y * blockDim.x) {
        const uint32_t c = idx >> 5, sl = (idx & 31) * 4;      // sequences sl .. sl + 3 of the device group: lanes sl / 2, sl / 2 + 1
        const uint32_t v = g * per + sl / vl, kk = sl % vl;
        const uint32_t len = v < vl_groups ? n[v] : 0u;
        const uint8_t *src = b + (size_t)(v < vl_groups ? disp[v] : 0u) + kk;
        uint32_t w[kChunkCols];
#pragma unroll
        for (int jj = 0; jj < kChunkCols; ++jj) {
            const uint32_t col = c * kChunkCols + jj;
            uint32_t x = 0x18181818u;                    // PREPROCESSED_DUMMY_ELEMENT, sequences.h:18
            if (col < len) {
                __builtin_memcpy(&x, src + (size_t)col * vl, 4);      // (aligned whenever the groups' offsets are multiples of the lane width, as the reference's are)
                const uint32_t over = ((x & 0x80808080u) | (((x & 0x7f7f7f7fu) + 0x67676767u) & 0x80808080u)) >> 7;   // 1 per byte > 24
                const uint32_t m = over * 0xffu;
                x = (x & ~m) | (0x18181818u & m);        // out-of-alphabet bytes score like padding
            }
            w[jj] = x;
        }
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // sequence q of the four: byte q of every column's word
            const uint32_t lo = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0000u | ((4u + q) << 8) | (uint32_t)q);
            const uint32_t hi = __builtin_amdgcn_perm(w[3], w[2], 0x0c0c0000u | ((4u + q) << 8) | (uint32_t)q);
            const uint32_t x = lo | (hi << 16);
            // true length of every sequence = 1 + its last non-padding column (the reference layout only carries group lengths)
            const uint32_t y = x ^ 0x18181818u;
            if (y) atomicMax(seq_len + (size_t)g * kGroupSeqs + sl + q, c * kChunkCols + ((39u - (uint32_t)__builtin_clz(y)) >> 3));
            o[q] = dev_code(x & 0xffu) | dev_code((x >> 8) & 0xffu) << 8 | dev_code((x >> 16) & 0xffu) << 16 | dev_code(x >> 24) << 24;
        }
        *(uint4 *)(tiled + goff[g] + (size_t)(c * 128 + sl) * 4) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// blocks per group of the two tiling kernels: eight (chunk, sequence) dwords per thread for the longest group, at most 32
static unsigned tile_slices(uint32_t max_cols)
{
    const uint32_t dwords = max_cols / kChunkCols * 128;
    return std::max(1u, std::min(32u, (dwords + 2047) / 2048));
}

hipError_t launch_retile(const uint8_t *b, const uint16_t *n, const uint32_t *disp, uint32_t vl_groups, uint32_t vl,
                         const uint64_t *goff, const uint32_t *gcols, uint32_t dev_groups, uint32_t max_cols, uint8_t *tiled, uint32_t *seq_len,
                         hipStream_t s)
{
    if (dev_groups == 0) return hipSuccess;
    if (vl % 4 == 0)
        hipLaunchKernelGGL(retile4_kernel, dim3(dev_groups, tile_slices(max_cols / 4)), dim3(256), 0, s, b, n, disp, vl_groups, vl, goff, gcols, tiled, seq_len);
    else
        hipLaunchKernelGGL(retile_kernel, dim3(dev_groups, tile_slices(max_cols)), dim3(256), 0, s, b, n, disp, vl_groups, vl, goff, gcols, tiled, seq_len);
    return hipGetLastError();
}

// ---- tile: sorted sequences as stored in the .seq file -> device groups ----------------------------
// The direct path of the `swimm` program: the concatenated residue codes and their offsets go to the GPU as they
// are and the device builds its own layout, instead of the host interleaving lanes first (sequences.c:506-526)
// and retile_kernel undoing it.  Group g holds sequences 128 g .. 128 g + 127 of the slab; lane l of chunk c gets
// columns 4c..4c+3 of sequence 2l (low dword) and of sequence 2l + 1 (high dword), code 24 past a sequence's end.
__global__ void tile_sequences_kernel(const uint8_t *__restrict__ codes, const uint16_t *__restrict__ lens, const uint32_t *__restrict__ gsrc,
                                      uint32_t n_seq, const uint64_t *__restrict__ goff, const uint32_t *__restrict__ gcols,
                                      uint8_t *__restrict__ tiled)
{
    __builtin_amdgcn_s_setprio(3);                    // (see retile_kernel)
    const uint32_t g = blockIdx.x;
    const uint32_t nch = gcols[g] / kChunkCols;
    // where the group's 128 sequences begin: the host hands over the lengths as the .seq file holds them and ONE offset per
    // group; the offsets within the group are a scan over 128 lengths here (the host's walk over 35 M sequences stays two
    // reductions per group -- swimm_hip_add_sequences)
    __shared__ uint32_t s_len[kGroupSeqs], s_beg[kGroupSeqs];
    if (threadIdx.x < kGroupSeqs) {
        const uint32_t sq = g * kGroupSeqs + threadIdx.x;
        const uint32_t l = sq < n_seq ? lens[sq] : 0u;
        s_len[threadIdx.x] = l;
        s_beg[threadIdx.x] = l;
    }
    __syncthreads();
    for (uint32_t d = 1; d < kGroupSeqs; d <<= 1) {          // inclusive scan, 7 steps
        uint32_t add = 0;
        if (threadIdx.x < kGroupSeqs && threadIdx.x >= d) add = s_beg[threadIdx.x - d];
        __syncthreads();
        if (threadIdx.x < kGroupSeqs) s_beg[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t base = gsrc[g];
    for (uint32_t idx = blockIdx.y * blockDim.x + threadIdx.x; idx < nch * 128; idx += gridDim.y * blockDim.x) {
        const uint32_t c = idx >> 7, sl = idx & 127;            // (one (chunk, sequence) dword per thread and iteration: see retile_kernel)
        uint32_t word = 0x18181818u;              // four padding codes (24)
        const uint32_t len = s_len[sl];
        const uint32_t col = c * kChunkCols;
        if (col < len) {
            const uint32_t b0 = base + s_beg[sl] - len;      // (exclusive = inclusive - own)
            if (col + kChunkCols <= len) {
                // four residues inside the sequence: one (unaligned) dword, codes above 24 clamped to the padding code
                uint32_t x;
                __builtin_memcpy(&x, codes + (size_t)b0 + col, 4);
                const uint32_t over = ((x & 0x80808080u) | (((x & 0x7f7f7f7fu) + 0x67676767u) & 0x80808080u)) >> 7;   // 1 per byte > 24
                const uint32_t m = over * 0xffu;
                x = (x & ~m) | (0x18181818u & m);
                word = dev_code(x & 0xffu) | dev_code((x >> 8) & 0xffu) << 8 | dev_code((x >> 16) & 0xffu) << 16 | dev_code(x >> 24) << 24;
            } else {
                word = 0;
#pragma unroll
                for (int jj = 0; jj < kChunkCols; ++jj) {
                    uint32_t code = 24;
                    if (col + jj < len) { code = codes[(size_t)b0 + col + jj]; if (code > 24) code = 24; }
                    word |= dev_code(code) << (8 * jj);
                }
            }
        }
        *(uint32_t *)(tiled + goff[g] + (size_t)idx * 4) = word;
    }
}

hipError_t launch_tile_sequences(const uint8_t *codes, const uint16_t *lens, const uint32_t *gsrc, uint32_t n_seq, const uint64_t *goff,
                                 const uint32_t *gcols, uint32_t dev_groups, uint32_t max_cols, uint8_t *tiled, hipStream_t s)
{
    if (dev_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_sequences_kernel, dim3(dev_groups, tile_slices(max_cols)), dim3(256), 0, s, codes, lens, gsrc, n_seq, goff, gcols, tiled);
    return hipGetLastError();
}

// ---- "this much of the item list has landed" ------------------------------------------------------
// A search that streams its database in for ONE query runs ONE pipeline launch over the whole item list, in the order the
// parts travel (sw_pipe_kernel, PipeParams::avail); this one-thread kernel follows every part's tiling kernel on the upload
// stream and publishes how many items of the list are on the device now.  The tiling kernel has ended by then -- its stores are
// written back at the kernel boundary -- and the store below is agent-scope (sc1: write-through); the consumer polls with
// agent-scope loads and takes an agent-scope acquire before the first load of the new items' bytes (form "producer release,
// then flag; consumer poll, acquire, barrier, plain loads" of MI355X_MICROARCH.md, inter-workgroup visibility).
__global__ void publish_items_kernel(uint32_t *avail, uint32_t value)
{
    __hip_atomic_store(avail, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_publish_items(uint32_t *avail, uint32_t value, hipStream_t s)
{
    hipLaunchKernelGGL(publish_items_kernel, dim3(1), dim3(1), 0, s, avail, value);
    return hipGetLastError();
}

// ---- a kernel that only lasts (swimm_hip.cpp, make_streams: which streams share a hardware queue) ----
__global__ void spin_kernel(uint32_t ticks)          // ticks of the 100 MHz constant clock
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

hipError_t launch_spin(uint32_t microseconds, hipStream_t s)
{
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, microseconds * 100u);
    return hipGetLastError();
}

// ---- saturation bookkeeping ----------------------------------------------------------------
// list[i] = slots whose first-tier best left the tier's exact range (>= thr: 32767 for int16, CPUsearch.c:820-824
// "overflow detection"; 2048 for f16);
// their scores are zeroed so that the int32 re-run can atomicMax its result in.  *count may exceed cap:
// the host then re-runs with a larger list.
__global__ void collect_saturated_kernel(int32_t *__restrict__ scores, uint64_t n, int thr, uint32_t *__restrict__ list,
                                         uint32_t *__restrict__ count, uint32_t cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && scores[i] >= thr) {
        const uint32_t k = atomicAdd(count, 1u);
        if (k < cap) { list[k] = (uint32_t)i; scores[i] = 0; }
    }
}

hipError_t launch_collect_saturated(int32_t *scores, uint64_t n, int thr, uint32_t *list, uint32_t *count, uint32_t cap, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(collect_saturated_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scores, n, thr, list, count, cap);
    return hipGetLastError();
}

// ---- device top-r candidates ------------------------------------------------------------------
// Replaces sort_scores + the print loop's first r rows (utils.c:71-86, swimm.c:151-160) for r <= 64:
// key = (score << 32) | global sorted index, so "larger key first" is exactly the reference order
// (score descending, ties by LARGER index first, utils.c:12,52).  Every wave keeps the 64 largest
// keys it has seen, one per lane in descending order; a tile of 64 new keys is merged in only when
// a wave-wide ballot finds one above the current 64th key (wavefront __shfl_xor bitonic network).
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    const unsigned lo = __shfl_xor((unsigned)v, mask, 64), hi = __shfl_xor((unsigned)(v >> 32), mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src)
{
    const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// sort 64 keys (one per lane) descending
__device__ __forceinline__ unsigned long long wave_sort_desc(unsigned long long v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, j);
            const bool desc = (lane & k) == 0;          // direction of this k-block (k == 64: all descending)
            const bool lower = (lane & j) == 0;         // lower lane of the pair keeps the larger key when descending
            const bool take_max = (desc == lower);
            v = take_max ? (v > o ? v : o) : (v < o ? v : o);
        }
    }
    return v;
}
// both sorted descending -> the 64 largest of the union, sorted descending
__device__ __forceinline__ unsigned long long wave_merge_desc(unsigned long long a, unsigned long long b, int lane)
{
    const unsigned long long br = shfl_u64(b, 63 - lane);
    unsigned long long v = a > br ? a : br;             // bitonic sequence holding the top 64
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const unsigned long long o = shfl_xor_u64(v, j);
        const bool lower = (lane & j) == 0;
        v = lower ? (v > o ? v : o) : (v < o ? v : o);
    }
    return v;
}

__global__ void __launch_bounds__(256) topk64_kernel(const int32_t *__restrict__ scores_all, uint64_t n_slots,
                                                     const int64_t *__restrict__ group_base,
                                                     const uint32_t *__restrict__ group_valid,
                                                     unsigned long long *__restrict__ out_keys_all)
{
    // blockIdx.y = query: its score row (n_slots apart) and its gridDim.x * 64 candidate keys
    const int32_t *__restrict__ scores = scores_all + (size_t)blockIdx.y * n_slots;
    unsigned long long *__restrict__ out_keys = out_keys_all + (size_t)blockIdx.y * gridDim.x * 64;
    __shared__ unsigned long long sh[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long top = 0;                          // key 0 = empty (real keys have index >= 0, score >= 0: key 0 only for (0, 0), handled by +1 below)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x + (uint64_t)wv * 64; base < n_slots; base += stride) {
        const uint64_t slot = base + lane;
        unsigned long long key = 0;
        if (slot < n_slots) {
            const uint32_t g = (uint32_t)(slot >> 7), off = (uint32_t)(slot & 127);
            if (off < group_valid[g]) {
                const unsigned long long idx = (unsigned long long)(group_base[g] + off);
                key = (((unsigned long long)(uint32_t)scores[slot]) << 32 | idx) + 1;   // +1: keep 0 as "empty"
            }
        }
        const unsigned long long kth = shfl_u64(top, 63);
        if (__ballot(key > kth) != 0ull) top = wave_merge_desc(top, wave_sort_desc(key, lane), lane);
    }
    sh[wv][lane] = top;
    __syncthreads();
    if (wv == 0) {
        for (int w = 1; w < 4; ++w) top = wave_merge_desc(top, sh[w][lane], lane);
        out_keys[(size_t)blockIdx.x * 64 + lane] = top;
    }
}

hipError_t launch_topk64(const int32_t *scores, uint64_t n_slots, const int64_t *group_base, const uint32_t *group_valid,
                         unsigned long long *out_keys, int n_blocks, uint32_t n_queries, hipStream_t s)
{
    for (uint32_t q0 = 0; q0 < n_queries; q0 += 65535) {          // (gridDim.y limit)
        const uint32_t nq = n_queries - q0 < 65535 ? n_queries - q0 : 65535;
        hipLaunchKernelGGL(topk64_kernel, dim3(n_blocks, nq), dim3(256), 0, s, scores + (size_t)q0 * n_slots, n_slots, group_base, group_valid,
                           out_keys + (size_t)q0 * n_blocks * 64);
    }
    return hipGetLastError();
}

}  // namespace swimm
