// valu_mix.hip -- do VALU instructions of the two issue classes of gfx950 overlap on one SIMD?
//
// tools/microbench/valu_rate.hip measured homogeneous streams only: every VOP3P packed / 3-source op issues over ~4.3 cycles
// per wave64 instruction, the 32-bit add / logic ops and the unpacked 16-bit VOP2 ops over ~2.45.  The Smith-Waterman column
// loop is all 4-cycle-class; this benchmark asks what a 2-cycle-class instruction costs BESIDE it:
//   mode A     every wave runs op A only                              (N instructions per wave)
//   mode B     every wave runs op B only
//   mode A|B   two of a SIMD's four waves run A, the other two run B   (different waves, same SIMD)
//   mode AB    every wave alternates A, B, A, B ...                    (N of each per wave)
// One 1024-thread workgroup per CU = 4 waves per SIMD (wave w of a workgroup sits on SIMD w % 4: checked through HW_REG_HW_ID
// and printed).  If the classes shared nothing, A|B would take max(2N cA, 2N cB) SIMD cycles; if they share the issue port /
// the ALU, 2N cA + 2N cB.  Output: SIMD cycles per instruction for A, for B, and the measured / serial / overlapped figures of
// the two mixed modes.
// hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 1024
#define UNROLL 16

__device__ unsigned long long g_clk[2];
__device__ unsigned int g_simd[16];

#define RUN(ASM)                                                                   \
    for (int it = 0; it < ITER; ++it) {                                            \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u)                         \
            asm volatile(ASM : "+v"(r[u & 7]) : "v"(y), "v"(z));                   \
    }
#define RUN2(ASMA, ASMB)                                                           \
    for (int it = 0; it < ITER; ++it) {                                            \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                       \
            asm volatile(ASMA : "+v"(r[u & 7]) : "v"(y), "v"(z));                  \
            asm volatile(ASMB : "+v"(q[u & 7]) : "v"(y), "v"(z));                  \
        }                                                                          \
    }

#define DEFINE_PAIR(NAME, ASMA, ASMB)                                                                          \
    __global__ void __launch_bounds__(1024) k_##NAME(uint32_t *out, uint32_t seed, int mode)                    \
    {                                                                                                          \
        uint32_t r[8], q[8], y = seed + threadIdx.x, z = seed * 3 + 1;                                         \
        const int wave = threadIdx.x >> 6;                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();     \
        for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 7 + i; q[i] = threadIdx.x * 5 + i; }                \
        if (mode == 0) { RUN(ASMA) }                                                                           \
        else if (mode == 1) { RUN(ASMB) }                                                                      \
        else if (mode == 2) { if ((wave >> 2) & 1) { RUN(ASMB) } else { RUN(ASMA) } }                          \
        else { RUN2(ASMA, ASMB) }                                                                              \
        uint32_t acc = 0;                                                                                      \
        for (int i = 0; i < 8; ++i) acc ^= r[i] ^ q[i];                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                      \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                                      \
            unsigned int hw;                                                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                   \
            g_simd[wave] = (hw >> 4) & 3u;                                                                     \
        }                                                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                             \
            g_clk[0] = __builtin_amdgcn_s_memtime() - t0;                                                      \
            g_clk[1] = __builtin_amdgcn_s_memrealtime() - w0;                                                  \
        }                                                                                                      \
    }

DEFINE_PAIR(pkadd_addf16, "v_pk_add_f16 %0, %0, %1", "v_add_f16 %0, %0, %1")
DEFINE_PAIR(pkmax3_maxf16, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_max_f16 %0, %0, %1")
DEFINE_PAIR(pkfma_addu32, "v_pk_fma_f16 %0, %0, %1, %2", "v_add_u32 %0, %0, %1")
DEFINE_PAIR(pkmax_maxi16, "v_pk_max_f16 %0, %0, %1", "v_max_i16 %0, %0, %1")
DEFINE_PAIR(pkmax3_mov, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_mov_b32 %0, %1")
DEFINE_PAIR(pkmax3_addf32, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_add_f32 %0, %0, %1")
DEFINE_PAIR(pkmax3_fmacf32, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_fmac_f32 %0, %1, %2")
DEFINE_PAIR(pkadd_pkadd, "v_pk_add_f16 %0, %0, %1", "v_pk_add_f16 %0, %0, %1")
DEFINE_PAIR(addf16_addf16, "v_add_f16 %0, %0, %1", "v_add_f16 %0, %0, %1")
DEFINE_PAIR(pkmax3_bfe, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_bfe_u32 %0, %0, %1, %2")
DEFINE_PAIR(pkmax3_mad24, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_PAIR(pkmax3_lshl_add, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_lshl_add_u32 %0, %0, 3, %1")
DEFINE_PAIR(pkmax3_and, "v_pk_maximum3_f16 %0, %0, %1, %2", "v_and_b32 %0, %0, %1")

typedef void (*kern_t)(uint32_t *, uint32_t, int);
struct Entry { const char *a, *b; kern_t k; };

int main()
{
    std::vector<Entry> ks = {
        {"v_pk_add_f16", "v_add_f16", k_pkadd_addf16},
        {"v_pk_maximum3_f16", "v_max_f16", k_pkmax3_maxf16},
        {"v_pk_fma_f16", "v_add_u32", k_pkfma_addu32},
        {"v_pk_max_f16", "v_max_i16", k_pkmax_maxi16},
        {"v_pk_maximum3_f16", "v_mov_b32", k_pkmax3_mov},
        {"v_pk_maximum3_f16", "v_add_f32", k_pkmax3_addf32},
        {"v_pk_maximum3_f16", "v_fmac_f32", k_pkmax3_fmacf32},
        {"v_pk_maximum3_f16", "v_bfe_u32", k_pkmax3_bfe},
        {"v_pk_maximum3_f16", "v_mad_u32_u24", k_pkmax3_mad24},
        {"v_pk_maximum3_f16", "v_lshl_add_u32", k_pkmax3_lshl_add},
        {"v_pk_maximum3_f16", "v_and_b32", k_pkmax3_and},
        {"v_pk_add_f16", "v_pk_add_f16", k_pkadd_pkadd},
        {"v_add_f16", "v_add_f16", k_addf16_addf16},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(uint32_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("device %s, %d CUs; one 1024-thread workgroup per CU = 4 waves per SIMD; N = %d instructions per wave and op\n", prop.gcnArchName, cus, ITER * UNROLL);
    printf("SIMD cycles per wave64 instruction at the measured shader clock.  A|B: waves 0-3, 8-11 run A, waves 4-7, 12-15 run B (two of each per SIMD); AB: every wave alternates A and B.\n");
    printf("%-20s %-16s %6s %6s | %7s %7s %7s | %7s %7s %7s\n", "A (4-cycle class)", "B", "A", "B", "A|B", "serial", "overlap", "AB", "serial", "overlap");
    bool printed_map = false;
    for (auto &e : ks) {
        double cyc[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; ++mode) {
            // (warm: the first launches after an idle moment run at a lower clock than the one block 0 reports at the end)
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(e.k, dim3(cus), dim3(1024), 0, 0, out, 1u, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(e.k, dim3(cus), dim3(1024), 0, 0, out, 1u, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long clk[2];
            hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
            const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
            // SIMD cycles of one launch; per instruction: modes 0-2 issue 4 N instructions per SIMD, mode 3 issues 8 N
            const double simd_cycles = ms * 1e-3 / 20.0 * ghz * 1e9;
            cyc[mode] = simd_cycles / ((mode == 3 ? 8.0 : 4.0) * ITER * UNROLL);
        }
        if (!printed_map) {
            unsigned int simd[16];
            hipMemcpyFromSymbol(simd, HIP_SYMBOL(g_simd), sizeof simd);
            printf("# SIMD of waves 0..15 of workgroup 0:");
            for (int i = 0; i < 16; ++i) printf(" %u", simd[i]);
            printf("\n");
            printed_map = true;
        }
        // A|B per instruction (4N per SIMD: 2N of A, 2N of B): serial = (cA + cB) / 2, full overlap = max(cA, cB) / 2
        const double ser = (cyc[0] + cyc[1]) / 2, ovl = (cyc[0] > cyc[1] ? cyc[0] : cyc[1]) / 2;
        printf("%-20s %-16s %6.2f %6.2f | %7.2f %7.2f %7.2f | %7.2f %7.2f %7.2f\n", e.a, e.b, cyc[0], cyc[1], cyc[2], ser, ovl, cyc[3], ser, ovl);
    }
    return 0;
}
