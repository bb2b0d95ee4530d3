// valu_rate.hip -- measures VALU issue cost (cycles per wave64 instruction per SIMD) on gfx950 for the
// instruction candidates of the Smith-Waterman cell update.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>

__device__ unsigned long long g_clk[2];
#define ITER 2048
#define OPS_PER_ITER 16

#define DEFINE_KERNEL(NAME, ASM3)                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t seed)                    \
    {                                                                                                \
        uint32_t r[8], y = seed + threadIdx.x, z = seed * 3 + 1;                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7 + i;                                      \
        for (int it = 0; it < ITER; ++it) {                                                          \
            _Pragma("unroll") for (int u = 0; u < OPS_PER_ITER; ++u)                                 \
                asm volatile(ASM3 : "+v"(r[u & 7]) : "v"(y), "v"(z));                                \
        }                                                                                            \
        uint32_t acc = 0;                                                                            \
        for (int i = 0; i < 8; ++i) acc ^= r[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                   \
            g_clk[0] = __builtin_amdgcn_s_memtime() - t0;                                            \
            g_clk[1] = __builtin_amdgcn_s_memrealtime() - w0;                                        \
        }                                                                                            \
    }

DEFINE_KERNEL(add_u32, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(max_i32, "v_max_i32 %0, %0, %1")
DEFINE_KERNEL(max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEFINE_KERNEL(sub_u32_clamp, "v_sub_u32 %0, %0, %1 clamp")
DEFINE_KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEFINE_KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEFINE_KERNEL(pk_add_i16_clamp, "v_pk_add_i16 %0, %0, %1 clamp")
DEFINE_KERNEL(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
DEFINE_KERNEL(perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEFINE_KERNEL(pk_add_f16, "v_pk_add_f16 %0, %0, %1")
DEFINE_KERNEL(pk_max_f16, "v_pk_max_f16 %0, %0, %1")
DEFINE_KERNEL(pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
DEFINE_KERNEL(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEFINE_KERNEL(max_i16, "v_max_i16 %0, %0, %1")
DEFINE_KERNEL(add_f32, "v_add_f32 %0, %0, %1")
DEFINE_KERNEL(max3_f32, "v_max3_f32 %0, %0, %1, %2")
DEFINE_KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(bfe_u32, "v_bfe_u32 %0, %0, %1, %2")
DEFINE_KERNEL(max_u16_sdwa, "v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")

DEFINE_KERNEL(pk_fmac_f16, "v_pk_fmac_f16 %0, %1, %2")
DEFINE_KERNEL(fmac_f16, "v_mac_f16 %0, %1, %2")
DEFINE_KERNEL(fmac_f32, "v_fmac_f32 %0, %1, %2")
DEFINE_KERNEL(max_f16, "v_max_f16 %0, %0, %1")
DEFINE_KERNEL(add_f16, "v_add_f16 %0, %0, %1")
DEFINE_KERNEL(max_f32, "v_max_f32 %0, %0, %1")
DEFINE_KERNEL(max_u32, "v_max_u32 %0, %0, %1")
DEFINE_KERNEL(min_i32, "v_min_i32 %0, %0, %1")
DEFINE_KERNEL(or_b32, "v_or_b32 %0, %0, %1")
DEFINE_KERNEL(and_b32, "v_and_b32 %0, %0, %1")
DEFINE_KERNEL(xor_b32, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
DEFINE_KERNEL(mov_b32, "v_mov_b32 %0, %1")
DEFINE_KERNEL(add_u16, "v_add_u16 %0, %0, %1")
DEFINE_KERNEL(sub_u16, "v_sub_u16 %0, %0, %1")
DEFINE_KERNEL(max_u16, "v_max_u16 %0, %0, %1")
DEFINE_KERNEL(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
DEFINE_KERNEL(pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
DEFINE_KERNEL(pk_min_f16, "v_pk_min_f16 %0, %0, %1")
DEFINE_KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFINE_KERNEL(dot2c_i32_i16, "v_dot2c_i32_i16 %0, %1, %2")
// third batch (round 2): ways to build the packed (A, B) score pair without the half-rate v_perm_b32
DEFINE_KERNEL(pack_b32_f16, "v_pack_b32_f16 %0, %0, %1")
DEFINE_KERNEL(pack_b32_f16_hi, "v_pack_b32_f16 %0, %0, %1 op_sel:[1,1,0]")
DEFINE_KERNEL(add_f16_sdwa_keep, "v_add_f16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0")
DEFINE_KERNEL(add_f16_sdwa_pad, "v_add_f16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
DEFINE_KERNEL(add_f16_sdwa_src, "v_add_f16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
DEFINE_KERNEL(bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(lshl_or_b32, "v_lshl_or_b32 %0, %0, 16, %1")
DEFINE_KERNEL(alignbit_b32, "v_alignbit_b32 %0, %0, %1, 16")
DEFINE_KERNEL(add_i16_opsel, "v_add_i16 %0, %0, %1 op_sel:[1,0,1]")
DEFINE_KERNEL(fma_f16_opsel, "v_fma_f16 %0, %0, 1.0, %1 op_sel:[1,0,0,1]")
DEFINE_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Entry { const char *name; kern_t k; };

int main()
{
    std::vector<Entry> ks = {
        {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32}, {"v_max3_i32", k_max3_i32}, {"v_sub_u32 clamp", k_sub_u32_clamp},
        {"v_add3_u32", k_add3_u32}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_add_i16 clamp", k_pk_add_i16_clamp},
        {"v_pk_sub_u16 clamp", k_pk_sub_u16_clamp}, {"v_perm_b32", k_perm_b32}, {"v_pk_add_f16", k_pk_add_f16},
        {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_maximum3_f16", k_pk_maximum3_f16}, {"v_pk_fma_f16", k_pk_fma_f16},
        {"v_max_i16", k_max_i16}, {"v_add_f32", k_add_f32}, {"v_max3_f32", k_max3_f32}, {"v_mad_u32_u24", k_mad_u32_u24},
        {"v_bfe_u32", k_bfe_u32}, {"v_max_i16_sdwa", k_max_u16_sdwa},
        {"v_pk_fmac_f16 (VOP2)", k_pk_fmac_f16}, {"v_mac_f16", k_fmac_f16}, {"v_fmac_f32", k_fmac_f32}, {"v_max_f16", k_max_f16},
        {"v_add_f16", k_add_f16}, {"v_max_f32", k_max_f32}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32}, {"v_or_b32", k_or_b32},
        {"v_and_b32", k_and_b32}, {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshlrev_b32}, {"v_mov_b32", k_mov_b32},
        {"v_add_u16", k_add_u16}, {"v_sub_u16", k_sub_u16}, {"v_max_u16", k_max_u16}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_sub_u32", k_sub_u32}, {"v_pk_mul_f16", k_pk_mul_f16}, {"v_pk_min_f16", k_pk_min_f16}, {"v_pk_add_u16", k_pk_add_u16},
        {"v_dot2c_i32_i16", k_dot2c_i32_i16},
        {"v_pack_b32_f16", k_pack_b32_f16}, {"v_pack_b32_f16 hi", k_pack_b32_f16_hi}, {"v_add_f16_sdwa keep", k_add_f16_sdwa_keep},
        {"v_add_f16_sdwa pad", k_add_f16_sdwa_pad}, {"v_add_f16_sdwa src", k_add_f16_sdwa_src}, {"v_bfi_b32", k_bfi_b32},
        {"v_and_or_b32", k_and_or_b32}, {"v_lshl_or_b32", k_lshl_or_b32}, {"v_alignbit_b32", k_alignbit_b32},
        {"v_add_i16 op_sel", k_add_i16_opsel}, {"v_fma_f16 op_sel", k_fma_f16_opsel}, {"v_mov_b32_dpp", k_mov_dpp},
        {"v_cndmask_b32", k_cndmask},
    };
    if (getenv("VALU_RATE_FROM")) ks.erase(ks.begin(), ks.begin() + atoi(getenv("VALU_RATE_FROM")));
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(uint32_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    printf("%-22s %8s %8s %8s %8s %8s %8s   cycles per wave64 instruction per SIMD at the MEASURED shader clock (s_memtime / s_memrealtime); last column: that clock in GHz at 8 w/SIMD\n", "instruction", "1 w/SIMD", "2", "3", "4", "6", "8");
    for (auto &e : ks) {
        printf("%-22s", e.name);
        double ghz = 0;
        for (int w : {1, 2, 3, 4, 6, 8}) {
            const int blocks = cus * w;   // 256-thread blocks: one wave per SIMD each
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = 5.0 * w * (double)ITER * OPS_PER_ITER;
            unsigned long long clk[2];
            hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
            ghz = (double)clk[0] / (double)clk[1] * 0.1;   // s_memrealtime ticks at 100 MHz
            const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
            printf(" %8.2f", cyc);
        }
        printf("   %.2f GHz\n", ghz);
    }
    return 0;
}
