// valu_rate.hip -- measures VALU issue cost (cycles per wave64 instruction per SIMD) on gfx950 for the
// instruction candidates of the Smith-Waterman cell update.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 2048
#define OPS_PER_ITER 16

#define DEFINE_KERNEL(NAME, ASM3)                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t seed)                    \
    {                                                                                                \
        uint32_t r[8], y = seed + threadIdx.x, z = seed * 3 + 1;                                     \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7 + i;                                      \
        for (int it = 0; it < ITER; ++it) {                                                          \
            _Pragma("unroll") for (int u = 0; u < OPS_PER_ITER; ++u)                                 \
                asm volatile(ASM3 : "+v"(r[u & 7]) : "v"(y), "v"(z));                                \
        }                                                                                            \
        uint32_t acc = 0;                                                                            \
        for (int i = 0; i < 8; ++i) acc ^= r[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                            \
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
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4 * sizeof(uint32_t));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    printf("%-22s %10s %10s %10s   (cycles per wave64 instruction per SIMD at the nominal 2.4 GHz)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto &e : ks) {
        printf("%-22s", e.name);
        for (int w : {1, 2, 4}) {
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
            const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
            printf(" %10.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
