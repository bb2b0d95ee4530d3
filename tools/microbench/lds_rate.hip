// lds_rate.hip -- LDS cost of the two ways a lane can fetch its profile scores on gfx950:
//   b128: one ds_read_b128 per residue and 8 query rows (the product kernel's lookup; pairs are combined by v_perm_b32)
//   u16 : one ds_read_u16_d16 (sequence A, low half) + one ds_read_u16_d16_hi (sequence B, high half) per query row, which
//         would deliver the packed (A, B) pair without any VALU instruction where d16 loads keep the other half of the
//         register (with SRAM-ECC on, as on MI355X, they write the whole register: LLVM's d16PreservesUnusedBits)
//   b32 : score-profile style (the reference's -p S): table[query residue][column][lane] holds the packed pair, one
//         lane-linear ds_read_b32 per query row, no v_perm_b32 -- but 5.9 KB of LDS per database column in flight
// Prints LDS clocks per wave-instruction per CU at 4..16 waves per CU, with random residue codes per lane.
// hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 512
__device__ unsigned long long g_clk[2];

template <int MODE>   // 0: b128 (codes 16 B apart mod 256), 1: u16 pairs (codes 4 B apart mod 128), 2: score-profile style b32 (lane-linear)
__global__ void __launch_bounds__(256) k_lds(uint32_t *out, const uint8_t *codes, int ps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 25 * ps / 4; i += blockDim.x) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t da = codes[(blockIdx.x * 256 + threadIdx.x) * 2] * ps, db = codes[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] * ps;
    uint32_t acc = 0;
    for (int it = 0; it < ITER; ++it) {
        const uint32_t r0 = (it & 3) * 32;    // 16 rows per iteration, 2 B each
        if (MODE == 0) {
            uint4 a0, a1, b0, b1;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(da + r0), "v"(db + r0));
            acc ^= a0.x ^ a0.y ^ a0.z ^ a0.w ^ a1.x ^ a1.y ^ a1.z ^ a1.w ^ b0.x ^ b0.y ^ b0.z ^ b0.w ^ b1.x ^ b1.y ^ b1.z ^ b1.w;
        } else if (MODE == 2) {
            // score profile (the reference's -p S, MICsearch.c:257-313): table[query residue][column][lane] holds the
            // packed (A, B) score pair, so a row's lookup is one lane-linear ds_read_b32 and needs no v_perm_b32
            uint32_t v[16];
            const uint32_t base = (threadIdx.x & 63) * 4 + (it & 3) * 256;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(base), "n"(r * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(v[r])); acc ^= v[r]; }
        } else {
            uint32_t v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                asm volatile("ds_read_u16_d16 %0, %1 offset:%3\n\tds_read_u16_d16_hi %0, %2 offset:%3" : "+v"(v[r]) : "v"(da + r0), "v"(db + r0), "n"(r * 2));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) { asm volatile("" : "+v"(v[r])); acc ^= v[r]; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - t0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - w0; }
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; uint8_t *codes;
    (void)hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(uint32_t));
    (void)hipMalloc(&codes, (size_t)cus * 4 * 256 * 2);
    {
        const size_t n = (size_t)cus * 4 * 256 * 2;
        uint8_t *h = (uint8_t *)malloc(n);
        uint64_t s = 12345;
        for (size_t i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (uint8_t)((s >> 33) % 20); }
        (void)hipMemcpy(codes, h, n, hipMemcpyHostToDevice);
        free(h);
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("device %s, %d CUs; 16 query rows x 2 residues per lane and iteration\n", prop.gcnArchName, cus);
    printf("%-34s %10s %10s %10s %10s   shader clocks per iteration per CU (all waves of the CU together), and per LDS instruction\n", "mode", "4 w/CU", "8", "12", "16");
    for (int mode = 0; mode < 3; ++mode) {
        const int ps = mode == 0 ? 192 * 2 + 16 : mode == 1 ? 192 * 2 + 4 : 704;    // 192 profile rows; code rows 16 B (mod 256) / 4 B (mod 128) apart; mode 2: 17.6 KB of lane-linear rows
        printf("%-34s", mode == 0 ? "4 x ds_read_b128" : mode == 1 ? "16 x (ds_read_u16_d16 + _d16_hi)" : "16 x ds_read_b32 lane-linear");
        for (int wg : {1, 2, 3, 4}) {
            const int blocks = cus * wg;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 25 * ps, 0, out, codes, ps);
                else if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 25 * ps, 0, out, codes, ps);
                else hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 25 * ps, 0, out, codes, ps);
            };
            launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long clk[2];
            (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
            const double ghz = (double)clk[0] / (double)clk[1] * 0.1;
            const double cyc_iter_cu = ms * 1e-3 * ghz * 1e9 / (5.0 * ITER) / (wg * 4);    // per wave-iteration, all of the CU's waves sharing the LDS
            printf(" %6.1f/%4.2f", cyc_iter_cu, cyc_iter_cu / (mode == 0 ? 4 : mode == 1 ? 32 : 16));
        }
        printf("\n");
    }
    return 0;
}
