// spin_probe.hip -- can a kernel that is launched AFTER a resident, polling kernel run beside it, and does the poller see what the
// newcomer publishes?  (The question behind the one-launch streaming search: a persistent pipeline launch waits for the
// database while the upload stream's tiling kernels deliver it.)
//   A  `blocks` workgroups of `threads` threads, `vgprs`-ish registers; wave 0 of every workgroup polls *flag (agent-scope loads,
//      s_sleep between polls) until it is non-zero or `limit` polls have passed; records what it saw and when
//   B  a small kernel on a second stream, launched once A is resident: writes a buffer (its completion is timed from the host)
//   P  a one-thread kernel on the second stream behind B: *flag = 1 (agent-scope store)
// Printed: when B and P completed relative to A's launch, when A ended, how many of A's workgroups saw the flag.
// hipcc --offload-arch=gfx950 -O3 spin_probe.hip -o spin_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void __launch_bounds__(1024) spin_kernel(const uint32_t *flag, uint32_t limit, uint32_t *seen, int heavy)
{
    __shared__ uint32_t lds[16384];      // 64 KB: like a pipeline workgroup
    if (heavy) lds[threadIdx.x] = threadIdx.x;
    uint32_t v = 0, polls = 0;
    if (threadIdx.x < 64) {
        for (; polls < limit; ++polls) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { seen[2 * blockIdx.x] = v; seen[2 * blockIdx.x + 1] = polls + (heavy ? lds[5] & 0 : 0); }
}

// the same poller with a full register file: 16 waves x 128 VGPRs fill a CU's 512 registers per SIMD lane, like a 16 x 24-row pipeline workgroup
__global__ void __launch_bounds__(1024) spin_kernel_fat(const uint32_t *flag, uint32_t limit, uint32_t *seen, int heavy)
{
    __shared__ uint32_t lds[24576];      // 96 KB
    float r[116];
#pragma unroll
    for (int i = 0; i < 116; ++i) r[i] = (float)(threadIdx.x + i * heavy);
    lds[threadIdx.x] = threadIdx.x;
    uint32_t v = 0, polls = 0;
    if (threadIdx.x < 64) {
        for (; polls < limit; ++polls) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 116; ++i) { asm volatile("" : "+v"(r[i])); acc += r[i] * (float)(v + i); }
    if (threadIdx.x == 0) { seen[2 * blockIdx.x] = v; seen[2 * blockIdx.x + 1] = polls + (acc == 12345.678f ? lds[5] : 0); }
}

// ... and shaped like a group-resident range launch: 256 threads, ~131 registers, 37 KB LDS, three per CU
__global__ void __launch_bounds__(256) spin_kernel_res(const uint32_t *flag, uint32_t limit, uint32_t *seen, int heavy)
{
    __shared__ uint32_t lds[9472];      // 37 KB
    float r[122];
#pragma unroll
    for (int i = 0; i < 122; ++i) r[i] = (float)(threadIdx.x + i * heavy);
    lds[threadIdx.x] = threadIdx.x;
    uint32_t v = 0, polls = 0;
    if (threadIdx.x < 64) {
        for (; polls < limit; ++polls) {
            v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 122; ++i) { asm volatile("" : "+v"(r[i])); acc += r[i] * (float)(v + i); }
    if (threadIdx.x == 0) { seen[2 * blockIdx.x] = v; seen[2 * blockIdx.x + 1] = polls + (acc == 12345.678f ? lds[5] : 0); }
}

__global__ void work_kernel(uint32_t *buf, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (uint32_t)i * 3u;
}

__global__ void publish_kernel(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 248, threads = argc > 2 ? atoi(argv[2]) : 1024, prio = argc > 3 ? atoi(argv[3]) : 0;
    const int heavy = argc > 4 ? atoi(argv[4]) : 1, h2d = argc > 5 ? atoi(argv[5]) : 0;
    uint32_t *flag, *seen, *buf;
    const size_t n = 8u << 20;
    hipMalloc(&flag, 64); hipMalloc(&seen, 2 * 4096 * sizeof(uint32_t)); hipMalloc(&buf, n * 4);
    hipMemset(flag, 0, 64); hipMemset(seen, 0, 2 * 4096 * sizeof(uint32_t));
    hipStream_t sa, sb;
    hipStreamCreate(&sa);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (prio) hipStreamCreateWithPriority(&sb, hipStreamDefault, hi); else hipStreamCreate(&sb);
    hipEvent_t eb, ep;
    hipEventCreate(&eb); hipEventCreate(&ep);
    uint32_t *hbuf = (uint32_t *)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) hbuf[i] = (uint32_t)i;
    uint32_t *pinned = nullptr;
    hipHostMalloc((void **)&pinned, 1 << 20, hipHostMallocDefault);
    hipDeviceSynchronize();
    const uint32_t limit = 400000;       // about half a second of polling
    const double t0 = now();
    if (heavy == 3) hipLaunchKernelGGL(spin_kernel_res, dim3(blocks), dim3(threads), 0, sa, flag, limit, seen, heavy);
    else if (heavy == 2) hipLaunchKernelGGL(spin_kernel_fat, dim3(blocks), dim3(threads), 0, sa, flag, limit, seen, heavy);
    else hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, sa, flag, limit, seen, heavy);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));      // A is resident and polling
    if (h2d == 1) hipMemcpyAsync(buf, hbuf, n * 4, hipMemcpyHostToDevice, sb);      // (pageable source, like the uploader's copies)
    if (h2d == 2) { hipMemcpyAsync(buf, pinned, 1 << 20, hipMemcpyHostToDevice, sb); hipStreamSynchronize(sb); }      // (1 MB from pinned memory + a wait, like a work list)
    const double t_copy = now();
    hipLaunchKernelGGL(work_kernel, dim3(512), dim3(256), 0, sb, buf, n);
    hipEventRecord(eb, sb);
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, sb, flag);
    hipEventRecord(ep, sb);
    hipEventSynchronize(eb);
    const double t_b = now();
    hipEventSynchronize(ep);
    const double t_p = now();
    hipStreamSynchronize(sa);
    const double t_a = now();
    uint32_t hs[2 * 4096];
    hipMemcpy(hs, seen, sizeof hs, hipMemcpyDeviceToHost);
    int saw = 0;
    uint32_t max_polls = 0;
    for (int b = 0; b < blocks; ++b) { saw += hs[2 * b] != 0; if (hs[2 * b + 1] > max_polls) max_polls = hs[2 * b + 1]; }
    printf("A: %d x %d threads%s, second stream %s%s: copy call returned %.2f ms, B done %.2f ms, publish done %.2f ms, A done %.2f ms after A's launch; %d of %d workgroups saw the flag (most polls %u of %u)\n",
           blocks, threads, heavy == 3 ? " x ~131 VGPRs + 37 KB LDS" : heavy == 2 ? " x ~128 VGPRs + 96 KB LDS" : heavy ? " + 64 KB LDS" : "", prio ? "high priority" : "plain", h2d == 1 ? ", 32 MB pageable H2D first" : h2d == 2 ? ", 1 MB pinned H2D + wait first" : "", (t_copy - t0) * 1e3, (t_b - t0) * 1e3, (t_p - t0) * 1e3, (t_a - t0) * 1e3, saw, blocks, max_polls, limit);
    return 0;
}
