// What the global -> LDS path of one CU sustains: every workgroup streams an L2-resident region into its LDS with
// global_load_lds (16 or 4 bytes per lane), NW waves per workgroup, one workgroup per CU, nothing else running.  Reports bytes per
// shader clock per CU and the implied cycles per wave-instruction -- the ceiling under the GEMM's staging (64 KiB per K-tile).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_rate.hip -o /tmp/ldsdma_rate && /tmp/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

template <int BYTES, int NW>
__global__ __launch_bounds__(NW * 64) void k(const char* src, int iters, long long* cycles, int region) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // each block reads its own `region`-byte window (L2 resident after the first pass), each wave a 64 * BYTES slice per step
    const char* base = src + (size_t)(blockIdx.x % 64) * region;
    char* dst = smem + wid * 8 * 64 * BYTES;
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned off = (unsigned)(wid * 64 + lane) * BYTES;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (BYTES == 16) __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + off), (LDS_AS void*)(dst + u * 64 * 16), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + off), (LDS_AS void*)(dst + u * 64 * 4), 4, 0, 0);
            off += NW * 64 * BYTES;
            off = off >= (unsigned)region ? off - (unsigned)region : off;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int BYTES, int NW>
static void run(const char* src, long long* dcyc, int region) {
    const int iters = 4000, blocks = 256;
    const size_t lds = (size_t)NW * 8 * 64 * BYTES;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<BYTES, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BYTES, NW>), dim3(blocks), dim3(NW * 64), lds, 0, src, 200, dcyc, region);  // warm up (L2, clocks)
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<BYTES, NW>), dim3(blocks), dim3(NW * 64), lds, 0, src, iters, dcyc, region);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(blocks);
    (void)hipMemcpy(cyc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= blocks;
    const double bytes_per_cu = (double)iters * 8 * NW * 64 * BYTES;
    // s_memtime ticks at a fixed 100 MHz-class reference on some parts: report both per-tick and per-ms figures
    printf("%2d B/lane, %d waves: %7.3f ms  %8.1f GB/s per CU  %6.2f TB/s chip   %.0f memtime ticks  (%.1f B/tick/CU, %.1f ticks per wave-instruction)\n",
           BYTES, NW, ms, bytes_per_cu / ms / 1e6, bytes_per_cu * blocks / ms / 1e9, mean, bytes_per_cu / mean, mean / (iters * 8.0));
}

int main() {
    const int region = getenv("REGION_KB") ? atoi(getenv("REGION_KB")) << 10 : 1 << 20;  // bytes per window, 64 windows (default 64 MiB: MALL; REGION_KB=64: L2)
    char* src; long long* dcyc;
    (void)hipMalloc((void**)&src, (size_t)64 * region);
    (void)hipMemset(src, 1, (size_t)64 * region);
    (void)hipMalloc((void**)&dcyc, 256 * sizeof(long long));
    run<16, 8>(src, dcyc, region);
    run<16, 4>(src, dcyc, region);
    run<16, 2>(src, dcyc, region);
    run<16, 1>(src, dcyc, region);
    run<4, 8>(src, dcyc, region);
    run<4, 4>(src, dcyc, region);
    run<4, 1>(src, dcyc, region);
    return 0;
}
