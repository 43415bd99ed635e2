// Does the SHAPE of a global_load_lds wave-instruction matter?  ldsdma_rate.hip streams 1 KiB of CONSECUTIVE bytes per
// wave-instruction (142 GB/s per CU from L2); the GEMM's staging pieces are 8 rows x 128 B at the operand's row stride, and
// staging alone runs at ~80 GB/s per CU inside gemm2_kernel.  This probe runs the probe's loop with the GEMM's piece shape:
// each workgroup (8 waves) streams K-tiles of a [256 x K] panel pair, piece = rows 8j .. 8j+7 x 128 B, row stride = K * 2 bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_pattern.hip -o /tmp/ldsdma_pattern && /tmp/ldsdma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

// MODE 0: consecutive 1 KiB per instruction (reference); 1: 8 rows x 128 B; 2: 4 rows x 256 B; 3: 16 rows x 64 B; 4: 2 rows x 512 B
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, int iters, int rowbytes, int panel_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(blockIdx.x % 32) * panel_bytes;  // 32 panels of 512 rows: L2/MALL resident depending on size
    char* dst = smem + wid * 8192;
    constexpr int RPP = MODE == 0 ? 1 : MODE == 1 ? 8 : MODE == 2 ? 4 : MODE == 3 ? 16 : 2;  // rows per piece
    constexpr int BPR = 1024 / RPP;                                                         // bytes per row per piece
    const int lrow = lane / (BPR / 16), lcol = (lane % (BPR / 16)) * 16;
    unsigned koff = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // piece u of wave wid: 512 rows per K-tile / (8 waves * 8 pieces) = 8 rows of 128 B each in the GEMM (MODE 1)
            unsigned off;
            if (MODE == 0) off = koff + (unsigned)((u * 8 + wid) * 1024 + lane * 16);
            else off = (unsigned)(((u * 8 + wid) * RPP + lrow) * rowbytes) + koff * (BPR / 128.0f > 0 ? 1 : 1) + lcol;
            __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + off), (LDS_AS void*)(dst + u * 1024), 16, 0, 0);
        }
        koff += MODE == 0 ? 65536 : BPR;  // next K-tile: the next BPR bytes of every row
        if (MODE == 0 ? koff >= (unsigned)panel_bytes : koff >= (unsigned)rowbytes) koff = 0;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
static void run(const char* name, const char* src, int rowbytes, int panel_bytes) {
    const int iters = 4000, blocks = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 65536, 0, src, 400, rowbytes, panel_bytes);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 65536, 0, src, iters, rowbytes, panel_bytes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * 65536;
    printf("%-28s row stride %6d B: %7.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  %.3f us per 64 KiB\n", name, rowbytes, ms,
           bytes_per_cu / ms / 1e6, bytes_per_cu * blocks / ms / 1e9, ms * 1e3 / iters);
}

int main() {
    char* src;
    const size_t total = (size_t)32 * 512 * 8192;  // 32 panels x 512 rows x 8 KiB = 128 MiB
    (void)hipMalloc((void**)&src, total + (64 << 20));  // slack: 16-row pieces walk 1024 rows of the last panel
    (void)hipMemset(src, 1, total + (64 << 20));
    for (int rowbytes : {2048, 8192, 2176}) {
        const int panel = 512 * rowbytes;
        printf("-- panels of 512 rows x %d B (%d KiB each, 32 panels = %d MiB)\n", rowbytes, panel >> 10, (32 * panel) >> 20);
        run<0>("consecutive 1 KiB", src, rowbytes, panel);
        run<1>("8 rows x 128 B (GEMM piece)", src, rowbytes, panel);
        run<2>("4 rows x 256 B", src, rowbytes, panel);
        run<4>("2 rows x 512 B", src, rowbytes, panel);
        run<3>("16 rows x 64 B", src, rowbytes, panel);
    }
    return 0;
}
