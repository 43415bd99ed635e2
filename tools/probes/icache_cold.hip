// Does a kernel pay for instruction fetch?  Each workgroup runs the same straight-line block of N_INSTR VALU instructions (8 bytes each)
// three times (a rolled loop around an unrolled body) and wave 0 times each pass with the 100 MHz wall clock.  Pass 1 - pass 2 = what the
// first execution of that much code costs a workgroup at the start of a launch (the launches are back to back, same kernel).
//   hipcc -O3 --offload-arch=gfx950 icache_cold.hip -o icache_cold && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef N_INSTR
#define N_INSTR 512
#endif
__device__ unsigned long long g_t[4];
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x = threadIdx.x * a, y = b;
    unsigned long long t[4];
    t[0] = wall_clock64();
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int i = 0; i < N_INSTR / 2; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0" : "+v"(x), "+v"(y));
        }
        t[pass + 1] = wall_clock64();
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < 3; ++i) atomicAdd(&g_t[i], t[i + 1] - t[i]);
        atomicAdd(&g_t[3], 1ull);
    }
    if (x == 12345.678f) out[0] = x + y;
}
int main() {
    float* buf;
    hipMalloc(&buf, 1 << 20);
    for (int grid : {256, 528, 704}) {
        for (int rep = 0; rep < 3; ++rep) {
            unsigned long long z[4] = {0, 0, 0, 0}, h[4];
            hipMemcpyToSymbol(HIP_SYMBOL(g_t), z, sizeof z);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, buf, 1.0001f, 0.5f);
            hipDeviceSynchronize();
            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof h);
            const double n = (double)h[3];
            printf("%d instructions (%d bytes), grid %d: pass 1 %.2f us, pass 2 %.2f us, pass 3 %.2f us per workgroup\n", N_INSTR, N_INSTR * 8, grid, h[0] / n / 100,
                   h[1] / n / 100, h[2] / n / 100);
        }
    }
    return 0;
}
