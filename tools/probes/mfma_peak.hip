// Sustained v_mfma_f32_32x32x16_f16 rate of the whole chip from registers only (no LDS, no global traffic in the loop):
// the practical ceiling the GEMM/attention numbers should be read against.  Random vs zero operands shows the DVFS effect.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k(const _Float16* src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = src[(lane * 16 + i * 8 + j) & 4095];
            b[i][j] = src[(lane * 16 + i * 8 + j + 2048) & 4095];
        }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[n & 1], b[(n >> 1) & 1], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 2, threads = 256, iters = 20000, NACC = 8;
    std::vector<_Float16> h(4096);
    _Float16* d; float* o;
    hipMalloc(&d, 4096 * 2); hipMalloc(&o, blocks * threads * 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 4096; ++i) h[i] = mode == 0 ? (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f) : (_Float16)0.f;
        hipMemcpy(d, h.data(), 4096 * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) k<NACC><<<blocks, threads>>>(d, o, iters);  // ~100+ ms warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<NACC><<<blocks, threads>>>(d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * NACC * 32768.0;
        printf("%s operands: %.1f TFLOP/s (%.2f ms)\n", mode == 0 ? "random" : "zero  ", flops / ms / 1e9, ms);
    }
    return 0;
}
