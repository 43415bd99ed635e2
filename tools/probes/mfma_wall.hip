// mfma_wall.hip -- where is the power wall of the matrix cores for each MFMA shape?  Register-only loops (no LDS, no global traffic),
// one wave per SIMD (256 accumulators, like the 4-wave GEMM probe gemm4w.hip) or two (128 accumulators each, like csrc/gemm2.hip), uniform
// random f16 operands in 8 + 8 distinct fragment registers (acc[i][j] += W[j] x X[i], the GEMM's operand pattern), whole chip.
// Prints TFLOP/s, matrix-pipe duty in shader cycles (s_memtime) and the clock the part sustained (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_wall.hip -o /tmp/mfma_wall && /tmp/mfma_wall
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_t[2];

template <int SHAPE, int WAVES>  // SHAPE 16: 16x16x32, 32: 32x32x16; WAVES per SIMD 1 or 2
__global__ __launch_bounds__(256 * WAVES) void k(const _Float16* src, float* out, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    f16x8 w[8], x[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            w[i][j] = src[(threadIdx.x * 131 + i * 8 + j) & 65535];
            x[i][j] = src[(threadIdx.x * 257 + i * 8 + j + 32768) & 65535];
        }
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        constexpr int NI = 8 / WAVES;  // 64 or 32 accumulators of 4
        f32x4 acc[NI][8];
        for (int i = 0; i < NI; ++i)
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], x[i], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < NI; ++i)
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        constexpr int NI = 4 / WAVES;  // 16 or 8 accumulators of 16
        f32x16 acc[NI][4];
        for (int i = 0; i < NI; ++i)
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j + 4 * ks], x[i + 4 * ks], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < NI; ++i)
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_t[0] = __builtin_readcyclecounter() - c0;
        g_t[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)smem[0];
}

template <int SHAPE, int WAVES>
static void run(const _Float16* src, float* out, const char* what, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(256), dim3(256 * WAVES), 100 * 1024, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, WAVES>), dim3(256), dim3(256 * WAVES), 100 * 1024, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long t[2];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_t), sizeof t);
    // per wave and iteration: 8 x (8 / WAVES) MFMAs of 16x16x32 (16 384 FLOP) or 2 x 4 x (4 / WAVES) of 32x32x16 (32 768 FLOP): 1 048 576 / WAVES
    const double flop_iter_wave = 1048576.0 / WAVES;
    const double flops = flop_iter_wave * iters * 4.0 * WAVES * 256.0;
    // matrix-pipe cycles per iteration and SIMD: 64 MFMA-slots of 16 cycles (= 1 048 576 FLOP / 1 024 FLOP/clk) regardless of shape
    const double duty = 1024.0 * iters / (double)t[0];
    printf("%-46s %8.1f TFLOP/s   duty %.3f   clock %.3f GHz   (%.3f ms)\n", what, flops / ms / 1e9, duty, t[0] / (t[1] * 10.0), ms);
}

int main() {
    std::vector<_Float16> h(65536);
    srand(1);
    for (auto& v : h) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
    _Float16* src; float* out;
    hipMalloc(&src, 65536 * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), 65536 * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 1>(src, out, "16x16x32, 1 wave/SIMD, 256 acc, random", 4000);
        run<32, 1>(src, out, "32x32x16, 1 wave/SIMD, 256 acc, random", 4000);
        run<16, 2>(src, out, "16x16x32, 2 waves/SIMD, 128 acc each, random", 4000);
        run<32, 2>(src, out, "32x32x16, 2 waves/SIMD, 128 acc each, random", 4000);
    }
    hipMemset(src, 0, 65536 * 2);
    run<16, 1>(src, out, "16x16x32, 1 wave/SIMD, ZERO operands", 4000);
    run<32, 1>(src, out, "32x32x16, 1 wave/SIMD, ZERO operands", 4000);
    return 0;
}
