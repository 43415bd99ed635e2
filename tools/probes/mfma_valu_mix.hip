// mfma_valu_mix.hip -- do the MFMAs of one wave overlap with the VALU work of the OTHER waves of its SIMD on gfx950?
// Every wave repeats an attention-like tile: phase A = 16 MFMA 32x32x16 (4 chains of 4 dependent MFMAs), phase B = NV VALU
// instructions (a third of them v_exp_f32).  MODE 0: both phases; 1: MFMAs only; 2: VALU only.  Reported: SIMD cycles per tile
// per wave (block span / (REP * waves per SIMD)).  If phases of different waves overlap, MODE 0 at 4 waves per SIMD costs
// ~max(MODE 1, MODE 2); if they do not, the sum.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix.bin mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define REP 128

template <int MODE, int DEP, int SHAPE = 32>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
    f16v acc[4];
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc4[4][2];
    for (int c = 0; c < 4; ++c) acc4[c][0] = acc4[c][1] = f4v{seed, 1.f, 2.f, 3.f};
    float r[32];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = seed * i;
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < REP; ++it) {
        if (MODE == 4) {  // barrier-staggered ping-pong: one wave of each SIMD in its MFMA section while its partner is in its VALU section
            if (it == 0 && (threadIdx.x >> 8) == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 (the SIMD partners) run one section behind
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[c]) : "v"(a));
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 31]));
#pragma unroll
            for (int i = 0; i < 32; i += 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[i + 1]));
            __builtin_amdgcn_s_barrier();
            if (it == REP - 1 && (threadIdx.x >> 8) == 0) __builtin_amdgcn_s_barrier();
            continue;
        }
        if (MODE == 3) {  // the same 16 MFMA 32x32x16 + 112 VALU, interleaved in program order: one MFMA, then 7 VALU (2 exp, 4 add, 1 cvt)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[m & 3]) : "v"(a));
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[2 * m]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[2 * m + 1]));
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[(4 * m + q) & 31]) : "v"(r[(4 * m + q + 1) & 31]));
                asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[(2 * m) & 31]) : "v"(r[(2 * m + 1) & 31]));
            }
            continue;
        }
        if (MODE != 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc[c]) : "v"(a));
                    else {  // the same flops as two 16x16x32 on two 4-register accumulators (8 chains of 4 dependent MFMAs per tile)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc4[c][0]) : "v"(a));
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc4[c][1]) : "v"(a));
                    }
                }
        }
        if (MODE != 1) {
            if (DEP) {  // the VALU phase reads the accumulators (as the softmax does): forces the wave to wait for its MFMAs
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (SHAPE == 32) asm volatile("s_nop 7\n s_nop 7\n v_add_f32 %0, %0, %1" : "+v"(r[c]) : "v"(acc[c][0]));
                    else asm volatile("s_nop 7\n s_nop 7\n v_add_f32 %0, %0, %1" : "+v"(r[c]) : "v"(acc4[c][1][0]));
                }
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 31]));
#pragma unroll
            for (int i = 0; i < 32; i += 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[i + 1]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    (void)0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += r[i];
    for (int c = 0; c < 4; ++c) s += acc[c][3] + acc4[c][0][1] + acc4[c][1][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int MODE, int DEP, int SHAPE = 32>
static void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 4 * 256); hipMalloc(&cyc, 8 * 256 * 32);
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4}) {
        hipLaunchKernelGGL((k<MODE, DEP, SHAPE>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, 0.5f);
        hipLaunchKernelGGL((k<MODE, DEP, SHAPE>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, 0.5f);
        hipDeviceSynchronize();
        std::vector<long long> c(256 * 32); hipMemcpy(c.data(), cyc, 8 * 256 * 32, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int b = 0; b < 256; ++b) {
            long long lo = c[b * 32], hi = c[b * 32 + 1];
            for (int w = 1; w < 4 * wps; ++w) { lo = std::min(lo, c[b * 32 + 2 * w]); hi = std::max(hi, c[b * 32 + 2 * w + 1]); }
            avg += (double)(hi - lo);
        }
        avg /= 256;
        printf("  wps=%d: %7.1f", wps, avg / (REP * (double)wps));
    }
    printf("   cycles per tile per wave\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 0>("16 MFMA only");
    run<2, 0>("112 VALU only (32 exp)");
    run<0, 0>("both, VALU independent of MFMA");
    run<0, 1>("both, VALU reads the accumulators");
    run<3, 0>("interleaved in ONE instruction stream");
    run<4, 0>("barrier-staggered ping-pong (wps=2 only)");
    run<1, 0, 16>("32 MFMA 16x16x32 only");
    run<0, 0, 16>("16x16x32: both, VALU independent");
    run<0, 1, 16>("16x16x32: both, VALU reads acc");
    return 0;
}
