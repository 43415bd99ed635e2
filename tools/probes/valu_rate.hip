// valu_rate.hip -- issue rate of the VALU instructions the attention softmax is made of, on gfx950, with 1..4 waves per SIMD.
// Each wave runs REP x 32 independent instructions of one kind; cycles per instruction per SIMD = (s_memtime delta) * waves_on_simd /
// (REP * 32 * waves_on_simd) ... reported as SIMD cycles per wave-instruction (all waves of the SIMD together).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 256
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
    float r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = seed + threadIdx.x * 1e-3f + i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < REP; ++it) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            if (KIND == 0) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 1) { asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 2) { asm volatile("v_max3_f32 %0, %0, %1, %1\n v_max3_f32 %1, %1, %0, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 3) { asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 4) { asm volatile("v_pk_add_f32 %0, %0, %0\n" : "+v"(*(double*)&r[i])); asm volatile("v_pk_add_f32 %0, %0, %0\n" : "+v"(*(double*)&r[i])); }
            if (KIND == 5) { asm volatile("v_pk_mul_f32 %0, %0, %0\n" : "+v"(*(double*)&r[i])); asm volatile("v_pk_mul_f32 %0, %0, %0\n" : "+v"(*(double*)&r[i])); }
            if (KIND == 6) { asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 7) { asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 8) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 9) { asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 10) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n" : "+v"(*(double*)&r[i])); asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n" : "+v"(*(double*)&r[i])); }
            if (KIND == 11) { asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 12) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 13) { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
            if (KIND == 15) {
                typedef float f16v __attribute__((ext_vector_type(16)));
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                static_assert(sizeof(f16v) == 64, "");
                f16v& acc = *(f16v*)&r[(i & 16)];  // two accumulators of 16 registers
                h8 a = {1, 2, 3, 4, 5, 6, 7, 8};
                if ((i & 15) == 0) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(a)); }
            }
            if (KIND == 14) { asm volatile("v_pk_mul_f16 %0, %0, %1\n v_pk_mul_f16 %1, %1, %0" : "+v"(r[i]), "+v"(r[i + 1])); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int KIND>
static void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 4 * 256); hipMalloc(&cyc, 8 * 256 * 32);
    printf("%-20s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD: block = 4 SIMDs x wps waves, one block per CU
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 0.5f);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, out, cyc, 0.5f);
        hipDeviceSynchronize();
        std::vector<long long> c(256 * 32); hipMemcpy(c.data(), cyc, 8 * 256 * 32, hipMemcpyDeviceToHost);
        double avg = 0;  // the arbiter favours the oldest wave: time the block from its first start to its last end
        for (int b = 0; b < 256; ++b) {
            long long lo = c[b * 32], hi = c[b * 32 + 1];
            for (int w = 1; w < 4 * wps; ++w) { lo = std::min(lo, c[b * 32 + 2 * w]); hi = std::max(hi, c[b * 32 + 2 * w + 1]); }
            avg += (double)(hi - lo);
        }
        avg /= 256;
        printf("  wps=%d: %6.2f cyc/instr/SIMD", wps, avg / (REP * 32.0 * wps));
    }
    printf("\n");
}
int main() {
    run<0>("v_exp_f32"); run<7>("v_exp_f16"); run<1>("v_add_f32"); run<6>("v_mul_f32"); run<9>("v_fma_f32"); run<2>("v_max3_f32");
    run<3>("v_cvt_pk_f16_f32"); run<8>("v_cvt_pk_bf16_f32"); run<4>("v_pk_add_f32"); run<5>("v_pk_mul_f32"); run<10>("v_pk_fma_f32");
    run<14>("v_pk_mul_f16"); run<11>("v_permlane32_swap"); run<12>("v_cndmask_b32"); run<13>("v_mov_b32");
    run<15>("mfma32x32x16 (x16)");  // 2 MFMAs per 32-instruction group: multiply the printed figure by 16
    return 0;
}
