// SURVEY 8(f) next-3, second half ("native low-precision compute"), closed by measurement.
//
// ggml multiplies a Q8_0 weight row with Q8_0-quantised activations block by block (/root/reference/dinov2.cpp:355-453 produce
// the files; ggml's vec_dot_q8_0_q8_0 does the arithmetic): for every 32-element K block
//     acc_f32 += (float)(sum_i8 w_q * x_q) * d_w[n, blk] * d_x[m, blk]
// A native path on gfx950 would use v_mfma_i32_32x32x32_i8 -- ONE instruction covers exactly one 32-deep block of a 32 x 32
// output tile -- and would then owe, per lane, 16 int->float conversions and 16 scaled accumulations (d_x is lane-local with
// the operand swap the f16 GEMM uses; the 16 d_w values differ per accumulator register) before the next block's integer sums
// may be added.  This probe prices exactly that, from registers (no global traffic; the d_w scales come from LDS as they would
// in a real kernel), in the regime most favourable to i8: everything the f16 kernel pays for besides the MFMAs is left out of
// BOTH arms.
//   arm A  f16   : 2 x v_mfma_f32_32x32x16_f16 per 32-deep block          (what the shipped GEMM issues)
//   arm B  i8    : 1 x v_mfma_i32_32x32x32_i8 per block, no rescale       (upper bound, not a usable result)
//   arm C  i8+q8 : arm B + the per-block rescale into an f32 accumulator  (the ggml Q8_0 x Q8_0 arithmetic)
// Reported in "f16-equivalent TFLOP/s" = 2 * 32 * 32 * 32 * blocks / time, i.e. the same K slice costs the same FLOPs in each arm.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/i8_q8_rescale.hip -o /tmp/i8_q8 && /tmp/i8_q8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 4;  // independent 32x32 output tiles per wave (the shipped kernel has 8; 4 keeps arm C inside 256 VGPRs)

template <int ARM>
__global__ __launch_bounds__(256, 2) void k(const int* src, float* out, int blocks_k) {
    __shared__ __attribute__((aligned(16))) float dw[2048];  // d_w scales, re-read every block like a staged scale tile
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) dw[i] = 1.0f + (float)(src[i & 1023] & 255) * (1.0f / 4096.0f);
    __syncthreads();
    i32x4 a8[2], b8[2];
    f16x8 a16[2], b16[2];
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 4; ++j) {
            a8[i][j] = src[(lane * 8 + i * 4 + j) & 1023];
            b8[i][j] = src[(lane * 8 + i * 4 + j + 512) & 1023];
        }
        a16[i] = __builtin_bit_cast(f16x8, a8[i]);
        b16[i] = __builtin_bit_cast(f16x8, b8[i]);
        for (int j = 0; j < 8; ++j) {  // keep the f16 operands finite and O(1)
            a16[i][j] = (_Float16)((float)((a8[i][j >> 1] >> (16 * (j & 1))) & 0xff) * (1.0f / 128.0f) - 1.0f);
            b16[i][j] = (_Float16)((float)((b8[i][j >> 1] >> (16 * (j & 1))) & 0xff) * (1.0f / 128.0f) - 1.0f);
        }
    }
    f32x16 acc[NT];
    for (int n = 0; n < NT; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float dx = 1.0f + (float)lane * (1.0f / 1024.0f);  // d_x[m, blk]: one row per lane
    for (int kb = 0; kb < blocks_k; ++kb) {
        if (ARM == 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[n & 1], b16[(n >> 1) & 1], acc[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[(n + 1) & 1], b16[(n >> 1) & 1], acc[n], 0, 0, 0);
            }
        } else if (ARM == 1) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                i32x16 c = __builtin_bit_cast(i32x16, acc[n]);
                c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a8[n & 1], b8[(n >> 1) & 1], c, 0, 0, 0);
                acc[n] = __builtin_bit_cast(f32x16, c);
            }
        } else {
            // one block: NT integer tiles, then their rescale.  The integer accumulators start from zero every block (that is
            // the Q8_0 arithmetic); the rescale of tile n can overlap the MFMA of tile n+1 (separate pipes).
            const int so = (kb & 15) * 128 + (lane >> 5) * 64;  // 16 scales per tile for this lane half, 4 x ds_read_b128
            dx = dx * 1.0001f;                                   // a new d_x per block
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                i32x16 z;
                for (int r = 0; r < 16; ++r) z[r] = 0;
                const i32x16 s = __builtin_amdgcn_mfma_i32_32x32x32_i8(a8[n & 1], b8[(n >> 1) & 1], z, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = *(const f32x4*)&dw[so + n * 16 + g * 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[n][g * 4 + e] = __builtin_fmaf((float)s[g * 4 + e], w[e] * dx, acc[n][g * 4 + e]);
                }
            }
        }
    }
    float s = 0.f;
    for (int n = 0; n < NT; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 2, threads = 256, kblocks = 40000;
    std::vector<int> h(1024);
    for (auto& v : h) v = rand() ^ (rand() << 16);
    int* d; float* o;
    hipMalloc(&d, 4096); hipMalloc(&o, blocks * threads * 4);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[3] = {"A f16 (2 x 32x32x16 per block)    ", "B i8 MFMA only (no rescale)       ", "C i8 MFMA + Q8_0 block rescale    "};
    double tf[3];
    for (int arm = 0; arm < 3; ++arm) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
            if (arm == 0) k<0><<<blocks, threads>>>(d, o, kblocks);
            else if (arm == 1) k<1><<<blocks, threads>>>(d, o, kblocks);
            else k<2><<<blocks, threads>>>(d, o, kblocks);
        };
        for (int w = 0; w < 3; ++w) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        tf[arm] = (double)blocks * (threads / 64) * kblocks * NT * 65536.0 / ms / 1e9;
        printf("%s %8.1f f16-equivalent TFLOP/s (%.2f ms)\n", names[arm], tf[arm], ms);
    }
    printf("C / A = %.3f   (a native Q8_0 path is worth building only above 1.2)\n", tf[2] / tf[0]);
    return 0;
}
