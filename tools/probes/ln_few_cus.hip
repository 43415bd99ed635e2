// ln_few_cus.hip -- how fast is LayerNorm (f32 [M, 1024] -> f16, one wave per row, statistics in double) when only G workgroups of
// 8 waves run it?  LayerNorm is HBM-bound (270 MB per launch, 45 us on the whole chip): if a few CUs could stream it at a good
// fraction of that rate, it could run beside an MFMA-bound GEMM on the remaining CUs instead of in a launch of its own
// (profiles/r03_gemm_notes.md section 3).   hipcc --offload-arch=gfx950 -O3 tools/probes/ln_few_cus.hip -o /tmp/ln_few_cus && /tmp/ln_few_cus
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int UNROLL>
__global__ __launch_bounds__(512) void ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                 _Float16* __restrict__ y, int rows) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 8 + (threadIdx.x >> 6), nw = gridDim.x * 8;
    float4 gw[4], gb[4];
    for (int j = 0; j < 4; ++j) { gw[j] = ((const float4*)w)[lane + 64 * j]; gb[j] = ((const float4*)b)[lane + 64 * j]; }
    for (int r0 = wave * UNROLL; r0 < rows; r0 += nw * UNROLL) {
        float4 v[UNROLL][4];
        for (int u = 0; u < UNROLL; ++u) {
            const int r = r0 + u < rows ? r0 + u : rows - 1;
            for (int j = 0; j < 4; ++j) v[u][j] = ((const float4*)(x + (size_t)r * 1024))[lane + 64 * j];
        }
        for (int u = 0; u < UNROLL; ++u) {
            if (r0 + u >= rows) break;
            double s = 0;
            for (int j = 0; j < 4; ++j) s += (double)v[u][j].x + (double)v[u][j].y + (double)v[u][j].z + (double)v[u][j].w;
            const float mean = (float)(wsum(s) / 1024);
            double q = 0;
            for (int j = 0; j < 4; ++j) {
                v[u][j].x -= mean; v[u][j].y -= mean; v[u][j].z -= mean; v[u][j].w -= mean;
                q += (double)(v[u][j].x * v[u][j].x) + (double)(v[u][j].y * v[u][j].y) + (double)(v[u][j].z * v[u][j].z) + (double)(v[u][j].w * v[u][j].w);
            }
            const float sc = 1.0f / sqrtf((float)(wsum(q) / 1024) + 1e-6f);
            for (int j = 0; j < 4; ++j) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;
                o[0] = (_Float16)(v[u][j].x * sc * gw[j].x + gb[j].x); o[1] = (_Float16)(v[u][j].y * sc * gw[j].y + gb[j].y);
                o[2] = (_Float16)(v[u][j].z * sc * gw[j].z + gb[j].z); o[3] = (_Float16)(v[u][j].w * sc * gw[j].w + gb[j].w);
                ((h4*)(y + (size_t)(r0 + u) * 1024))[lane + 64 * j] = o;
            }
        }
    }
}

int main() {
    const int rows = 43968;
    float *x, *w, *b; _Float16* y;
    hipMalloc(&x, (size_t)rows * 4096); hipMalloc(&w, 4096); hipMalloc(&b, 4096); hipMalloc(&y, (size_t)rows * 2048);
    std::vector<float> h((size_t)rows * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(w, 0, 4096); hipMemset(b, 0, 4096);
    // a second large buffer touched between timed launches so that x does not sit in the 256 MB Infinity Cache
    char* flush; hipMalloc(&flush, (size_t)1 << 30);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int unroll = 1; unroll <= 4; unroll *= 2)
        for (int g : {8, 16, 32, 64, 128, 256, 1024}) {
            float best = 1e9f, sum = 0;
            for (int it = 0; it < 6; ++it) {
                hipMemsetAsync(flush, it, (size_t)1 << 30);
                hipEventRecord(e0);
                if (unroll == 1) ln_kernel<1><<<g, 512>>>(x, w, b, y, rows);
                else if (unroll == 2) ln_kernel<2><<<g, 512>>>(x, w, b, y, rows);
                else ln_kernel<4><<<g, 512>>>(x, w, b, y, rows);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it) { best = ms < best ? ms : best; sum += ms; }
            }
            printf("rows/wave-iter %d  workgroups %4d: best %.1f us  mean %.1f us  (%.2f TB/s)\n", unroll, g, best * 1e3, sum / 5 * 1e3, 270.1e6 / (best * 1e-3) / 1e12);
        }
    return 0;
}
