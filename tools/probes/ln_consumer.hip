// ln_consumer.hip -- VERDICT r3 item 4, priced before built: what would it cost a batch-1 GEMM workgroup to compute the LayerNorm statistics
// of its OWN 128-row panel in its tile prologue (consumer-side fusion: no cross-workgroup hand-off, but every one of the N / 256 column
// tiles of a row panel repeats the pass)?  The probe runs only that pass -- WG (panel, column tile) reads its 128 x H f32 rows (the
// residual stream, L2 / MALL resident at batch 1: 5.6 MB), accumulates sum and sum of squares per row in double like layernorm_kernel, and
// writes mean / rstd -- for the two batch-1 launches that would carry it: QKV (11 panels x 12 column tiles = 132 workgroups) and FFN-in
// (11 x 16 = 176), 512 threads each like the 128-row GEMM plan.  To beat: the LayerNorm launch it would remove, 5.1 us + one launch
// boundary (~1.6 us) -- before the GEMM pays for staging an f32 operand through registers instead of LDS-DMA.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ln_consumer.hip -o /tmp/ln_consumer && /tmp/ln_consumer
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(512) void stats(const float* __restrict__ x, float* __restrict__ out, int M, int H, int ncol) {
    const int panel = blockIdx.x / ncol;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 8 waves x 16 rows: one wave per row at a time, the row in registers (H = 1024: 4 float4 per lane), statistics in double
    for (int r = 0; r < 16; ++r) {
        int row = panel * 128 + wid * 16 + r;
        row = row < M ? row : M - 1;
        const float4* p = (const float4*)(x + (size_t)row * H);
        double s = 0.0, q = 0.0;
        for (int c = lane; c < H / 4; c += 64) {
            const float4 v = p[c];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (int o = 32; o; o >>= 1) {
            s += __shfl_xor(s, o);
            q += __shfl_xor(q, o);
        }
        if (lane == 0) {
            const double mean = s / H, var = q / H - mean * mean;
            out[((size_t)blockIdx.x * 128 + wid * 16 + r) * 2] = (float)mean;
            out[((size_t)blockIdx.x * 128 + wid * 16 + r) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-6));
        }
    }
}

__global__ void nop() {}

int main() {
    const int M = 1374, H = 1024;
    float *x, *out;
    hipMalloc(&x, (size_t)M * H * 4);
    hipMalloc(&out, 256 * 128 * 2 * 4);
    hipMemset(x, 0x3c, (size_t)M * H * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ncol : {12, 16}) {
        const int grid = ((M + 127) / 128) * ncol;
        for (int i = 0; i < 50; ++i) stats<<<grid, 512>>>(x, out, M, H, ncol);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) stats<<<grid, 512>>>(x, out, M, H, ncol);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) nop<<<grid, 512>>>();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms0; hipEventElapsedTime(&ms0, e0, e1);
        printf("statistics pass alone, %3d workgroups (%d column tiles per panel): %.2f us per launch (empty launch of the same grid %.2f us) -> %.2f us of work\n",
               grid, ncol, ms / 200 * 1e3, ms0 / 200 * 1e3, (ms - ms0) / 200 * 1e3);
    }
    return 0;
}
