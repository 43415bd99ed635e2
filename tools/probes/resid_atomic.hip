// resid_atomic.hip -- can the residual epilogue (x_new = x + v: read 4 B, write 4 B per element) be a fire-and-forget f32 atomic add
// executed at the L2, and at what rate?  180 MB of f32 (the batch-32 ViT-L residual stream), each element touched once.
//   (a) load + add + store    (b) no-return global_atomic_add_f32    (c) store only    (d) pk: global_atomic_pk_add n/a for f32
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o resid_atomic.bin resid_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ x, const float* __restrict__ v, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = ((const float4*)v)[i & 0xFFFF];  // the "accumulator": small, cache-resident
        if (MODE == 0) {
            float4 b = ((float4*)x)[i];
            b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
            ((float4*)x)[i] = b;
        } else if (MODE == 1) {
            float* p = x + 4 * i;
            unsafeAtomicAdd(p, a.x); unsafeAtomicAdd(p + 1, a.y); unsafeAtomicAdd(p + 2, a.z); unsafeAtomicAdd(p + 3, a.w);
        } else if (MODE == 2) {
            ((float4*)x)[i] = a;
        } else if (MODE == 3) {  // lane-contiguous scalar atomics: lane l adds element 64 * j + l (one 256-B line per instruction)
            const size_t wbase = (i & ~(size_t)63) * 4;
            const int l = threadIdx.x & 63;
            unsafeAtomicAdd(x + wbase + l, a.x); unsafeAtomicAdd(x + wbase + 64 + l, a.y);
            unsafeAtomicAdd(x + wbase + 128 + l, a.z); unsafeAtomicAdd(x + wbase + 192 + l, a.w);
        }
    }
}

template <int MODE>
static void run(const char* name, float* x, float* v, size_t n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, v, n / 4);
        hipEventRecord(e0);
        const int it = 20;
        for (int w = 0; w < it; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, v, n / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        printf("%-44s grid %5d: %7.3f ms  %6.2f TB/s of read+write-equivalent traffic (8 B per element)\n", name, grid, ms, n * 8.0 / ms / 1e9);
    }
}
int main() {
    const size_t n = (size_t)43968 * 1024;  // 180 MB
    float *x, *v; hipMalloc(&x, n * 4); hipMalloc(&v, 1 << 20);
    hipMemset(x, 0, n * 4); hipMemset(v, 0, 1 << 20);
    run<0>("load + add + store (float4)", x, v, n);
    run<1>("no-return atomic add, 4 consecutive per lane", x, v, n);
    run<3>("no-return atomic add, lane-contiguous", x, v, n);
    run<2>("store only (4 B per element written)", x, v, n);
    return 0;
}
