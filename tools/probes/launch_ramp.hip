// What a kernel costs before and after its workgroups' own work: a graph of 200 dependent launches of a kernel that does nothing
// (or touches `bytes` of output per launch), for a grid / block / LDS shape.  us per launch = dispatch ramp + end-of-kernel release.
//   hipcc -O3 --offload-arch=gfx950 launch_ramp.hip -o launch_ramp && ./launch_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(float* out, int words_per_thread) {
    extern __shared__ char lds[];
    if (words_per_thread) {
        float4* o = (float4*)out + (size_t)blockIdx.x * blockDim.x * words_per_thread + threadIdx.x;  // coalesced: a wave writes 1 KB runs
        for (int i = 0; i < words_per_thread; ++i) o[(size_t)i * blockDim.x] = make_float4(1.f, 2.f, 3.f, (float)i);
    }
}
static float run(int grid, int block, int lds, int wpt, float* buf) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, st, buf, wpt);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    hipStreamDestroy(st);
    return ms * 1000.f / 1000.f;  // us per launch
}
int main() {
    float* buf;
    hipMalloc(&buf, (size_t)1 << 30);
    struct C { int grid, block, lds, wpt; };
    std::vector<C> cs = {{1, 64, 0, 0},        {256, 256, 0, 0},       {256, 512, 0, 0},        {528, 256, 0, 0},     {528, 256, 48 << 10, 0},
                         {704, 256, 48 << 10, 0}, {176, 512, 144 << 10, 0}, {344, 256, 0, 0},  {2048, 256, 0, 0}, {4096, 256, 0, 0},
                         {528, 256, 48 << 10, 1}, {528, 256, 48 << 10, 2}, {528, 256, 48 << 10, 4}, {528, 256, 48 << 10, 16}, {176, 512, 144 << 10, 8}, {704, 256, 48 << 10, 8}, {2048, 256, 0, 16}, {8192, 256, 0, 16}};
    for (auto c : cs)
        printf("grid %5d block %4d lds %6d  out %8.2f MB : %7.2f us / launch\n", c.grid, c.block, c.lds, (double)c.grid * c.block * c.wpt * 16 / 1e6,
               run(c.grid, c.block, c.lds, c.wpt, buf));
    return 0;
}
