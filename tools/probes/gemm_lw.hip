// gemm_lw.hip -- probe: the 256 x 256 x 64 GEMM with LOADER WAVES.  profiles/r02_gemm_kloop.md: in every symmetric structure tried
// (gemm2_kernel, the 8-phase template, W-direct) MFMA time and VMEM time ADD, because a wave sits in the issue of a global_load*
// while the texture path drains the queue and issues no MFMA meanwhile.  Here the eight MFMA waves never issue a VMEM instruction:
// a workgroup is 12 waves -- 8 compute (2 x 4, wave tile 32 XREP x 64) + 4 loaders (one per SIMD) that do nothing but
// global_load_lds and wait for it.  Three waves per SIMD means <= 168 registers per wave (uniform allocation), hence fragment
// registers are single-buffered (the SIMD's other compute wave covers the LDS latency) and XREP = 3 (192-row tiles) is the
// comfortable configuration, XREP = 4 the tight one.
// One barrier per K-tile: at barrier t the loaders have waited for K-tile t (stage t & 1) and every compute wave has finished
// K-tile t - 1, so the loaders may refill stage (t + 1) & 1 while the compute waves work on stage t & 1.
//   hipcc -O3 --offload-arch=gfx950 -DXREP=3 tools/probes/gemm_lw.hip -o /tmp/gemm_lw && /tmp/gemm_lw
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

#ifndef XREP
#define XREP 3  // 32-row blocks per compute wave along M: tile = 64 XREP rows x 256 columns
#endif
#ifndef VARIANT
#define VARIANT 0  // timing only: 8 no staging, 16 no fragment reads, 32 no MFMA
#endif
constexpr int BM = 64 * XREP, ROWS = BM + 256, STAGE = ROWS * 128;  // bytes per K-tile stage
constexpr int NPIECE = ROWS / 8;                                    // 1-KiB pieces per K-tile (56 or 64)
constexpr int PPL = (NPIECE + 3) / 4;                               // pieces per loader wave

static __device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(768) void gemm_lw(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ C,
                                               int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = N / 256, ntm = M / BM, nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    constexpr int GM = 8;
    const int g = lid / (GM * ntn), r0 = lid - g * (GM * ntn);
    const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
    const int tn = r0 / gm, tmi = r0 - tn * gm;
    const int m0 = (g * GM + tmi) * BM, n0 = tn * 256;
    const int nk = K / 64;

    if (wid >= 8) {
        // ================= loader wave l: pieces l, l + 4, l + 8, ... of every K-tile (piece = 8 rows x 128 B; X rows first) =========
        const int l = wid - 8;
        unsigned src[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int p = j * 4 + l;
            const int r = (p < NPIECE ? p : NPIECE - 1) * 8 + (lane >> 3);
            const int ch = (lane & 7) ^ ((r >> 1) & 7);
            src[j] = (r < BM ? (unsigned)(m0 + r) : (unsigned)(n0 + r - BM)) * (unsigned)(K * 2) + ch * 16;
        }
        auto stage = [&](int kt) {
            if (VARIANT & 8) return;
            char* dst = smem + (kt & 1) * STAGE;
            const size_t ko = (size_t)kt * 128;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const int p = j * 4 + l;
                if (p < NPIECE) glds16((p * 8 < BM ? (const char*)A : (const char*)W) + ko + src[j], dst + p * 1024);
            }
        };
        stage(0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K-tile kt is in LDS (this wave's share)
            __builtin_amdgcn_s_barrier();                      // ... everybody's share; stage (kt + 1) & 1 is free
            if (kt + 1 < nk) stage(kt + 1);
        }
        return;
    }

    // ================= compute wave (wr, wc): rows wr * 32 XREP .., columns wc * 64 .. ====================================================
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 31, hh = lane >> 5, sw = (fr >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    const unsigned xbase = lds0 + (unsigned)((wr * 32 * XREP + fr) * 128);
    const unsigned wbase = lds0 + (unsigned)((BM + wc * 64 + fr) * 128);
    f32x16 acc[XREP][2];
#pragma unroll
    for (int i = 0; i < XREP; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 xf[XREP], wf[2];
    if (VARIANT & 16) {
        for (int i = 0; i < XREP; ++i) xf[i] = u32x4{(unsigned)tid, 1u, 2u, 3u};
        wf[0] = wf[1] = u32x4{(unsigned)tid, 5u, 6u, 7u};
    }
#define DSR(DST, ADDR, OFF) \
    if (!(VARIANT & 16)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))

    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned so = (unsigned)(kt & 1) * (unsigned)STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned ch = (unsigned)(((ks * 2 + hh) ^ sw) << 4) + so;
            const unsigned xa = xbase + ch, wa = wbase + ch;
            DSR(wf[0], wa, 0);
            DSR(wf[1], wa, 4096);
            DSR(xf[0], xa, 0);
            DSR(xf[1], xa, 4096);
            DSR(xf[2], xa, 8192);
            if (XREP == 4) DSR(xf[XREP - 1], xa, 12288);
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < XREP; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (VARIANT & 32) acc[i][j][0] += __builtin_bit_cast(float, xf[i][0]) * __builtin_bit_cast(float, wf[j][0]);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[i]), __builtin_bit_cast(f16x8, wf[j]), acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < XREP; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 32 * XREP + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                C[(size_t)m * N + n0 + wc * 64 + j * 32 + fr] = acc[i][j][r];
            }
}

__global__ void ref_kernel(const _Float16* A, const _Float16* W, const int* ms, const int* ns, float* out, int K, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)ms[i] * K + k] * (float)W[(size_t)ns[i] * K + k];
    out[i] = s;
}

__global__ void fill(_Float16* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (_Float16)((float)(x & 0xffffff) * (2.0f / 16777216.0f) - 1.0f);
    }
}

static void run(int M, int N, int K, int iters) {
    M = M / BM * BM;
    _Float16 *A, *W;
    float* C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4);
    fill<<<1024, 256>>>(A, (size_t)M * K, 1u); fill<<<1024, 256>>>(W, (size_t)N * K, 2u);
    const int lds = 2 * STAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lw), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const dim3 grid((M / BM) * (N / 256)), block(768);
    hipLaunchKernelGGL(gemm_lw, grid, block, lds, 0, A, W, C, M, N, K);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
    const int ns_ = 4096;
    std::vector<int> hm(ns_), hn(ns_);
    for (int i = 0; i < ns_; ++i) { hm[i] = (int)(((unsigned)rand() * 2654435761u) % (unsigned)M); hn[i] = (int)(((unsigned)rand() * 40503u + 17) % (unsigned)N); }
    int *dm, *dn; float* dr;
    hipMalloc(&dm, ns_ * 4); hipMalloc(&dn, ns_ * 4); hipMalloc(&dr, ns_ * 4);
    hipMemcpy(dm, hm.data(), ns_ * 4, hipMemcpyHostToDevice); hipMemcpy(dn, hn.data(), ns_ * 4, hipMemcpyHostToDevice);
    ref_kernel<<<(ns_ + 255) / 256, 256>>>(A, W, dm, dn, dr, K, ns_);
    std::vector<float> href(ns_);
    hipMemcpy(href.data(), dr, ns_ * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < ns_; ++i) {
        float v;
        hipMemcpy(&v, C + (size_t)hm[i] * N + hn[i], 4, hipMemcpyDeviceToHost);
        worst = std::fmax(worst, std::fabs((double)v - href[i]));
    }
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(gemm_lw, grid, block, lds, 0, A, W, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_lw, grid, block, lds, 0, A, W, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double rounds = std::ceil((double)((M / BM) * (N / 256)) / 256.0);
    printf("gemm_lw XREP=%d v%d M=%d N=%d K=%d: %.4f ms  %.1f TFLOP/s  (%.3f us per K-tile-round, %.3f scaled to 256 rows)  refcheck max|d| = %.3g %s\n", XREP, VARIANT, M, N, K,
           ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / ((K / 64) * rounds), ms * 1e3 / ((K / 64) * rounds) * 4.0 / XREP, worst,
           worst < 2e-2 * std::sqrt((double)K / 1024) ? "OK" : "MISMATCH");
    hipFree(A); hipFree(W); hipFree(C); hipFree(dm); hipFree(dn); hipFree(dr);
}

int main() {
    run(BM, 256, 128, 1);
    run(BM * 3, 768, 1024, 10);
    run(BM * 16, 4096, 4096, 50);   // 256 tiles: one round
    run(BM * 32, 8192, 8192, 10);
    run(43776, 4096, 1024, 50);
    return 0;
}
