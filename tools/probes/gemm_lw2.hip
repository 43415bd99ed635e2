// gemm_lw2.hip -- probe: the 256 x 256 x 64 GEMM with LOADER WAVES that stage through their own REGISTERS, two K-tiles ahead.  profiles/r02_gemm_kloop.md: in every symmetric structure tried
// (gemm2_kernel, the 8-phase template, W-direct) MFMA time and VMEM time ADD, because a wave sits in the issue of a global_load*
// while the texture path drains the queue and issues no MFMA meanwhile.  Here the eight MFMA waves never issue a VMEM instruction:
// a workgroup is 12 waves -- 8 compute (2 x 4, wave tile 32 XREP x 64) + 4 loaders (one per SIMD) that do nothing but
// global_load_lds and wait for it.  Three waves per SIMD means <= 168 registers per wave (uniform allocation), hence fragment
// registers are single-buffered (the SIMD's other compute wave covers the LDS latency) and XREP = 3 (192-row tiles) is the
// comfortable configuration, XREP = 4 the tight one.
// gemm_lw.hip (loaders issuing global_load_lds, one K-tile ahead) measured 2.21 us per K-tile: with two LDS stages the loaders
// drain (vmcnt(0)) every K-tile, so every K-tile pays a full memory latency -- and a third 64-KiB stage does not fit in 160 KiB.
// The loaders' REGISTERS do: 168 VGPRs x 64 lanes x 4 loaders = 172 KiB.  Each loader keeps two K-tiles of its 16 pieces in
// registers (global_load_dwordx4, counted vmcnt), and between barrier t - 1 and barrier t it only moves K-tile t from registers
// to stage t & 1 (ds_write_b128, no memory latency involved) and re-issues the loads of K-tile t + 2 into the freed registers.
// Bonus: the XOR swizzle moves to the per-lane ds_write address, so the global reads are plain full lines.
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

__global__ __launch_bounds__(768) void gemm_lw2(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ C,
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
        // piece p = 4 j + l covers image rows 8 p .. 8 p + 7: X rows for j < XJ, W rows after.  Per lane ONE 32-bit offset (its row
        // inside the loader's 8-row group and its 16-byte chunk); the piece's base is scalar: rows advance by 32 per j.
        constexpr int XJ = BM / 32;  // X pieces per loader
        static_assert(PPL == XJ + 8, "every loader has XJ X pieces and 8 W pieces");
        const unsigned voff = (unsigned)(l * 8 + (lane >> 3)) * (unsigned)(K * 2) + (lane & 7) * 16;
        const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
        const int r0_ = l * 8 + (lane >> 3);  // image row of piece j = 0; + 32 j for piece j: the swizzle term (r >> 1) & 7 does not change
        const unsigned dst0 = lds0 + (unsigned)(r0_ * 128 + (((lane & 7) ^ ((r0_ >> 1) & 7)) << 4));
        const char* const xg = (const char*)A + (size_t)m0 * K * 2;
        const char* const wg = (const char*)W + (size_t)n0 * K * 2;
        const size_t jstride = (size_t)32 * K * 2;
        u32x4 ra[PPL], rb[PPL];  // K-tiles 0, 2, 4 ... / 1, 3, 5 ...
#define GLD(R, J, KT)                                                                                                        \
    {                                                                                                                        \
        const char* b__ = ((J) < XJ ? xg + (J) * jstride : wg + ((J) - XJ) * jstride) + (size_t)(KT) * 128;                  \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(R[J]) : "v"(voff), "s"(b__));                                   \
    }
#define LOADS(R, KT)                                                             \
    {                                                                            \
        const int kk__ = (KT) < nk ? (KT) : nk - 1; /* past the end: re-fetch */ \
        _Pragma("unroll") for (int j = 0; j < PPL; ++j) GLD(R, j, kk__)          \
        __builtin_amdgcn_sched_barrier(0);                                       \
    }
#define WRITES(R, KT)                                                                                             \
    {                                                                                                             \
        const unsigned a__ = dst0 + (unsigned)((KT) & 1) * (unsigned)STAGE;                                       \
        _Pragma("unroll") for (int j = 0; j < PPL; ++j)                                                           \
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a__), "v"(R[j]), "n"(j * 4096) : "memory");       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
#define STEP(R, KT)                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPL) : "memory"); /* K-tile KT landed; KT + 1 may be in flight */ \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (!(VARIANT & 8)) WRITES(R, KT)                                                                         \
        LOADS(R, (KT) + 2)                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
        LOADS(ra, 0)
        LOADS(rb, 1)
        for (int kt = 0; kt < nk; kt += 2) {
            STEP(ra, kt)
            STEP(rb, kt + 1)
        }
        // the padding loads of the last two steps are still in flight: wait, then keep their registers alive past the wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < PPL; ++j) asm volatile("" ::"v"(ra[j]), "v"(rb[j]));
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

#define READ_KS(XF, WF, KS, SO)                                                  \
    {                                                                            \
        const unsigned ch__ = (unsigned)((((KS) * 2 + hh) ^ sw) << 4) + (SO);    \
        const unsigned xa__ = xbase + ch__, wa__ = wbase + ch__;                 \
        DSR(WF[0], wa__, 0);                                                     \
        DSR(WF[1], wa__, 4096);                                                  \
        DSR(XF[0], xa__, 0);                                                     \
        DSR(XF[1], xa__, 4096);                                                  \
        DSR(XF[2], xa__, 8192);                                                  \
        if (XREP == 4) DSR(XF[XREP - 1], xa__, 12288);                           \
    }
#define MMA_KS(XF, WF)                                                                                                   \
    {                                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        _Pragma("unroll") for (int i = 0; i < XREP; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                 \
            if (VARIANT & 32) acc[i][j][0] += __builtin_bit_cast(float, XF[i][0]) * __builtin_bit_cast(float, WF[j][0]); \
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, XF[i]), __builtin_bit_cast(f16x8, WF[j]), acc[i][j], 0, 0, 0); \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
#if VARIANT & 64  // fragments double-buffered inside the K-tile (one k-step of lookahead, as gemm2_kernel has)
    u32x4 xg[XREP], wg2[2];
    constexpr int NR = XREP + 2;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned so = (unsigned)(kt & 1) * (unsigned)STAGE;
        READ_KS(xf, wf, 0, so)
        READ_KS(xg, wg2, 1, so)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR));
        MMA_KS(xf, wf)
        READ_KS(xf, wf, 2, so)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR));
        MMA_KS(xg, wg2)
        READ_KS(xg, wg2, 3, so)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NR));
        MMA_KS(xf, wf)
        asm volatile("s_waitcnt lgkmcnt(0)");
        MMA_KS(xg, wg2)
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned so = (unsigned)(kt & 1) * (unsigned)STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            READ_KS(xf, wf, ks, so)
            asm volatile("s_waitcnt lgkmcnt(0)");
            MMA_KS(xf, wf)
        }
    }
#endif
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
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lw2), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const dim3 grid((M / BM) * (N / 256)), block(768);
    hipLaunchKernelGGL(gemm_lw2, grid, block, lds, 0, A, W, C, M, N, K);
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
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(gemm_lw2, grid, block, lds, 0, A, W, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_lw2, grid, block, lds, 0, A, W, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double rounds = std::ceil((double)((M / BM) * (N / 256)) / 256.0);
    printf("gemm_lw2 XREP=%d v%d M=%d N=%d K=%d: %.4f ms  %.1f TFLOP/s  (%.3f us per K-tile-round, %.3f scaled to 256 rows)  refcheck max|d| = %.3g %s\n", XREP, VARIANT, M, N, K,
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
