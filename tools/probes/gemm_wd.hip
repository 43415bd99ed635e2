// gemm_wd.hip -- probe: the 256 x 256 x 64 GEMM with the WEIGHT operand taken straight from L2 into registers ("W direct").
//
// Why: profiles/r02_gemm_kloop.md -- the K loop of gemm2_kernel is co-bound by the CU's LDS: per K-tile 64 KiB of LDS-DMA writes
// (>= 16 cycles per KiB) + 192 KiB of fragment reads ~ 1 800 LDS cycles against 2 048 MFMA cycles, and staging + reads + MFMA do not
// overlap (1.93 us per K-tile against 1.21 MFMA-paced).  Weights are STATIC: they can be laid out at load time in MFMA-fragment order
// (one 1-KiB block per 32 output columns x 16 k: lane l's 16 bytes at l * 16), so a wave fetches a W fragment with ONE fully
// coalesced global_load_dwordx4 and never touches LDS for it.  Per K-tile and CU: LDS-DMA 32 KiB + fragment reads 128 KiB
// (~1 000 LDS cycles), texture path 32 LDS-DMA pieces + 64 fragment loads (the two waves that share a wave column load the same
// blocks: L1 hits) instead of 64 pieces.
//
// Structure: 8 waves 2 (M) x 4 (N), wave tile 128 x 64, 32x32x16 MFMA, ONE barrier per K-tile.  X: 3-stage LDS ring of 32 KiB,
// global_load_lds two K-tiles ahead.  W: 8 fragments per K-tile and wave, loaded one K-tile ahead into a second register set.
// All VMEM return in order: per iteration the W loads are issued BEFORE the X pieces, so `vmcnt(4)` at the top of the next
// iteration retires W (and every older X piece) and leaves the newest X K-tile in flight.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_wd.hip -o /tmp/gemm_wd && /tmp/gemm_wd
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

#ifndef VARIANT
#define VARIANT 0  // timing-only (wrong results): 8 no X staging, 16 no X reads, 32 no MFMA, 64 no W loads
#endif

static __device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const GLB_AS void*)g, (LDS_AS void*)l, 16, 0, 0);
}

// W [N, K] row-major -> fragment-major: block (nb, kk) = rows 32 nb .. +31, k 16 kk .. +15, 1 KiB, lane l = (k half l >> 5, row l & 31)
__global__ void pack_w(const _Float16* __restrict__ W, _Float16* __restrict__ Wp, int N, int K) {
    const size_t nblk = (size_t)(N / 32) * (K / 16);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nblk * 64; i += (size_t)gridDim.x * blockDim.x) {
        const size_t blk = i >> 6;
        const int l = (int)(i & 63);
        const size_t nb = blk / (K / 16), kk = blk % (K / 16);
        const _Float16* s = W + (nb * 32 + (l & 31)) * (size_t)K + kk * 16 + (l >> 5) * 8;
        _Float16* d = Wp + blk * 512 + (size_t)l * 8;
        for (int e = 0; e < 8; ++e) d[e] = s[e];
    }
}

__global__ __launch_bounds__(512) void gemm_wd(const _Float16* __restrict__ A, const _Float16* __restrict__ Wp, float* __restrict__ C,
                                               int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int ntn = N / 256, ntm = M / 256, nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, q = nwg >> 3, rr = nwg & 7;
    const int lid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    constexpr int GM = 8;
    const int g = lid / (GM * ntn), r0 = lid - g * (GM * ntn);
    const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
    const int tn = r0 / gm, tmi = r0 - tn * gm;
    const int m0 = (g * GM + tmi) * 256, n0 = tn * 256;
    const int nk = K / 64;  // even (the K loop is unrolled by two so that the W register sets are named statically)

    // ---- X staging: 32 pieces of 8 rows per K-tile; this wave issues pieces wid, wid + 8, wid + 16, wid + 24
    unsigned xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 8 + wid) * 8 + (lane >> 3);
        xsrc[j] = (unsigned)(m0 + r) * (unsigned)(K * 2) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto stage_x = [&](int kt, int j) {  // j literal
        if (VARIANT & 8) return;
        glds16((const char*)A + (size_t)kt * 128 + xsrc[j], smem + (kt % 3) * 32768 + (j * 8 + wid) * 1024);
    };

    // ---- W fragments: per-lane pointer to block (nb0 + j, 4 kt + ks) of the fragment-major copy
    const char* wptr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wptr[j] = (const char*)Wp + ((size_t)((n0 + wc * 64) / 32 + j) * (K / 16)) * 1024 + lane * 16;
    // wptr names the NEXT pair of k-steps to fetch (2 KiB per pair and column block); never inside a branch: an asm load in one arm
    // of an if makes hipcc merge the arms with register copies -- of registers whose data has not landed yet
#define WLD(DST, J, KS) \
    if (!(VARIANT & 64)) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(wptr[J]), "n"((KS) * 1024))

    const int fr = lane & 31, hh = lane >> 5, sw = (fr >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    unsigned xa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xa[ks] = lds0 + (unsigned)((wr * 128 + fr) * 128) + (unsigned)(((ks * 2 + hh) ^ sw) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // W fragments: wP0[j][s] = k-steps 0, 1 of the current K-tile, wP1[j][s] = k-steps 2, 3 (j = 32-column block).  wP1 is loaded
    // while k-steps 0, 1 run, wP0 (for the NEXT K-tile) while k-steps 2, 3 run: half a K-tile of lookahead in 32 registers
    // (a full K-tile of lookahead, 64 registers, spilled).
    u32x4 xf0[4], xf1[4], wP0[2][2], wP1[2][2];
    if (VARIANT & (16 | 64))
        for (int i = 0; i < 4; ++i) {
            xf0[i] = xf1[i] = u32x4{(unsigned)tid, 1u, 2u, 3u};
            wP0[i >> 1][i & 1] = wP1[i >> 1][i & 1] = u32x4{(unsigned)tid, 5u, 6u, 7u};
        }

#define DSR(DST, ADDR, OFF) \
    if (!(VARIANT & 16)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define READ_X(XF, SO, KS)                      \
    {                                           \
        const unsigned a__ = xa[KS] + (SO);     \
        DSR(XF[0], a__, 0);                     \
        DSR(XF[1], a__, 4096);                  \
        DSR(XF[2], a__, 8192);                  \
        DSR(XF[3], a__, 12288);                 \
    }
#define WAIT_LGKM(N)                                           \
    {                                                          \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));        \
        __builtin_amdgcn_sched_barrier(0);                     \
    }
#define WAIT_VM(N)                                            \
    {                                                         \
        __builtin_amdgcn_sched_barrier(0);                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));         \
        __builtin_amdgcn_sched_barrier(0);                    \
    }
#define MMA1(XF, WP, S, I, J)                                                                                          \
    if (VARIANT & 32) acc[I][J][0] += __builtin_bit_cast(float, XF[I][0]) * __builtin_bit_cast(float, WP[J][S][0]);   \
    else acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, XF[I]), __builtin_bit_cast(f16x8, WP[J][S]), acc[I][J], 0, 0, 0);
#define MMA8(XF, WP, S)                                                          \
    {                                                                            \
        MMA1(XF, WP, S, 0, 0) MMA1(XF, WP, S, 0, 1) MMA1(XF, WP, S, 1, 0) MMA1(XF, WP, S, 1, 1) \
        MMA1(XF, WP, S, 2, 0) MMA1(XF, WP, S, 2, 1) MMA1(XF, WP, S, 3, 0) MMA1(XF, WP, S, 3, 1) \
        __builtin_amdgcn_sched_barrier(0);                                       \
    }
#define WLD4(WP, ADV)                                                             \
    {                                                                             \
        WLD(WP[0][0], 0, 0); WLD(WP[0][1], 0, 1); WLD(WP[1][0], 1, 0); WLD(WP[1][1], 1, 1); \
        __builtin_amdgcn_sched_barrier(0);                                        \
        if (!(VARIANT & 256)) wptr[0] += (ADV);                                    \
        if (!(VARIANT & 256)) wptr[1] += (ADV);                                    \
    }

    // ---- prologue: issue order X(0) x4, X(1) x2, W(0, k-steps 0-1) x4, X(1) x2 -- the order every later iteration leaves behind.
    // Loads are never conditional (see WLD): past the end of K the X pieces re-fetch the last K-tile into a ring slot nobody reads
    // any more and the W loads re-fetch the last pair, so every wait in the loop is the same counted vmcnt(2).
    stage_x(0, 0); stage_x(0, 1); stage_x(0, 2); stage_x(0, 3);
    stage_x(1, 0); stage_x(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    WLD4(wP0, 2048)
    stage_x(1, 2); stage_x(1, 3);
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned so = (unsigned)(kt % 3) * 32768u;
        const int kx = kt + 2 < nk ? kt + 2 : nk - 1;             // K-tile the X pieces of this iteration fetch
        char* const xdst = smem + ((kt + 2) % 3) * 32768;          // always the slot two ahead
#if VARIANT & 128
        // conservative waits: every load has half a K-tile to land and every wait is vmcnt(0) -- no assumption about the order in
        // which LDS-DMA and register loads retire
        WAIT_VM(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        READ_X(xf0, so, 0);
        READ_X(xf1, so, 1);
        if (!(VARIANT & 8)) {
            glds16((const char*)A + (size_t)kx * 128 + xsrc[0], xdst + (0 * 8 + wid) * 1024);
            glds16((const char*)A + (size_t)kx * 128 + xsrc[1], xdst + (1 * 8 + wid) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        WLD4(wP1, kt + 1 < nk ? 2048 : 0)
        WAIT_LGKM(4);
        MMA8(xf0, wP0, 0)
        READ_X(xf0, so, 2);
        WAIT_LGKM(4);
        MMA8(xf1, wP0, 1)
        READ_X(xf1, so, 3);
        WAIT_VM(0)
        if (!(VARIANT & 8)) {
            glds16((const char*)A + (size_t)kx * 128 + xsrc[2], xdst + (2 * 8 + wid) * 1024);
            glds16((const char*)A + (size_t)kx * 128 + xsrc[3], xdst + (3 * 8 + wid) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        WLD4(wP0, kt + 1 < nk ? 2048 : 0)
        WAIT_LGKM(4);
        MMA8(xf0, wP1, 0)
        WAIT_LGKM(0);
        MMA8(xf1, wP1, 1)
#else
        // top: W(kt, k-steps 0-1) and every older load have landed; only the two newest X pieces may still be in flight
        WAIT_VM(2)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        READ_X(xf0, so, 0);
        READ_X(xf1, so, 1);
        WLD4(wP1, kt + 1 < nk ? 2048 : 0)  // this K-tile's k-steps 2, 3 (the last pair of all is fetched twice: see above)
        WAIT_LGKM(4);
        MMA8(xf0, wP0, 0)
        READ_X(xf0, so, 2);
        if (!(VARIANT & 8)) {
            glds16((const char*)A + (size_t)kx * 128 + xsrc[0], xdst + (0 * 8 + wid) * 1024);
            glds16((const char*)A + (size_t)kx * 128 + xsrc[1], xdst + (1 * 8 + wid) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        WAIT_LGKM(4);
        MMA8(xf1, wP0, 1)
        READ_X(xf1, so, 3);
        WAIT_VM(2)                                  // wP1 landed (newer: the two X pieces just issued)
        WLD4(wP0, kt + 1 < nk ? 2048 : 0)           // next K-tile's k-steps 0, 1 (wP0 was last read by the MFMAs above)
        WAIT_LGKM(4);
        MMA8(xf0, wP1, 0)
        if (!(VARIANT & 8)) {
            glds16((const char*)A + (size_t)kx * 128 + xsrc[2], xdst + (2 * 8 + wid) * 1024);
            glds16((const char*)A + (size_t)kx * 128 + xsrc[3], xdst + (3 * 8 + wid) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        WAIT_LGKM(0);
        MMA8(xf1, wP1, 1)
#endif
    }
    // The padding loads of the last iteration are still in flight and hipcc believes their destination registers are dead: it
    // computed epilogue store addresses in them, the late data overwrote the addresses, the stores faulted.  Wait, then "use" the
    // registers, so that their live ranges reach past the wait and nothing else can be allocated there before the data has landed.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(wP0[0][0]), "v"(wP0[0][1]), "v"(wP0[1][0]), "v"(wP0[1][1]), "v"(wP1[0][0]), "v"(wP1[0][1]), "v"(wP1[1][0]), "v"(wP1[1][1]));
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
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
    _Float16 *A, *W, *Wp;
    float* C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&Wp, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4);
    fill<<<1024, 256>>>(A, (size_t)M * K, 1u); fill<<<1024, 256>>>(W, (size_t)N * K, 2u);
    pack_w<<<1024, 256>>>(W, Wp, N, K);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wd), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const dim3 grid((M / 256) * (N / 256)), block(512);
    hipLaunchKernelGGL(gemm_wd, grid, block, 98304, 0, A, Wp, C, M, N, K);
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
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(gemm_wd, grid, block, 98304, 0, A, Wp, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_wd, grid, block, 98304, 0, A, Wp, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    printf("gemm_wd v%d M=%d N=%d K=%d: %.4f ms  %.1f TFLOP/s  (%.3f us per K-tile-round)  refcheck max|d| = %.3g %s\n", VARIANT, M, N, K, ms,
           2.0 * M * N * K / ms / 1e9, ms * 1e3 / ((double)(K / 64) * (((M / 256) * (N / 256) + 255) / 256)), worst,
           worst < 2e-2 * std::sqrt((double)K / 1024) ? "OK" : "MISMATCH");
    hipFree(A); hipFree(W); hipFree(Wp); hipFree(C); hipFree(dm); hipFree(dn); hipFree(dr);
}

int main() {
    run(256, 256, 128, 1);
    run(512, 768, 1024, 10);
    run(4096, 4096, 4096, 50);
    run(8192, 8192, 8192, 10);
    run(43776, 4096, 1024, 50);
    return 0;
}
