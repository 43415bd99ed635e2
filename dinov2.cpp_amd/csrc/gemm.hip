// gemm.hip -- f16/bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// Replaces every weight `ggml_mul_mat` + the elementwise nodes that follow it in the reference graph
// (/root/reference/dinov2.cpp:471-474 qkv, :546-551 out-proj + :708-714 LayerScale/residual, :561-567 fc1+GELU,
//  :570-573 fc2 + :744-749, :582-605 weights_in+SwiGLU, :608-611 weights_out, :636-671 patch-embed + pos-embed).
// Numerics contract = ggml CPU mul_mat with F16 weights: activations rounded to the weight type, products and
// accumulation in f32 (v_mfma_f32_16x16x32_{f16,bf16}).
//
// Structure (MI355X-first, not a CUDA tiling):
//   * C[M,N] = A[M,K] * W[N,K]^T; both operands are K-contiguous, so A and B fragments are plain 16-byte
//     ds_read_b128 of a row-major [rows][64 k] LDS image (128-byte rows).
//   * 256x256x64 block tile, 8 wave64 (2 M x 4 N), each wave owns 128x64 = 4x2 MFMA 32x32 accumulators
//     (128 acc VGPRs); 128x128x64 / 4 waves variant for small grids.
//   * staging by global_load_lds_dwordx4 (HBM -> LDS without a VGPR round trip), double-buffered; the LDS image
//     is lane-linear per wave instruction, so the bank swizzle is applied on the per-lane SOURCE address and
//     again on the ds_read address: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7), which makes every
//     16-lane ds_read_b128 group hit 16 distinct bank slots.
//   * 1-D grid with an XCD-aware remap so tiles that share an A row-panel run on the same XCD (shared L2).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "device_types.h"
#include "kernels.h"

namespace dinov2 {

// -DDINO_PREC=<bits> (tuning builds for profiles/r05_parity_attribution.md; the small-tile kernel only -- run them with the "gemm_tile" = 128
// and "attn_v" = 1 switches): which roundings of the HIP path are REMOVED, one stage at a time, to attribute its distance to exact arithmetic.
//   1 / 8 / 16  q / k / v keep a second f16 word (lo = f16(x - f16(x)), columns [3H, 6H) of a 6H-wide qkv row) and attention adds the
//               cross terms K.Q_lo / K_lo.Q / V_lo.P on the matrix core: q, k, v to ~ 22 bits instead of 11
//   2           the un-normalised probabilities enter PV as hi + lo f16 words as well (P to ~ 22 bits)
//   4           GELU table entries from a double-precision tanh instead of v_exp_f32 / v_rcp_f32
#ifndef DINO_PREC
#define DINO_PREC 0
#endif

// KSUB = 64-wide K sub-tiles per LDS stage (1, or 2 for the few-tile shapes of a small batch: their K loop is a serial chain of
// wait -> barrier -> issue -> read -> MFMA per stage, and a stage twice as deep halves the number of links; same K order).
// -DDINO_GEMM_SPROF (tuning builds): wall-clock (100 MHz) sums of wave 0 of every workgroup -- [0] set-up and the first stages' issue, [1] counted
// wait + barrier, [2] issuing the next stage, [3] fragment reads + MFMAs, [4] epilogue, [5] workgroups -- printed after each launch.
#ifdef DINO_GEMM_SPROF
__device__ unsigned long long g_sprof[8];
__device__ unsigned long long g_sprof_t[2 * 1024];  // wall clock (100 MHz) at entry / exit of each workgroup
#define DINO_SP_INIT const unsigned long long sp_w0 = wall_clock64(); unsigned long long sp_t = sp_w0, sp_acc[5] = {0, 0, 0, 0, 0};
#define DINO_SP(i) { const unsigned long long t__ = wall_clock64(); sp_acc[i] += t__ - sp_t; sp_t = t__; }
#define DINO_SP_FLUSH if (tid == 0) { for (int i__ = 0; i__ < 5; ++i__) atomicAdd(&g_sprof[i__], sp_acc[i__]); atomicAdd(&g_sprof[5], 1ull); \
        if (blockIdx.x < 1024) { g_sprof_t[2 * blockIdx.x] = sp_w0; g_sprof_t[2 * blockIdx.x + 1] = wall_clock64(); } }
#else
#define DINO_SP_INIT
#define DINO_SP(i)
#define DINO_SP_FLUSH
#endif

template <typename T, int BM, int BN, int WM, int WN, int NST, int EPI, int KSUB = 1>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs p) {
#pragma clang fp contract(off)  // position-independent results: see gemm2.hip
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    constexpr int NW = WM * WN;
    constexpr int BK = 64;
    constexpr int ROWB = BK * 2;  // bytes per LDS row
    constexpr int SUBT = (BM + BN) * ROWB;  // one 64-wide K sub-tile: A rows, then W rows
    constexpr int STAGE = KSUB * SUBT;
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    constexpr int MREP = WTM / 16, NREP = WTN / 16;  // 16 x 16 accumulator blocks of the wave tile (MFMA 16x16x32)
    constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // glds wave-instructions per wave per tile
    static_assert(NREP == 4 || (EPI != EPI_SWIGLU && EPI != EPI_SWIGLU_LN), "the SwiGLU pairing assumes a 64-wide wave tile");
    // LN fold (kernels.h): LNC = this GEMM consumes T(gamma x) and applies the LayerNorm in its epilogue; EB = the epilogue it specialises
    constexpr bool LNC = EPI == EPI_QKV_LN || EPI == EPI_GELU_LN || EPI == EPI_SWIGLU_LN;
    constexpr int EB = EPI == EPI_RESID_LN ? EPI_RESID : EPI == EPI_QKV_LN ? EPI_QKV : EPI == EPI_GELU_LN ? EPI_GELU : EPI == EPI_SWIGLU_LN ? EPI_SWIGLU : EPI;

    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    DINO_SP_INIT

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    const size_t lda = p.lda ? p.lda : K, ldw = p.ldw ? p.ldw : K;

    const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
    const int lid = xcd_remap(blockIdx.x, ntn * ntm);
    const int m0 = (lid / ntn) * BM, n0 = (lid % ntn) * BN;

    // ---- per-lane staging sources (row clamped to the matrix: out-of-range rows re-read the last row) ----
    const char* asrc[AI];
    const char* bsrc[BI];
    const int srow = lane >> 3;  // row within an 8-row wave-instruction
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int row = (j * NW + wid) * 8 + srow;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        asrc[j] = (const char*)p.A + ((size_t)gm * lda) * 2 + lc * 16;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int row = (j * NW + wid) * 8 + srow;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        bsrc[j] = (const char*)p.W + ((size_t)gn * ldw) * 2 + lc * 16;
    }

    char* const smem = smem_all;  // the LDS ring (the launcher guarantees (K / 64) % KSUB == 0)
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int sb = 0; sb < KSUB; ++sb) {
            char* sA = smem + buf * STAGE + sb * SUBT;
            char* sB = sA + BM * ROWB;
            const size_t koff = (size_t)(kt * KSUB + sb) * (BK * 2);
#pragma unroll
            for (int j = 0; j < AI; ++j) glds16(asrc[j] + koff, sA + (j * NW + wid) * 8 * ROWB);
#pragma unroll
            for (int j = 0; j < BI; ++j) glds16(bsrc[j] + koff, sB + (j * NW + wid) * 8 * ROWB);
        }
    };

    // ---- fragment read offsets ----
    const int wm = wid / WN, wn = wid % WN;
    const int fr = lane & 15;             // row within a 16-row MFMA block
    const int fh = lane >> 4;             // which 8-wide k quarter of a 32-wide k step
    const int sw = (fr >> 1) & 7;         // swizzle term (block bases are multiples of 16 rows)
    const int aoff = (wm * WTM + fr) * ROWB;
    const int boff = BM * ROWB + (wn * WTN + fr) * ROWB;

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue operands, fetched BEFORE the K loop: these kernels live for 10-30 us, and a bias / LayerScale / residual fetch issued
    // after the last MFMA costs a whole memory round trip (2-4 k cycles of a 20-60 k cycle workgroup) that the K loop can hide.
    // vec_ok (uniform): every column of the tile exists and all row starts are 16-byte aligned -> vector loads and stores, no column guards.
    // acc[i][j][r] is C[row, col]: row = m0 + wm*WTM + i*16 + (lane&15), col = n0 + wn*WTN + j*16 + 4*(lane>>4) + r.
    const int colb = n0 + wn * WTN + 4 * fh;
    const int rowb = m0 + wm * WTM + fr;
    const bool vec_ok = EB != EPI_SWIGLU && EPI != EPI_PATCH && n0 + BN <= N && (p.ldo & 3) == 0 && (p.qcols & 3) == 0 &&
                        (((size_t)p.bias | (size_t)p.aux | (size_t)p.ln_s | (size_t)p.ln_c | (size_t)p.ln_gamma) & 15) == 0;
    // bias4: the additive per-column term (LN consumers: c[n], which contains the bias); lns4: LN consumers' s[n]
    f32x4 bias4[NREP], aux4[NREP], lns4[LNC ? NREP : 1], gam4[EPI == EPI_RESID_LN ? NREP : 1], xin4[EB == EPI_RESID ? MREP : 1][NREP];
    if (vec_ok) {
#pragma unroll
        for (int jn = 0; jn < NREP; ++jn) {
            const int col0 = colb + jn * 16;
            if constexpr (LNC) {
                bias4[jn] = *(const f32x4*)(p.ln_c + col0);
                lns4[jn] = *(const f32x4*)(p.ln_s + col0);
            } else {
                bias4[jn] = p.bias ? *(const f32x4*)(p.bias + col0) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (EPI == EPI_RESID_LN) gam4[jn] = *(const f32x4*)(p.ln_gamma + col0);
            if constexpr (EB == EPI_RESID) {
                aux4[jn] = *(const f32x4*)(p.aux + col0);
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    const int row = rowb + i * 16 < M ? rowb + i * 16 : M - 1;  // (clamped: rows past M are never stored)
                    xin4[i][jn] = *(const f32x4*)((const float*)p.out + (size_t)row * p.ldo + col0);
                }
            }
        }
    }

    // NST-stage LDS ring, tiles kt+1 .. kt+NST-1 in flight while tile kt is multiplied: with few workgroups per CU (small M)
    // nothing else hides the global -> LDS latency.  One barrier per K tile: (a) every wave's loads of tile kt have landed
    // (each wave waits for its own with a counted vmcnt -- loads return in order -- before the barrier), (b) every wave is
    // done reading the buffer that tile kt+NST-1 is about to overwrite.  Raw s_barrier: __syncthreads() would drain vmcnt.
    constexpr int LPT = KSUB * (AI + BI);  // glds instructions per wave per stage
    static_assert((NST - 2) * LPT < 64, "vmcnt is 6 bits");
    static_assert(!LNC || (NST - 1) * LPT < 64, "vmcnt is 6 bits (LN consumers wait with every first stage in flight)");
    const int nk = (K / BK) / KSUB;
    // LN consumers: the partial sums of this tile's rows ([BM][K / 64] x (sum, sum of squares)) go to LDS behind the ring by LDS-DMA,
    // AHEAD of the first stages (so every counted wait of the K loop covers them); they are turned into coefficients after the loop.
    float2* const lnst = (float2*)(smem_all + NST * STAGE);
    if constexpr (LNC) {
        const int hg = p.ln_gs / 2;  // 16-byte chunks per row of the statistics buffer
        const int nch = BM * hg;
        for (int f = wid * 64 + lane; f < nch; f += NW * 64) {
            const int r_ = f / hg, c_ = f - r_ * hg;
            const int gm = m0 + r_ < M ? m0 + r_ : M - 1;
            glds16(p.stats + ((size_t)gm * hg + c_) * 4, (char*)lnst + (size_t)(f - lane) * 16);
        }
    }
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) stage(t, t);
    // LN consumers: the LayerNorm coefficients (r, -mean r) of this lane's MREP rows, computed while the first stages are still in flight:
    // the statistics were requested before them, so a wait that leaves exactly the stages' requests outstanding (and a barrier: other
    // waves fetched other rows) is enough.  The four 16-lane groups each finalise ONE of the wave tile's 16-row blocks; the others come
    // by lane exchange.
    float lnr[LNC ? MREP : 1], lnn[LNC ? MREP : 1];
    if constexpr (LNC) {
        if (nk >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NST - 1) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        float r1, n1;
        ln_row_coeffs_lds(lnst + (size_t)(wm * WTM + (fh % MREP) * 16 + fr) * p.ln_gs, p.ln_gs, 1.0f / (float)K, p.ln_eps, r1, n1);
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            lnr[i] = __shfl(r1, fr + 16 * i);
            lnn[i] = __shfl(n1, fr + 16 * i);
        }
    }
    int buf = 0, nbuf = NST - 1;
    DINO_SP(0)
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;  // younger tiles that may still be in flight
        if (NST == 2 || ahead == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPT) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPT < 64 ? 2 * LPT : 0) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * LPT < 64 ? 3 * LPT : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * LPT < 64 ? 4 * LPT : 0) : "memory");
        static_assert(NST <= 6, "extend the vmcnt dispatch above");
        DINO_SP(1)
        if (kt + NST - 1 < nk) stage(nbuf, kt + NST - 1);
        DINO_SP(2)
        const char* s0 = smem + buf * STAGE;
        nbuf = buf;  // the buffer just consumed is the next to be refilled
        buf = buf + 1 == NST ? 0 : buf + 1;
#pragma unroll
        for (int sk = 0; sk < 2 * KSUB; ++sk) {  // k-steps of 32 in ascending K: the order of gemm2.hip, so both kernels give a row the same bits
            const char* s = s0 + (sk >> 1) * SUBT;
            const int ks = sk & 1;
            const int ch = ((ks * 4 + fh) ^ sw) << 4;
            vec8 af[MREP], bf[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) af[i] = *(const vec8*)(s + aoff + i * 16 * ROWB + ch);
#pragma unroll
            for (int j = 0; j < NREP; ++j) bf[j] = *(const vec8*)(s + boff + j * 16 * ROWB + ch);
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) {
                    acc[i][j] = E::mfma16(bf[j], af[i], acc[i][j]);  // operand swap (as gemm2.hip): a lane owns one row, four columns
                }
        }
#ifdef DINO_GEMM_SPROF
        asm volatile("s_nop 0" : "+v"(acc[0][0]));  // the section ends when the last MFMA chain's first link has issued, not retired
#endif
        DINO_SP(3)
    }

    // ---- epilogue: acc[i][j][r] is C[row, col] with
    //      row = m0 + wm*WTM + i*16 + (lane&15),  col = n0 + wn*WTN + j*16 + 4*(lane>>4) + r   (r = 0..3: four consecutive columns)
    // Edge-guarded per element in M and N (N = 1000-style shapes take the scalar path for their last columns).
    if constexpr (EB == EPI_SWIGLU) {
        // W rows interleaved in 32-blocks: columns 0..31 of the wave's 64 hold x1[32 units], columns 32..63 x2 of the same units
        T* out = (T*)p.out;
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c1 = colb + jh * 16 + r, c2 = c1 + 32;
                const float* const bsrc_ = LNC ? p.ln_c : p.bias;
                const float b1 = (bsrc_ && c1 < N) ? bsrc_[c1] : 0.f;
                const float b2 = (bsrc_ && c2 < N) ? bsrc_[c2] : 0.f;
                float s1 = 0.f, s2 = 0.f;
                if constexpr (LNC) {
                    s1 = c1 < N ? p.ln_s[c1] : 0.f;
                    s2 = c2 < N ? p.ln_s[c2] : 0.f;
                }
                const int hu = ((n0 + wn * WTN) >> 6) * 32 + jh * 16 + 4 * fh + r;  // hidden unit index
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    const int row = rowb + i * 16;
                    if (row < M && c2 < N) {
                        float h1, h2;
                        if constexpr (LNC) {
                            h1 = __builtin_fmaf(lnr[i], acc[i][jh][r], __builtin_fmaf(lnn[i], s1, b1));
                            h2 = __builtin_fmaf(lnr[i], acc[i][jh + 2][r], __builtin_fmaf(lnn[i], s2, b2));
                        } else {
                            h1 = acc[i][jh][r] + b1;
                            h2 = acc[i][jh + 2][r] + b2;
                        }
                        float sl = h1 * __builtin_amdgcn_rcpf(1.0f + __expf(-h1)) * h2;  // silu(x1) * x2, dinov2.cpp:605
                        asm("" : "+v"(sl));
                        out[(size_t)row * p.ldo + hu] = E::from_f32(sl);
                    }
                }
            }
        return;
    } else if (vec_ok) {
        // the common case: whole tile inside N, aligned rows; the operands were fetched before the K loop.  Arithmetic identical, expression
        // by expression, to the guarded path below (and to gemm2.hip).
        // One unconditional use of every prefetched register first: the compiler then waits for those loads HERE, once.  Left to the
        // row-guarded blocks below it puts an `s_waitcnt vmcnt(0)` into each of them, and on gfx9 vmcnt also counts stores: every
        // block would wait for the previous block's store to retire (measured: 8.8 us of a 20 us FFN-in launch at M = 1 374).
#pragma unroll
        for (int jn = 0; jn < NREP; ++jn) {
            asm volatile("" ::"v"(bias4[jn]));
            if constexpr (LNC) asm volatile("" ::"v"(lns4[jn]));
            if constexpr (EPI == EPI_RESID_LN) asm volatile("" ::"v"(gam4[jn]));
            if constexpr (EB == EPI_RESID) {
                asm volatile("" ::"v"(aux4[jn]));
#pragma unroll
                for (int i = 0; i < MREP; ++i) asm volatile("" ::"v"(xin4[i][jn]));
            }
        }
        // EPI_RESID_LN: per-row partial sums of the 16-column blocks go through LDS (behind the K loop's ring: other waves may still be
        // reading it), where the 64-column groups are completed in the fixed pairwise order
        float2* const red = (float2*)(smem_all + NST * STAGE);  // [BM][BN / 16]
#pragma unroll
        for (int jn = 0; jn < NREP; ++jn) {
            const int col0 = colb + jn * 16;
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int row = rowb + i * 16;
                if (row >= M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (LNC) v[r] = __builtin_fmaf(lnr[i], acc[i][jn][r], __builtin_fmaf(lnn[i], lns4[jn][r], bias4[jn][r]));
                    else v[r] = acc[i][jn][r] + bias4[jn][r];
                    asm("" : "+v"(v[r]));
                }
                if constexpr (EB == EPI_RESID) {
                    float* x = (float*)p.out + (size_t)row * p.ldo + col0;
                    const f32x4 xi = xin4[i][jn], au = aux4[jn];
                    const f32x4 xn = f32x4{v[0] * au[0] + xi[0], v[1] * au[1] + xi[1], v[2] * au[2] + xi[2], v[3] * au[3] + xi[3]};
                    *(f32x4*)x = xn;
                    if constexpr (EPI == EPI_RESID_LN) {
                        typename E::vec4 og;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float g = xn[r] * gam4[EPI == EPI_RESID_LN ? jn : 0][r];
                            asm("" : "+v"(g));  // f32 product first, then the rounding (as everywhere)
                            og[r] = E::from_f32(g);
                        }
                        *(typename E::vec4*)((T*)p.xg + (size_t)row * p.ldo + col0) = og;
                        float s4, q4;
                        ln_leaf4(xn[0], xn[1], xn[2], xn[3], s4, q4);
                        s4 += __shfl_xor(s4, 16);  // 8 columns
                        q4 += __shfl_xor(q4, 16);
                        s4 += __shfl_xor(s4, 32);  // 16 columns
                        q4 += __shfl_xor(q4, 32);
                        if (fh == 0) red[(wm * WTM + i * 16 + fr) * (BN / 16) + wn * NREP + jn] = make_float2(s4, q4);
                    }
                } else if constexpr (EPI == EPI_PLAIN_F32) {
                    float* x = (float*)p.out + (size_t)row * p.ldo + col0;
                    *(f32x4*)x = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    typename E::vec4 o;
                    if constexpr (EB == EPI_QKV) {
                        const float qs = col0 < p.qcols ? p.qscale : 1.0f;  // (qcols % 4 == 0: one answer for the lane's four columns)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float vq = v[r] * qs;
                            asm("" : "+v"(vq));
                            o[r] = E::from_f32(vq);
                        }
#if DINO_PREC & 25
                        if (p.ldo >= 2 * N) {  // second word of q | k | v: columns [N, 2N) of the (6H-wide) row model.cpp allocates in these builds
                                               // (a launch with a dense output -- dinov2_hip_op_gemm -- has nowhere to put it: ADVICE r5)
                            typename E::vec4 lo;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float vq = v[r] * qs;
                                asm("" : "+v"(vq));
                                lo[r] = E::from_f32(vq - E::to_f32(o[r]));
                            }
                            *(typename E::vec4*)((T*)p.out + (size_t)row * p.ldo + N + col0) = lo;
                        }
#endif
                    } else {
                        // two columns per instruction (v_pk_*_f32; v_exp / v_rcp / the conversions per element), exactly as gemm2.hip
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const f32x2 vv = {v[2 * e2], v[2 * e2 + 1]};
                            const f32x2 xr = __builtin_convertvector(__builtin_convertvector(vv, f16x2), f32x2);
                            const f32x2 c1 = {-0.1029432397f, -0.1029432397f}, c2 = {-2.302208199f, -2.302208199f};
                            const f32x2 t = xr * __builtin_elementwise_fma(xr * xr, c1, c2);  // -2 log2(e) u
                            const f32x2 den = f32x2{1.0f, 1.0f} + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                            f32x2 gl = xr * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
#if DINO_PREC & 4
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const double xd = (double)xr[e];
                                gl[e] = xd <= -10.0 ? 0.0f : xd >= 10.0 ? xr[e] : (float)(0.5 * xd * (1.0 + tanh(0.79788456080286535587989211986876 * xd * (1.0 + 0.044715 * xd * xd))));
                            }
#endif
                            asm("" : "+v"(gl));
                            o[2 * e2] = E::from_f32((float)(_Float16)gl[0]);
                            o[2 * e2 + 1] = E::from_f32((float)(_Float16)gl[1]);
                        }
                    }
                    *(typename E::vec4*)((T*)p.out + (size_t)row * p.ldo + col0) = o;
                }
            }
        }
    } else {
#pragma unroll
        for (int jn = 0; jn < NREP; ++jn) {
            const int col0 = colb + jn * 16;
            if (col0 >= N) continue;
            const bool full = col0 + 3 < N && (p.ldo & 3) == 0;  // the lane's four columns exist and rows are 16-byte aligned: vector path
            float bias[4], auxv[4], lnsv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + r < N ? col0 + r : N - 1;
                bias[r] = LNC ? p.ln_c[col] : p.bias ? p.bias[col] : 0.f;
                lnsv[r] = LNC ? p.ln_s[col] : 0.f;
                auxv[r] = 0.f;
                if constexpr (EB == EPI_RESID) auxv[r] = p.aux[col];
                if constexpr (EB == EPI_QKV) auxv[r] = col < p.qcols ? p.qscale : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int row = rowb + i * 16;
                if (row >= M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (LNC) v[r] = __builtin_fmaf(lnr[i], acc[i][jn][r], __builtin_fmaf(lnn[i], lnsv[r], bias[r]));
                    else v[r] = acc[i][jn][r] + bias[r];
                    asm("" : "+v"(v[r]));  // f32 value first, then any f16 rounding (no v_fma_mix fusion: gemm2.hip)
                }
                if constexpr (EPI == EPI_PATCH) {
                    const int b = row / p.P, pp = row - b * p.P;
                    float* x = (float*)p.out + ((size_t)b * p.T + 1 + p.R + pp) * p.ldo;
                    const float* pe = p.aux + (size_t)(1 + pp) * N;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < N) x[col0 + r] = v[r] + pe[col0 + r];
                } else if constexpr (EB == EPI_RESID) {  // (EPI_RESID_LN never gets here: launch_gemm only accepts shapes whose tiles are all vec_ok)
                    float* x = (float*)p.out + (size_t)row * p.ldo;
                    if (full) {
                        const float4 xin = *(const float4*)(x + col0);
                        *(float4*)(x + col0) = make_float4(v[0] * auxv[0] + xin.x, v[1] * auxv[1] + xin.y, v[2] * auxv[2] + xin.z, v[3] * auxv[3] + xin.w);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col0 + r < N) x[col0 + r] = v[r] * auxv[r] + x[col0 + r];
                    }
                } else if constexpr (EPI == EPI_PLAIN_F32) {
                    float* x = (float*)p.out + (size_t)row * p.ldo;
                    if (full) *(float4*)(x + col0) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col0 + r < N) x[col0 + r] = v[r];
                    }
                } else {  // EPI_QKV, EPI_GELU: 2-byte outputs
                    typename E::vec4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (EB == EPI_QKV) {
                            float vq = v[r] * auxv[r];
                            asm("" : "+v"(vq));
                            o[r] = E::from_f32(vq);
#if DINO_PREC & 25
                            if (col0 + r < N && p.ldo >= 2 * N) ((T*)p.out)[(size_t)row * p.ldo + N + col0 + r] = E::from_f32(vq - E::to_f32(o[r]));
#endif
                        } else {
                            // ggml_gelu = f16 lookup table: table[f16(x)] = f16(gelu(f32(f16(x)))).  EXACTLY the expression of
                            // gemm2.hip (same constants, same operation order; the x <= -10 / x >= 10 branches fall out of it):
                            // a token must get the same bits from either kernel, whatever batch it arrives in.
                            const float xr = (float)(_Float16)v[r];
                            const float t = xr * __builtin_fmaf(xr * xr, -0.1029432397f, -2.302208199f);  // -2 log2(e) u
                            float g = xr * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
#if DINO_PREC & 4
                            {
                                const double xd = (double)xr;
                                g = xd <= -10.0 ? 0.0f : xd >= 10.0 ? xr : (float)(0.5 * xd * (1.0 + tanh(0.79788456080286535587989211986876 * xd * (1.0 + 0.044715 * xd * xd))));
                            }
#endif
                            asm("" : "+v"(g));
                            o[r] = E::from_f32((float)(_Float16)g);
                        }
                    }
                    T* y = (T*)p.out + (size_t)row * p.ldo;
                    if (full) *(typename E::vec4*)(y + col0) = o;
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col0 + r < N) y[col0 + r] = o[r];
                    }
                }
            }
        }
    }
    if constexpr (EPI == EPI_RESID_LN) {
        // the 64-column groups of this tile's rows: (b0 + b1) + (b2 + b3) over the four 16-column blocks, one (row, group) per thread
        __syncthreads();
        const float2* const red = (const float2*)(smem_all + NST * STAGE);
        constexpr int GPT = BN / 64;
        if (tid < BM * GPT) {
            const int row = tid / GPT, g = tid % GPT;
            if (m0 + row < M) {
#pragma clang fp contract(off)
                const float2 b0 = red[row * (BN / 16) + 4 * g], b1 = red[row * (BN / 16) + 4 * g + 1], b2 = red[row * (BN / 16) + 4 * g + 2],
                             b3 = red[row * (BN / 16) + 4 * g + 3];
                const float2 o = make_float2((b0.x + b1.x) + (b2.x + b3.x), (b0.y + b1.y) + (b2.y + b3.y));
                *(float2*)(p.stats + ((size_t)(m0 + row) * p.ln_gs + (n0 >> 6) + g) * 2) = o;
            }
        }
    }
    DINO_SP(4)
    DINO_SP_FLUSH
}

template <typename T, int BM, int BN, int WM, int WN, int NST, int KSUB = 1>
static hipError_t launch_cfg(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM;
    const dim3 grid(ntn * ntm), block(WM * WN * 64);
    const size_t lds = NST * KSUB * (size_t)(BM + BN) * 128 + (epi == EPI_RESID_LN ? (size_t)BM * (BN / 16) * 8 : 0) +
                       (epi_ln_consumer(epi) ? (size_t)BM * a.ln_gs * 8 : 0);
#define DINO_LAUNCH(E)                                                                             \
    case E:                                                                                        \
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, NST, E, KSUB>), grid, block, lds, st, a); \
        break;
    switch (epi) {
        DINO_LAUNCH(EPI_PATCH)
        DINO_LAUNCH(EPI_QKV)
        DINO_LAUNCH(EPI_RESID)
        DINO_LAUNCH(EPI_GELU)
        DINO_LAUNCH(EPI_PLAIN_F32)
        DINO_LAUNCH(EPI_RESID_LN)
        DINO_LAUNCH(EPI_QKV_LN)
        DINO_LAUNCH(EPI_GELU_LN)
        case EPI_SWIGLU:  // (its column pairing needs 64-wide wave tiles)
            if constexpr (BN / WN == 64) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, NST, EPI_SWIGLU, KSUB>), grid, block, lds, st, a);
            else return hipErrorInvalidValue;
            break;
        case EPI_SWIGLU_LN:
            if constexpr (BN / WN == 64) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, NST, EPI_SWIGLU_LN, KSUB>), grid, block, lds, st, a);
            else return hipErrorInvalidValue;
            break;
    }
#undef DINO_LAUNCH
#ifdef DINO_GEMM_SPROF
    {
        unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipDeviceSynchronize();
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sprof), sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sprof), z, sizeof z);
        const double n = h[5] ? (double)h[5] : 1.0;
        static int shown = 0;
        if (shown++ % 50 == 49) {  // (every 50th launch: the bench loops)
            static unsigned long long tt[2 * 1024];
            (void)hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_sprof_t), sizeof tt);
            const int nb = (int)grid.x < 1024 ? (int)grid.x : 1024;
            unsigned long long t0 = ~0ull, t1 = 0, slast = 0;
            double life = 0, lmax = 0;
            for (int b = 0; b < nb; ++b) {
                t0 = tt[2 * b] < t0 ? tt[2 * b] : t0;
                t1 = tt[2 * b + 1] > t1 ? tt[2 * b + 1] : t1;
                slast = tt[2 * b] > slast ? tt[2 * b] : slast;
                const double l = (double)(tt[2 * b + 1] - tt[2 * b]);
                life += l;
                lmax = l > lmax ? l : lmax;
            }
            fprintf(stderr, "gemm_sprof wall (us): first start -> last end %.2f, last start at +%.2f, workgroup lifetime avg %.2f max %.2f\n", (t1 - t0) / 100.0,
                    (slast - t0) / 100.0, life / nb / 100.0, lmax / 100.0);
            fprintf(stderr, "gemm_sprof %dx%d w%dx%d nst%d ksub%d epi%d M=%d N=%d K=%d: (us) set-up %.2f  wait %.2f  issue %.2f  mma %.2f  epilogue %.2f  (%.0f WGs)\n", BM, BN, WM,
                    WN, NST, KSUB, (int)epi, a.M, a.N, a.K, h[0] / n / 100, h[1] / n / 100, h[2] / n / 100, h[3] / n / 100, h[4] / n / 100, n);
        }
    }
#endif
    return hipGetLastError();
}

template <typename T, int BM, int BN, int WM, int WN, int NST, int KSUB = 1>
static hipError_t set_attr_cfg() {
    const int lds = NST * KSUB * (BM + BN) * 128 + BM * LN_MAX_GROUPS * 8;  // (+ the LN-fold variants' row statistics, <= 24 KiB)
    hipError_t e = hipSuccess;
#define DINO_ATTR(E)                                                                                          \
    if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, BM, BN, WM, WN, NST, E, KSUB>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DINO_ATTR(EPI_PATCH)
    DINO_ATTR(EPI_QKV)
    DINO_ATTR(EPI_RESID)
    DINO_ATTR(EPI_GELU)
    DINO_ATTR(EPI_PLAIN_F32)
    DINO_ATTR(EPI_RESID_LN)
    DINO_ATTR(EPI_QKV_LN)
    DINO_ATTR(EPI_GELU_LN)
    if constexpr (BN / WN == 64) DINO_ATTR(EPI_SWIGLU)
    if constexpr (BN / WN == 64) DINO_ATTR(EPI_SWIGLU_LN)
#undef DINO_ATTR
    return e;
}

#ifdef DINO_GEMM_SWEEP
template <int BM, int BN, int WM, int WN, int NST, int KSUB>
static void hipFuncSetAttribute_all() { (void)set_attr_cfg<_Float16, BM, BN, WM, WN, NST, KSUB>(); }
#endif

hipError_t launch_gemm2(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st);      // gemm2.hip, 256-row tiles
hipError_t launch_gemm2_192(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st);  // gemm2.hip, 192-row tiles
hipError_t launch_gemm2_128(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st);  // gemm2.hip, 128-row tiles, one per workgroup
hipError_t launch_gemm2_mixed(DType dt, Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st);
hipError_t gemm2_init();
// gemm4.hip: the same tiles on four waves with a hand-ordered K loop (256-row tiles; whole rounds of 256-row + a round of 192-row tiles)
bool gemm4_ok(Epilogue epi, const GemmArgs& a);
hipError_t launch_gemm4(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st);
hipError_t launch_gemm4_mixed(DType dt, Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st);
hipError_t launch_gemm4_short(DType dt, Epilogue epi, const GemmArgs& a, int ni, hipStream_t st);  // 32 ni-row tiles, one per workgroup
hipError_t gemm4_init();

hipError_t gemm_init() {
    hipError_t e = set_attr_cfg<_Float16, 128, 128, 2, 2, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 128, 128, 2, 2, 2>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 128, 2, 2, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 128, 2, 2, 2>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 128, 2, 2, 3>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 128, 2, 2, 3>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 128, 4, 2, 3>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 128, 4, 2, 3>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 128, 4, 2, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 128, 4, 2, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 128, 2, 4, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 128, 2, 4, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 64, 64, 2, 4, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 64, 64, 2, 4, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<_Float16, 32, 64, 1, 4, 3, 2>();
    if (e == hipSuccess) e = set_attr_cfg<__bf16, 32, 64, 1, 4, 3, 2>();
    if (e == hipSuccess) e = gemm2_init();
    if (e == hipSuccess) e = gemm4_init();
    return e;
}

// ---- testing aids, read from the environment ONCE (ADVICE r4 / VERDICT r4 item 7a: they used to be getenv() calls on every launch) -------
namespace {
std::atomic<int> g_tune[TUNE_COUNT];
int g_tune_env[TUNE_COUNT];  // what the environment said at first use: what a negative value restores (tests leave the process as they found it)
std::once_flag g_tune_once;
void tune_init() {
    static const char* const names[TUNE_COUNT] = {"DINOV2_HIP_GEMM_GEN", "DINOV2_HIP_GEMM_TILE", "DINOV2_HIP_ATTN_V", "DINOV2_HIP_ATTN_NWV"};
    for (int k = 0; k < TUNE_COUNT; ++k) {
        const char* e = getenv(names[k]);
        g_tune_env[k] = e ? atoi(e) : 0;
        g_tune[k].store(g_tune_env[k], std::memory_order_relaxed);
    }
}
}  // namespace
int tune_get(TuneKey k) {
    std::call_once(g_tune_once, tune_init);
    return g_tune[k].load(std::memory_order_relaxed);
}
void tune_set(TuneKey k, int v) {
    std::call_once(g_tune_once, tune_init);
    g_tune[k].store(v < 0 ? g_tune_env[k] : v, std::memory_order_relaxed);
}

// ---- plan recording (gemm_plan_describe): with a sink installed the leaves of launch_gemm note their kernel instead of launching it ------
namespace {
thread_local std::string* t_plan_sink = nullptr;
}
static bool plan_note(const char* fmt, ...) {
    if (!t_plan_sink) return false;
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (!t_plan_sink->empty()) *t_plan_sink += ';';
    *t_plan_sink += buf;
    return true;
}
template <typename T, int BM, int BN, int WM, int WN, int NST, int KSUB = 1>
static hipError_t leaf_cfg(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    if (BN / WN != 64 && epi_base(epi) == EPI_SWIGLU) return hipErrorInvalidValue;  // (refused plans leave no text behind: ADVICE r5)
    if (plan_note("small<%dx%d,w%dx%d,st%d,ks%d>", BM, BN, WM, WN, NST, KSUB)) return hipSuccess;
    return launch_cfg<T, BM, BN, WM, WN, NST, KSUB>(epi, a, st);
}
#define DINO_LEAF(CALL, ...) (plan_note(__VA_ARGS__) ? hipSuccess : (CALL))

// Kernel choice: the 256x256 kernel (gemm2.hip) needs N % 256 == 0 (tiles must not straddle q|k|v or a SwiGLU pair,
// and it has no N edge guards) and enough tiles to fill the 256 CUs; everything else takes the 128x128 kernel here.
// ---- which kernel(s) for a shape --------------------------------------------------------------------------------------
// The persistent kernel (gemm2.hip; needs N % 256 == 0 and an even K / 64) works in rounds of 256 tiles, and a partial last
// round costs a whole one (QKV at batch 32: 2 064 tiles of 256 rows = 8 rounds + 16 tiles -> 9; out-proj / FFN-out: 688 =
// 2.69 -> 3; ViT-g at batch 8: 258 -> 2).  Every kernel here produces the same bits for a row
// (test_gemm_small_and_large_m_agree_bit_for_bit), so the plan is chosen purely by estimated time, in units of one round of
// 256-row tiles; a round of 192-row tiles is 0.79 (0.75 of the MFMA work, the same weight panel staged), the small-tile
// kernel runs at about half the persistent kernel's rate:
//   A  256-row tiles only                                   ceil(t256 / 256)
//   B  192-row tiles only                                   ceil(t192 / 256) * 0.79
//   C  whole rounds of 256-row tiles, rest 192-row tiles,   R + ceil(tail192 / 256) * 0.79      (gemm2_mixed_kernel)
//      one launch
//   D  whole rounds of 256-row tiles, rest small-tile       R + tail share of a round / 0.5 + 0.1 launch
//      kernel (two launches; pays for a few left-over tiles)
//   E  small-tile kernel only                               outputs / (256 tiles) / 0.5
// Measured (M = 43 968): QKV 0.292 -> 0.285 ms (D), out-proj 0.129 -> 0.125 (C), FFN-out 0.393 -> 0.374 (C); ViT-g bf16 batch 8
// 221 -> 244 images/s.  DINOV2_HIP_GEMM_TILE=128|256 forces E / A (testing aid: include/dinov2_hip.h, "Environment").
hipError_t launch_gemm(DType dt, Epilogue epi, const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    if (a.K % 64 != 0 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
    // 2-byte outputs much larger than the chip's L2s (8 x 4 MiB = 32 MiB) leave with non-temporal stores: they would only evict the operand
    // panels, and their consumer (another kernel) reads them through the memory side; small ones (batch 1) are found in L2 by the next
    // kernel.  The threshold, 48 MiB = 1.5 x the L2s, sits between the two regimes that were MEASURED (ViT-L batch 1: outputs <= 11 MB,
    // ordinary stores win by 3 % of p50; batch 32: outputs >= 90 MB, non-temporal stores win by 0.7 % of the forward -- profiles/r04_gemm4w.md
    // section 5b); nothing in between was.  Decided ONCE per logical output, at the top-level call: the parts of a split launch (`sub`)
    // inherit it, so one output never leaves under two store policies.
    const Epilogue eb = epi_base(epi);  // the LN-fold variants are dispatched like the epilogue they specialise
    const bool ln = eb != epi;
    if (!a.sub) {
        a.nt_out = 0;
        if (eb == EPI_QKV || eb == EPI_GELU || eb == EPI_SWIGLU)
            a.nt_out = (size_t)a.M * (size_t)(eb == EPI_SWIGLU ? a.N / 2 : a.N) * 2 > ((size_t)48 << 20) ? 1 : 0;
        a.clk_slot = DINO_CLK_GEMM_SLOT((int)epi, a.N, a.K);
        if (ln) {
            // LN fold: statistics are kept per 64 columns of the LayerNorm's rows, at most LN_MAX_GROUPS of them; the producer writes full
            // 64-column groups with vector stores from every kernel it may be split over
            const int hcols = epi == EPI_RESID_LN ? a.N : a.K;
            if (hcols % 128 != 0 || hcols / LN_GROUP > LN_MAX_GROUPS || (epi == EPI_RESID_LN && a.ldo != a.N)) return hipErrorInvalidValue;
            if (a.ln_gs == 0) a.ln_gs = ln_stat_slots(hcols);
            if (a.ln_gs != ln_stat_slots(hcols)) return hipErrorInvalidValue;
            if (!t_plan_sink) {  // (plan queries carry no pointers)
                if (!a.stats) return hipErrorInvalidValue;
                if (epi == EPI_RESID_LN && (!a.ln_gamma || !a.xg || (((size_t)a.ln_gamma | (size_t)a.aux | (size_t)a.bias) & 15))) return hipErrorInvalidValue;
                if (epi != EPI_RESID_LN && (!a.ln_s || !a.ln_c || (((size_t)a.ln_s | (size_t)a.ln_c) & 15))) return hipErrorInvalidValue;
            }
        }
    }
    // staging cursors are 32-bit byte offsets from A and W (dinov2_hip_predict splits batches long before this)
    const size_t lda_ = a.lda ? a.lda : a.K, ldw_ = a.ldw ? a.ldw : a.K;
    // rows are fetched with 16-byte global -> LDS DMA pieces and 16-byte vector loads: strides must keep rows 16-byte aligned
    if (a.lda < 0 || a.ldw < 0 || lda_ % 8 != 0 || ldw_ % 8 != 0 || lda_ < (size_t)a.K || ldw_ < (size_t)a.K) return hipErrorInvalidValue;
    if ((size_t)a.M * lda_ * 2 >= ((size_t)1 << 32) || (size_t)a.N * ldw_ * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    const int forced = tune_get(TUNE_GEMM_TILE);
    const int gen = tune_get(TUNE_GEMM_GEN);
#ifdef DINO_GEMM_SWEEP  // tuning builds only: DINOV2_HIP_GEMM_CFG=<n> forces one small-tile configuration (f16)
    {
        const char* e = getenv("DINOV2_HIP_GEMM_CFG");
        const int c = e ? atoi(e) : -1;
        if (c >= 0 && t_plan_sink) return hipErrorInvalidValue;  // (plan queries must not launch: ADVICE r5)
        if (c >= 0 && dt == DT_F16) {
            switch (c) {
#define DINO_SW(n, ...) case n: { static bool once = (hipFuncSetAttribute_all<__VA_ARGS__>(), true); (void)once; return launch_cfg<_Float16, __VA_ARGS__>(epi, a, st); }
                DINO_SW(0, 64, 128, 2, 2, 2, 1)
                DINO_SW(1, 64, 128, 2, 2, 3, 1)
                DINO_SW(2, 64, 128, 4, 2, 3, 2)
                DINO_SW(3, 64, 128, 4, 2, 6, 1)
                DINO_SW(4, 64, 128, 4, 2, 3, 1)
                DINO_SW(5, 64, 128, 4, 2, 4, 1)
                DINO_SW(6, 128, 128, 4, 2, 2, 1)
                DINO_SW(7, 128, 128, 4, 2, 3, 1)
                DINO_SW(8, 128, 128, 4, 2, 4, 1)
                DINO_SW(9, 192, 128, 4, 2, 3, 1)
                DINO_SW(10, 192, 128, 4, 2, 4, 1)
                DINO_SW(11, 64, 128, 2, 2, 6, 1)
                DINO_SW(12, 128, 128, 2, 2, 2, 1)
                DINO_SW(13, 64, 128, 2, 2, 4, 1)
                DINO_SW(14, 128, 128, 4, 2, 5, 1)
                DINO_SW(15, 64, 128, 2, 4, 3, 2)
                DINO_SW(16, 64, 128, 2, 4, 3, 1)
                DINO_SW(17, 128, 128, 2, 4, 2, 1)
                DINO_SW(18, 128, 128, 2, 2, 3, 1)
                DINO_SW(19, 128, 128, 2, 4, 3, 1)
                DINO_SW(20, 64, 64, 2, 2, 3, 2)
                DINO_SW(21, 64, 64, 2, 4, 3, 2)
                DINO_SW(22, 32, 128, 1, 4, 3, 2)
                DINO_SW(23, 32, 64, 1, 4, 3, 2)
                DINO_SW(24, 32, 64, 2, 2, 3, 2)
                DINO_SW(25, 64, 64, 2, 2, 4, 2)
                DINO_SW(26, 64, 64, 2, 2, 3, 1)
#undef DINO_SW
                default: return hipErrorInvalidValue;
            }
        }
    }
#endif
    // N not a multiple of 256 (ViT-S: 384, 1 152): the persistent kernel takes the leading multiple of 256 columns, the small-tile
    // kernel the remaining ones (two launches; every kernel gives a row the same bits, so the cut is invisible in the results).
    // Only where the persistent part fills the chip; not for the patch / SwiGLU epilogues (row -> token scatter indexed with N,
    // interleaved column pairs).
    {
        const int nrem = a.N % 256, n1 = a.N - nrem;
        const bool epi_ok = eb == EPI_QKV || eb == EPI_RESID || eb == EPI_GELU || eb == EPI_PLAIN_F32;
        if (!a.small_only && forced != 128 && nrem != 0 && n1 >= 256 && epi_ok && (a.K / 64) % 2 == 0 &&
            (long)(n1 / 256) * ((a.M + 191) / 192) >= 192) {
            const size_t osz = (eb == EPI_RESID || eb == EPI_PLAIN_F32) ? 4 : 2;
            GemmArgs a1 = a, a2 = a;
            a1.sub = a2.sub = 1;
            if (epi == EPI_RESID_LN) {  // (row strides of xg and stats come from ldo, which the parts keep)
                a2.ln_gamma = a.ln_gamma + n1;
                a2.xg = (char*)a.xg + (size_t)n1 * 2;
                a2.stats = a.stats + (size_t)(n1 / LN_GROUP) * 2;
            } else if (ln) {
                a2.ln_s = a.ln_s + n1;
                a2.ln_c = a.ln_c + n1;
            }
            a1.N = n1;
            a1.qcols = a.qcols < n1 ? a.qcols : n1;
            a2.N = nrem;
            a2.W = (const char*)a.W + (size_t)n1 * ldw_ * 2;
            a2.bias = a.bias ? a.bias + n1 : nullptr;
            a2.aux = a.aux ? a.aux + n1 : nullptr;  // (EPI_RESID: LayerScale [N]; unused by the other three)
            a2.out = (char*)a.out + (size_t)n1 * osz;
            a2.qcols = a.qcols > n1 ? a.qcols - n1 : 0;
            a2.small_only = 1;
            const hipError_t e = launch_gemm(dt, epi, a1, st);
            if (e != hipSuccess) return e;
            return launch_gemm(dt, epi, a2, st);
        }
    }
    constexpr bool split_ok = true;
    // (the patch-embed epilogue maps row -> (image, patch): no row splits for it)
    const bool is_patch = epi == EPI_PATCH;
    // (the LN-fold epilogues exist in gemm4.hip and in the small-tile kernel, not in gemm2.hip)
    const bool big_ok = !a.small_only && forced != 128 && a.N % 256 == 0 && (a.K / 64) % 2 == 0 && (!ln || (gen != 2 && gemm4_ok(epi, a)));
    if (big_ok) {
        const int ntn = a.N / 256;
        const long t256 = (long)ntn * ((a.M + 255) / 256), t192 = (long)ntn * ((a.M + 191) / 192);
        auto rnd = [](long t) { return (double)((t + 255) / 256); };
        const double unit = 256.0 * 65536.0;  // outputs of one round of 256-row tiles
        const double costE = (double)a.M * a.N / unit / 0.5;
        char plan = 'A';
        double best = rnd(t256);
        if (forced == 256) {
            best = -1.0;
        } else {
            if (t256 < 192 && costE < best) { plan = 'E'; best = costE; }  // far from filling the chip with big tiles
            if (split_ok && t192 >= 192 && rnd(t192) * 0.79 < best - 0.02) { plan = 'B'; best = rnd(t192) * 0.79; }
        }
        const long R = t256 / 256;
        const int panels1 = (int)(R * 256 / ntn);
        const int M1 = panels1 * 256;
        GemmArgs a1 = a, a2 = a;
        if (split_ok && !is_patch && forced != 256 && R >= 1 && M1 > 0 && M1 < a.M) {
            const size_t osz = (eb == EPI_RESID || eb == EPI_PLAIN_F32) ? 4 : 2;
            a1.M = M1;
            a2.M = a.M - M1;
            a2.A = (const char*)a.A + (size_t)M1 * lda_ * 2;
            a2.out = (char*)a.out + (size_t)M1 * a.ldo * osz;
            if (epi == EPI_RESID_LN) {
                a2.xg = (char*)a.xg + (size_t)M1 * a.ldo * 2;
                a2.stats = a.stats + (size_t)M1 * a.ln_gs * 2;
            } else if (ln) {
                a2.stats = a.stats + (size_t)M1 * a.ln_gs * 2;
            }
            a1.sub = a2.sub = 1;
            const long tail192 = (long)ntn * ((a2.M + 191) / 192);
            const double costC = (double)R + rnd(tail192) * 0.79;
            const double costD = (double)R + (double)a2.M * a.N / unit / 0.5 + 0.1;
            if (costC < best - 0.02) { plan = 'C'; best = costC; }
            if (costD < best - 0.02) { plan = 'D'; best = costD; }
        }
#ifdef DINO_GEMM_SWEEP
        if (forced == 192) plan = 'B';
        if (forced == 129 && !is_patch) return DINO_LEAF(launch_gemm2_128(dt, epi, a, st), "gemm2<128>");
#endif
        // F: 128-row tiles of the persistent kernel's schedule, one per workgroup, when they give 112 ... 256 workgroups: too few rows to
        // fill the chip with 256- or 192-row tiles, enough columns that 128 x 256 tiles do (ViT-L batch 1 at 518 x 518: QKV 132 tiles
        // 18.0 -> 15.0 us, FFN-in 176 tiles 21.1 -> 17.6; ViT-g QKV 42 -> 24; batch 4 FFN-out 64 -> 51).  Below ~112 tiles the
        // small-tile kernel's many workgroups win, above 256 a second round starts (profiles/r03_small_m_gemm.md section 9).
        // F4 (round 4): the same idea on gemm4.hip with the tile HEIGHT chosen for the launch: the shortest of 128 / 96 / 64 rows that still
        // gives at most one tile per CU (QKV at ViT-L batch 1: 180 tiles of 96 rows instead of 132 of 128; FFN-in 240 instead of 176),
        // for the 2-byte epilogues (profiles/r04_gemm4w.md section 6).  DINOV2_HIP_GEMM_GEN=2 keeps plan F.
        {
            const bool two_byte = eb == EPI_QKV || eb == EPI_GELU || eb == EPI_SWIGLU;
            if ((two_byte || epi == EPI_RESID_LN) && !forced && gen != 2 && gemm4_ok(epi, a) && (gen == 4 || ln || a.K >= 1024)) {
                int ni = 0;
                for (int c = 2; c <= 4 && !ni; ++c)
                    if ((long)ntn * ((a.M + 32 * c - 1) / (32 * c)) <= 256) ni = c;
                const long tiles = ni ? (long)ntn * ((a.M + 32 * ni - 1) / (32 * ni)) : 0;
                // (EPI_RESID_LN: only where the plain residual epilogue would leave the small-tile kernel too, at >= 112 tiles of 128 rows.
                //  Its epilogue is twice as long as the plain one and one wave per SIMD issues it; measured on ViT-L with the fold on:
                //  batch 2 (88 such tiles) small-tile kernel - 1.5 % of the forward against the unfolded path, 172 tiles of 64 rows - 8 %,
                //  116 of 96 rows - 13 %; batch 3 (132) small-tile - 16 %, 172 tiles of 96 rows - 3.5 %; batch 4 (172) small-tile - 15 %,
                //  232 tiles of 96 rows - 2.5 %, 172 of 128 rows - 5.6 % -- profiles/r06_ln_fold.md)
                if (epi == EPI_RESID_LN && (long)ntn * ((a.M + 127) / 128) < 112) ni = 0;
                if (ni && tiles >= 112) return DINO_LEAF(launch_gemm4_short(dt, epi, a, ni, st), "gemm4_short<%d>", 32 * ni);
            }
        }
        {
            const long t128r = (long)ntn * ((a.M + 127) / 128);
            if (!is_patch && !ln && !forced && t128r >= 112 && t128r <= 256) return DINO_LEAF(launch_gemm2_128(dt, epi, a, st), "gemm2<128>");
        }
        if (is_patch) plan = (plan == 'E' || t192 < 192) ? 'E' : 'B';  // only the 192-row instantiation exists for this epilogue
        // which generation runs the 256-row / mixed plans: gemm4.hip (four waves, hand-ordered K loop) where it applies, unless
        // DINOV2_HIP_GEMM_GEN=2 asks for gemm2.hip (testing aid: the bit-equality tests compare the two)
        // default: gemm4.hip wherever it applies (in the model: FFN-out - 4 %, QKV and FFN-in within 0.5 %, attn-out + 2 %; forward + 0.8 %
        // over gemm2.hip everywhere, same box, interleaved runs -- profiles/r04_gemm4w.md)
        // ... for K >= 1 024: with fewer K-tiles per output tile the four-wave kernel's longer epilogue (one wave per SIMD issues it alone)
        // outweighs its K loop -- ViT-B / ViT-S (K = 768 / 384) measured 1.4 % / 3 % faster on gemm2.hip, ViT-L / ViT-g on gemm4.hip.
        // DINOV2_HIP_GEMM_GEN=4 forces gemm4.hip wherever it can run (tests).
        // ... and not the residual epilogue at K < 2 048 (attn-out: its read-modify-write burst is 36 % of a tile and eight waves keep more
        // of it in flight: 0.131 against 0.133 ms in the model, four interleaved runs).
        const bool g4 = ln || (gen != 2 && gemm4_ok(epi, a) && (gen == 4 || (a.K >= 1024 && !(epi == EPI_RESID && a.K < 2048))));
        switch (plan) {
            case 'A': return g4 ? DINO_LEAF(launch_gemm4(dt, epi, a, st), "gemm4<256>") : DINO_LEAF(launch_gemm2(dt, epi, a, st), "gemm2<256>");
            case 'B': {
                if (!ln) return DINO_LEAF(launch_gemm2_192(dt, epi, a, st), "gemm2<192>");
                GemmArgs a0 = a;  // (the LN-fold epilogues: gemm4.hip's two-height launch with an empty 256-row part)
                a0.M = 0;
                return DINO_LEAF(launch_gemm4_mixed(dt, epi, a0, a, st), "gemm4_mixed<0+192>");
            }
            case 'C':
                return g4 ? DINO_LEAF(launch_gemm4_mixed(dt, epi, a1, a2, st), "gemm4_mixed<256+192>")
                          : DINO_LEAF(launch_gemm2_mixed(dt, epi, a1, a2, st), "gemm2_mixed<256+192>");
            case 'D': {
                const hipError_t e = g4 ? DINO_LEAF(launch_gemm4(dt, epi, a1, st), "gemm4<256>") : DINO_LEAF(launch_gemm2(dt, epi, a1, st), "gemm2<256>");
                if (e != hipSuccess) return e;
                a2.small_only = 1;  // the tail of a split goes straight to the small-tile kernel below
                return launch_gemm(dt, epi, a2, st);
            }
            default: break;  // 'E'
        }
    }
    // small problems (batch 1: M = 1374): 128x128 tiles leave most CUs idle and one workgroup per CU cannot hide the
    // global -> LDS latency of its K loop.  64x128 tiles with 2 LDS stages (48 KiB: three workgroups per CU hide each other's
    // latency) when there are enough tiles, with 3 stages (72 KiB) when there are not (N = 1024 at batch 1: 176 tiles; the
    // K = 4096 GEMM 39.7 -> 33.8 us).  Measured at M = 1374: 128x128 / 64x128x2 / 64x128x3 = qkv 21.8 / 19.5 / 27.1 us,
    // ffn-in 31.2 / 24.9 / 34.0, ffn-out 49.0 / 39.7 / 33.8.
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long t64 = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    const int cfg = t128 >= 512 ? 0 : t64 >= 384 ? 1 : 2;
    // Fewer tiles than CUs (batch 1: the two N = hidden GEMMs at 518 x 518, every GEMM at 224 x 224): one workgroup per CU walks
    // a serial K loop (wait -> barrier -> issue -> fragment reads -> MFMAs per stage) and nothing overlaps it, so what counts is how
    // short one link is (profiles/r03_small_m_gemm.md):
    //  * eight waves, two 64-wide K sub-tiles per LDS stage (half the links);
    //  * wave tiles of 32 x 32 (2 x 4 waves) rather than 16 x 64 (4 x 2): the loop is bound by LDS reads -- a 16 x 64 wave tile reads
    //    five 1 KiB fragments per four MFMAs, a 32 x 32 one four -- and the LDS moves 128 B/clk (FFN-out at M = 1 374: 28.0 -> 25.4 us);
    //  * the SMALLEST tile that still gives at most one workgroup per CU: every CU that joins shortens everybody's chain (M = 261:
    //    attn-out 9.8 -> 5.7 us, FFN-out 27.3 -> 14.5 on 32 x 64 tiles; QKV 8.9 -> 6.3 on 64 x 64); two workgroups per CU lose again.
    // Nothing about the arithmetic changes -- each output's MFMA chain runs over K in the same order in every configuration.
    if (cfg == 2 && t64 < 256 && (a.K / 64) % 2 == 0 && a.K >= 256 && eb != EPI_SWIGLU) {
        const long t6464 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64), t3264 = (long)((a.M + 31) / 32) * ((a.N + 63) / 64);
        if (t3264 <= 256) return dt == DT_F16 ? leaf_cfg<_Float16, 32, 64, 1, 4, 3, 2>(epi, a, st) : leaf_cfg<__bf16, 32, 64, 1, 4, 3, 2>(epi, a, st);
        if (t6464 <= 256) return dt == DT_F16 ? leaf_cfg<_Float16, 64, 64, 2, 4, 3, 2>(epi, a, st) : leaf_cfg<__bf16, 64, 64, 2, 4, 3, 2>(epi, a, st);
        return dt == DT_F16 ? leaf_cfg<_Float16, 64, 128, 2, 4, 3, 2>(epi, a, st) : leaf_cfg<__bf16, 64, 128, 2, 4, 3, 2>(epi, a, st);
    }
    if (cfg == 2 && t64 < 256 && (a.K / 64) % 2 == 0 && a.K >= 256)  // (SwiGLU pairs columns inside a 64-wide wave tile)
        return dt == DT_F16 ? leaf_cfg<_Float16, 64, 128, 4, 2, 3, 2>(epi, a, st) : leaf_cfg<__bf16, 64, 128, 4, 2, 3, 2>(epi, a, st);
    if (cfg == 2 && t64 < 256)
        return dt == DT_F16 ? leaf_cfg<_Float16, 64, 128, 4, 2, 3>(epi, a, st) : leaf_cfg<__bf16, 64, 128, 4, 2, 3>(epi, a, st);
    if (cfg == 1) return dt == DT_F16 ? leaf_cfg<_Float16, 64, 128, 2, 2, 2>(epi, a, st) : leaf_cfg<__bf16, 64, 128, 2, 2, 2>(epi, a, st);
    if (cfg == 2) return dt == DT_F16 ? leaf_cfg<_Float16, 64, 128, 2, 2, 3>(epi, a, st) : leaf_cfg<__bf16, 64, 128, 2, 2, 3>(epi, a, st);
    return dt == DT_F16 ? leaf_cfg<_Float16, 128, 128, 2, 2, 2>(epi, a, st) : leaf_cfg<__bf16, 128, 128, 2, 2, 2>(epi, a, st);
}

hipError_t gemm_plan_describe(DType dt, Epilogue epi, const GemmArgs& a, char* out, size_t cap) {
    if (!out || cap == 0) return hipErrorInvalidValue;
    std::string s;
    t_plan_sink = &s;
    const hipError_t e = launch_gemm(dt, epi, a, nullptr);
    t_plan_sink = nullptr;
    snprintf(out, cap, "%s", s.c_str());
    return e;
}

}  // namespace dinov2
