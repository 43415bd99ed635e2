// gemm3_pingpong.hip -- gemm2.hip's persistent 256x256x64 GEMM with a PING-PONG K loop.  NOT part of the library build:
// measured 7-11 % slower than gemm2.hip on MI355X (profiles/r01_gemm_tuning.md section 9), kept as the starting point for a
// later attempt.  To try it: copy next to csrc/gemm2.hip as gemm3.hip, add it to SRC in the Makefile, declare
// launch_gemm3{,_192,_mixed} / gemm3_init in csrc/gemm.hip and route plans A-D of launch_gemm to them.  Bit-identical
// results to gemm2.hip (the GEMM tests of tests/test_gpu_ops.py pass on it).
//
// Tile, LDS image, fragment layout, accumulator layout, K order and every epilogue are those of gemm2.hip (same bits).
// What differs is who does what when inside the K loop.  The eight waves form two groups (waves 0-3 / 4-7: the two
// waves that share each SIMD) that run HALF A PHASE APART, separated by workgroup barriers:
//
//     interval:   k          k+1         k+2         k+3
//     group 0:    MEM(p)     MFMA(p)     MEM(p+1)    MFMA(p+1)
//     group 1:    MFMA(p-1)  MEM(p)      MFMA(p)     MEM(p+1)
//
// so each SIMD always has one wave issuing nothing but MFMAs (at raised priority) and one wave doing the stall-prone work:
// the fragment ds_reads of its next quadrant and two global_load_lds (one HALF-TILE of a later K-tile).  A K-tile is four
// phases; phase = one quadrant of the wave's 128 x 64 result (two 32-row blocks x one 32-column block) x all four k-steps
// = 8 MFMAs on fragments read in that phase's MEM section (X rows only when the block pair changes, W rows when the column
// block changes: 12 + 4 + 8 + 4 reads).  A K-tile is staged as four half-tiles (XA = blocks 0,1; XB = blocks 2,3; W0; W1),
// each re-staged as soon as its last readers have retired their reads and passed a barrier -- up to 1.5 K-tiles before it
// is needed.  Ordering rules followed (cdna_hip_programming.md, "256^2 8-phase template"):
//   WAR  a half-tile is re-staged only after a barrier that every reader passed after its `s_waitcnt lgkmcnt(0)`;
//   RAW  every wave waits (counted vmcnt) for its pieces of K-tile t+1 during K-tile t, early enough that a reader passes
//        TWO barriers between that wait and its first read when the issuing wave is in the other group, one otherwise.
#include "device_types.h"
#include "kernels.h"

// tuning aid, compile-time only (make variant): 2 = skip in-loop staging, 4 = skip MFMA
#ifndef DINO_GEMM_DBG
#define DINO_GEMM_DBG 0
#endif

namespace dinov2 {

// XREP = 32-row MFMA blocks per wave along M: 4 -> 256-row tiles (the main configuration), 3 -> 192-row tiles, used by the
// dispatcher for the LAST partial round of a launch (688 tiles of 256 rows on 256 CUs are 2.69 rounds -> 3; two rounds of
// 256-row tiles plus one round of 192-row tiles cover the same rows in 2.79).  Same instruction schedule minus the fourth
// activation fragment; same K order, so a row's bits do not depend on the tile height.
template <typename T, int EPI, int XREP>
static __device__ __forceinline__ void gemm3_body(const GemmArgs& p, char* smem) {
    // No implicit mul+add -> fma contraction anywhere in this kernel: the unrolled epilogue instances would otherwise be
    // contracted differently, making an output element's last f32 bit (and, after the f16 rounding, occasionally its
    // value) depend on WHERE its row sits in the tile.  B images must equal B independent forwards bit for bit.
#pragma clang fp contract(off)
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    using vec4 = typename E::vec4;
    constexpr int BM = 64 * XREP, BN = 256, BK = 64, NW = 8;
    constexpr int ROWB = BK * 2;
    constexpr int STAGE = 512 * ROWB;  // 64 KiB per K-tile (X rows at 0, W rows at BM * ROWB), two stages
    constexpr int WREP = 2;                  // wave tile: 32 * XREP tokens x 64 output columns
    constexpr int WOFF = BM * ROWB;          // LDS offset of the weight rows inside a stage


    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // opaque: when two bodies run back to back (gemm3_mixed_kernel) nothing lane-derived is
                                     // shared between them and kept live across the first one's loops
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int ntiles = ntn * ntm;
    // Block b sits on XCD b % 8 (observed placement; affects speed only): each XCD walks a contiguous chunk of the tile
    // order, its blocks side by side, so concurrently running tiles share operand panels in that XCD's L2.
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int nb_x = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);  // blocks of this grid on my XCD
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int chunk0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int chunkn = tq + (xcd < tr ? 1 : 0);

    // logical tile id -> (m0, n0): groups of GM row panels are swept column by column, so the 32 tiles an XCD runs side
    // by side form an 8 x 4 patch (8 activation panels + 4 weight panels live in its L2) instead of 2 x 16
    // (2 + 16 panels): ~1.5x less refill traffic per K step.  Pure speed choice.
    constexpr int GM = 8;
    auto tile_mn = [&](int lid, int& m0, int& n0) {
        const int g = lid / (GM * ntn), r = lid - g * (GM * ntn);
        const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
        const int n = r / gm, mi = r - n * gm;
        m0 = (g * GM + mi) * BM;
        n0 = n * BN;
    };

    // ---- staging: a K-tile = four half-tiles, two (XB of a 192-row tile: one) global_load_lds_dwordx4 per wave each ----
    constexpr int H_XA = 0, H_XB = 1, H_W0 = 2, H_W1 = 3;
    const int srow = lane >> 3;
    auto piece_row0 = [&](int h, int u) {  // first tile row (of eight) that piece u of half-tile h covers for this wave
        const int rh = (u * NW + wid) * 8;
        if (h == H_XA) return (rh >> 6) * (32 * XREP) + (rh & 63);
        if (h == H_XB) return XREP == 4 ? (rh >> 6) * 128 + 64 + (rh & 63) : (rh >> 5) * 96 + 64 + (rh & 31);
        return (rh >> 5) * 64 + (h == H_W1 ? 32 : 0) + (rh & 31);
    };
    auto npieces = [](int h) { return (h == H_XB && XREP == 3) ? 1 : 2; };
    unsigned hsrc[4][2];  // byte offset from p.A / p.W of this lane's 16 bytes (both far below 4 GiB)
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u >= npieces(h)) continue;
                const int row = piece_row0(h, u) + srow;
                const int lc = (lane & 7) ^ ((row >> 1) & 7);
                if (h < 2) {
                    int gm = m0 + row;
                    gm = gm < M ? gm : M - 1;
                    hsrc[h][u] = (unsigned)gm * (unsigned)(K * 2) + lc * 16;
                } else {
                    hsrc[h][u] = (unsigned)(n0 + row) * (unsigned)(K * 2) + lc * 16;
                }
            }
    };
    // half-tile h of K-tile kt into stage buf (h is a literal at every call)
    auto half = [&](int h, int buf, int kt) {
        const char* base = (h < 2 ? (const char*)p.A : (const char*)p.W) + (size_t)kt * (BK * 2);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u >= npieces(h)) continue;
            glds16(base + hsrc[h][u], smem + buf * STAGE + (h < 2 ? 0 : BM * ROWB) + piece_row0(h, u) * ROWB);
        }
    };

    const int wx = wid >> 2, ww = wid & 3;
    const int grp = wid >> 2;  // waves 0-3 / 4-7: the two waves that share each SIMD
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    const int xoff = (wx * (32 * XREP) + fr) * ROWB;
    const int woff = (ww * 64 + fr) * ROWB;  // + WOFF goes into the instruction offset

    const unsigned lds0 = (unsigned)(uintptr_t)(DINO_LDS_AS char*)smem;
    unsigned xaddr[4], waddr[4];  // per k-step LDS byte address of this lane's first X / W fragment row (stage 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned ch = (unsigned)(((ks * 2 + fh) ^ sw) << 4);
        xaddr[ks] = lds0 + (unsigned)xoff + ch;
        waddr[ks] = lds0 + (unsigned)woff + ch;
    }

#if DINO_GEMM_DBG & 8  // tuning aid: block 0 / thread 0 stamps s_memtime at phase boundaries of its first tiles
    int tsn = 0;
#define DINO_TS() \
    if (p.ts && blockIdx.x == 0 && tid == 0 && tsn < 64) p.ts[tsn++] = (long long)__builtin_amdgcn_s_memtime();
#else
#define DINO_TS()
#endif

    const int nk = K / BK;  // even (checked by the launcher): the last K-tile sits in stage 1, stage 0 is free for the
                            // next tile's first K-tile while the epilogue works in stage 1
    if (bidx < chunkn) {
        int pm0, pn0;
        tile_mn(chunk0 + bidx, pm0, pn0);
        set_tile(pm0, pn0);
        half(H_XA, 0, 0);
        half(H_W0, 0, 0);
        half(H_W1, 0, 0);
        half(H_XB, 0, 0);
    }
    for (int tix = bidx; tix < chunkn; tix += nb_x) {
        DINO_TS();
        int m0, n0;
        tile_mn(chunk0 + tix, m0, n0);

        f32x16 acc[WREP][4];  // [.][3] untouched (and eliminated) when XREP == 3
#pragma unroll
        for (int j = 0; j < WREP; ++j)
#pragma unroll
            for (int i = 0; i < XREP; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

        u32x4 xq[2][4], wq[4];  // this phase's fragments: two activation blocks x four k-steps, one weight block x four k-steps

        // ---- main loop (see the header) ----------------------------------------------------------------------------------
#define PP_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define PP_READ_X(IP)                                                         \
    _Pragma("unroll") for (int ks__ = 0; ks__ < 4; ++ks__) {                  \
        const unsigned xa__ = xaddr[ks__] + cur;                              \
        PP_DSR(xq[0][ks__], xa__, (2 * (IP)) * 4096);                         \
        if (2 * (IP) + 1 < XREP) PP_DSR(xq[1][ks__], xa__, (2 * (IP) + 1) * 4096); \
    }
#define PP_READ_W(J)                                                          \
    _Pragma("unroll") for (int ks__ = 0; ks__ < 4; ++ks__) {                  \
        const unsigned wa__ = waddr[ks__] + cur;                              \
        PP_DSR(wq[ks__], wa__, WOFF + (J) * 4096);                            \
    }
        // barrier, then this wave's fragment reads must have returned; nothing may be scheduled across either
#define PP_SYNC_READS()                                                       \
    __builtin_amdgcn_sched_barrier(0);                                        \
    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);
#define PP_BARRIER()                                                          \
    __builtin_amdgcn_sched_barrier(0);                                        \
    asm volatile("s_barrier" ::: "memory");                                   \
    __builtin_amdgcn_sched_barrier(0);
#define PP_MFMAS(IP, J)                                                       \
    __builtin_amdgcn_s_setprio(1);                                            \
    _Pragma("unroll") for (int ks__ = 0; ks__ < 4; ++ks__) {                  \
        acc[J][2 * (IP)] = E::mfma32(__builtin_bit_cast(vec8, wq[ks__]), __builtin_bit_cast(vec8, xq[0][ks__]), acc[J][2 * (IP)]); \
        if (2 * (IP) + 1 < XREP)                                              \
            acc[J][2 * (IP) + 1] = E::mfma32(__builtin_bit_cast(vec8, wq[ks__]), __builtin_bit_cast(vec8, xq[1][ks__]), acc[J][2 * (IP) + 1]); \
    }                                                                         \
    __builtin_amdgcn_s_setprio(0);
#define PP_WAIT_STAGED(HAS1, F2)                                              \
    if (HAS1) {                                                               \
        if (F2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
    }

        __syncthreads();  // K-tile 0 of this tile has landed (vmcnt(0) precedes the barrier; also drains the previous
                          // tile's stores) and every wave has left the previous tile's epilogue slices in stage 1
        // the part of K-tile 1 that the steady-state schedule would have issued during "K-tile -1"
        half(H_XA, 1, 1);
        half(H_W1, 1, 1);
        if (grp == 1) {
            half(H_XB, 1, 1);
            PP_BARRIER();  // group 1 runs one barrier (half a phase) behind group 0 from here to the end of the K loop
        }
        const bool fetch = tix + nb_x < chunkn;  // another output tile follows: its first K-tile is staged at the tail
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned cur = (unsigned)(kt & 1) * STAGE;
            const int cb = kt & 1, ob = cb ^ 1;
            const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
            const bool f1 = has1 || fetch;                    // "K-tile kt+1" (or K-tile 0 of the next output tile) -> stage ob
            const bool f2 = has2 || (kt + 2 == nk && fetch);  // "K-tile kt+2" (or K-tile 0 of the next output tile) -> stage cb
            const int k1 = has1 ? kt + 1 : 0, k2 = has2 ? kt + 2 : 0;

            // ---- phase 0: blocks i = 0,1 x j = 0 ----
            PP_READ_X(0);
            PP_READ_W(0);
            if (grp == 0) { if (f1) half(H_XB, ob, k1); }
            else          { if (f1) half(H_W0, ob, k1); }
            if (grp == 1 && kt + 2 == nk && fetch) {  // from here on group 1 stages the next output tile
                int nm0, nn0;
                tile_mn(chunk0 + tix + nb_x, nm0, nn0);
                set_tile(nm0, nn0);
            }
            PP_SYNC_READS();
            PP_MFMAS(0, 0);
            PP_BARRIER();
            // ---- phase 1: blocks i = 0,1 x j = 1 ----
            PP_READ_W(1);
            if (grp == 0) { if (f1) half(H_W0, ob, k1); }
            else          { if (f2) half(H_XA, cb, k2); }
            if (grp == 0 && kt + 2 == nk && fetch) {  // ... and group 0 from here
                int nm0, nn0;
                tile_mn(chunk0 + tix + nb_x, nm0, nn0);
                set_tile(nm0, nn0);
            }
            PP_SYNC_READS();
            PP_MFMAS(0, 1);
            PP_BARRIER();
            // ---- phase 2: blocks i = 2,3 x j = 1 ----
            PP_READ_X(1);
            if (grp == 0) { if (f2) half(H_XA, cb, k2); }
            else          { if (f2) half(H_W1, cb, k2); }
            PP_SYNC_READS();
            PP_MFMAS(1, 1);
            if (grp == 1) PP_WAIT_STAGED(has1, f2);  // group 1's pieces of K-tile kt+1: two barriers before group 0 reads them
            PP_BARRIER();
            // ---- phase 3: blocks i = 2,3 x j = 0 ----
            PP_READ_W(0);
            if (grp == 0) { if (f2) half(H_W1, cb, k2); }
            else          { if (f2) half(H_XB, cb, k2); }
            PP_SYNC_READS();
            PP_MFMAS(1, 0);
            if (grp == 0) PP_WAIT_STAGED(has1, f2);  // group 0's pieces: one barrier before group 0, two before group 1 reads
            PP_BARRIER();
        }
        if (grp == 0) { PP_BARRIER(); }  // re-align the groups
#undef PP_DSR
#undef PP_READ_X
#undef PP_READ_W
#undef PP_SYNC_READS
#undef PP_BARRIER
#undef PP_MFMAS
#undef PP_WAIT_STAGED

        // ---- epilogue ----------------------------------------------------------------------------------------------
        // acc[j][i][4g + e] = C[m, n] with  m = m0 + wx*128 + i*32 + (lane & 31)
        //                                   n = n0 + ww*64 + j*32 + 8g + 4*(lane >> 5) + e
        // LDS slice image: 64 rows x 128 B, 16-byte slot s of row r stored at slot s ^ (r & 7) (conflict-free reads).
        // No block barrier is needed before writing the slices: they lie in stage 1, which nobody reads after the last
        // K-tile barrier (all k-step-3 fragments were in registers before it).
        // `el` launders the lane id: without it LICM hoists ~40 loop-invariant epilogue addresses out of the persistent
        // tile loop, they stay live across the K loop and the kernel spills (fatal next to the asm-loaded fragments).
        int el = lane;
        asm volatile("" : "+v"(el));
        const int er = el & 31, eh = el >> 5;
        char* const ep = smem + STAGE + wid * 8192;
        const int mbase = m0 + wx * (32 * XREP);
        const int ncol = n0 + ww * 64 + 4 * eh;

        float4 bs[WREP][4];
#pragma unroll
        for (int j = 0; j < WREP; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bs[j][g] = p.bias ? *(const float4*)(p.bias + ncol + j * 32 + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);

        if constexpr (EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_SWIGLU) {
            // 2-byte outputs: two passes of 64 rows x 64 columns (SwiGLU: x 32)
            const float qs = (EPI == EPI_QKV && n0 < p.qcols) ? p.qscale : 1.0f;  // tiles never straddle q|k|v
            constexpr int JN = EPI == EPI_SWIGLU ? 1 : WREP;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int j = 0; j < JN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float bb[4] = {bs[j][g].x, bs[j][g].y, bs[j][g].z, bs[j][g].w};
                        const float b2[4] = {bs[1][g].x, bs[1][g].y, bs[1][g].z, bs[1][g].w};
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int i = 2 * q + ii;
                            if (i >= XREP) continue;  // 192-row tiles: the second pass has one 32-row block
                            vec4 o;
#ifndef DINO_GELU_SCALAR
                            if constexpr (EPI == EPI_GELU) {
                                // Two columns per instruction: the bias add, x^2, the cubic, 1 + 2^t and the final product
                                // run as v_pk_*_f32 (IEEE results identical to the scalar ops of gemm.hip, so both kernels
                                // still agree bit for bit); v_exp / v_rcp / the f16 conversions stay per element.  The GELU
                                // epilogue was ~24 % of this kernel: 9.5 VALU + 2 transcendental instructions per element.
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                                for (int e2 = 0; e2 < 2; ++e2) {
                                    f32x2 v = {acc[j][i][4 * g + 2 * e2], acc[j][i][4 * g + 2 * e2 + 1]};
                                    v += f32x2{bb[2 * e2], bb[2 * e2 + 1]};
                                    asm volatile("" : "+v"(v));  // f32 sums first (no v_fma_mix fusion), then the f16 rounding
                                    const f32x2 xr = {(float)(_Float16)v[0], (float)(_Float16)v[1]};
                                    const f32x2 c1 = {-0.1029432397f, -0.1029432397f}, c2 = {-2.302208199f, -2.302208199f};
                                    const f32x2 t = xr * __builtin_elementwise_fma(xr * xr, c1, c2);  // -2 log2(e) u
                                    const f32x2 den = f32x2{1.0f, 1.0f} + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                                    f32x2 gl = xr * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                                    asm volatile("" : "+v"(gl));
                                    o[2 * e2] = E::from_f32((float)(_Float16)gl[0]);
                                    o[2 * e2 + 1] = E::from_f32((float)(_Float16)gl[1]);
                                }
                            } else
#endif
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[j][i][4 * g + e] + bb[e];
                                // keep the f32 sum a real register value: hipcc otherwise fuses "add, then round to f16"
                                // into v_fma_mixlo_f16 for SOME unrolled instances (single rounding instead of the
                                // reference's f32-then-f16 double rounding), which made results depend on the row's
                                // position in the tile
                                asm volatile("" : "+v"(v));
                                if constexpr (EPI == EPI_QKV) {
                                    float vq = v * qs;
                                    asm volatile("" : "+v"(vq));
                                    o[e] = E::from_f32(vq);
                                } else if constexpr (EPI == EPI_SWIGLU) {
                                    // W rows interleaved in 32-blocks: j = 0 holds x1[32q..], j = 1 holds x2[32q..]
                                    const float h2 = acc[1][i][4 * g + e] + b2[e];
                                    float sg = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)) * h2;  // silu(x1) * x2
                                    asm volatile("" : "+v"(sg));
                                    o[e] = E::from_f32(sg);
                                } else {
                                    // EPI_GELU, ggml semantics: y = table[f16(x)], table[h] = f16(gelu_tanh(f32(h))).
                                    // 0.5 x (1 + tanh u) == x / (1 + exp(-2u)); the reference's x <= -10 -> 0 and
                                    // x >= 10 -> x branches fall out of the formula after the f16 roundings (exp -> inf
                                    // gives -0, exp -> 0 gives x), so no compares are needed.
                                    const float xr = (float)(_Float16)v;
                                    const float t = xr * __builtin_fmaf(xr * xr, -0.1029432397f, -2.302208199f);  // -2 log2(e) u
                                    float gl = xr * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
                                    asm volatile("" : "+v"(gl));
                                    o[e] = E::from_f32((float)(_Float16)gl);
                                }
                            }
                            const int row = ii * 32 + er;
                            const int slot = (4 * j + g) ^ (row & 7);
                            *(vec4*)(ep + row * 128 + slot * 16 + eh * 8) = o;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                if constexpr (EPI == EPI_SWIGLU) {
                    const int hid0 = ((n0 + ww * 64) >> 6) * 32;  // 32 hidden units = 64 B per row: 4 lanes per row
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 16 + (el >> 2), slot = el & 3;
                        const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int m = mbase + q * 64 + row;
                        if (m < M && (XREP == 4 || q * 64 + row < 32 * XREP))
                            *(u32x4*)((T*)p.out + (size_t)m * p.ldo + hid0 + slot * 8) = v;
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = it * 8 + (el >> 3), slot = el & 7;
                        const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int m = mbase + q * 64 + row;
                        if (m < M && (XREP == 4 || q * 64 + row < 32 * XREP))
                            *(u32x4*)((T*)p.out + (size_t)m * p.ldo + n0 + ww * 64 + slot * 8) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // 4-byte outputs: four passes of 64 rows x 32 columns (128 B per row).  All loads of a pass (residual stream /
            // pos-embed rows) are issued before its LDS transposition and long before its first store.
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int j = ps >> 1, q = ps & 1;
                const int nb = n0 + ww * 64 + j * 32 + (el & 7) * 4;
                float4 add[8];
                if constexpr (EPI == EPI_RESID || EPI == EPI_PATCH) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        int m = mbase + q * 64 + it * 8 + (el >> 3);
                        m = m < M ? m : M - 1;
                        if constexpr (EPI == EPI_PATCH) {
                            const int pp = m % p.P;
                            add[it] = *(const float4*)(p.aux + (size_t)(1 + pp) * N + nb);
                        } else {
                            add[it] = *(const float4*)((const float*)p.out + (size_t)m * p.ldo + nb);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 ls = make_float4(1.f, 1.f, 1.f, 1.f);
                    if constexpr (EPI == EPI_RESID) ls = *(const float4*)(p.aux + ncol + j * 32 + 8 * g);
                    const float4 b4 = bs[j][g];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int i = 2 * q + ii;
                        if (i >= XREP) continue;
                        const int row = ii * 32 + er;
                        const int slot = (2 * g + eh) ^ (row & 7);
                        *(float4*)(ep + row * 128 + slot * 16) =
                            make_float4((acc[j][i][4 * g + 0] + b4.x) * ls.x, (acc[j][i][4 * g + 1] + b4.y) * ls.y,
                                        (acc[j][i][4 * g + 2] + b4.z) * ls.z, (acc[j][i][4 * g + 3] + b4.w) * ls.w);
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 8 + (el >> 3), slot = el & 7;
                    float4 v = *(const float4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                    if constexpr (EPI == EPI_RESID || EPI == EPI_PATCH)
                        v = make_float4(v.x + add[it].x, v.y + add[it].y, v.z + add[it].z, v.w + add[it].w);
                    const int m = mbase + q * 64 + row;
                    if (m < M && (XREP == 4 || q * 64 + row < 32 * XREP)) {
                        size_t o;
                        if constexpr (EPI == EPI_PATCH) {
                            const int b = m / p.P, pp = m - b * p.P;
                            o = ((size_t)b * p.T + 1 + p.R + pp) * p.ldo + nb;
                        } else {
                            o = (size_t)m * p.ldo + nb;
                        }
                        *(float4*)((float*)p.out + o) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        DINO_TS();
    }  // persistent tile loop
#undef DINO_TS
}

template <typename T, int EPI, int XREP>
__global__ __launch_bounds__(512) void gemm3_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm3_body<T, EPI, XREP>(p, smem);
}

// One launch, two tile heights: every block first walks its share of the 256-row tiles of `p` (whole rounds), then its share
// of the 192-row tiles of `q` (the remaining rows).  No grid-wide barrier in between -- a block that is done with its
// 256-row tiles starts on the 192-row ones at once -- which is what two back-to-back launches lacked (they were slower
// than the plain kernel for K = 1024).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm3_mixed_kernel(GemmArgs p, GemmArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm3_body<T, EPI, 4>(p, smem);
    gemm3_body<T, EPI, 3>(q, smem);
}

template <typename T, int XREP>
static hipError_t launch3_t(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int tiles = (a.N / 256) * ((a.M + 64 * XREP - 1) / (64 * XREP));
    const dim3 grid(tiles < 256 ? tiles : 256), block(512);
    const size_t lds = 2 * 512 * 128;
#define DINO_L2(E)                                                         \
    case E:                                                                \
        hipLaunchKernelGGL((gemm3_kernel<T, E, XREP>), grid, block, lds, st, a); \
        break;
    switch (epi) {
        case EPI_PATCH:  // the 256-row instantiation spills (the pos-embed prefetch on top of 128 accumulators); 192-row does not
            if (XREP == 4) return hipErrorInvalidValue;
            hipLaunchKernelGGL((gemm3_kernel<T, EPI_PATCH, 3>), grid, block, lds, st, a);
            break;
        DINO_L2(EPI_QKV)
        DINO_L2(EPI_RESID)
        DINO_L2(EPI_GELU)
        DINO_L2(EPI_SWIGLU)
        DINO_L2(EPI_PLAIN_F32)
    }
#undef DINO_L2
    return hipGetLastError();
}

// requires N % 256 == 0 and (K / 64) even
hipError_t launch_gemm3(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch3_t<_Float16, 4>(epi, a, st) : launch3_t<__bf16, 4>(epi, a, st);
}

template <typename T>
static hipError_t launch3_mixed_t(Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    const dim3 grid(256), block(512);
    const size_t lds = 2 * 512 * 128;
#define DINO_LM(E)                                                                  \
    case E:                                                                         \
        hipLaunchKernelGGL((gemm3_mixed_kernel<T, E>), grid, block, lds, st, a, b); \
        break;
    switch (epi) {
        DINO_LM(EPI_QKV)
        DINO_LM(EPI_RESID)
        DINO_LM(EPI_GELU)
        DINO_LM(EPI_SWIGLU)
        DINO_LM(EPI_PLAIN_F32)
        default: return hipErrorInvalidValue;
    }
#undef DINO_LM
    return hipGetLastError();
}

// 256-row tiles for `a` (must be >= 256 tiles), then 192-row tiles for `b`, in one launch (see gemm3_mixed_kernel)
hipError_t launch_gemm3_mixed(DType dt, Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    return dt == DT_F16 ? launch3_mixed_t<_Float16>(epi, a, b, st) : launch3_mixed_t<__bf16>(epi, a, b, st);
}

// same kernel with 192-row tiles (see gemm3_kernel)
hipError_t launch_gemm3_192(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch3_t<_Float16, 3>(epi, a, st) : launch3_t<__bf16, 3>(epi, a, st);
}

template <typename T, int XREP>
static hipError_t attr3_t() {
    hipError_t e = hipSuccess;
    const int lds = 2 * 512 * 128;
#define DINO_A2(E)                                                                  \
    if (e == hipSuccess)                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<T, E, XREP>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess && XREP == 3)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<T, EPI_PATCH, 3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DINO_A2(EPI_QKV)
    DINO_A2(EPI_RESID)
    DINO_A2(EPI_GELU)
    DINO_A2(EPI_SWIGLU)
    DINO_A2(EPI_PLAIN_F32)
#undef DINO_A2
#define DINO_A3(E)                                                                        \
    if (e == hipSuccess && XREP == 4)                                                     \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_mixed_kernel<T, E>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DINO_A3(EPI_QKV)
    DINO_A3(EPI_RESID)
    DINO_A3(EPI_GELU)
    DINO_A3(EPI_SWIGLU)
    DINO_A3(EPI_PLAIN_F32)
#undef DINO_A3
    return e;
}

hipError_t gemm3_init() {
    hipError_t e = attr3_t<_Float16, 4>();
    if (e == hipSuccess) e = attr3_t<__bf16, 4>();
    if (e == hipSuccess) e = attr3_t<_Float16, 3>();
    if (e == hipSuccess) e = attr3_t<__bf16, 3>();
    return e;
}

}  // namespace dinov2
