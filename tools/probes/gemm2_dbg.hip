// gemm2.hip -- the main f16/bf16 MFMA GEMM of the forward pass (256x256x64 tile, persistent), second generation.
//
// Same contract and epilogues as gemm.hip (which stays as the 128x128 edge-guarded kernel for small / odd shapes).
// Every structural choice below comes from a measurement on MI355X (profiles/r01_gemm_tuning.md):
//   * PERSISTENT workgroups (grid = min(tiles, 256), one per CU).  Retiring and relaunching a 512-thread / 128 KiB-LDS
//     workgroup per tile cost about as much as the whole K = 1024 loop (fixed 31 us per tile -> 15 us).
//   * OPERAND SWAP: the weight tile is the MFMA A operand and the activation tile the B operand, so a 32x32
//     accumulator block holds C^T: lane l owns ONE token row (l & 31) and, per 4-register group, FOUR CONSECUTIVE
//     output columns -> vector bias / LayerScale loads and 8/16-byte LDS writes in the epilogue.
//   * REGISTER DOUBLE-BUFFERED FRAGMENTS with HAND-COUNTED waits: the six ds_read_b128 of k-step s+1 are issued (inline
//     asm) before the eight MFMAs of k-step s, across the K-tile boundary too, and waited for with s_waitcnt lgkmcnt(6).
//     hipcc's own wait insertion put lgkmcnt(0) right behind freshly issued reads.
//   * LOADS TWO K-TILES AHEAD, ONE BARRIER PER K-TILE: global_load_lds for K-tile t+2 is issued right after the barrier
//     that publishes K-tile t+1; the NEXT output tile's first K-tile is issued after the last barrier of the current one,
//     so its HBM/L2 latency (8k cycles when exposed) hides under the epilogue.
//   * FULL-LINE EPILOGUE THROUGH LDS: each wave transposes its 128x64 result through a private 8 KiB slice of the idle
//     second LDS stage and moves whole 128-byte lines (lane l owns 16 B of row 8*it + (l >> 3)); residual-stream reads are
//     issued one pass ahead of the stores that would otherwise force a vmcnt(0) drain (gfx950 counts stores on vmcnt).
// Fragment layouts, LDS swizzle and the XCD-aware tile order are those of gemm.hip.
#include "device_types.h"
#include "kernels.h"

// tuning aid, compile-time only (make variant): 2 = skip in-loop staging, 4 = skip MFMA;
// 1024 = single-owner staging experiment (one wave group stages per K-loop iteration; measured slower, see profiles/r01_gemm_tuning.md)
#ifndef DINO_GEMM_DBG
#define DINO_GEMM_DBG 0
#endif

namespace dinov2 {

// XREP = 32-row MFMA blocks per wave along M: 4 -> 256-row tiles (the main configuration), 3 -> 192-row tiles, used by the
// dispatcher for the LAST partial round of a launch (688 tiles of 256 rows on 256 CUs are 2.69 rounds -> 3; two rounds of
// 256-row tiles plus one round of 192-row tiles cover the same rows in 2.79).  Same instruction schedule minus the fourth
// activation fragment; same K order, so a row's bits do not depend on the tile height.
template <typename T, int EPI, int XREP>
static __device__ __forceinline__ void gemm2_body(const GemmArgs& p, char* smem) {
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
    asm volatile("" : "+v"(tid));  // opaque: when two bodies run back to back (gemm2_mixed_kernel) nothing lane-derived is
                                     // shared between them and kept live across the first one's loops
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    const unsigned lda2 = (unsigned)(p.lda ? p.lda : K) * 2u, ldw2 = (unsigned)(p.ldw ? p.ldw : K) * 2u;  // row strides in bytes
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

    // ---- staging: 4 + 4 global_load_lds_dwordx4 per thread per K-tile, rows clamped to M ----
    unsigned xsrc[4], wsrc[4];  // byte offsets from p.A / p.W (both far below 4 GiB)
#if DINO_GEMM_DBG & 1024  // experiment: in K-loop iteration kt only the wave group (kt & 1) stages -- its rows AND its SIMD partner's
    unsigned xsrc2[4], wsrc2[4];
#endif
    const int srow = lane >> 3;
    auto set_tile = [&](int m0, int n0) {
#if DINO_GEMM_DBG & 256  // timing experiment only: every tile stages tile (0,0) -> operands stay L2-resident
        m0 = 0;
        n0 = 0;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (j * NW + wid) * 8 + srow;
#if DINO_GEMM_DBG & 32768  // timing experiment only (wrong results): un-swizzled source, lanes 0-7 of a row read its 128 B in order
            const int lc = lane & 7;
#else
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
#endif
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            if (j < XREP) xsrc[j] = (unsigned)gm * lda2 + lc * 16;
            wsrc[j] = (unsigned)(n0 + row) * ldw2 + lc * 16;
#if DINO_GEMM_DBG & 1024
            {
                const int row2 = (j * NW + (wid ^ 4)) * 8 + srow;
                const int lc2 = (lane & 7) ^ ((row2 >> 1) & 7);
                int gm2 = m0 + row2;
                gm2 = gm2 < M ? gm2 : M - 1;
                if (j < XREP) xsrc2[j] = (unsigned)gm2 * lda2 + lc2 * 16;
                wsrc2[j] = (unsigned)(n0 + row2) * ldw2 + lc2 * 16;
            }
#endif
        }
    };
    auto stage = [&](int buf, int kt) {
        char* sX = smem + buf * STAGE;
        char* sW = sX + BM * ROWB;
        const char* ga = (const char*)p.A + (size_t)kt * (BK * 2);
        const char* gw = (const char*)p.W + (size_t)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < XREP; ++j) glds16(ga + xsrc[j], sX + (j * NW + wid) * 8 * ROWB);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(gw + wsrc[j], sW + (j * NW + wid) * 8 * ROWB);
    };

    // one of the 8 wave-instructions of a K-tile (0-3: activation rows, 4-7: weight rows); j is a literal at every call
    auto piece = [&](int buf, int kt, int j) {
        if (DINO_GEMM_DBG & 2) return;
        if (j < 4 && j >= XREP) return;  // 192-row tiles have three activation pieces
        char* dst = smem + buf * STAGE + (j < 4 ? 0 : BM * ROWB) + ((j & 3) * NW + wid) * 8 * ROWB;
        const char* src = (j < 4 ? (const char*)p.A + xsrc[j & 3] : (const char*)p.W + wsrc[j & 3]) + (size_t)kt * (BK * 2);
        glds16(src, dst);
    };

#if DINO_GEMM_DBG & 1024
    auto piece2 = [&](int buf, int kt, int j) {  // the same piece of the SIMD partner (wave wid ^ 4)
        if (j < 4 && j >= XREP) return;
        char* dst = smem + buf * STAGE + (j < 4 ? 0 : BM * ROWB) + ((j & 3) * NW + (wid ^ 4)) * 8 * ROWB;
        const char* src = (j < 4 ? (const char*)p.A + xsrc2[j & 3] : (const char*)p.W + wsrc2[j & 3]) + (size_t)kt * (BK * 2);
        glds16(src, dst);
    };
#endif
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
        stage(0, 0);
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

        u32x4 xf0[4], wf0[WREP], xf1[4], wf1[WREP];
#if DINO_GEMM_DBG & 2048  // timing experiment only: no fragment ds_reads (MFMAs run on whatever these registers hold)
        for (int i = 0; i < 4; ++i) xf0[i] = xf1[i] = u32x4{(unsigned)tid, 1u, 2u, 3u};
        for (int i = 0; i < WREP; ++i) wf0[i] = wf1[i] = u32x4{(unsigned)tid, 5u, 6u, 7u};
#endif

        // ---- main loop ---------------------------------------------------------------------------------------------
        // Rules followed for the inline-asm reads (cdna_hip_programming.md 5.7): every asm read is waited for by an asm
        // s_waitcnt before its first consumer, and a sched_barrier(0) follows each wait so no MFMA is hoisted above it.
#define DINO_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DINO_LOAD_FRAGS(XF, WF, BUFOFF, KS)                                      \
    if (!(DINO_GEMM_DBG & 2048)) {                                               \
        const unsigned xa__ = xaddr[KS] + (BUFOFF), wa__ = waddr[KS] + (BUFOFF); \
        DINO_DSR(XF[0], xa__, 0);                                                \
        DINO_DSR(XF[1], xa__, 4096);                                             \
        DINO_DSR(XF[2], xa__, 8192);                                             \
        if (XREP == 4) DINO_DSR(XF[3], xa__, 12288);                             \
        DINO_DSR(WF[0], wa__, WOFF);                                             \
        DINO_DSR(WF[1], wa__, WOFF + 4096);                                      \
    }
#define DINO_WAIT_LGKM(N)                                         \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");    \
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NRD = XREP + WREP;  // fragment reads per k-step
#if DINO_GEMM_DBG & 512  // experiment: raise the wave's priority around its MFMA runs
#define DINO_PRIO(P) __builtin_amdgcn_s_setprio(P)
#else
#define DINO_PRIO(P)
#endif
#if DINO_GEMM_DBG & 4  // timing experiment only: no MFMA (one VALU op keeps the fragment registers and the accumulator live)
#define DINO_MFMA1(XF, WF, I, J) \
    acc[J][I][0] += __builtin_bit_cast(float, WF[J][0]) * __builtin_bit_cast(float, XF[I][0]);
#else
#define DINO_MFMA1(XF, WF, I, J) \
    acc[J][I] = E::mfma32(__builtin_bit_cast(vec8, WF[J]), __builtin_bit_cast(vec8, XF[I]), acc[J][I]);
#endif
    // Eight MFMAs of one k-step with staging instructions in four slots: S0 before the 1st MFMA, S1 after the 3rd, S2
    // after the 6th, S3 after the 8th.  A global_load_lds occupies its wave's issue port for ~60-185 cycles, so wave
    // group 0 (waves 0-3) stages in S0,S1,S2 and group 1 (waves 4-7, their SIMD partners) in S1,S2,S3: the two waves of a
    // SIMD are not both stuck in a staging instruction at the same moment.  Measured honestly: a burst of eight behind
    // the barrier vs. these slots is worth ~5 % in the micro-benchmark and nothing in-model; with the staging removed
    // altogether the kernel runs 786 -> 1182 TFLOP/s (cycles -20 %, clock +17 %), and neither the load latency (no-wait
    // experiment), nor the LDS reads (free), nor the barrier (6 %) explains it -- see profiles/r01_gemm_tuning.md.
#define DINO_MFMAS_P(XF, WF, S0, S1, S2, S3)       \
    {                                              \
        S0;                                        \
        __builtin_amdgcn_sched_barrier(0);         \
        DINO_PRIO(1);                              \
        DINO_MFMA1(XF, WF, 0, 0)                   \
        DINO_MFMA1(XF, WF, 0, 1)                   \
        DINO_MFMA1(XF, WF, 1, 0)                   \
        DINO_PRIO(0);                              \
        __builtin_amdgcn_sched_barrier(0);         \
        S1;                                        \
        __builtin_amdgcn_sched_barrier(0);         \
        DINO_PRIO(1);                              \
        DINO_MFMA1(XF, WF, 1, 1)                   \
        DINO_MFMA1(XF, WF, 2, 0)                   \
        DINO_MFMA1(XF, WF, 2, 1)                   \
        DINO_PRIO(0);                              \
        __builtin_amdgcn_sched_barrier(0);         \
        S2;                                        \
        __builtin_amdgcn_sched_barrier(0);         \
        DINO_PRIO(1);                              \
        if (XREP == 4) {                           \
            DINO_MFMA1(XF, WF, 3, 0)               \
            DINO_MFMA1(XF, WF, 3, 1)               \
        }                                          \
        DINO_PRIO(0);                              \
        __builtin_amdgcn_sched_barrier(0);         \
        S3;                                        \
        __builtin_amdgcn_sched_barrier(0);         \
    }
    // slot helpers: piece ja for group 0 / piece jb for group 1 of K-tile KT into stage BUF when COND holds
#if DINO_GEMM_DBG & 1024
#define DINO_SLOT_A(COND, BUF, KT, JA) if ((COND) && own) { piece(BUF, KT, JA); piece2(BUF, KT, JA); }
#define DINO_SLOT_B(COND, BUF, KT, JB)
#define DINO_SLOT_AB(COND, BUF, KT, JA, JB) DINO_SLOT_A(COND, BUF, KT, JA)
#else
#define DINO_SLOT_A(COND, BUF, KT, JA) if ((COND) && grp == 0) piece(BUF, KT, JA)
#define DINO_SLOT_B(COND, BUF, KT, JB) if ((COND) && grp == 1) piece(BUF, KT, JB)
#define DINO_SLOT_AB(COND, BUF, KT, JA, JB) \
    if (COND) {                             \
        if (grp == 0) piece(BUF, KT, JA);   \
        else piece(BUF, KT, JB);            \
    }
#endif
#define DINO_MFMAS(XF, WF) DINO_MFMAS_P(XF, WF, , , , )

        __syncthreads();  // K-tile 0 of this tile has landed (vmcnt(0) precedes the barrier; also drains the previous
                          // tile's stores) and every wave has left the previous tile's epilogue slices in stage 1
        DINO_TS();
        piece(1, 1, 0);
        piece(1, 1, 1);
        piece(1, 1, 2);
        DINO_LOAD_FRAGS(xf0, wf0, 0u, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned cur = (unsigned)(kt & 1) * STAGE, nxt = (unsigned)((kt + 1) & 1) * STAGE;
            const int nb = (kt + 1) & 1;
#if DINO_GEMM_DBG & 1024
            const bool own = (kt & 1) == grp;
#endif
            const bool more = kt + 1 < nk;      // K-tile kt+1 exists: its pieces 0-2 were issued behind the last barrier,
                                                // pieces 3-7 go out with the first two MFMA groups of this iteration
            DINO_LOAD_FRAGS(xf1, wf1, cur, 1);  // 12 reads in flight at most
            DINO_WAIT_LGKM(NRD);                  // the older six (k-step 0) have returned
            DINO_MFMAS_P(xf0, wf0, DINO_SLOT_A(more, nb, kt + 1, 3), DINO_SLOT_AB(more, nb, kt + 1, 4, 3),
                         DINO_SLOT_AB(more, nb, kt + 1, 5, 4), DINO_SLOT_B(more, nb, kt + 1, 5));
            DINO_LOAD_FRAGS(xf0, wf0, cur, 2);
            DINO_WAIT_LGKM(NRD);
            DINO_MFMAS_P(xf1, wf1, DINO_SLOT_A(more, nb, kt + 1, 6), DINO_SLOT_AB(more, nb, kt + 1, 7, 6),
                         DINO_SLOT_B(more, nb, kt + 1, 7), );
            DINO_LOAD_FRAGS(xf1, wf1, cur, 3);
            DINO_WAIT_LGKM(NRD);
            DINO_MFMAS(xf0, wf0);               // no staging here: slack for K-tile kt+1 to land before the barrier
            // k-step-3 fragments (issued one MFMA group ago) must be in registers before the barrier: after it nobody
            // reads stage kt&1 any more, so K-tile kt+2 may overwrite it.  __syncthreads adds vmcnt(0): this wave's part
            // of K-tile kt+1 has landed; after the barrier everyone's has.
            DINO_WAIT_LGKM(0);
#if DINO_GEMM_DBG & 8192  // timing experiment only (wrong results): the newest 8 pieces stay in flight across the barrier
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#elif DINO_GEMM_DBG & 16384  // timing experiment only (wrong results): 16 pieces (two K-tiles) stay in flight
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#elif DINO_GEMM_DBG & 64  // timing experiment only: no barrier at all
#elif DINO_GEMM_DBG & 16  // timing experiment only (wrong results): barrier WITHOUT waiting for the in-flight K-tile
            __builtin_amdgcn_s_barrier();
#else
            __syncthreads();
#endif
            DINO_LOAD_FRAGS(xf0, wf0, nxt, 0);  // after the last K-tile this reads LDS that is never used
            __builtin_amdgcn_sched_barrier(0);
            // MFMAs are never inside a branch (the accumulators would be copied at the join): only the staging
            // instructions are conditional.  pk = K-tile to fetch (kt+2, or 0 of the NEXT output tile after the last
            // barrier, when stage 0 is idle), pb = its stage.
            const bool last = kt + 1 == nk;
            const bool fetch = last ? (tix + nb_x < chunkn) : (kt + 2 < nk);
            if (last && fetch) {
                int nm0, nn0;
                tile_mn(chunk0 + tix + nb_x, nm0, nn0);
                set_tile(nm0, nn0);
            }
            const int pk = last ? 0 : kt + 2, pb = last ? 0 : (kt & 1);
            DINO_MFMAS_P(xf1, wf1, DINO_SLOT_A(fetch, pb, pk, 0), DINO_SLOT_AB(fetch, pb, pk, 1, 0), DINO_SLOT_AB(fetch, pb, pk, 2, 1),
                         DINO_SLOT_B(fetch, pb, pk, 2));
        }
        if (tix + nb_x < chunkn) {  // rest of the next tile's first K-tile: lands under the epilogue
            piece(0, 0, 3);
            piece(0, 0, 4);
            piece(0, 0, 5);
            piece(0, 0, 6);
            piece(0, 0, 7);
        }
        DINO_WAIT_LGKM(0);
        DINO_TS();
#undef DINO_DSR
#undef DINO_LOAD_FRAGS
#undef DINO_WAIT_LGKM
#undef DINO_MFMAS
#undef DINO_SLOT_A
#undef DINO_SLOT_B
#undef DINO_SLOT_AB
#undef DINO_MFMAS_P
#undef DINO_MFMA1

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
__global__ __launch_bounds__(512) void gemm2_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm2_body<T, EPI, XREP>(p, smem);
}

// One launch, two tile heights: every block first walks its share of the 256-row tiles of `p` (whole rounds), then its share
// of the 192-row tiles of `q` (the remaining rows).  No grid-wide barrier in between -- a block that is done with its
// 256-row tiles starts on the 192-row ones at once -- which is what two back-to-back launches lacked (they were slower
// than the plain kernel for K = 1024).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm2_mixed_kernel(GemmArgs p, GemmArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm2_body<T, EPI, 4>(p, smem);
    gemm2_body<T, EPI, 3>(q, smem);
}

template <typename T, int XREP>
static hipError_t launch2_t(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int tiles = (a.N / 256) * ((a.M + 64 * XREP - 1) / (64 * XREP));
    const dim3 grid(tiles < 256 ? tiles : 256), block(512);
    const size_t lds = 2 * 512 * 128;
#define DINO_L2(E)                                                         \
    case E:                                                                \
        hipLaunchKernelGGL((gemm2_kernel<T, E, XREP>), grid, block, lds, st, a); \
        break;
    switch (epi) {
        case EPI_PATCH:  // the 256-row instantiation spills (the pos-embed prefetch on top of 128 accumulators); 192-row does not
            if (XREP == 4) return hipErrorInvalidValue;
            hipLaunchKernelGGL((gemm2_kernel<T, EPI_PATCH, 3>), grid, block, lds, st, a);
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
hipError_t launch_gemm2(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch2_t<_Float16, 4>(epi, a, st) : launch2_t<__bf16, 4>(epi, a, st);
}

template <typename T>
static hipError_t launch2_mixed_t(Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    const dim3 grid(256), block(512);
    const size_t lds = 2 * 512 * 128;
#define DINO_LM(E)                                                                  \
    case E:                                                                         \
        hipLaunchKernelGGL((gemm2_mixed_kernel<T, E>), grid, block, lds, st, a, b); \
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

// 256-row tiles for `a` (must be >= 256 tiles), then 192-row tiles for `b`, in one launch (see gemm2_mixed_kernel)
hipError_t launch_gemm2_mixed(DType dt, Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    return dt == DT_F16 ? launch2_mixed_t<_Float16>(epi, a, b, st) : launch2_mixed_t<__bf16>(epi, a, b, st);
}

// same kernel with 192-row tiles (see gemm2_kernel)
hipError_t launch_gemm2_192(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch2_t<_Float16, 3>(epi, a, st) : launch2_t<__bf16, 3>(epi, a, st);
}

template <typename T, int XREP>
static hipError_t attr2_t() {
    hipError_t e = hipSuccess;
    const int lds = 2 * 512 * 128;
#define DINO_A2(E)                                                                  \
    if (e == hipSuccess)                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<T, E, XREP>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess && XREP == 3)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<T, EPI_PATCH, 3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DINO_A2(EPI_QKV)
    DINO_A2(EPI_RESID)
    DINO_A2(EPI_GELU)
    DINO_A2(EPI_SWIGLU)
    DINO_A2(EPI_PLAIN_F32)
#undef DINO_A2
#define DINO_A3(E)                                                                        \
    if (e == hipSuccess && XREP == 4)                                                     \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_mixed_kernel<T, E>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DINO_A3(EPI_QKV)
    DINO_A3(EPI_RESID)
    DINO_A3(EPI_GELU)
    DINO_A3(EPI_SWIGLU)
    DINO_A3(EPI_PLAIN_F32)
#undef DINO_A3
    return e;
}

hipError_t gemm2_init() {
    hipError_t e = attr2_t<_Float16, 4>();
    if (e == hipSuccess) e = attr2_t<__bf16, 4>();
    if (e == hipSuccess) e = attr2_t<_Float16, 3>();
    if (e == hipSuccess) e = attr2_t<__bf16, 3>();
    return e;
}

}  // namespace dinov2
