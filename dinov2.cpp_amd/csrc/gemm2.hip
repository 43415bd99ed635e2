// gemm2.hip -- the main f16/bf16 MFMA GEMM of the forward pass (256x256x64 tile, persistent), third generation.
//
// Same contract and epilogues as gemm.hip (which stays as the edge-guarded small-tile kernel for small / odd shapes).
// Every structural choice below comes from a measurement on MI355X (profiles/r01_gemm_tuning.md, profiles/r02_gemm_kloop.md):
//   * PERSISTENT workgroups (grid = min(tiles, 256), one per CU).  Retiring and relaunching a 512-thread / 128 KiB-LDS
//     workgroup per tile cost about as much as the whole K = 1024 loop (fixed 31 us per tile -> 15 us).
//   * MFMA 16x16x32, NOT 32x32x16: the same 8-phase schedule measured 1 066 / 1 137 TFLOP/s (4 096^3 / 8 192^3, f16, random
//     operands) with the 32x32x16 shape and 1 242 / 1 333 with 16x16x32; the previous generation of this kernel (32x32x16, one
//     barrier per K-tile, register double-buffered fragments) 1 098 / 1 110.
//   * OPERAND SWAP: the weight fragment is the MFMA A operand and the activation fragment the B operand, so an accumulator
//     block holds C^T: lane l owns ONE token (l & 15) and FOUR CONSECUTIVE output columns -> vector bias / LayerScale loads
//     and 8/16-byte LDS writes in the epilogue.
//   * THE LOCAL GUIDE'S 8-PHASE SCHEDULE: quadrant phases, half-tile staging by global_load_lds one and a half K-tiles ahead with
//     ONE counted s_waitcnt vmcnt(4) per K-tile, raw s_barrier (no vmcnt(0) drain), the two waves of a SIMD half a phase apart.
//     The NEXT output tile's first K-tile is staged during the last phases, so its HBM/L2 latency hides under the epilogue.
//   * FULL-LINE EPILOGUE THROUGH LDS: each wave transposes its 128x64 result through a private 8 KiB slice of the idle
//     second LDS buffer and moves whole 128-byte lines (lane l owns 16 B of row 8*it + (l >> 3)); residual-stream reads are
//     issued one pass ahead of the stores that would otherwise force a vmcnt(0) drain (gfx950 counts stores on vmcnt).
// LDS swizzle and the XCD-aware tile order are those of gemm.hip.
#include <cstdio>

#include "device_types.h"
#include "kernels.h"

// The timing-only experiment switches this kernel carried while it was tuned (no staging / no MFMA / no fragment reads / deeper
// prefetch / L2-resident operands ...) live in tools/probes/gemm2_dbg.hip; what they measured: profiles/r01_gemm_tuning.md,
// profiles/r02_gemm_kloop.md.  The product kernel has one schedule.

namespace dinov2 {

// 16-byte store of the 2-byte epilogues: non-temporal for outputs larger than the L2s (GemmArgs::nt_out; see gemm4.hip, DINO4_ST16 --
// ViT-B / ViT-S at batch 32, which run on this kernel: + 1.7 % / + 2.6 % images/s)
#define DINO2_ST16(PTR, V)                               \
    {                                                    \
        if (nt_out) __builtin_nontemporal_store((V), (PTR)); \
        else *(PTR) = (V);                               \
    }

// -DDINO_GEMM_PROF (tuning builds): s_memtime sums per workgroup -- [0] tile prologue (both barriers + the counted wait), [1] K loop,
// [2] epilogue, [3] tiles -- of wave 0 and of wave 4 (+ 4), printed by the launcher after each launch.
#ifdef DINO_GEMM_PROF
__device__ unsigned long long g_gemm_prof[256 * 8];
#define DINO_GP_INIT unsigned long long gp_t = __builtin_readcyclecounter(), gp_acc[4] = {0, 0, 0, 0};
#define DINO_GP(i) { const unsigned long long t__ = __builtin_readcyclecounter(); gp_acc[i] += t__ - gp_t; gp_t = t__; }
#define DINO_GP_FLUSH if (lane == 0 && (wid == 0 || wid == 4)) for (int i__ = 0; i__ < 4; ++i__) g_gemm_prof[blockIdx.x * 8 + (wid ? 4 : 0) + i__] += gp_acc[i__];
#else
#define DINO_GP_INIT
#define DINO_GP(i)
#define DINO_GP_FLUSH
#endif

// XREP = 32-token blocks per wave along M: 4 -> 256-row tiles (the main configuration), 3 -> 192-row tiles, used by the
// dispatcher for the LAST partial round of a launch (688 tiles of 256 rows on 256 CUs are 2.69 rounds -> 3; two rounds of
// 256-row tiles plus one round of 192-row tiles cover the same rows in 2.79).  Same schedule with a half-height second token
// half; same K order, so a row's bits do not depend on the tile height.
// XREP = 2 -> 128-row tiles for launches of at most ~256 of them (batch 1 at 518 x 518: QKV 132, FFN-in 176 tiles; ONE tile per
// workgroup, no persistence): there is no second token half, so a K-tile is TWO phases -- (0,0) and (0,1) -- and three half-tile slots
// (X0, W0, W1: 48 KiB); K-tiles rotate through a ring of three such buffers and K-tile t + 2 is staged under K-tile t, so that a
// K-tile is still two K-tile times in flight although a K-tile takes half as long.
template <typename T, int EPI, int XREP>
static __device__ __forceinline__ void gemm2_body(const GemmArgs& p, char* smem) {
    // No implicit mul+add -> fma contraction anywhere in this kernel: the unrolled epilogue instances would otherwise be
    // contracted differently, making an output element's last f32 bit (and, after the f16 rounding, occasionally its
    // value) depend on WHERE its row sits in the tile.  B images must equal B independent forwards bit for bit.
#pragma clang fp contract(off)
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    using vec4 = typename E::vec4;
    constexpr int BM = 64 * XREP, BN = 256, BK = 64;
    constexpr bool ONE = XREP == 2;          // 128-row tiles: one token half, one tile per workgroup, three-buffer ring
    constexpr int BUF = ONE ? 49152 : 65536; // one K-tile: half-tile slots of 16 KiB -- X0, X1 (token halves; 128-row tiles: X0 only), W0, W1 (column halves)
    constexpr unsigned WOFF = ONE ? 16384u : 32768u;  // W0's slot in a buffer (W1 follows it)
    constexpr int RX1 = 32 * (XREP - 2);     // tokens per wave-row in the second token half: 64 (256-row tiles), 32 (192-row tiles) or none
    constexpr int NI1 = RX1 / 16;            // 16-token blocks of the second half: 4 or 2

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // opaque: when two bodies run back to back (gemm2_mixed_kernel) nothing lane-derived is
                                     // shared between them and kept live across the first one's loops
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N, K = p.K;
    const bool nt_out = p.nt_out != 0;
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
#ifndef DINO_GEMM_GM
#define DINO_GEMM_GM 8  // (tuning builds: -DDINO_GEMM_GM=4|16|32; traffic and time of each in profiles/r03_gemm_notes.md section 4)
#endif
    constexpr int GM = DINO_GEMM_GM;
    auto tile_mn = [&](int lid, int& m0, int& n0) {
        const int g = lid / (GM * ntn), r = lid - g * (GM * ntn);
        const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
        const int n = r / gm, mi = r - n * gm;
        m0 = (g * GM + mi) * BM;
        n0 = n * BN;
    };

    const int wx = wid >> 2, ww = wid & 3;  // wave tile: 32 * XREP tokens (wave-row wx) x 64 output columns (wave-column ww)

    // ---- staging: a half-tile is an image of 8-row x 128-byte pieces (one global_load_lds_dwordx4 wave-instruction each, lane ->
    // row lane >> 3, 16-byte chunk lane & 7, the chunk XOR-swizzled on the SOURCE side).  Image row r of token half a belongs to
    // wave-row r / RX (RX = 64, or 32 for the second half of a 192-row tile), local token r % RX; image row r of column half b
    // to wave-column r >> 5, local column r & 31.  A wave issues pieces 2 wid and 2 wid + 1 of a 16-piece half-tile (piece wid of
    // the 8-piece one).  Rows are clamped to M.
    unsigned src[4][2];  // byte offsets from p.A / p.W of this wave's pieces: [X0, X1, W0, W1][piece]
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const bool half8 = h == 1 && RX1 == 32;  // the 8-piece half-tile
                if (half8 && pc == 1) continue;
                if (h == 1 && ONE) continue;
                const int r = (half8 ? wid : 2 * wid + pc) * 8 + (lane >> 3);
                const int ch = (lane & 7) ^ ((r >> 1) & 7);
                if (h < 2) {
                    const int rx = (h == 0 || ONE) ? 64 : RX1;
                    int gm = m0 + (r / rx) * (32 * XREP) + h * 64 + (r % rx);
                    gm = gm < M ? gm : M - 1;
                    src[h][pc] = (unsigned)gm * lda2 + ch * 16;
                } else {
                    src[h][pc] = (unsigned)(n0 + (r >> 5) * 64 + (h - 2) * 32 + (r & 31)) * ldw2 + ch * 16;
                }
            }
    };
    auto stage = [&](int h, int kt, int buf) {  // h is a literal at every call site
        const char* base = (h < 2 ? (const char*)p.A : (const char*)p.W) + (size_t)kt * (BK * 2);
        const bool half8 = h == 1 && RX1 == 32;
        if (h == 1 && ONE) return;
        char* dst = smem + buf * BUF + (ONE && h >= 2 ? h - 1 : h) * 16384 + (half8 ? wid : 2 * wid) * 1024;
#ifdef DINO_GEMM2_NTX  // tuning builds: non-temporal LDS-DMA for the X pieces of the residual epilogue (profiles/r05_gemm4_nt_loads.txt)
        if (EPI == EPI_RESID && h < 2) {
            __builtin_amdgcn_global_load_lds((const DINO_GLOBAL_AS void*)(base + src[h][0]), (DINO_LDS_AS void*)dst, 16, 0, 2);
            if (!half8) __builtin_amdgcn_global_load_lds((const DINO_GLOBAL_AS void*)(base + src[h][1]), (DINO_LDS_AS void*)(dst + 1024), 16, 0, 2);
            return;
        }
#endif
        glds16(base + src[h][0], dst);
        if (!half8) glds16(base + src[h][1], dst + 1024);
    };

    // ---- fragment addresses (16x16x32 MFMA: lane -> row lane & 15, k quarter lane >> 4; 16-byte chunk (4 ks + kq) ^ swizzle)
    const int fr = lane & 15, kq = lane >> 4, sw = (fr >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(DINO_LDS_AS char*)smem;
    unsigned xa[2][2], wa[2];  // [token half][k-step] / [k-step]: this lane's fragment row in buffer 0 (16-row blocks at + 2048 each)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned ch = (unsigned)(((ks * 4 + kq) ^ sw) << 4);
        xa[0][ks] = lds0 + (unsigned)((wx * 64 + fr) * 128) + ch;
        xa[1][ks] = lds0 + 16384u + (unsigned)((wx * RX1 + fr) * 128) + ch;
        wa[ks] = lds0 + WOFF + (unsigned)((ww * 32 + fr) * 128) + ch;  // column half 1 at + 16384
    }

    const int nk = K / BK;  // even (checked by the launcher): the last K-tile sits in buffer 1, where the epilogue's slices go, and
                            // buffer 0 is free for the next output tile's first K-tile
    if (bidx < chunkn) {
        int pm0, pn0;
        tile_mn(chunk0 + bidx, pm0, pn0);
        set_tile(pm0, pn0);
        stage(0, 0, 0);
        stage(2, 0, 0);
        stage(1, 0, 0);
        stage(3, 0, 0);
        if (ONE && nk > 1) {  // (the launcher gives every workgroup exactly one tile)
            stage(0, 1, 1);
            stage(2, 1, 1);
            stage(3, 1, 1);
        }
    }
    DINO_GP_INIT
    for (int tix = bidx; tix < chunkn; tix += nb_x) {
        int m0, n0;
        tile_mn(chunk0 + tix, m0, n0);
        DINO_GP(2)  // (time since the end of the previous tile's stamps: none)
        const bool has_next = tix + nb_x < chunkn;

        // acc[a][b][i][j][e] = C[m0 + wx * 32 XREP + 64 a + 16 i + (lane & 15)][n0 + ww * 64 + 32 b + 16 j + 4 (lane >> 4) + e]
        // (OPERAND SWAP: the weight fragment is the MFMA A operand, so a lane owns one token and four CONSECUTIVE output columns:
        // vector bias / LayerScale loads and 8 / 16-byte LDS writes in the epilogue)
        f32x4 acc[2][2][4][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xf[4][2], wf[2][2];  // fragments of ONE quadrant (64 tokens x 32 columns x the whole K-tile): 48 registers

        // ---- main loop: the local guide's 8-phase schedule with its own MFMA shape --------------------------------------------
        // One phase = one quadrant (token half a, column half b) over the whole K-tile: its fragment reads and one half-tile of
        // staging in the MEM section, its 16 MFMAs 16x16x32 in the MMA section, a workgroup barrier after each section.  The two
        // wave-rows (= the two waves of every SIMD) run half a phase apart (wave-row 1 takes one extra barrier up front), so one
        // wave per SIMD is in its MMA section while its partner reads and stages.  Per K-tile t:
        //   phase 0  reads X0(t) 8 + W0(t) 4 -> quadrant (0,0) -> stages X1(t+1)
        //   phase 1  reads W1(t) 4           -> quadrant (0,1) -> stages W0(t+1)
        //   phase 2  reads X1(t) 8           -> quadrant (1,1) -> stages X0(t+2)
        //   phase 3  reads W0(t) 4           -> quadrant (1,0) -> stages W1(t+2), then s_waitcnt vmcnt(4): K-tile t+1 has landed
        // A half-tile slot is re-staged two phases after its last read, and a K-tile is read one phase after the counted wait that
        // retires it.  Past the end of this tile's K the staging continues with the NEXT output tile's K-tile 0 (buffer 0), so its
        // latency hides under the last phases and the epilogue; that tile's X0 / W1 of K-tile 1 follow after the epilogue, whose
        // LDS slices live in buffer 1.
        // Measured (tools/probes/gemm8p.hip, gemm8p16.hip; profiles/r02_gemm_kloop.md): this schedule with 32x32x16 MFMAs runs at the
        // level of the previous one-barrier-per-K-tile loop (1 066 / 1 137 TFLOP/s at 4 096^3 / 8 192^3); with 16x16x32 MFMAs
        // 1 242 / 1 333.
#define DINO_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DINO_READ_X(A_, BO)                                                   \
    {                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                    \
            const unsigned a__ = xa[A_][ks] + (BO);                           \
            DINO_DSR(xf[0][ks], a__, 0);                                      \
            DINO_DSR(xf[1][ks], a__, 2048);                                   \
            if ((A_) == 0 || NI1 == 4) {                                      \
                DINO_DSR(xf[2][ks], a__, 4096);                               \
                DINO_DSR(xf[3][ks], a__, 6144);                               \
            }                                                                 \
        }                                                                     \
    }
#define DINO_READ_W(B_, BO)                                                   \
    {                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                    \
            const unsigned a__ = wa[ks] + (BO) + (B_) * 16384u;               \
            DINO_DSR(wf[0][ks], a__, 0);                                      \
            DINO_DSR(wf[1][ks], a__, 2048);                                   \
        }                                                                     \
    }
#define DINO_BAR()                              \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    }
// (no s_setprio around the MFMAs: measured -1 % on the K = 1024 GEMMs with it, neutral on the others)
#define DINO_MMA(A_, B_)                                                                                                     \
    {                                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
        _Pragma("unroll") for (int i = 0; i < ((A_) == 0 ? 4 : NI1); ++i)                                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                        \
            acc[A_][B_][i][j] = E::mfma16(__builtin_bit_cast(vec8, wf[j][ks]), __builtin_bit_cast(vec8, xf[i][ks]), acc[A_][B_][i][j]); \
        /* pin the MFMAs INSIDE this section: pure register ops otherwise sink below the barrier that ends it */             \
        _Pragma("unroll") for (int i = 0; i < ((A_) == 0 ? 4 : NI1); ++i)                                                    \
            asm volatile("" : "+v"(acc[A_][B_][i][0]), "+v"(acc[A_][B_][i][1]));                                             \
    }
#define DINO_PHASE_END(A_, B_)                                  \
    DINO_BAR()                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);                          \
    DINO_MMA(A_, B_)                                            \
    DINO_BAR()

        // every wave has left the previous tile's epilogue slices (buffer 1); K-tile 0 of this tile is in flight or in buffer 0
        DINO_BAR()
        if constexpr (ONE) {
            if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but K-tile 1's six pieces: K-tile 0 has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            stage(0, 1, 1);
            stage(3, 1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but those four: K-tile 0 has landed (and the previous tile's stores)
        }
        DINO_BAR()
        if (wx == 1) DINO_BAR()  // wave-row 1 runs one barrier behind wave-row 0
        DINO_GP(0)

        if constexpr (ONE) {
            // two phases per K-tile; K-tile t + 2 goes into the buffer K-tile t - 1 has left (last read one phase pair ago by either wave-row)
            unsigned bo = 0;
            int b2 = 2;
            for (int t = 0; t < nk; ++t) {
                const bool s2 = t + 2 < nk;
                DINO_READ_X(0, bo)
                DINO_READ_W(0, bo)
                if (s2) {
                    stage(0, t + 2, b2);
                    stage(2, t + 2, b2);
                }
                DINO_PHASE_END(0, 0)
                DINO_READ_W(1, bo)
                if (s2) {
                    stage(3, t + 2, b2);
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but K-tile t + 2: K-tile t + 1 is in LDS
                } else if (t + 1 < nk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                DINO_PHASE_END(0, 1)
                bo = bo == 2u * BUF ? 0u : bo + BUF;
                b2 = b2 == 2 ? 0 : b2 + 1;
            }
        } else
        for (int t = 0; t < nk; ++t) {
            const unsigned bo = (unsigned)(t & 1) * (unsigned)BUF;
            const int b1 = (t + 1) & 1, b2 = t & 1;
            const bool s01 = t + 1 < nk || has_next;                 // phases 0 / 1 stage K-tile t + 1, or the next tile's K-tile 0
            const int k01 = t + 1 < nk ? t + 1 : 0;
            const bool s23 = t + 2 < nk || (t + 2 == nk && has_next);  // phases 2 / 3: K-tile t + 2, or the next tile's K-tile 0
            const int k23 = t + 2 < nk ? t + 2 : 0;
            // phase 0
            DINO_READ_X(0, bo)
            DINO_READ_W(0, bo)
            if (s01) stage(1, k01, b1);
            DINO_PHASE_END(0, 0)
            // phase 1
            DINO_READ_W(1, bo)
            if (s01) stage(2, k01, b1);
            DINO_PHASE_END(0, 1)
            if (t + 2 == nk && has_next) {  // everything staged from here on belongs to the next output tile
                int nm0, nn0;
                tile_mn(chunk0 + tix + nb_x, nm0, nn0);
                set_tile(nm0, nn0);
            }
            // phase 2
            DINO_READ_X(1, bo)
            if (s23) stage(0, k23, b2);
            DINO_PHASE_END(1, 1)
            // phase 3
            DINO_READ_W(0, bo)
            if (s23) {
                stage(3, k23, b2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but X0, W1 of K-tile t + 2: K-tile t + 1 is in LDS
            } else if (t + 1 < nk) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            DINO_PHASE_END(1, 0)
        }
        if (wx == 0) DINO_BAR()
        DINO_GP(1)
#undef DINO_DSR
#undef DINO_READ_X
#undef DINO_READ_W
#undef DINO_BAR
#undef DINO_MMA
#undef DINO_PHASE_END

        // ---- epilogue ----------------------------------------------------------------------------------------------
        // Each wave transposes its result through a private 8 KiB slice of buffer 1 (nobody reads buffer 1 after the last
        // barrier above) and moves whole 128-byte lines.  LDS slice image: 64 rows x 128 B, 16-byte slot s of row r stored at
        // slot s ^ (r & 7) (conflict-free reads).
        // `el` launders the lane id: without it LICM hoists ~40 loop-invariant epilogue addresses out of the persistent
        // tile loop, they stay live across the K loop and the kernel spills (fatal next to the asm-loaded fragments).
        int el = lane;
        asm volatile("" : "+v"(el));
        const int er = el & 15, eq = el >> 4;
        char* const ep = smem + 65536 + wid * 8192;  // (buffer 1 of the two-buffer layouts; the 128-row ring is idle by now)
        const int mbase = m0 + wx * (32 * XREP);
        const int ncol = n0 + ww * 64 + 4 * eq;  // + 32 b + 16 j: this lane's four consecutive columns of block (b, j)

        float4 bs[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bs[b][j] = p.bias ? *(const float4*)(p.bias + ncol + b * 32 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);

        if constexpr (EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_SWIGLU) {
            // 2-byte outputs: two passes (token halves) of 64 rows x 64 columns (SwiGLU: x 32)
            // per WAVE (64 columns), not per tile: q|k|v boundaries are multiples of the head size 64, but with hidden = 384 they
            // fall inside a 256-column tile
            const float qs = (EPI == EPI_QKV && n0 + ww * 64 < p.qcols) ? p.qscale : 1.0f;
            constexpr int BN_ = EPI == EPI_SWIGLU ? 1 : 2;
            // SUBP sub-passes per token half.  1: 64 rows through the wave's whole 8 KiB slice.  2 (the epilogues with real VALU work
            // per element): 32 rows through alternating 4 KiB halves of the slice, so that the stores of one sub-pass and the
            // arithmetic of the next are independent and the two waves of a SIMD -- which share its VALU -- fall out of step.
#ifdef DINO_EPI_SPLIT
            constexpr int SUBP = EPI == EPI_SWIGLU ? 1 : 2;
#else
            constexpr int SUBP = 1;
#endif
            constexpr int IPS = 4 / SUBP;  // 16-row blocks per sub-pass
#pragma unroll
            for (int sp = 0; sp < 2 * SUBP; ++sp) {
                const int q = sp / SUBP, ih = sp % SUBP;
                if (q == 1 && ih * IPS >= NI1) continue;  // 192-row tiles: the second token half has 32 rows
                // GELU: the two waves of a SIMD share its VALU, the older one wins every arbitration and the younger one then runs its last
                // third alone at half the rate.  The younger one (wave-row 1) takes priority for its first token half and gives it back for
                // the second: FFN-in -1 % in the model (profiles/r03_gemm_notes.md section 6).  Scheduling only, no effect on the bits.
                if (EPI == EPI_GELU && wx == 1) {
                    if (q == 0) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                char* const eh = ep + (SUBP == 2 ? (sp & 1) * 4096 : 0);
#pragma unroll
                for (int b = 0; b < BN_; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float bb[4] = {bs[b][j].x, bs[b][j].y, bs[b][j].z, bs[b][j].w};
                        const float b2[4] = {bs[1][j].x, bs[1][j].y, bs[1][j].z, bs[1][j].w};
#pragma unroll
                        for (int i = ih * IPS; i < (ih + 1) * IPS; ++i) {
                            if (q == 1 && i >= NI1) continue;  // 192-row tiles: the second pass has 32 rows
                            vec4 o;
#ifndef DINO_GELU_SCALAR
                            if constexpr (EPI == EPI_GELU) {
                                // Two columns per instruction: the bias add, x^2, the cubic, 1 + 2^t and the final product
                                // run as v_pk_*_f32 (IEEE results identical to the scalar ops of gemm.hip, so both kernels
                                // still agree bit for bit); v_exp / v_rcp / the f16 conversions stay per element.
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                                for (int e2 = 0; e2 < 2; ++e2) {
                                    f32x2 v = {acc[q][b][i][j][2 * e2], acc[q][b][i][j][2 * e2 + 1]};
                                    v += f32x2{bb[2 * e2], bb[2 * e2 + 1]};
                                    asm("" : "+v"(v));  // f32 sums first (no v_fma_mix fusion), then the f16 rounding
                                    // (vector converts: one v_cvt_pk_f16_f32 + v_cvt_f32_f16 / its SDWA form instead of four scalar converts)
                                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                                    const f32x2 xr = __builtin_convertvector(__builtin_convertvector(v, f16x2), f32x2);
                                    const f32x2 c1 = {-0.1029432397f, -0.1029432397f}, c2 = {-2.302208199f, -2.302208199f};
                                    const f32x2 t = xr * __builtin_elementwise_fma(xr * xr, c1, c2);  // -2 log2(e) u
                                    const f32x2 den = f32x2{1.0f, 1.0f} + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                                    f32x2 gl = xr * f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                                    asm("" : "+v"(gl));
                                    o[2 * e2] = E::from_f32((float)(_Float16)gl[0]);
                                    o[2 * e2 + 1] = E::from_f32((float)(_Float16)gl[1]);
                                }
                            } else
#endif
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[q][b][i][j][e] + bb[e];
                                // keep the f32 sum a real register value: hipcc otherwise fuses "add, then round to f16"
                                // into v_fma_mixlo_f16 for SOME unrolled instances (single rounding instead of the
                                // reference's f32-then-f16 double rounding), which made results depend on the row's
                                // position in the tile
                                asm("" : "+v"(v));
                                if constexpr (EPI == EPI_QKV) {
                                    float vq = v * qs;
                                    asm("" : "+v"(vq));
                                    o[e] = E::from_f32(vq);
                                } else if constexpr (EPI == EPI_SWIGLU) {
                                    // W rows interleaved in 32-blocks: column half 0 holds x1[32 units], half 1 holds x2 of the same units
                                    const float h2 = acc[q][1][i][j][e] + b2[e];
                                    float sg = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)) * h2;  // silu(x1) * x2
                                    asm("" : "+v"(sg));
                                    o[e] = E::from_f32(sg);
                                } else {
                                    // EPI_GELU, ggml semantics: y = table[f16(x)], table[h] = f16(gelu_tanh(f32(h))).
                                    // 0.5 x (1 + tanh u) == x / (1 + exp(-2u)); the reference's x <= -10 -> 0 and
                                    // x >= 10 -> x branches fall out of the formula after the f16 roundings (exp -> inf
                                    // gives -0, exp -> 0 gives x), so no compares are needed.
                                    const float xr = (float)(_Float16)v;
                                    const float t = xr * __builtin_fmaf(xr * xr, -0.1029432397f, -2.302208199f);  // -2 log2(e) u
                                    float gl = xr * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
                                    asm("" : "+v"(gl));
                                    o[e] = E::from_f32((float)(_Float16)gl);
                                }
                            }
                            const int row = (i - ih * IPS) * 16 + er;  // row within the sub-pass
                            const int slot = (4 * b + 2 * j + (eq >> 1)) ^ (row & 7);  // 8 columns (16 B) per slot
                            *(vec4*)(eh + row * 128 + slot * 16 + (eq & 1) * 8) = o;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                if constexpr (EPI == EPI_SWIGLU) {
                    const int hid0 = ((n0 + ww * 64) >> 6) * 32;  // 32 hidden units = 64 B per row: 4 lanes per row
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 16 + (el >> 2), slot = el & 3;
                        const u32x4 v = *(const u32x4*)(eh + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int m = mbase + q * 64 + row;
                        if (m < M && (XREP == 4 || q * 64 + row < 32 * XREP))
                            DINO2_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + hid0 + slot * 8), v);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 8 / SUBP; ++it) {
                        const int row = it * 8 + (el >> 3), slot = el & 7;
                        const u32x4 v = *(const u32x4*)(eh + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int tr = q * 64 + ih * (64 / SUBP) + row;  // token row within the wave's 128
                        const int m = mbase + tr;
                        if (m < M && (XREP == 4 || tr < 32 * XREP))
                            DINO2_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + n0 + ww * 64 + slot * 8), v);
                    }
                }
                // (two halves: the next sub-pass writes the OTHER half, and a wave's LDS operations execute in order, so only the
                //  write -> read fence above is needed)
                if (SUBP == 1) __builtin_amdgcn_wave_barrier();
            }
        } else {
            // 4-byte outputs: four passes of 64 rows x 32 columns (128 B per row).  All loads of a pass (residual stream /
            // pos-embed rows) are issued before its LDS transposition and long before its first store.
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int b = ps >> 1, q = ps & 1;
                if (q == 1 && NI1 == 0) continue;  // 128-row tiles
                const int nb = n0 + ww * 64 + b * 32 + (el & 7) * 4;
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
                for (int j = 0; j < 2; ++j) {
                    float4 ls = make_float4(1.f, 1.f, 1.f, 1.f);
                    if constexpr (EPI == EPI_RESID) ls = *(const float4*)(p.aux + ncol + b * 32 + j * 16);
                    const float4 b4 = bs[b][j];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (q == 1 && i >= NI1) continue;
                        const int row = i * 16 + er;
                        const int slot = (4 * j + eq) ^ (row & 7);  // 4 columns (16 B) per slot
                        *(float4*)(ep + row * 128 + slot * 16) =
                            make_float4((acc[q][b][i][j][0] + b4.x) * ls.x, (acc[q][b][i][j][1] + b4.y) * ls.y,
                                        (acc[q][b][i][j][2] + b4.z) * ls.z, (acc[q][b][i][j][3] + b4.w) * ls.w);
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
                            const int bb_ = m / p.P, pp = m - bb_ * p.P;
                            o = ((size_t)bb_ * p.T + 1 + p.R + pp) * p.ldo + nb;
                        } else {
                            o = (size_t)m * p.ldo + nb;
                        }
                        *(float4*)((float*)p.out + o) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        DINO_GP(2)
#ifdef DINO_GEMM_PROF
        gp_acc[3] += 1;
#endif
    }  // persistent tile loop
    DINO_GP_FLUSH
}

// Clock probe slots of this file's kernels (device_types.h, "clock probe")
__device__ unsigned long long g_clk2[CLK_SLOTS * 4];
hipError_t gemm_clock_probe_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk2), sizeof(unsigned long long) * CLK_SLOTS * 4);
}

template <typename T, int EPI, int XREP>
__global__ __launch_bounds__(512) void gemm2_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DINO_CLK_BEGIN()
    gemm2_body<T, EPI, XREP>(p, smem);
    DINO_CLK_END(g_clk2, p.clk_slot)
}

// One launch, two tile heights: every block first walks its share of the 256-row tiles of `p` (whole rounds), then its share
// of the 192-row tiles of `q` (the remaining rows).  No grid-wide barrier in between -- a block that is done with its
// 256-row tiles starts on the 192-row ones at once -- which is what two back-to-back launches lacked (they were slower
// than the plain kernel for K = 1024).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm2_mixed_kernel(GemmArgs p, GemmArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DINO_CLK_BEGIN()
    gemm2_body<T, EPI, 4>(p, smem);
    gemm2_body<T, EPI, 3>(q, smem);
    DINO_CLK_END(g_clk2, p.clk_slot)
}

#ifdef DINO_GEMM_PROF
static void gemm_prof_dump(const char* what, int nblocks) {
    static unsigned long long h[256 * 8], z[256 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_prof), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof z);
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
        for (int i = 0; i < 8; ++i) a[i] += (double)h[b * 8 + i];
    const double t0 = a[3] > 0 ? a[3] : 1, t4 = a[7] > 0 ? a[7] : 1;
    fprintf(stderr, "gemm_prof %s: cycles per tile, wave 0: prologue %.0f  K loop %.0f  epilogue %.0f (%.1f tiles/block) | wave 4: %.0f %.0f %.0f\n", what,
            a[0] / t0, a[1] / t0, a[2] / t0, a[3] / nblocks, a[4] / t4, a[5] / t4, a[6] / t4);
}
#endif

template <typename T, int XREP>
static hipError_t launch2_t(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int tiles = (a.N / 256) * ((a.M + 64 * XREP - 1) / (64 * XREP));
    const dim3 grid(XREP == 2 ? tiles : tiles < 256 ? tiles : 256), block(512);  // (128-row tiles: one tile per workgroup)
    const size_t lds = XREP == 2 ? 3 * 49152 : 2 * 512 * 128;
#define DINO_L2(E)                                                         \
    case E:                                                                \
        hipLaunchKernelGGL((gemm2_kernel<T, E, XREP>), grid, block, lds, st, a); \
        break;
    switch (epi) {
        case EPI_PATCH:  // the 256-row instantiation spills (the pos-embed prefetch on top of 128 accumulators); 192-row does not
            if (XREP != 3) return hipErrorInvalidValue;
            hipLaunchKernelGGL((gemm2_kernel<T, EPI_PATCH, 3>), grid, block, lds, st, a);
            break;
        DINO_L2(EPI_QKV)
        DINO_L2(EPI_RESID)
        DINO_L2(EPI_GELU)
        DINO_L2(EPI_SWIGLU)
        DINO_L2(EPI_PLAIN_F32)
        default: return hipErrorInvalidValue;  // (the LN-fold epilogues live in gemm4.hip and the small-tile kernel)
    }
#undef DINO_L2
#ifdef DINO_GEMM_PROF
    gemm_prof_dump(epi == EPI_QKV ? "plain-launch QKV" : "plain-launch", (int)grid.x);
#endif
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
#ifdef DINO_GEMM_PROF
    gemm_prof_dump(epi == EPI_GELU ? "mixed GELU" : epi == EPI_RESID ? "mixed RESID" : epi == EPI_QKV ? "mixed QKV" : "mixed", 256);
#endif
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
// 128-row tiles, one per workgroup (see gemm2_body)
hipError_t launch_gemm2_128(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch2_t<_Float16, 2>(epi, a, st) : launch2_t<__bf16, 2>(epi, a, st);
}

template <typename T, int XREP>
static hipError_t attr2_t() {
    hipError_t e = hipSuccess;
    const int lds = XREP == 2 ? 3 * 49152 : 2 * 512 * 128;
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
    if (e == hipSuccess) e = attr2_t<_Float16, 2>();
    if (e == hipSuccess) e = attr2_t<__bf16, 2>();
    return e;
}

}  // namespace dinov2
