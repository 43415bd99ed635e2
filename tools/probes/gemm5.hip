// gemm5.hip (PARKED: not part of the product library; opt-in build `make -C dinov2.cpp_amd g5`; why: profiles/r05_gemm5.md) -- fifth generation of the f16/bf16 MFMA GEMM: a 192 x 128 x 64 tile on four waves (one per SIMD, 96 accumulators in AGPRs +
// <= 160 VGPRs each, 80 KiB of LDS), so that TWO INDEPENDENT workgroups are resident per CU.  Same contract, epilogue expressions, MFMA
// instruction and K order as gemm.hip / gemm2.hip / gemm4.hip: a row's bits do not depend on which generation computes it.
//
// Why (profiles/r04_gemm4w.md section 5b): a K = 1 024 tile of the 256 x 256 kernels spends 25 - 30 % of its time OUTSIDE
// its K loop (GELU arithmetic, the store burst, the f32 read-modify-write of the residual stream), and nothing can run beside it: gemm4.hip
// holds all 512 registers of a SIMD lane in one wave, gemm2.hip's two waves per SIMD belong to one workgroup and reach their epilogues
// together.  Here the second wave of every SIMD belongs to ANOTHER workgroup with its own tile, barriers and phase: while one workgroup
// converts, transposes and stores its tile the other one owns the matrix pipe.  Nothing synchronises the two; the hardware's
// oldest-first arbitration lets the older workgroup run its K loop at nearly full rate, so the pair drifts out of phase by itself.
// The price is the smaller tile: 1.67 x the global -> LDS bytes and 1.67 x the fragment reads per MFMA of the 256 x 256 tile.
//
//   LDS (80 KiB): K-tile buffer b at 40 960 b: X rows (192 x 128 B) at + 0, W rows (128 x 128 B) at + 24 576; 16-byte chunks XOR-swizzled by
//   (row >> 1) & 7 on the SOURCE side of the LDS-DMA and on the fragment reads (gemm2.hip's layout).  The epilogue's four 8 KiB transposition
//   slices live inside buffer 1, which is idle from barrier A of the last K-tile on.
//   Wave (wr, wc) = (wid >> 1, wid & 1): tokens [96 wr, + 96) x columns [64 wc, + 64), 6 x 4 blocks of 16 x 16, acc[i][j] in AGPRs; OPERAND
//   SWAP as in gemm2.hip (weight fragment = MFMA A operand): a lane owns one token and four consecutive output columns.
//   Fragment registers: P = k-step 0 of a K-tile (6 X + 4 W fragments of 4 VGPRs), Q = k-step 1.  MFMA index m = 24 ks + 6 j + i (column
//   block j outer, token block i inner), 48 per K-tile and wave.
//   K-tile t in buffer b = t & 1:
//     m = 0 .. 9       one ds_read_b128 of Q(t) per MFMA (W fragments first)
//     m = PA           s_waitcnt lgkmcnt(0); s_barrier      [A] every wave has read all of buffer b
//     m = S0 + SP k    the ten LDS-DMA pieces (global_load_lds_dwordx4, 8 rows x 128 B) of K-tile t + 2 -> buffer b
//     m = PB           s_waitcnt vmcnt(pieces issued since A); s_barrier   [B] K-tile t + 1 is in buffer b ^ 1 for everyone
//     m = PB + 1 ..    one ds_read_b128 of P(t + 1) per MFMA; s_waitcnt lgkmcnt(0) at the end
//   Tiles are handed out DYNAMICALLY (one ticket counter per XCD, p.sched): the hardware's oldest-first arbitration gives the older
//   workgroup of a CU the matrix pipe whenever both want it, so with a static share of the tiles the younger one needs ~ 40 % longer per tile
//   and decides the kernel time (profiles/r05_gemm5.md); with tickets the faster workgroup simply takes more tiles.  Wave 0 pulls tickets two
//   tiles ahead (the atomic's latency hides under a whole K loop) and passes them to the other waves through eight idle bytes of buffer 1
//   during the epilogue; tickets are taken in tile order, so the workgroups of an XCD keep working on neighbouring tiles (shared L2).
//   Tile hand-over: K-tile 0 of the NEXT output tile is staged into buffer 0 behind barrier A of K-tile nk - 2; the last K-tile stages
//   nothing and has no barrier B; every wave waits for its own pieces (vmcnt(0)) before its epilogue, so ONE barrier behind the epilogue
//   hands the slices back and publishes K-tile 0; K-tile 1 is staged right behind it.  What a lone workgroup would lose there (the
//   barrier, the first fragment reads, a short first staging distance) the co-resident workgroup fills with its own MFMAs.
#include <cstdio>
#include <type_traits>
#include <utility>

#include "device_types.h"
#include "kernels.h"

namespace dinov2 {

template <int... Is, class F>
static __device__ __forceinline__ void static_for5_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
static __device__ __forceinline__ void static_for5(F&& f) {
    static_for5_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// The K loop's instructions as asm statements (volatile statements keep their program order; hipcc only allocates the registers).
#define DINO5_MFMA_F16(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define DINO5_MFMA_F16_Z(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(ACC) : "v"(W), "v"(X))
#define DINO5_MFMA_BF16(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define DINO5_MFMA_BF16_Z(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(W), "v"(X))
#define DINO5_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DINO5_GLDS(VOFF, SBASE, LDSBASE, IMM)                                                                                   \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSBASE), "n"(IMM) \
                 : "memory", "scc")

#define DINO5_ST16(PTR, V)                                   \
    {                                                        \
        if (nt_out) __builtin_nontemporal_store((V), (PTR)); \
        else *(PTR) = (V);                                   \
    }

// Schedule constants (MFMA index within a K-tile, 0 .. 47); tuning builds override them with -D.
#ifndef DINO_GEMM5_PA
#define DINO_GEMM5_PA 11
#endif
#ifndef DINO_GEMM5_S0
#define DINO_GEMM5_S0 12
#endif
#ifndef DINO_GEMM5_SP
#define DINO_GEMM5_SP 2
#endif
#ifndef DINO_GEMM5_PB
#define DINO_GEMM5_PB 33
#endif
#ifndef DINO_GEMM5_GM
#define DINO_GEMM5_GM 8
#endif
#ifndef DINO_GEMM5_PF
#define DINO_GEMM5_PF 2
#endif
#ifndef DINO_GEMM5_GRID
#define DINO_GEMM5_GRID 512  // resident workgroups (tuning builds: 256 = one per CU, the same kernel without its twin)
#endif
#ifndef DINO_GEMM5_PRIO
#define DINO_GEMM5_PRIO 0  // 1: s_setprio 1 around the K loop (the epilogue of the co-resident workgroup then never delays an MFMA)
#endif

// -DDINO_GEMM5_PROF (tuning builds): s_memtime sums per workgroup of wave 0 -- [0] K loops, [1] epilogues, [2] hand-overs (barrier + first
// fragments), [3] tiles, [4] 100 MHz ticks of all of it, [5] / [6] / [7] shader cycles, 100 MHz ticks and count of the MIDDLE K-tiles (2 ..
// nk - 3: steady state) -- printed by the launcher after each launch.
#ifdef DINO_GEMM5_PROF
__device__ unsigned long long g_gemm5_prof[512 * 8];
#define DINO5_GP_DECL unsigned long long gp_t = 0, gp_r = 0, gp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DINO5_GP_START { gp_t = __builtin_readcyclecounter(); gp_r = __builtin_amdgcn_s_memrealtime(); }
#define DINO5_GP(i) { const unsigned long long t__ = __builtin_readcyclecounter(); gp_acc[i] += t__ - gp_t; gp_t = t__; }
#define DINO5_GP_TILE { gp_acc[3] += 1; const unsigned long long r__ = __builtin_amdgcn_s_memrealtime(); gp_acc[4] += r__ - gp_r; gp_r = r__; }
#define DINO5_GP_FLUSH if (threadIdx.x == 0) for (int i__ = 0; i__ < 8; ++i__) g_gemm5_prof[blockIdx.x * 8 + i__] += gp_acc[i__];
#else
#define DINO5_GP_DECL
#define DINO5_GP_START
#define DINO5_GP(i)
#define DINO5_GP_TILE
#define DINO5_GP_FLUSH
#endif

constexpr int G5_NI = 6, G5_NJ = 4;                  // 16 x 16 blocks per wave: token blocks x column blocks
constexpr int G5_BM = 32 * G5_NI, G5_BN = 32 * G5_NJ;  // 192 x 128
constexpr int G5_XBYTES = G5_BM * 128, G5_WBYTES = G5_BN * 128, G5_BUF = G5_XBYTES + G5_WBYTES;  // 24 576 + 16 384 = 40 960
constexpr size_t G5_LDS = 2 * G5_BUF;                // 81 920: two workgroups per CU

template <typename T, int EPI>
static __device__ __forceinline__ void gemm5_body(const GemmArgs& p, char* smem) {
    // (no implicit mul+add -> fma contraction: an element's bits must not depend on where its row sits in a tile -- see gemm2.hip)
#pragma clang fp contract(off)
    using E = Elem<T>;
    using vec4 = typename E::vec4;
    constexpr bool F16 = std::is_same<T, _Float16>::value;
    constexpr int NI = G5_NI, NJ = G5_NJ, BM = G5_BM, BN = G5_BN;
    constexpr int NM = 2 * NI * NJ;    // MFMAs per K-tile and wave
    constexpr int NP = NI + NJ;        // LDS-DMA pieces per wave and K-tile: NI of X, NJ of W
    constexpr int PA = DINO_GEMM5_PA, S0 = DINO_GEMM5_S0, SP = DINO_GEMM5_SP, PB = DINO_GEMM5_PB;
    constexpr int NBS = PB >= S0 ? (PB - S0) / SP + 1 : 0;  // slots up to barrier B
    constexpr int NB = NBS < NP ? NBS : NP;                 // pieces issued between the barriers: what barrier B's vmcnt leaves in flight
    static_assert(NP - 1 < PA && PA < S0 && S0 + SP * (NP - 1) < NM && PB + NP < NM && PB > PA, "K-tile schedule");

    DINO5_GP_DECL
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int M = p.M, N = p.N, K = p.K;
    const bool nt_out = p.nt_out != 0;
    const unsigned lda2 = (unsigned)(p.lda ? p.lda : K) * 2u, ldw2 = (unsigned)(p.ldw ? p.ldw : K) * 2u;
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int ntiles = ntn * ntm;
    const int nk = K / 64;  // even, >= 4 (checked by the launcher)

    // XCD-aware tile order (gemm2.hip): block b sits on XCD b % 8 (observed placement; used for L2 locality only), every XCD owns a contiguous
    // chunk of the tile order (patches of GM row panels, column tiles outer) and hands its tiles out through its own ticket counter
    const int xcd = blockIdx.x & 7;
    const int nb_x = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);  // workgroups that share this counter
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int chunk0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int chunkn = tq + (xcd < tr ? 1 : 0);
    unsigned* const head = p.sched + 32 * xcd;  // [0] next ticket, [1] workgroups that have left; one 128-byte line per XCD
    constexpr int GM = DINO_GEMM5_GM;
    auto tile_mn = [&](int lid, int& m0, int& n0) {
        const int g = lid / (GM * ntn), r = lid - g * (GM * ntn);
        const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
        const int n = r / gm, mi = r - n * gm;
        m0 = (g * GM + mi) * BM;
        n0 = n * BN;
    };

    // ---- staging.  Piece pc of a wave (0 .. 9): X and W alternate while W has pieces left (pc < 2 NJ: operand pc & 1, row block pc >> 1),
    // then X only (row block pc - NJ).  Wave w moves X rows [48 w, + 48) and W rows [32 w, + 32) of the tile; a piece is 8 rows x 128 B
    // (lane -> row lane >> 3, 16-byte chunk lane & 7, XOR-swizzled on the source side); X rows are clamped to M.
    unsigned so[NP];  // byte offsets from p.A / p.W of this lane's 16 bytes of every piece (K-tile 0; + 128 kt through the scalar base)
    auto piece_off = [&](int pc, int m0, int n0) -> unsigned {
        const bool isw = pc < 2 * NJ && (pc & 1);
        const int rb = pc < 2 * NJ ? pc >> 1 : pc - NJ;
        const int r = (isw ? 8 * NJ : 8 * NI) * wid + 8 * rb + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        if (isw) return (unsigned)(n0 + r) * ldw2 + ch * 16;
        int gm = m0 + r;
        gm = gm < M ? gm : M - 1;
        return (unsigned)gm * lda2 + ch * 16;
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(DINO_LDS_AS char*)smem;
    const unsigned ldsx = lds0 + (unsigned)wid * (NI * 1024u), ldsw = lds0 + (unsigned)G5_XBYTES + (unsigned)wid * (NJ * 1024u);

    // ---- fragment addresses: lane -> row lane & 15 of a 16-row block, 16-byte chunk (4 ks + (lane >> 4)) ^ ((row >> 1) & 7)
    const int fr = lane & 15, kq = lane >> 4, sw = (fr >> 1) & 7;
    unsigned xa[2], wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned ch = (unsigned)(((ks * 4 + kq) ^ sw) << 4);
        xa[ks] = lds0 + (unsigned)((wr * (16 * NI) + fr) * 128) + ch;
        wa[ks] = lds0 + (unsigned)G5_XBYTES + (unsigned)((wc * (16 * NJ) + fr) * 128) + ch;
    }

    // acc[i][j][e] = C[m0 + 96 wr + 16 i + (lane & 15)][n0 + 64 wc + 16 j + 4 (lane >> 4) + e]
    f32x4 acc[NI][NJ];
    u32x4 Px[NI], Pw[NJ], Qx[NI], Qw[NJ];

    const char* const Ab = (const char*)p.A;
    const char* const Wb = (const char*)p.W;

#define DINO5_PIECE(PC, KT, BUF)                                                                                         \
    {                                                                                                                    \
        constexpr int pc__ = (PC);                                                                                       \
        constexpr bool isw__ = pc__ < 2 * NJ && (pc__ & 1);                                                              \
        constexpr int rb__ = pc__ < 2 * NJ ? pc__ >> 1 : pc__ - NJ;                                                      \
        if constexpr (isw__) DINO5_GLDS(so[pc__], Wb + (size_t)(KT) * 128, ldsw, (BUF) * G5_BUF + rb__ * 1024);          \
        else DINO5_GLDS(so[pc__], Ab + (size_t)(KT) * 128, ldsx, (BUF) * G5_BUF + rb__ * 1024);                           \
    }

    // One K-tile in buffer B.  FIRST: the accumulators start from zero.  LAST: the last K-tile of an output tile -- no staging, no barrier
    // B, no fragment reads of a successor.  Otherwise the slots behind barrier A carry the pieces of K-tile kt_post (-> THIS buffer; if
    // post_on), `at_a()` runs at A (it switches the piece offsets to the next output tile where the staging crosses over), and P of the
    // next K-tile is read behind barrier B.
    auto ktile = [&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsx, &ldsw, Ab, Wb](auto bc, auto firstc, auto lastc, auto&& at_a, int kt_post,
                                                                                bool post_on) {
        static_for5<NM>([&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsx, &ldsw, Ab, Wb, &at_a, &kt_post, &post_on, bc, firstc, lastc](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int b = decltype(bc)::value;
            constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
            constexpr int ks = m / (NI * NJ), j = (m % (NI * NJ)) / NI, i = m % NI;
            if constexpr (ks == 0) {
                if constexpr (FIRST) {
                    if constexpr (F16) DINO5_MFMA_F16_Z(acc[i][j], Pw[j], Px[i]);
                    else DINO5_MFMA_BF16_Z(acc[i][j], Pw[j], Px[i]);
                } else {
                    if constexpr (F16) DINO5_MFMA_F16(acc[i][j], Pw[j], Px[i]);
                    else DINO5_MFMA_BF16(acc[i][j], Pw[j], Px[i]);
                }
            } else {
                if constexpr (F16) DINO5_MFMA_F16(acc[i][j], Qw[j], Qx[i]);
                else DINO5_MFMA_BF16(acc[i][j], Qw[j], Qx[i]);
            }
            // Q(t): k-step 1 of this K-tile (W fragments first: their registers have been free longest)
            if constexpr (m < NP) {
                if constexpr (m < NJ) DINO5_DSR(Qw[m], wa[1], m * 2048 + b * G5_BUF);
                else DINO5_DSR(Qx[m - NJ], xa[1], (m - NJ) * 2048 + b * G5_BUF);
            }
            if constexpr (m == PA) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                at_a();
            }
            if constexpr (!LAST) {
                if constexpr (m >= S0 && (m - S0) % SP == 0 && (m - S0) / SP < NP) {
                    if (post_on) DINO5_PIECE((m - S0) / SP, kt_post, b)
                }
                if constexpr (m == PB) {
                    if (post_on) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                }
                // P(t + 1): k-step 0 of the next K-tile, from the other buffer
                if constexpr (m > PB && m <= PB + NP) {
                    constexpr int q = m - PB - 1;
                    if constexpr (q < NJ) DINO5_DSR(Pw[q], wa[0], q * 2048 + (b ^ 1) * G5_BUF);
                    else DINO5_DSR(Px[q - NJ], xa[0], (q - NJ) * 2048 + (b ^ 1) * G5_BUF);
                }
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using TT = std::integral_constant<bool, true>;
    using TF = std::integral_constant<bool, false>;
    auto nop = [] {};

    // ---- tickets.  Wave 0 (lane 0) pulls: two before the first tile (this tile, the next one), then one per tile, at the start of tile i
    // for tile i + 2; the value travels to the other waves through slot[1] (bytes of buffer 1 that neither a K-tile in flight nor an
    // epilogue slice uses between barrier A of the last K-tile and the hand-over), written before the epilogue of tile i, read between
    // the two barriers of the hand-over to tile i + 1.  Every workgroup pulls exactly (its tiles + 2) tickets, all of them before it leaves.
    int* const slot = (int*)(smem + G5_BUF + 32768);
    const bool puller = wid == 0 && lane == 0;
    auto pull = [&]() -> int { return (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (puller) {
        slot[0] = pull();
        slot[1] = pull();
    }
    __syncthreads();
    int t_cur = __builtin_amdgcn_readfirstlane(*(volatile int*)&slot[0]);
    int pend = 0;  // (lane 0 of wave 0) the ticket requested at the start of the current tile
    if (t_cur < chunkn) {
        int pm0, pn0;
        tile_mn(chunk0 + t_cur, pm0, pn0);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) so[pc] = piece_off(pc, pm0, pn0);
        // K-tile 0 of the first tile -> buffer 0 (every later tile's arrives under its predecessor's last K-tiles)
        static_for5<NP>([&so, &ldsx, &ldsw, Ab, Wb](auto pcc) { DINO5_PIECE(decltype(pcc)::value, 0, 0) });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    while (t_cur < chunkn) {
        int m0, n0;
        tile_mn(chunk0 + t_cur, m0, n0);
        DINO5_GP_START

        // ---- hand-over: every wave has waited for its own pieces of K-tile 0 and is done with its epilogue slice (buffer 1); the next
        // tile's ticket is read between two barriers (the second one frees buffer 1 for K-tile 1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int t_nxt = __builtin_amdgcn_readfirstlane(*(volatile int*)&slot[1]);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool has_next = t_nxt < chunkn;
        if (puller) pend = pull();
        static_for5<NP>([&so, &ldsx, &ldsw, Ab, Wb](auto pcc) { DINO5_PIECE(decltype(pcc)::value, 1, 1) });
        static_for5<NP>([&Px, &Pw, &xa, &wa](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < NJ) DINO5_DSR(Pw[q], wa[0], q * 2048);
            else DINO5_DSR(Px[q - NJ], xa[0], (q - NJ) * 2048);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DINO5_GP(2)

        // K-tiles 0, 1 (accumulators from zero), the middle, and the last two, under which the staging crosses over to the next output tile
        if (DINO_GEMM5_PRIO) asm volatile("s_setprio 1");
        ktile(T0{}, TT{}, TF{}, nop, 2, true);
        ktile(T1{}, TF{}, TF{}, nop, 3, true);
#ifdef DINO_GEMM5_PROF
        const unsigned long long kc0 = __builtin_readcyclecounter(), kr0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int t = 2; t < nk - 2; t += 2) {
            ktile(T0{}, TF{}, TF{}, nop, t + 2, true);
            ktile(T1{}, TF{}, TF{}, nop, t + 3, true);
        }
#ifdef DINO_GEMM5_PROF
        gp_acc[5] += __builtin_readcyclecounter() - kc0;
        gp_acc[6] += __builtin_amdgcn_s_memrealtime() - kr0;
        gp_acc[7] += (unsigned long long)(nk - 4);
#endif
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_mn(chunk0 + t_nxt, nm0, nn0);
        ktile(T0{}, TF{}, TF{},
              [&] {
                  if (has_next) {
#pragma unroll
                      for (int pc = 0; pc < NP; ++pc) so[pc] = piece_off(pc, nm0, nn0);
                  }
              },
              0, has_next);
        ktile(T1{}, TF{}, TT{}, nop, 0, false);
        if (DINO_GEMM5_PRIO) asm volatile("s_setprio 0");
        // this wave's pieces of the next tile's K-tile 0 have landed; the last MFMAs' results are architecturally visible to the reads below
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
        if (puller) slot[1] = pend;  // the ticket for the tile after next (requested a whole K loop ago)
        DINO5_GP(0)

        // ---- epilogue (gemm2.hip's / gemm4.hip's expressions): each wave transposes its 96 x 64 block through a private 8 KiB LDS slice and
        // moves whole 128-byte lines.  Slice image: 64 rows x 128 B, 16-byte slot s of row r stored at s ^ (r & 7).
        // `el` launders the lane id so that the loop-invariant epilogue addresses are not hoisted out of the persistent tile loop.
        int el = lane;
        asm volatile("" : "+v"(el));
        const int er = el & 15, eq = el >> 4;
        char* const ep = smem + G5_BUF + wid * 8192;
        const int mbase = m0 + wr * (16 * NI);
        const int nw0 = n0 + wc * (16 * NJ);  // first column of the wave's 64
#define DINO5_ACC(Q_, B_, I_, J_) acc[((Q_)*4 + (I_)) < NI ? ((Q_)*4 + (I_)) : 0][(B_)*2 + (J_)]  /* (the clamp only ever acts in dead code) */
        if constexpr (EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_SWIGLU) {
            const int ncol = nw0 + 4 * eq;  // + 32 b + 16 j: this lane's four consecutive columns of block (b, j)
            float4 bs[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bs[b][j] = p.bias ? *(const float4*)(p.bias + ncol + b * 32 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            // 2-byte outputs: two passes (token rows [0, 64) and [64, 96) of the wave's block) of <= 64 rows x 64 columns (SwiGLU: x 32)
            const float qs = (EPI == EPI_QKV && nw0 < p.qcols) ? p.qscale : 1.0f;
            constexpr int BN_ = EPI == EPI_SWIGLU ? 1 : 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int b = 0; b < BN_; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float bb[4] = {bs[b][j].x, bs[b][j].y, bs[b][j].z, bs[b][j].w};
                        const float b2[4] = {bs[1][j].x, bs[1][j].y, bs[1][j].z, bs[1][j].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (q * 4 + i >= NI) continue;
                            vec4 o;
                            if constexpr (EPI == EPI_GELU) {
                                // ggml semantics: y = table[f16(x)], table[h] = f16(gelu_tanh(f32(h))); two columns per instruction
                                // (v_pk_*_f32: IEEE results identical to the scalar ops of gemm.hip, so the kernels agree bit for bit)
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                                for (int e2 = 0; e2 < 2; ++e2) {
                                    f32x2 v = {DINO5_ACC(q, b, i, j)[2 * e2], DINO5_ACC(q, b, i, j)[2 * e2 + 1]};
                                    v += f32x2{bb[2 * e2], bb[2 * e2 + 1]};
                                    asm("" : "+v"(v));  // f32 sums first (no v_fma_mix fusion), then the f16 rounding
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
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float v = DINO5_ACC(q, b, i, j)[e] + bb[e];
                                    asm("" : "+v"(v));  // a real f32 sum: no "add, then round" fusion into v_fma_mixlo_f16
                                    if constexpr (EPI == EPI_QKV) {
                                        float vq = v * qs;
                                        asm("" : "+v"(vq));
                                        o[e] = E::from_f32(vq);
                                    } else {
                                        // EPI_SWIGLU: W rows interleaved in 32-blocks: column half 0 holds x1[32 units], half 1 x2 of the same units
                                        const float h2 = DINO5_ACC(q, 1, i, j)[e] + b2[e];
                                        float sg = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)) * h2;  // silu(x1) * x2
                                        asm("" : "+v"(sg));
                                        o[e] = E::from_f32(sg);
                                    }
                                }
                            }
                            const int row = i * 16 + er;
                            const int slot = (4 * b + 2 * j + (eq >> 1)) ^ (row & 7);  // 8 columns (16 B) per slot
                            *(vec4*)(ep + row * 128 + slot * 16 + (eq & 1) * 8) = o;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
                if constexpr (EPI == EPI_SWIGLU) {
                    const int hid0 = (nw0 >> 6) * 32;  // 32 hidden units = 64 B per row: 4 lanes per row
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (q * 64 + it * 16 >= 16 * NI) continue;
                        const int row = it * 16 + (el >> 2), slot = el & 3;
                        const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int m = mbase + q * 64 + row;
                        if (m < M) DINO5_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + hid0 + slot * 8), v);
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        if (q * 64 + it * 8 >= 16 * NI) continue;
                        const int row = it * 8 + (el >> 3), slot = el & 7;
                        const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                        const int m = mbase + q * 64 + row;
                        if (m < M) DINO5_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + nw0 + slot * 8), v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (EPI == EPI_RESID || EPI == EPI_PLAIN_F32) {
            // 4-byte outputs: four passes (column half b, token rows [0, 64) / [64, 96)) of <= 64 rows x 32 columns (128 B per row), each
            // transposed through the wave's LDS slice and moved as whole lines.  The residual-stream rows of a pass are requested PF passes
            // ahead, before the stores of the current pass (gfx950 retires loads and stores through one in-order counter: gemm4.hip).
            constexpr int PF = DINO_GEMM5_PF;
            float4 add[4][8];
            auto issue_loads = [&](int ps) {
                if constexpr (EPI == EPI_RESID) {
                    const int b = ps >> 1, q = ps & 1;
                    const int nb = nw0 + b * 32 + (el & 7) * 4;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        if (q * 64 + it * 8 >= 16 * NI) continue;
                        int m = mbase + q * 64 + it * 8 + (el >> 3);
                        m = m < M ? m : M - 1;
                        add[ps][it] = *(const float4*)((const float*)p.out + (size_t)m * p.ldo + nb);
                    }
                }
            };
#pragma unroll
            for (int ps = 0; ps < PF; ++ps) issue_loads(ps);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int b = ps >> 1, q = ps & 1;
                asm volatile("" : "+v"(el));  // per pass: row pointers are recomputed, not kept live across the passes
                const int ncol = nw0 + 4 * (el >> 4);
                const int nb = nw0 + b * 32 + (el & 7) * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 ls = make_float4(1.f, 1.f, 1.f, 1.f);
                    if constexpr (EPI == EPI_RESID) ls = *(const float4*)(p.aux + ncol + b * 32 + j * 16);
                    const float4 b4 = p.bias ? *(const float4*)(p.bias + ncol + b * 32 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (q * 4 + i >= NI) continue;
                        const int row = i * 16 + (el & 15);
                        const int slot = (4 * j + (el >> 4)) ^ (row & 7);  // 4 columns (16 B) per slot
                        const f32x4 a = DINO5_ACC(q, b, i, j);
                        *(float4*)(ep + row * 128 + slot * 16) =
                            make_float4((a[0] + b4.x) * ls.x, (a[1] + b4.y) * ls.y, (a[2] + b4.z) * ls.z, (a[3] + b4.w) * ls.w);
                    }
                }
                if (ps + PF < 4) issue_loads(ps + PF);  // ahead of this pass's stores
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (q * 64 + it * 8 >= 16 * NI) continue;
                    const int row = it * 8 + (el >> 3), slot = el & 7;
                    float4 v = *(const float4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                    if constexpr (EPI == EPI_RESID)
                        v = make_float4(v.x + add[ps][it].x, v.y + add[ps][it].y, v.z + add[ps][it].z, v.w + add[ps][it].w);
                    const int m = mbase + q * 64 + row;
                    if (m < M) *(float4*)((float*)p.out + (size_t)m * p.ldo + nb) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#undef DINO5_ACC
        DINO5_GP(1)
        DINO5_GP_TILE
        t_cur = t_nxt;
    }  // persistent tile loop
    // the last workgroup to leave puts the XCD's counters back to zero for the next launch on this stream (hipGraph replays included)
    if (puller) {
        const unsigned left = __hip_atomic_fetch_add(head + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)nb_x - 1u) {
            __hip_atomic_store(head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(head + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    DINO5_GP_FLUSH
#undef DINO5_PIECE
}

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm5_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm5_body<T, EPI>(p, smem);
}

#ifdef DINO_GEMM5_PROF
static void gemm5_prof_dump(int epi, int nblocks, const GemmArgs& g) {
    static unsigned long long h[512 * 8], z[512 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm5_prof), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm5_prof), z, sizeof z);
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // per-workgroup lifetime (100 MHz ticks) and tile time, first and second half of the grid apart (dispatch order: the second half is
    // the YOUNGER workgroup of every CU)
    double life[2][3] = {{1e30, 0, 0}, {1e30, 0, 0}}, per_tile[2] = {0, 0}, ntl[2] = {0, 0};
    for (int b = 0; b < nblocks; ++b) {
        for (int i = 0; i < 8; ++i) a[i] += (double)h[b * 8 + i];
        const int hf = b >= nblocks / 2 ? 1 : 0;
        const double l = (double)h[b * 8 + 4] * 0.01;
        life[hf][0] = l < life[hf][0] ? l : life[hf][0];
        life[hf][1] += l / (nblocks / 2.0);
        life[hf][2] = l > life[hf][2] ? l : life[hf][2];
        per_tile[hf] += l;
        ntl[hf] += (double)h[b * 8 + 3];
    }
    const double t = a[3] > 0 ? a[3] : 1, kt = a[7] > 0 ? a[7] : 1;
    static int shown = 0;
    if (shown++ % 25 == 24)
        fprintf(stderr, "gemm5_prof epi %d N %d K %d grid %d: per tile (wave 0): K loop %.0f cycles, epilogue %.0f, hand-over %.0f, %.2f us in all (%.2f tiles per workgroup) -> clock %.3f GHz | "
                "middle K-tiles: %.0f cycles each (one wave's 48 MFMAs = 768), %.3f us, clock %.3f GHz | workgroup busy time us min/avg/max: older half %.1f/%.1f/%.1f (%.2f us per tile), younger half %.1f/%.1f/%.1f (%.2f us per tile)\n",
                epi, g.N, g.K, nblocks, a[0] / t, a[1] / t, a[2] / t, a[4] / t * 0.01, a[3] / nblocks, (a[0] + a[1] + a[2]) / (a[4] * 10.0), a[5] / kt, a[6] / kt * 0.01,
                a[5] / (a[6] * 10.0), life[0][0], life[0][1], life[0][2], per_tile[0] / (ntl[0] > 0 ? ntl[0] : 1), life[1][0], life[1][1], life[1][2],
                per_tile[1] / (ntl[1] > 0 ? ntl[1] : 1));
}
#endif

static int g5_wgs_per_cu = -1;  // the occupancy query's answer, worst instantiation (the design needs 2); -1 = not asked yet (no device)

// Default ticket counters (one set per device: a module global), for callers that bring none (GemmArgs::sched == nullptr: the diagnostic
// entry points, which launch on one stream at a time).  Launches that may overlap on one device -- sessions on different streams -- must
// bring their own 1 KiB of zero-initialised device memory each: two kernels pulling from one counter would split one tile list.
__device__ unsigned g5_sched[8 * 32];

template <typename T>
static hipError_t launch5_t(Epilogue epi, const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    if (!a.sched) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g5_sched));
        if (e != hipSuccess) return e;
        a.sched = (unsigned*)ptr;
    }
    const int tiles = (a.N / G5_BN) * ((a.M + G5_BM - 1) / G5_BM);
    const dim3 grid(tiles < DINO_GEMM5_GRID ? tiles : DINO_GEMM5_GRID), block(256);
#define DINO_L5(E)                                                            \
    case E:                                                                   \
        hipLaunchKernelGGL((gemm5_kernel<T, E>), grid, block, G5_LDS, st, a); \
        break;
    switch (epi) {
        DINO_L5(EPI_QKV)
        DINO_L5(EPI_RESID)
        DINO_L5(EPI_GELU)
        DINO_L5(EPI_SWIGLU)
        DINO_L5(EPI_PLAIN_F32)
        default: return hipErrorInvalidValue;
    }
#undef DINO_L5
#ifdef DINO_GEMM5_PROF
    gemm5_prof_dump((int)epi, (int)grid.x, a);
#endif
    return hipGetLastError();
}

// requires N % 128 == 0, K / 64 even and >= 4, an epilogue other than EPI_PATCH, and two resident workgroups per CU (unknown before
// gemm5_init has asked the device: the plan query of a machine without one assumes the design point)
bool gemm5_ok(Epilogue epi, const GemmArgs& a) {
    return (g5_wgs_per_cu < 0 || g5_wgs_per_cu >= 2) && epi != EPI_PATCH && a.N % G5_BN == 0 && a.K % 128 == 0 && a.K >= 256;
}
hipError_t launch_gemm5(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch5_t<_Float16>(epi, a, st) : launch5_t<__bf16>(epi, a, st);
}
int gemm5_wgs_per_cu() { return g5_wgs_per_cu; }

template <typename T>
static hipError_t attr5_t() {
    hipError_t e = hipSuccess;
#define DINO_A5(E) \
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm5_kernel<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G5_LDS);
    DINO_A5(EPI_QKV)
    DINO_A5(EPI_RESID)
    DINO_A5(EPI_GELU)
    DINO_A5(EPI_SWIGLU)
    DINO_A5(EPI_PLAIN_F32)
#undef DINO_A5
    return e;
}

hipError_t gemm5_init() {
    hipError_t e = attr5_t<_Float16>();
    if (e == hipSuccess) e = attr5_t<__bf16>();
    if (e == hipSuccess) {
        // the design needs two resident workgroups per CU (registers <= 256 per lane, LDS <= 80 KiB): ask, and keep the slowest answer
        int n = 0, worst = 1 << 30;
#define DINO_O5(T, E)                                                                                                                        \
    if (e == hipSuccess) {                                                                                                                   \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&gemm5_kernel<T, E>), 256, G5_LDS);             \
        worst = n < worst ? n : worst;                                                                                                       \
    }
        DINO_O5(_Float16, EPI_QKV)
        DINO_O5(_Float16, EPI_RESID)
        DINO_O5(_Float16, EPI_GELU)
        DINO_O5(_Float16, EPI_SWIGLU)
        DINO_O5(_Float16, EPI_PLAIN_F32)
        DINO_O5(__bf16, EPI_QKV)
        DINO_O5(__bf16, EPI_RESID)
        DINO_O5(__bf16, EPI_GELU)
        DINO_O5(__bf16, EPI_SWIGLU)
        DINO_O5(__bf16, EPI_PLAIN_F32)
#undef DINO_O5
        if (e == hipSuccess) {
            g5_wgs_per_cu = worst;
            if (worst < 2) fprintf(stderr, "dinov2_hip: gemm5 kernels get %d workgroup(s) per CU instead of 2; generation 5 disabled\n", worst);
        }
    }
    return e;
}

}  // namespace dinov2
