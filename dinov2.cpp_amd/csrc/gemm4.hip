// gemm4.hip -- fourth generation of the f16/bf16 MFMA GEMM: the 256 x 256 x 64 tile on FOUR waves (one per SIMD, 512 registers each: 256
// accumulators in AGPRs + fragments in VGPRs), every wave a 128 x 128 output block, the K loop ONE hand-ordered instruction stream per
// wave.  Same contract, epilogues, K order (a row's bits are those of gemm.hip / gemm2.hip) and XCD-aware persistent tile walk as
// gemm2.hip, which stays for the launches this kernel is not given (K < 1 024, where its shorter epilogue wins -- launch_gemm in gemm.hip --
// the patch-embed epilogue, the 192-row-only plan).
//
// Why (profiles/r04_gemm4w.md): gemm2.hip's K loop runs barrier-separated MEM / MMA sections shared by the two waves of a SIMD and
// reaches ~75 % matrix-pipe duty; the vendor's assembly kernel of the same macro tile is a single wave per SIMD issuing its 128 MFMAs per
// K-tile back to back with the memory instructions in their shadow.  Rebuilt here from scratch (standalone probe:
// tools/probes/gemm4w.hip, where the schedule was measured): 88 % duty in the K loop -- at which point the part's power limit, not
// the schedule, sets the rate (the shader clock drops from 1.87 to 1.66 GHz; duty x clock moves + 3 %) -- no tile prologue (the next
// tile's first two K-tiles are staged under the last two of the current one and its first fragments are in registers before the
// epilogue starts), and an epilogue that no second wave of the SIMD competes with.
//
//   LDS (160 KiB): X buffers 0 / 1 at 0 / 32 KiB, W buffers 0 / 1 at 64 / 96 KiB (a buffer = 256 rows x 128 B, one K-tile of one operand,
//   16-byte chunks XOR-swizzled by (row >> 1) & 7 on the SOURCE side of the LDS-DMA and on the fragment reads); 4 x 8 KiB epilogue slices at
//   128 KiB.
//   Wave (wr, wc) = (wid >> 1, wid & 1): tokens [16 NI wr, + 16 NI) x columns [128 wc, + 128), NI x 8 blocks of 16 x 16 (NI = 8: 256-row
//   tiles, 6: 192-row tiles for the partial last round, same launch -- gemm4_mixed_kernel; 2 / 3 / 4: 64- / 96- / 128-row tiles, one per
//   workgroup, for launches that cannot fill the chip with taller ones -- batch 1), acc[i][j] in AGPRs; OPERAND SWAP as in
//   gemm2.hip (weight fragment = MFMA A operand): a lane owns one token and four consecutive output columns.
//   Fragment registers: P = k-step 0 of a K-tile (NI X + 8 W fragments of 4 VGPRs), Q = k-step 1.  MFMA order: column block j outer,
//   token block i inner; MFMA index m = 64 ks + 8 j + i.
//   K-tile t in buffer b = t & 1:
//     m = 0, 2 .. 30   one ds_read_b128 of Q(t) behind every second MFMA (W fragments first)
//     m = 8 k          one LDS-DMA piece (global_load_lds_dwordx4, 8 rows x 128 B) per 8 MFMAs (short tiles: per 4): before barrier A the LAST pieces of K-tile
//                      t + 1 (-> buffer b ^ 1), after it the FIRST pieces of K-tile t + 2 (-> buffer b): the staging of a K-tile spans one
//                      whole K-tile time, and every piece has >= 1 000 matrix-pipe cycles to land
//     m = 39           s_waitcnt lgkmcnt(0); s_barrier      [A] every wave has read all of buffer b
//     m = 103          s_waitcnt vmcnt(8); s_barrier        [B] K-tile t + 1 is in buffer b ^ 1 for everyone (8 = pieces issued since A)
//     m = 104 .. 119   one ds_read_b128 of P(t + 1) per MFMA; s_waitcnt lgkmcnt(0) at the end
//   Hazards (MI355X_MICROARCH.md, "Two waves per SIMD" item 7): LDS-DMA data is read only after the issuing waves' counted vmcnt AND a
//   barrier [B]; a buffer is re-staged only after an lgkmcnt(0) + barrier [A] behind its last read.
#include <cstdio>
#include <type_traits>
#include <utility>

#include "device_types.h"
#include "kernels.h"

namespace dinov2 {

template <int... Is, class F>
static __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
static __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// The K loop's instructions, as asm statements: volatile statements keep their program order, so the stream below is emitted exactly as
// written; hipcc only allocates the registers ("a": accumulator file).  Nothing else in this kernel uses M0 (the LDS-DMA destination),
// which is written in the same statement that reads it.
#define DINO4_MFMA_F16(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define DINO4_MFMA_F16_Z(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(ACC) : "v"(W), "v"(X))
#define DINO4_MFMA_BF16(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define DINO4_MFMA_BF16_Z(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(W), "v"(X))
#define DINO4_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
// (M0 = one wave-uniform base + an immediate: sixteen pieces x two buffers would otherwise sit in thirty-two SGPRs.  M0 is a RESERVED
// register for hipcc: it never keeps a value in it across statements and re-materialises it before each of its own uses (LDS-direct /
// movrel forms), and it rejects "m0" in a clobber list with -Winline-asm "clobber list contains reserved registers" -- ADVICE r4)
#define DINO4_GLDS(VOFF, SBASE, LDSBASE, IMM)                                                                                   \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSBASE), "n"(IMM) \
                 : "memory", "scc")
// (tuning builds, -DDINO_GEMM4_NT=1|2|3|4: non-temporal LDS-DMA for the X pieces / the W pieces / both / the X pieces of the residual epilogue only -- profiles/r05_gemm4_nt_loads.txt)
#define DINO4_GLDS_NT(VOFF, SBASE, LDSBASE, IMM)                                                                                \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(VOFF), "s"(SBASE), "s"(LDSBASE), "n"(IMM) \
                 : "memory", "scc")
#ifndef DINO_GEMM4_NT
#define DINO_GEMM4_NT 0
#endif
// -DDINO_LN_ABL=<bits> (TIMING-ONLY tuning builds, results are wrong): which parts of the LN-fold epilogues are left out, to price them
// (profiles/r06_ln_fold.md).  Consumers: 1 no next-tile statistics prefetch, 2 no s / c lane exchange, 4 no row-coefficient lane exchange,
// 8 plain add instead of the two fused multiply-adds.  Producer: 16 no row statistics (DPP reduction + store), 32 no xg conversion / store,
// 64 three passes of residual rows in flight instead of two.
#ifndef DINO_LN_ABL
#define DINO_LN_ABL 0
#endif

// Clock probe slots of this file's kernels (device_types.h, "clock probe")
__device__ unsigned long long g_clk4[CLK_SLOTS * 4];
hipError_t gemm4_clock_probe_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk4), sizeof(unsigned long long) * CLK_SLOTS * 4);
}

constexpr int G4_PA = 39, G4_PB = 103;  // barrier A / B behind these MFMA indices

// 16-byte store of the 2-byte epilogues.  NON-TEMPORAL when the launcher says so (GemmArgs::nt_out: outputs larger than the chip's 32 MiB of
// L2), so that a tile's 128 KiB of output does not push the operand panels out of the XCD's 4 MiB L2 (32 CUs x 128 KiB = all of it); the
// consumer is another kernel and reads it through the memory side anyway.  Same box, interleaved: QKV 0.2496 -> 0.2378 ms in the
// micro-benchmark, forward 927 - 930 -> 933 - 935 images/s (the FFN-out GEMM that reads the FFN hidden buffer gains most: 0.331 -> 0.325 ms).
// Small outputs (batch 1) stay ordinary stores: their consumer finds them in L2 (p50 2.50 against 2.58 ms with non-temporal stores).
#define DINO4_ST16(PTR, V)                               \
    {                                                    \
        if (nt_out) __builtin_nontemporal_store((V), (PTR)); \
        else *(PTR) = (V);                               \
    }

// -DDINO_GEMM4_PROF (tuning builds): s_memtime sums per workgroup of wave 0 -- [0] K loops, [1] epilogues, [2] tiles, [3] the 100 MHz ticks of
// both -- printed by the launcher after each launch.
#ifdef DINO_GEMM4_PROF
__device__ unsigned long long g_gemm4_prof[256 * 4];
#define DINO4_GP_DECL unsigned long long gp_t = 0, gp_r = 0, gp_acc[4] = {0, 0, 0, 0};
#define DINO4_GP_START { gp_t = __builtin_readcyclecounter(); gp_r = __builtin_amdgcn_s_memrealtime(); }
#define DINO4_GP(i) { const unsigned long long t__ = __builtin_readcyclecounter(); gp_acc[i] += t__ - gp_t; gp_t = t__; }
#define DINO4_GP_TILE { gp_acc[2] += 1; const unsigned long long r__ = __builtin_amdgcn_s_memrealtime(); gp_acc[3] += r__ - gp_r; gp_r = r__; }
#define DINO4_GP_FLUSH if (threadIdx.x == 0) for (int i__ = 0; i__ < 4; ++i__) g_gemm4_prof[blockIdx.x * 4 + i__] += gp_acc[i__];
#else
#define DINO4_GP_DECL
#define DINO4_GP_START
#define DINO4_GP(i)
#define DINO4_GP_TILE
#define DINO4_GP_FLUSH
#endif

template <typename T, int EPI, int NI>
static __device__ __forceinline__ void gemm4_body(const GemmArgs& p, char* smem) {
    // (no implicit mul+add -> fma contraction: an element's bits must not depend on where its row sits in a tile -- see gemm2.hip)
#pragma clang fp contract(off)
    using E = Elem<T>;
    using vec4 = typename E::vec4;
    constexpr bool F16 = std::is_same<T, _Float16>::value;
    // LN fold (kernels.h): LNC = this GEMM consumes T(gamma x) and applies the LayerNorm in its epilogue; EB = the epilogue it specialises
    constexpr bool LNC = EPI == EPI_QKV_LN || EPI == EPI_GELU_LN || EPI == EPI_SWIGLU_LN;
    constexpr int EB = EPI == EPI_RESID_LN ? EPI_RESID : EPI == EPI_QKV_LN ? EPI_QKV : EPI == EPI_GELU_LN ? EPI_GELU : EPI == EPI_SWIGLU_LN ? EPI_SWIGLU : EPI;
    constexpr int BM = 32 * NI, BN = 256;
    constexpr int NP = NI + 8;         // LDS-DMA pieces per wave and K-tile: NI of X, 8 of W
    // staging slots at m = SP k: one per 8 MFMA indices for the tall tiles (a K-tile's staging spans one K-tile time); one per 4 for the
    // short ones, whose K-tiles have 32 - 64 MFMAs (i < NI only) and give a piece little time to land: all of theirs go out right behind A
    constexpr int SP = NI <= 4 ? 4 : 8;
    constexpr int G4_NPRE = G4_PA / SP + 1;    // slots up to barrier A
    constexpr int NSLOT = 128 / SP - G4_NPRE;  // slots behind barrier A
    constexpr int NPOST = NP < NSLOT ? NP : NSLOT;  // pieces 0 .. NPOST - 1 of the K-tile two ahead go out behind barrier A, the rest before the next one
    constexpr int NBS = (G4_PB - SP * G4_NPRE) / SP + 1;  // slots between the barriers
    constexpr int G4_NB = NBS < NPOST ? NBS : NPOST;      // pieces issued between them: what barrier B's vmcnt leaves in flight
    static_assert(NP - NPOST <= G4_NPRE, "staging schedule");

    DINO4_GP_DECL
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // opaque: nothing lane-derived is shared between the two bodies of gemm4_mixed_kernel
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int M = p.M, N = p.N, K = p.K;
    const bool nt_out = p.nt_out != 0;
    const unsigned lda2 = (unsigned)(p.lda ? p.lda : K) * 2u, ldw2 = (unsigned)(p.ldw ? p.ldw : K) * 2u;
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int ntiles = ntn * ntm;
    const int nk = K / 64;  // even, >= 4 (checked by the launcher)

    // XCD-aware persistent tile walk (gemm2.hip): block b sits on XCD b % 8, every XCD walks a contiguous chunk of the tile order in
    // patches of GM row panels x 32 / GM column tiles
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int nb_x = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int chunk0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int chunkn = tq + (xcd < tr ? 1 : 0);
#ifndef DINO_GEMM4_GM
#define DINO_GEMM4_GM 8  // (tuning builds: 4 | 16)
#endif
    constexpr int GM = DINO_GEMM4_GM;
    auto tile_mn = [&](int lid, int& m0, int& n0) {
        const int g = lid / (GM * ntn), r = lid - g * (GM * ntn);
        const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
        const int n = r / gm, mi = r - n * gm;
        m0 = (g * GM + mi) * BM;
        n0 = n * BN;
    };

    // ---- staging.  Piece pc of a wave (0 .. NP - 1): X and W alternate while X has pieces left (pc < 2 NI: operand pc & 1, row block
    // pc >> 1), then W only (row block pc - NI).  Wave w moves X rows [8 NI w, + 8 NI) and W rows [64 w, + 64) of the tile; a piece is 8
    // rows x 128 B (lane -> row lane >> 3, 16-byte chunk lane & 7, XOR-swizzled on the source side); X rows are clamped to M.
    unsigned so[NP];  // byte offsets from p.A / p.W of this lane's 16 bytes of every piece (K-tile 0; + 128 kt through the scalar base)
    auto piece_off = [&](int pc, int m0, int n0) -> unsigned {
        const bool isw = pc >= 2 * NI || (pc & 1);
        const int rb = pc >= 2 * NI ? pc - NI : pc >> 1;
        // (LN consumers: their epilogue needs the registers that the loop-invariant parts of these sixteen offsets would otherwise occupy
        //  across the whole tile loop -- they spilled, and were reloaded from scratch inside the K loop; recomputed at the tile change instead)
        int pl = lane;
        if constexpr (LNC) asm volatile("" : "+v"(pl));
        const int r = (isw ? 64 : 8 * NI) * wid + 8 * rb + (pl >> 3);
        const int ch = (pl & 7) ^ ((r >> 1) & 7);
        if (isw) return (unsigned)(n0 + r) * ldw2 + ch * 16;
        int gm = m0 + r;
        gm = gm < M ? gm : M - 1;
        return (unsigned)gm * lda2 + ch * 16;
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(DINO_LDS_AS char*)smem;
    const unsigned ldsx = lds0 + (unsigned)wid * (NI * 1024u), ldsw = lds0 + 65536u + (unsigned)wid * 8192u;

    // ---- fragment addresses: lane -> row lane & 15 of a 16-row block, 16-byte chunk (4 ks + (lane >> 4)) ^ ((row >> 1) & 7)
    const int fr = lane & 15, kq = lane >> 4, sw = (fr >> 1) & 7;
    unsigned xa[2], wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned ch = (unsigned)(((ks * 4 + kq) ^ sw) << 4);
        xa[ks] = lds0 + (unsigned)((wr * (16 * NI) + fr) * 128) + ch;
        wa[ks] = lds0 + 65536u + (unsigned)((wc * 128 + fr) * 128) + ch;
    }

    // acc[i][j][e] = C[m0 + 16 NI wr + 16 i + (lane & 15)][n0 + 128 wc + 16 j + 4 (lane >> 4) + e]
    f32x4 acc[NI][8];
    u32x4 Px[NI], Pw[8], Qx[NI], Qw[8];

    const char* const Ab = (const char*)p.A;
    const char* const Wb = (const char*)p.W;

#define DINO4_PIECE(PC, KT, BUF)                                                                                            \
    {                                                                                                                       \
        constexpr int pc__ = (PC);                                                                                          \
        constexpr bool isw__ = pc__ >= 2 * NI || (pc__ & 1);                                                                \
        constexpr int rb__ = pc__ >= 2 * NI ? pc__ - NI : pc__ >> 1;                                                        \
        if constexpr (isw__) {                                                                                              \
            if constexpr (DINO_GEMM4_NT & 2) DINO4_GLDS_NT(so[pc__], Wb + (size_t)(KT) * 128, ldsw, (BUF) * 32768 + rb__ * 1024);   \
            else DINO4_GLDS(so[pc__], Wb + (size_t)(KT) * 128, ldsw, (BUF) * 32768 + rb__ * 1024);                          \
        } else {                                                                                                            \
            if constexpr ((DINO_GEMM4_NT & 1) || ((DINO_GEMM4_NT & 4) && EPI == EPI_RESID))                                 \
                DINO4_GLDS_NT(so[pc__], Ab + (size_t)(KT) * 128, ldsx, (BUF) * 32768 + rb__ * 1024);                        \
            else DINO4_GLDS(so[pc__], Ab + (size_t)(KT) * 128, ldsx, (BUF) * 32768 + rb__ * 1024);                          \
        }                                                                                                                   \
    }

    // One K-tile in buffer B.  FIRST: the accumulators start from zero (first K-tile of an output tile).  Staging under it: the slots up
    // to barrier A carry the last NP - NPOST pieces of the K-tile that goes into buffer b ^ 1 (kt_pre; if pre_on); at A `at_a()` runs
    // (it switches the piece offsets to the next output tile where the staging crosses over); the slots behind A carry the first NPOST
    // pieces of the K-tile that goes into THIS buffer (kt_post; if post_on).  do_readp: read P of the next K-tile behind barrier B.
    auto ktile = [&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsx, &ldsw, Ab, Wb](auto bc, auto firstc, int kt_pre, bool pre_on, auto&& at_a,
                                                                                int kt_post, bool post_on, bool do_readp) {
        static_for<128>([&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsx, &ldsw, Ab, Wb, &kt_pre, &pre_on, &at_a, &kt_post, &post_on, &do_readp,
                         bc, firstc](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int b = decltype(bc)::value;
            constexpr bool FIRST = decltype(firstc)::value;
            constexpr int ks = m >> 6, j = (m >> 3) & 7, i = m & 7;
            if constexpr (i < NI) {
                if constexpr (ks == 0) {
                    if constexpr (FIRST) {
                        if constexpr (F16) DINO4_MFMA_F16_Z(acc[i][j], Pw[j], Px[i]);
                        else DINO4_MFMA_BF16_Z(acc[i][j], Pw[j], Px[i]);
                    } else {
                        if constexpr (F16) DINO4_MFMA_F16(acc[i][j], Pw[j], Px[i]);
                        else DINO4_MFMA_BF16(acc[i][j], Pw[j], Px[i]);
                    }
                } else {
                    if constexpr (F16) DINO4_MFMA_F16(acc[i][j], Qw[j], Qx[i]);
                    else DINO4_MFMA_BF16(acc[i][j], Qw[j], Qx[i]);
                }
            }
            // Q(t): k-step 1 of this K-tile (W fragments first: their registers have been free longest)
            if constexpr (m % 2 == 0 && m / 2 < 8 + NI) {
                constexpr int q = m / 2;
                if constexpr (q < 8) DINO4_DSR(Qw[q], wa[1], q * 2048 + b * 32768);
                else DINO4_DSR(Qx[q - 8], xa[1], (q - 8) * 2048 + b * 32768);
            }
            // staging slots
            if constexpr (m % SP == 0) {
                constexpr int k = m / SP;
                if constexpr (k < G4_NPRE) {
                    if constexpr (NPOST + k < NP) {
                        if (pre_on) DINO4_PIECE(NPOST + k, kt_pre, b ^ 1)
                    }
                } else if constexpr (k - G4_NPRE < NPOST) {
                    if (post_on) DINO4_PIECE(k - G4_NPRE, kt_post, b)
                }
            }
            if constexpr (m == G4_PA) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                at_a();
            }
            if constexpr (m == G4_PB) {
                if (post_on) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G4_NB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            // P(t + 1): k-step 0 of the next K-tile, from the other buffer
            if constexpr (m > G4_PB && m <= G4_PB + 8 + NI) {
                constexpr int q = m - G4_PB - 1;
                if (do_readp) {
                    if constexpr (q < 8) DINO4_DSR(Pw[q], wa[0], q * 2048 + (b ^ 1) * 32768);
                    else DINO4_DSR(Px[q - 8], xa[0], (q - 8) * 2048 + (b ^ 1) * 32768);
                }
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    static_assert(2 * (8 + 8) - 2 < G4_PA && G4_PB + 16 <= 127, "fragment reads fit their windows");

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using TT = std::integral_constant<bool, true>;
    using TF = std::integral_constant<bool, false>;
    auto nop = [] {};

    // LN consumers: st_r / st_n[h] = the LayerNorm coefficients (r, -mean r) of row 64 h + lane of this wave's 16 NI rows of the CURRENT tile.
    // They are requested one tile ahead -- for the first tile here, ahead of the staging (the first K-tile's wait covers them), for the
    // others from inside the previous tile's epilogue -- and handed to the lanes that need them through the wave's LDS slice at the
    // start of the epilogue.
    constexpr int LNH = !LNC ? 0 : NI > 4 ? 2 : 1;
    float st_r[LNH ? LNH : 1], st_n[LNH ? LNH : 1];
    const int ln_gs = p.ln_gs;  // slots per row of the statistics buffer: 12 or 24
    const float ln_inv_h = 1.0f / (float)K;
    auto ln_row_of = [&](int m0_, int h) {
        int ll = lane;
        asm volatile("" : "+v"(ll));  // (opaque: nothing of this is to be hoisted out of the persistent tile loop)
        int rl = h * 64 + ll;
        rl = rl < 16 * NI ? rl : 16 * NI - 1;
        const int row = m0_ + wr * (16 * NI) + rl;
        return row < M ? row : M - 1;
    };

    if (bidx < chunkn) {
        int pm0, pn0;
        tile_mn(chunk0 + bidx, pm0, pn0);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) so[pc] = piece_off(pc, pm0, pn0);
        // every wave has left the previous body's epilogue slices and buffers (gemm4_mixed_kernel runs two bodies back to back)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        LnRaw lraw[LNH ? LNH : 1], lraw1[LNH ? LNH : 1];  // (both halves of the rows' slots at once: one round trip, under the staging)
        if constexpr (LNC) {
#pragma unroll
            for (int h = 0; h < LNH; ++h) {
                ln_row_load<0>(p.stats, ln_row_of(pm0, h), ln_gs, lraw[h]);
                if (ln_gs > 12) ln_row_load<1>(p.stats, ln_row_of(pm0, h), ln_gs, lraw1[h]);
            }
        }
        // K-tile 0 whole, K-tile 1's first NPOST pieces (its last ones go out before barrier A of K-tile 0, like everywhere else)
        static_for<NP>([&so, &ldsx, &ldsw, Ab, Wb](auto pcc) { DINO4_PIECE(decltype(pcc)::value, 0, 0) });
        static_for<NPOST>([&so, &ldsx, &ldsw, Ab, Wb](auto pcc) { DINO4_PIECE(decltype(pcc)::value, 1, 1) });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPOST) : "memory");
        asm volatile("s_barrier" ::: "memory");
        static_for<8 + NI>([&Px, &Pw, &xa, &wa](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q < 8) DINO4_DSR(Pw[q], wa[0], q * 2048);
            else DINO4_DSR(Px[q - 8], xa[0], (q - 8) * 2048);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (LNC) {
#pragma unroll
            for (int h = 0; h < LNH; ++h) {
                LnAcc la;
                ln_row_add<0>(lraw[h], la);
                if (ln_gs > 12) ln_row_add<1>(lraw1[h], la);
                ln_row_finish(la, ln_inv_h, p.ln_eps, st_r[h], st_n[h]);
            }
        }
    }
    for (int tix = bidx; tix < chunkn; tix += nb_x) {
        int m0, n0;
        tile_mn(chunk0 + tix, m0, n0);
        const bool has_next = tix + nb_x < chunkn;
        DINO4_GP_START

        // K-tiles 0, 1 (accumulators from zero), the middle, and the last two, under which the staging crosses over to the next output tile
        ktile(T0{}, TT{}, 1, true, nop, 2, true, true);
        ktile(T1{}, TF{}, 2, true, nop, 3, true, true);
        for (int t = 2; t < nk - 2; t += 2) {
            ktile(T0{}, TF{}, t + 1, true, nop, t + 2, true, true);
            ktile(T1{}, TF{}, t + 2, true, nop, t + 3, true, true);
        }
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_mn(chunk0 + tix + nb_x, nm0, nn0);
        ktile(T0{}, TF{}, nk - 1, true,
              [&] {
                  if (has_next) {
                      if constexpr (LNC) {
                          // (LN consumers keep none of piece_off's loop-invariant parts in registers: inside the matrix, the next tile's
                          //  offsets are this tile's plus two scalars; only a tile on the M edge -- clamped rows -- takes the full computation)
                          if (m0 + BM <= M && nm0 + BM <= M) {
                              const unsigned dx = (unsigned)(nm0 - m0) * lda2, dw = (unsigned)(nn0 - n0) * ldw2;
#pragma unroll
                              for (int pc = 0; pc < NP; ++pc) so[pc] += (pc >= 2 * NI || (pc & 1)) ? dw : dx;
                          } else {
#pragma unroll
                              for (int pc = 0; pc < NP; ++pc) so[pc] = piece_off(pc, nm0, nn0);
                          }
                      } else {
#pragma unroll
                          for (int pc = 0; pc < NP; ++pc) so[pc] = piece_off(pc, nm0, nn0);
                      }
                  }
              },
              0, has_next, true);
        // LN consumers: the wave's 128 columns of s[n] (lanes 0 .. 31, four each) and c[n] (lanes 32 .. 63), requested under the last K-tile.
        // The epilogue's passes fetch their 16-column blocks from these lanes (ds_bpermute): a global load inside a pass would have to
        // wait for the previous pass's STORES to retire (gfx950 retires loads and stores through one in-order counter) -- measured: QKV
        // + 15 %, FFN-in + 18 % with per-block loads.
        float4 lnsc = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LNC) {
            int ll = lane;
            asm volatile("" : "+v"(ll));
            lnsc = *(const float4*)((ll < 32 ? p.ln_s : p.ln_c) + n0 + wc * 128 + 4 * (ll & 31));
        }
        // (LN consumers read the next tile's first fragments AFTER the epilogue, not under this K-tile: their epilogue has no 64 registers to
        //  carry them in, and a fragment the compiler spills is spilled right behind the asm ds_read that requests it -- before the data has
        //  arrived.  Round 6: that is what happened to Pw[3] of EPI_QKV_LN, and every tile lost its first k-step in columns 48 .. 63.)
        ktile(T1{}, TF{}, 0, has_next, nop, 1, has_next, has_next && !LNC);
        // The compiler's hazard recogniser cannot see into the asm statements: make the last MFMAs' results architecturally visible to the
        // v_accvgpr_read of the epilogue by hand (16x16x32: 8 passes; the epilogue starts with acc[0][0], written 127 MFMAs ago, but nothing in
        // the source guarantees that order) -- 24 idle cycles per tile
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        // ... and pin every accumulator read BEHIND those wait states: a v_accvgpr_read is an ordinary instruction with a data dependence on
        // the LAST asm statement that wrote its register only, so the scheduler is free to hoist it into the K loop's stream, right behind
        // that MFMA (round 6: it did, in the EPI_QKV_LN instantiation -- columns 48 .. 63 of every first column group came out wrong,
        // timing-dependent).  Volatile statements keep their order, and each of these redefines its accumulator after the s_nop.
        static_for<NI * 8>([&acc](auto ic) { asm volatile("" : "+a"(acc[decltype(ic)::value / 8][decltype(ic)::value % 8])); });
        DINO4_GP(0)

        // ---- epilogue (gemm2.hip's, per 64-column group cg of the wave's 128 columns): each wave transposes its result through a private
        // 8 KiB LDS slice and moves whole 128-byte lines.  Slice image: 64 rows x 128 B, 16-byte slot s of row r stored at s ^ (r & 7).
        // `el` launders the lane id: without it LICM hoists the loop-invariant epilogue addresses out of the persistent tile loop and
        // they stay live across the K loop.
        int el = lane;
        asm volatile("" : "+v"(el));
        const int er = el & 15, eq = el >> 4;
        char* const ep = smem + 131072 + wid * 8192;
        const int mbase = m0 + wr * (16 * NI);
        // LN consumers: the coefficients of the CURRENT tile move aside (st_r / st_n are refilled for the next tile during this epilogue);
        // every pass fetches those of its four row blocks from the lanes that hold them (ds_bpermute: no LDS storage involved)
        float cu_r[LNH ? LNH : 1], cu_n[LNH ? LNH : 1];
        if constexpr (LNC) {
#pragma unroll
            for (int h = 0; h < LNH; ++h) {
                cu_r[h] = st_r[h];
                cu_n[h] = st_n[h];
            }
        }
#define DINO4_ACC(Q_, B_, I_, J_) acc[((Q_)*4 + (I_)) < NI ? ((Q_)*4 + (I_)) : 0][cg * 4 + (B_)*2 + (J_)]  /* (the clamp only ever acts in dead code) */
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const int nw0 = n0 + wc * 128 + cg * 64;  // first column of this 64-column group
            const int ncol = nw0 + 4 * eq;            // + 32 b + 16 j: this lane's four consecutive columns of block (b, j)
            // bs: the additive per-column term, fetched ahead for the whole column group.  (LN consumers fetch theirs -- c[n], which contains
            // the bias, and s[n] -- per 16-column block inside the passes: their epilogue has no registers to park 32 values in)
            float4 bs[2][2];
            if constexpr (!LNC && EPI != EPI_RESID_LN) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        bs[b][j] = p.bias ? *(const float4*)(p.bias + ncol + b * 32 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // LN consumers: this column group's s[n] / c[n] blocks from the lanes that hold them (one batch of ds_bpermute per column group:
            // with one wave per SIMD every dependent LDS round trip inside a pass is exposed)
            float4 lsv[LNC ? 2 : 1][2], lcv[LNC ? 2 : 1][2];
            if constexpr (LNC) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int f = cg * 16 + b * 8 + j * 4 + eq;  // this lane's float4 of block (b, j), as an index into the wave's 128 columns
                        if constexpr (DINO_LN_ABL & 2) {
                            lsv[b][j] = lnsc;
                            lcv[b][j] = lnsc;
                            continue;
                        }
                        lsv[b][j] = make_float4(__shfl(lnsc.x, f), __shfl(lnsc.y, f), __shfl(lnsc.z, f), __shfl(lnsc.w, f));
                        lcv[b][j] = make_float4(__shfl(lnsc.x, 32 + f), __shfl(lnsc.y, 32 + f), __shfl(lnsc.z, 32 + f), __shfl(lnsc.w, 32 + f));
                    }
            }
            // LN consumers with a next tile: request the partial sums of its rows 64 cg + lane now, turn them into coefficients when this
            // column group is done (st_r / st_n of the current tile were handed out above)
            LnRaw lraw;
            LnAcc lacc;
            const bool ln_next = LNC && cg < LNH && has_next && !(DINO_LN_ABL & 1);
            if constexpr (LNC) {
                if (ln_next) ln_row_load<0>(p.stats, ln_row_of(nm0, cg), ln_gs, lraw);
            }

            if constexpr (EB == EPI_QKV || EB == EPI_GELU || EB == EPI_SWIGLU) {
                // 2-byte outputs: two passes (token halves) of 64 rows x 64 columns (SwiGLU: x 32)
                const float qs = (EB == EPI_QKV && nw0 < p.qcols) ? p.qscale : 1.0f;
                constexpr int BN_ = EB == EPI_SWIGLU ? 1 : 2;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (LNC) {
                        if (q == 1 && ln_next) {  // between the token halves: first twelve groups in, the rest requested
                            ln_row_add<0>(lraw, lacc);
                            if (ln_gs > 12) ln_row_load<1>(p.stats, ln_row_of(nm0, cg), ln_gs, lraw);
                        }
                    }
                    if (q * 4 >= NI) continue;  // (tiles of at most 128 rows: one token half)
                    float lnr[LNC ? 4 : 1], lnn[LNC ? 4 : 1];  // rows 64 q + 16 i + (lane & 15): held by lane 16 i + (lane & 15) as its row-half q
                    if constexpr (LNC) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (DINO_LN_ABL & 4) {
                                lnr[i] = cu_r[q < LNH ? q : 0];
                                lnn[i] = cu_n[q < LNH ? q : 0];
                                continue;
                            }
                            lnr[i] = __shfl(cu_r[q < LNH ? q : 0], 16 * i + er);
                            lnn[i] = __shfl(cu_n[q < LNH ? q : 0], 16 * i + er);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < BN_; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float4 bbv, b2v, sbv = make_float4(0.f, 0.f, 0.f, 0.f), s2v = make_float4(0.f, 0.f, 0.f, 0.f);
                            if constexpr (LNC) {
                                sbv = lsv[b][j];
                                bbv = lcv[b][j];
                                b2v = lcv[1][j];
                                s2v = lsv[1][j];
                            } else {
                                bbv = bs[b][j];
                                b2v = bs[1][j];
                            }
                            const float bb[4] = {bbv.x, bbv.y, bbv.z, bbv.w};
                            const float b2[4] = {b2v.x, b2v.y, b2v.z, b2v.w};
                            const float sb[4] = {sbv.x, sbv.y, sbv.z, sbv.w};
                            const float s2[4] = {s2v.x, s2v.y, s2v.z, s2v.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (q * 4 + i >= NI) continue;  // shorter tiles: fewer 16-token blocks
                                const int ir = i;  // (index of this block's row coefficients)
                                vec4 o;
                                if constexpr (EB == EPI_GELU) {
                                    // ggml semantics: y = table[f16(x)], table[h] = f16(gelu_tanh(f32(h))); two columns per instruction
                                    // (v_pk_*_f32: IEEE results identical to the scalar ops of gemm.hip, so the kernels agree bit for bit)
                                    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                                    for (int e2 = 0; e2 < 2; ++e2) {
                                        f32x2 v = {DINO4_ACC(q, b, i, j)[2 * e2], DINO4_ACC(q, b, i, j)[2 * e2 + 1]};
                                        if constexpr (LNC && !(DINO_LN_ABL & 8)) {  // r (acc - mean s[n]) + c[n] as two fused multiply-adds per column (v_pk_fma_f32)
                                            const f32x2 d = __builtin_elementwise_fma(f32x2{lnn[LNC ? ir : 0], lnn[LNC ? ir : 0]}, f32x2{sb[2 * e2], sb[2 * e2 + 1]},
                                                                                      f32x2{bb[2 * e2], bb[2 * e2 + 1]});
                                            v = __builtin_elementwise_fma(f32x2{lnr[LNC ? ir : 0], lnr[LNC ? ir : 0]}, v, d);
                                        } else {
                                            v += f32x2{bb[2 * e2], bb[2 * e2 + 1]};
                                        }
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
                                        float v;
                                        if constexpr (LNC && !(DINO_LN_ABL & 8)) v = __builtin_fmaf(lnr[LNC ? ir : 0], DINO4_ACC(q, b, i, j)[e], __builtin_fmaf(lnn[LNC ? ir : 0], sb[e], bb[e]));
                                        else v = DINO4_ACC(q, b, i, j)[e] + bb[e];
                                        asm("" : "+v"(v));  // a real f32 sum: no "add, then round" fusion into v_fma_mixlo_f16
                                        if constexpr (EB == EPI_QKV) {
                                            float vq = v * qs;
                                            asm("" : "+v"(vq));
                                            o[e] = E::from_f32(vq);
                                        } else {
                                            // EPI_SWIGLU: W rows interleaved in 32-blocks: column half 0 holds x1[32 units], half 1 x2 of the same units
                                            float h2;
                                            if constexpr (LNC) h2 = __builtin_fmaf(lnr[LNC ? ir : 0], DINO4_ACC(q, 1, i, j)[e], __builtin_fmaf(lnn[LNC ? ir : 0], s2[e], b2[e]));
                                            else h2 = DINO4_ACC(q, 1, i, j)[e] + b2[e];
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
                    if constexpr (EB == EPI_SWIGLU) {
                        const int hid0 = (nw0 >> 6) * 32;  // 32 hidden units = 64 B per row: 4 lanes per row
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int row = it * 16 + (el >> 2), slot = el & 3;
                            const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                            const int m = mbase + q * 64 + row;
                            if (m < M && (NI == 8 || q * 64 + row < 16 * NI))
                                DINO4_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + hid0 + slot * 8), v);
                        }
                    } else {
#pragma unroll
                        for (int it = 0; it < 8; ++it) {
                            const int row = it * 8 + (el >> 3), slot = el & 7;
                            const u32x4 v = *(const u32x4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                            const int trw = q * 64 + row;  // token row within the wave's 16 NI
                            const int m = mbase + trw;
                            if (m < M && (NI == 8 || trw < 16 * NI))
                                DINO4_ST16((u32x4*)((T*)p.out + (size_t)m * p.ldo + nw0 + slot * 8), v);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                // (4-byte outputs are handled after this loop: their passes are software-pipelined across both column groups)
            }
            if constexpr (LNC) {
                if (ln_next) {
                    if (ln_gs > 12) ln_row_add<1>(lraw, lacc);
                    ln_row_finish(lacc, ln_inv_h, p.ln_eps, st_r[cg < LNH ? cg : 0], st_n[cg < LNH ? cg : 0]);
                }
            }
        }
        if constexpr (EPI == EPI_RESID || EPI == EPI_PLAIN_F32) {
            // 4-byte outputs: eight passes (column group cg, column half b, token half q) of 64 rows x 32 columns (128 B per row), each
            // transposed through the wave's LDS slice and moved as whole lines.  The residual-stream rows of a pass are requested THREE
            // passes ahead (the first three before anything else): with one wave per SIMD the read-modify-write burst is bound by how many
            // lines a CU has in flight, and the loads of pass k + 3 are issued before the stores of pass k, so that waiting for them does
            // not drain those stores (gfx950 retires loads and stores through one in-order counter).
            constexpr int PF = 3;  // passes of residual rows in flight (x 32 VGPRs: four spill next to the next tile's 64 fragment registers)
            float4 add[8][8];
            auto pass_cols = [&](int ps8, int& nb, int& q) {
                const int cg = ps8 >> 2, b = (ps8 >> 1) & 1;
                q = ps8 & 1;
                nb = n0 + wc * 128 + cg * 64 + b * 32 + (el & 7) * 4;
            };
            auto issue_loads = [&](int ps8) {
                if constexpr (EPI == EPI_RESID) {
                    int nb, q;
                    pass_cols(ps8, nb, q);
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        int m = mbase + q * 64 + it * 8 + (el >> 3);
                        m = m < M ? m : M - 1;
                        add[ps8][it] = *(const float4*)((const float*)p.out + (size_t)m * p.ldo + nb);
                    }
                }
            };
#pragma unroll
            for (int ps8 = 0; ps8 < PF; ++ps8) issue_loads(ps8);
#pragma unroll
            for (int ps8 = 0; ps8 < 8; ++ps8) {
                const int cg = ps8 >> 2, b = (ps8 >> 1) & 1, q = ps8 & 1;
                if (q * 4 >= NI) continue;
                asm volatile("" : "+v"(el));  // per pass: row pointers are recomputed, not kept live across the eight passes
                const int ncol = n0 + wc * 128 + cg * 64 + 4 * (el >> 4);
                const int nb = n0 + wc * 128 + cg * 64 + b * 32 + (el & 7) * 4;
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
                        const f32x4 a = DINO4_ACC(q, b, i, j);
                        *(float4*)(ep + row * 128 + slot * 16) =
                            make_float4((a[0] + b4.x) * ls.x, (a[1] + b4.y) * ls.y, (a[2] + b4.z) * ls.z, (a[3] + b4.w) * ls.w);
                    }
                }
                if (ps8 + PF < 8) issue_loads(ps8 + PF);  // ahead of this pass's stores
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 8 + (el >> 3), slot = el & 7;
                    float4 v = *(const float4*)(ep + row * 128 + ((slot ^ (row & 7)) << 4));
                    if constexpr (EPI == EPI_RESID)
                        v = make_float4(v.x + add[ps8][it].x, v.y + add[ps8][it].y, v.z + add[ps8][it].z, v.w + add[ps8][it].w);
                    const int m = mbase + q * 64 + row;
                    if (m < M && (NI == 8 || q * 64 + row < 16 * NI)) *(float4*)((float*)p.out + (size_t)m * p.ldo + nb) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (EPI == EPI_RESID_LN) {
            // The residual epilogue that also feeds the LayerNorm behind it (LN fold, kernels.h): x += ls (acc + bias) as above, plus the
            // next GEMM's operand xg = T(x gamma) and, per row, (sum, sum of squares) of x over this 64-column group.  Eight passes (column
            // group cg, token half q, 32-row half ih) of 32 rows x 64 columns -- 256 B of f32 per row, so that ONE pass holds a whole
            // statistics group of its rows: 16 lanes per row, reduced with row-local DPP in the fixed pairwise order of ln_leaf4 (4 -> 8 ->
            // 16 -> 32 -> 64 columns; the small-tile kernel produces the same bits), and xg leaves as whole 128-byte lines.  LDS slice image:
            // 32 rows x 256 B, 16-byte slot s of row r at s ^ (r & 15).  Residual rows are requested PF passes ahead, as above.
            constexpr int PF = (DINO_LN_ABL & 64) ? 3 : 2;
            float4 add[8][8];
            auto pass_of = [&](int ps8, int& cg, int& q, int& ih) {
                cg = ps8 >> 2;
                q = (ps8 >> 1) & 1;
                ih = ps8 & 1;
            };
            auto pass_live = [&](int ps8) {
                const int q = (ps8 >> 1) & 1, ih = ps8 & 1;
                return q * 4 + ih * 2 < NI;
            };
            auto issue_loads = [&](int ps8) {
                int cg, q, ih;
                pass_of(ps8, cg, q, ih);
                const int nb = n0 + wc * 128 + cg * 64 + (el & 15) * 4;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    int m = mbase + q * 64 + ih * 32 + it * 4 + (el >> 4);
                    m = m < M ? m : M - 1;
                    add[ps8][it] = *(const float4*)((const float*)p.out + (size_t)m * p.ldo + nb);
                }
            };
            // (NI = 6: passes with q = 1, ih = 1 do not exist; the pipeline simply skips them)
            int issued = 0;
#pragma unroll
            for (int ps8 = 0; ps8 < 8; ++ps8)
                if (pass_live(ps8) && issued < PF) {
                    issue_loads(ps8);
                    ++issued;
                }
#pragma unroll
            for (int ps8 = 0; ps8 < 8; ++ps8) {
                if (!pass_live(ps8)) continue;
                int cg, q, ih;
                pass_of(ps8, cg, q, ih);
                asm volatile("" : "+v"(el));
                const int ncol = n0 + wc * 128 + cg * 64 + 4 * (el >> 4);   // acc layout: + 32 b + 16 j
                const int nb = n0 + wc * 128 + cg * 64 + (el & 15) * 4;    // line layout: this lane's four columns of the 64
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 ls = *(const float4*)(p.aux + ncol + b * 32 + j * 16);
                        const float4 b4 = p.bias ? *(const float4*)(p.bias + ncol + b * 32 + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            if (q * 4 + ih * 2 + ii >= NI) continue;
                            const int row = ii * 16 + (el & 15);
                            const int slot = (b * 8 + j * 4 + (el >> 4)) ^ (row & 15);
                            const f32x4 a = DINO4_ACC(q, b, ih * 2 + ii, j);
                            *(float4*)(ep + row * 256 + slot * 16) =
                                make_float4((a[0] + b4.x) * ls.x, (a[1] + b4.y) * ls.y, (a[2] + b4.z) * ls.z, (a[3] + b4.w) * ls.w);
                        }
                    }
                const float4 gam = *(const float4*)(p.ln_gamma + nb);
                {  // the loads of the pass PF live passes ahead, before this pass's stores
                    int ahead = 0;
#pragma unroll
                    for (int nx = ps8 + 1; nx < 8; ++nx)
                        if (pass_live(nx) && ++ahead == PF) issue_loads(nx);
                }
                __builtin_amdgcn_wave_barrier();
                float ssum = 0.f, ssq = 0.f;  // this lane's row of the pass: lanes (el & 15) < 8 own row 4 (el & 15) + (el >> 4)
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 4 + (el >> 4), slot = el & 15;
                    float4 v = *(const float4*)(ep + row * 256 + ((slot ^ (row & 15)) << 4));
                    v = make_float4(v.x + add[ps8][it].x, v.y + add[ps8][it].y, v.z + add[ps8][it].z, v.w + add[ps8][it].w);
                    const int rl = q * 64 + ih * 32 + row;
                    const int m = mbase + rl;
                    const bool live = m < M && (NI == 8 || rl < 16 * NI);
                    if (live) *(float4*)((float*)p.out + (size_t)m * p.ldo + nb) = v;
                    vec4 og;
                    if constexpr (!(DINO_LN_ABL & 32)) {
                        float g0 = v.x * gam.x, g1 = v.y * gam.y, g2 = v.z * gam.z, g3 = v.w * gam.w;
                        asm("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));  // f32 products first, then the rounding
                        og[0] = E::from_f32(g0);
                        og[1] = E::from_f32(g1);
                        og[2] = E::from_f32(g2);
                        og[3] = E::from_f32(g3);
                    }
                    if constexpr (!(DINO_LN_ABL & 32)) {
                        if (live) *(vec4*)((T*)p.xg + (size_t)m * p.ldo + nb) = og;
                    }
                    if constexpr (DINO_LN_ABL & 16) continue;
                    float s4, q4;
                    ln_leaf4(v.x, v.y, v.z, v.w, s4, q4);
                    s4 += dpp_f32<0xB1>(s4);   // 8 columns  (quad_perm [1,0,3,2])
                    q4 += dpp_f32<0xB1>(q4);
                    s4 += dpp_f32<0x4E>(s4);   // 16         (quad_perm [2,3,0,1])
                    q4 += dpp_f32<0x4E>(q4);
                    s4 += dpp_f32<0x141>(s4);  // 32         (row_half_mirror)
                    q4 += dpp_f32<0x141>(q4);
                    s4 += dpp_f32<0x140>(s4);  // 64         (row_mirror)
                    q4 += dpp_f32<0x140>(q4);
                    if ((el & 15) == it) {
                        ssum = s4;
                        ssq = q4;
                    }
                }
                {
                    const int rl = q * 64 + ih * 32 + (el & 15) * 4 + (el >> 4);
                    const int m = mbase + rl;
                    if ((el & 15) < 8 && m < M && (NI == 8 || rl < 16 * NI))
                        *(float2*)(p.stats + ((size_t)m * p.ln_gs + ((n0 + wc * 128 + cg * 64) >> 6)) * 2) = make_float2(ssum, ssq);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#undef DINO4_ACC
        if constexpr (LNC) {
            if (has_next) {  // the next tile's K-tile 0 sits in buffer 0 since barrier B of the last K-tile
                static_for<8 + NI>([&Px, &Pw, &xa, &wa](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (q < 8) DINO4_DSR(Pw[q], wa[0], q * 2048);
                    else DINO4_DSR(Px[q - 8], xa[0], (q - 8) * 2048);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        DINO4_GP(1)
        DINO4_GP_TILE
    }  // persistent tile loop
    DINO4_GP_FLUSH
#undef DINO4_PIECE
}

template <typename T, int EPI, int NI>
__global__ __launch_bounds__(256) void gemm4_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DINO_CLK_BEGIN()
    gemm4_body<T, EPI, NI>(p, smem);
    DINO_CLK_END(g_clk4, p.clk_slot)
}

// One launch, two tile heights (gemm2_mixed_kernel's plan): every workgroup first walks its share of the 256-row tiles of `p` (whole
// rounds), then its share of the 192-row tiles of `q` (the remaining rows); no grid-wide barrier in between.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm4_mixed_kernel(GemmArgs p, GemmArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DINO_CLK_BEGIN()
#ifdef DINO_GEMM4_DEPHASE
    // XCDs of odd index take their 192-row tiles FIRST: a quarter of a tile time out of phase with the even ones for the rest of the launch, so
    // that the chip's epilogue bursts (stores, residual read-modify-write) come as two half-size ones.  All workgroups of an XCD stay in step
    // (they share operand panels in its L2).  Scheduling only: every tile is computed by the same code either way.
    if ((blockIdx.x & 7) & 1) {
        gemm4_body<T, EPI, 6>(q, smem);
        gemm4_body<T, EPI, 8>(p, smem);
    } else
#endif
    {
        gemm4_body<T, EPI, 8>(p, smem);
        gemm4_body<T, EPI, 6>(q, smem);
    }
    DINO_CLK_END(g_clk4, p.clk_slot)
}

constexpr size_t G4_LDS = 163840;

#ifdef DINO_GEMM4_PROF
static void gemm4_prof_dump(const char* what, int epi, int nblocks) {
    static unsigned long long h[256 * 4], z[256 * 4];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm4_prof), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm4_prof), z, sizeof z);
    double a[4] = {0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
        for (int i = 0; i < 4; ++i) a[i] += (double)h[b * 4 + i];
    const double t = a[2] > 0 ? a[2] : 1;
    fprintf(stderr, "gemm4_prof %s epi %d: per tile (wave 0): K loop %.0f cycles, epilogue %.0f cycles, %.2f us in all (%.2f tiles per workgroup) -> clock %.3f GHz\n", what, epi,
            a[0] / t, a[1] / t, a[3] / t * 0.01, a[2] / nblocks, (a[0] + a[1]) / (a[3] * 10.0));
}
#endif

template <typename T, int NI>
static hipError_t launch4_t(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int tiles = (a.N / 256) * ((a.M + 32 * NI - 1) / (32 * NI));
    const dim3 grid(tiles < 256 ? tiles : 256), block(256);
#define DINO_L4(E)                                                                \
    case E:                                                                       \
        hipLaunchKernelGGL((gemm4_kernel<T, E, NI>), grid, block, G4_LDS, st, a); \
        break;
    switch (epi) {
        DINO_L4(EPI_QKV)
        DINO_L4(EPI_RESID)
        DINO_L4(EPI_GELU)
        DINO_L4(EPI_SWIGLU)
        DINO_L4(EPI_PLAIN_F32)
        DINO_L4(EPI_RESID_LN)
        DINO_L4(EPI_QKV_LN)
        DINO_L4(EPI_GELU_LN)
        DINO_L4(EPI_SWIGLU_LN)
        default: return hipErrorInvalidValue;
    }
#undef DINO_L4
#ifdef DINO_GEMM4_PROF
    gemm4_prof_dump("plain launch", (int)epi, (int)grid.x);
#endif
    return hipGetLastError();
}

template <typename T>
static hipError_t launch4_mixed_t(Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    const dim3 grid(256), block(256);
#define DINO_LM4(E)                                                                    \
    case E:                                                                            \
        hipLaunchKernelGGL((gemm4_mixed_kernel<T, E>), grid, block, G4_LDS, st, a, b); \
        break;
    switch (epi) {
        DINO_LM4(EPI_QKV)
        DINO_LM4(EPI_RESID)
        DINO_LM4(EPI_GELU)
        DINO_LM4(EPI_SWIGLU)
        DINO_LM4(EPI_PLAIN_F32)
        DINO_LM4(EPI_RESID_LN)
        DINO_LM4(EPI_QKV_LN)
        DINO_LM4(EPI_GELU_LN)
        DINO_LM4(EPI_SWIGLU_LN)
        default: return hipErrorInvalidValue;
    }
#undef DINO_LM4
#ifdef DINO_GEMM4_PROF
    gemm4_prof_dump("mixed launch", (int)epi, 256);
#endif
    return hipGetLastError();
}

// both require N % 256 == 0, K / 64 even and >= 4, and an epilogue other than EPI_PATCH (gemm4_ok)
bool gemm4_ok(Epilogue epi, const GemmArgs& a) {
    return epi != EPI_PATCH && a.N % 256 == 0 && a.K % 128 == 0 && a.K >= 256;
}
hipError_t launch_gemm4(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t st) {
    return dt == DT_F16 ? launch4_t<_Float16, 8>(epi, a, st) : launch4_t<__bf16, 8>(epi, a, st);
}
// 256-row tiles for `a` (whole rounds), then 192-row tiles for `b`, in one launch
hipError_t launch_gemm4_mixed(DType dt, Epilogue epi, const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
    return dt == DT_F16 ? launch4_mixed_t<_Float16>(epi, a, b, st) : launch4_mixed_t<__bf16>(epi, a, b, st);
}

// Short tiles (64 / 96 / 128 rows), one per workgroup: for launches whose 256-row tiles would leave most CUs idle (batch 1: M = 1 374 ->
// QKV 15 panels of 96 rows x 12 column tiles = 180 workgroups instead of 132 of 128 rows; FFN-in 240 instead of 176).  The 2-byte
// epilogues (the plain f32 ones go to the small-tile kernel or to gemm2.hip's 128-row tiles at these sizes) and EPI_RESID_LN, which has no
// gemm2.hip form (batch 4: without it the LN fold fell back to the small-tile kernel where the default path runs 128-row tiles: - 15 %).
// Same K order, same bits.
template <typename T, int NI>
static hipError_t launch4_short_t(Epilogue epi, const GemmArgs& a, hipStream_t st) {
    const int tiles = (a.N / 256) * ((a.M + 32 * NI - 1) / (32 * NI));
    if (tiles > 256) return hipErrorInvalidValue;  // (one tile per workgroup, one workgroup per CU)
    const dim3 grid(tiles), block(256);
    switch (epi) {
        case EPI_QKV: hipLaunchKernelGGL((gemm4_kernel<T, EPI_QKV, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_GELU: hipLaunchKernelGGL((gemm4_kernel<T, EPI_GELU, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm4_kernel<T, EPI_SWIGLU, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_QKV_LN: hipLaunchKernelGGL((gemm4_kernel<T, EPI_QKV_LN, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_GELU_LN: hipLaunchKernelGGL((gemm4_kernel<T, EPI_GELU_LN, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_SWIGLU_LN: hipLaunchKernelGGL((gemm4_kernel<T, EPI_SWIGLU_LN, NI>), grid, block, G4_LDS, st, a); break;
        case EPI_RESID_LN: hipLaunchKernelGGL((gemm4_kernel<T, EPI_RESID_LN, NI>), grid, block, G4_LDS, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// rows per tile = 32 ni, ni in {2, 3, 4}
hipError_t launch_gemm4_short(DType dt, Epilogue epi, const GemmArgs& a, int ni, hipStream_t st) {
    if (dt == DT_F16) return ni == 2 ? launch4_short_t<_Float16, 2>(epi, a, st) : ni == 3 ? launch4_short_t<_Float16, 3>(epi, a, st) : launch4_short_t<_Float16, 4>(epi, a, st);
    return ni == 2 ? launch4_short_t<__bf16, 2>(epi, a, st) : ni == 3 ? launch4_short_t<__bf16, 3>(epi, a, st) : launch4_short_t<__bf16, 4>(epi, a, st);
}

template <typename T, int NI>
static hipError_t attr4_short_t() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_QKV, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_GELU, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_SWIGLU, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_QKV_LN, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_GELU_LN, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_SWIGLU_LN, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, EPI_RESID_LN, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    return e;
}

template <typename T>
static hipError_t attr4_t() {
    hipError_t e = hipSuccess;
#define DINO_A4(E)                                                                                                                        \
    if (e == hipSuccess)                                                                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<T, E, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS); \
    if (e == hipSuccess)                                                                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_mixed_kernel<T, E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G4_LDS);
    DINO_A4(EPI_QKV)
    DINO_A4(EPI_RESID)
    DINO_A4(EPI_GELU)
    DINO_A4(EPI_SWIGLU)
    DINO_A4(EPI_PLAIN_F32)
    DINO_A4(EPI_RESID_LN)
    DINO_A4(EPI_QKV_LN)
    DINO_A4(EPI_GELU_LN)
    DINO_A4(EPI_SWIGLU_LN)
#undef DINO_A4
    return e;
}

hipError_t gemm4_init() {
    hipError_t e = attr4_t<_Float16>();
    if (e == hipSuccess) e = attr4_t<__bf16>();
    if (e == hipSuccess) e = attr4_short_t<_Float16, 2>();
    if (e == hipSuccess) e = attr4_short_t<_Float16, 3>();
    if (e == hipSuccess) e = attr4_short_t<_Float16, 4>();
    if (e == hipSuccess) e = attr4_short_t<__bf16, 2>();
    if (e == hipSuccess) e = attr4_short_t<__bf16, 3>();
    if (e == hipSuccess) e = attr4_short_t<__bf16, 4>();
    return e;
}

}  // namespace dinov2
