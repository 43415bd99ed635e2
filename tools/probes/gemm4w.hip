// gemm4w.hip -- round-4 structural probe: the 256 x 256 x 64 tile with FOUR waves (one per SIMD, 512 registers each: 256 accumulators
// in AGPRs + fragments in VGPRs), every wave a 128 x 128 output block, and ONE hand-ordered instruction stream per K-tile instead of
// barrier-separated MEM / MMA sections shared by two waves of a SIMD (csrc/gemm2.hip).
//
// Why (profiles/r03_vendor_gemm_yardstick.md): the vendor's assembly kernel of the same macro tile and MFMA shape is
// 14 % ahead at 4 096^3 (1 479 vs 1 295 TFLOP/s).  Its code object says how: 256 threads, 249 VGPRs + the accumulator file, 128 MFMAs per
// K-tile per wave issued back to back with one or two memory / scalar instructions in each MFMA's shadow, fragments of the next k-step
// read under the current one's MFMAs, LDS-DMA staging counted with vmcnt(13), three barriers per K-tile.  This probe rebuilds that
// STRUCTURE from scratch with this library's own layout (XOR-swizzled 128-byte rows, global_load_lds pieces of 8 rows, operand swap so
// that a lane owns four consecutive output columns) and its own schedule (two barriers per K-tile):
//
//   LDS (128 KiB): X buffers 0 / 1 at 0 / 32 KiB, W buffers 0 / 1 at 64 / 96 KiB; a buffer = 256 rows x 128 B (one K-tile of one operand).
//   Wave (wr, wc) = (wid >> 1, wid & 1) owns tokens [128 wr, +128) x columns [128 wc, +128): 8 x 8 blocks of 16 x 16, acc[i][j] in AGPRs.
//   Fragment registers: P = k-step 0 (8 X + 8 W fragments of 4 VGPRs), Q = k-step 1.  MFMA order in a k-step: j (column block) outer, i inner.
//   K-tile t in buffer b = t & 1, MFMA index m = 0 .. 127:
//     m   0 .. 15   k-step 0 (P); one ds_read_b128 of Q(t) per MFMA (W fragments first: their registers were free longest)
//     m  31         s_waitcnt lgkmcnt(0); s_barrier                 [A] every wave has read all of buffer b
//     m  32 .. 92   one global_load_lds piece of K-tile t + 2 -> buffer b every 4th MFMA (16 pieces per wave)
//     m  64 ..      k-step 1 (Q)
//     m  95         s_waitcnt vmcnt(16); s_barrier                  [B] K-tile t + 1 (staged one K-tile ago) is in buffer b ^ 1 for everyone
//     m  96 .. 111  one ds_read_b128 of P(t + 1) per MFMA
//     end           s_waitcnt lgkmcnt(0)
//   A staged K-tile has >= one whole K-tile time (>= 2 048 matrix-pipe cycles) to land; the pieces of the NEXT output tile's K-tiles 0 / 1
//   are staged under the last two K-tiles of the current one, so the epilogue never waits for HBM.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gemm4w.hip -o /tmp/gemm4w && /tmp/gemm4w
//   -DVARIANT=<bits>: timing only (wrong results): 1 no staging, 2 no fragment reads, 4 no barriers
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

#ifndef VARIANT
#define VARIANT 0
#endif

template <int... Is, class F>
static __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
static __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#define MFMA_ACC(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(W), "v"(X))
#define MFMA_ZERO(ACC, W, X) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(ACC) : "v"(W), "v"(X))
#define DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
// LDS-DMA: 64 lanes x 16 B -> LDS [M0 .. M0 + 1024); M0 is written in the same statement that reads it (the compiler reserves M0 and does
// not preserve it around an asm statement; nothing else in this kernel uses M0)
#define GLDS(VOFF, SBASE, LDSADDR) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")

// Schedule constants (MFMA index m = 0 .. 127 within a K-tile):
//   Q(t) reads at m = 0, QSTRIDE, 2 QSTRIDE ... (16 of them); barrier A after MFMA BAR_A; pieces one per 8 MFMAs at m = 8 k + STAG * W
//   (W = wave id: the four waves of the workgroup present their pieces to the CU's one texture path STAG MFMAs apart instead of all
//   at once); slots after A carry pieces 0 .. of K-tile t + 2, slots before A the last pieces of K-tile t + 1 (its first ones went out
//   after A of the previous K-tile: the staging of a K-tile spans exactly one K-tile time); barrier B after MFMA BAR_B with
//   vmcnt(pieces issued since A); P(t + 1) reads one per MFMA from BAR_B + 1.
#ifndef QSTRIDE
#define QSTRIDE 2
#endif
#ifndef STAG
#define STAG 2
#endif
#ifndef PA
#define PA 39
#endif
#ifndef PB
#define PB 103
#endif

// -DPROF: shader-clock (s_memtime) and 100 MHz wall-clock (s_memrealtime) stamps around the middle K-tiles of every tile, summed per
// workgroup by wave 0: cycles per K-tile and the clock the CU actually ran at.
__device__ unsigned long long g_prof[256 * 4];
template <int WV>
static __device__ __forceinline__ void gemm4w_body(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ C,
                                                   int M, int N, int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int wid = WV;
    constexpr int wr = wid >> 1, wc = wid & 1;
    const int ntn = N / 256, ntm = M / 256, ntiles = ntn * ntm;
    const int nk = K / 64;  // even, >= 4
    const unsigned lda2 = (unsigned)K * 2u;

    // XCD-aware persistent tile walk (that of csrc/gemm2.hip): block b sits on XCD b % 8, every XCD walks a contiguous chunk of the tile order
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    const int nb_x = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int chunk0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int chunkn = tq + (xcd < tr ? 1 : 0);
    constexpr int GM = 8;
    auto tile_mn = [&](int lid, int& m0, int& n0) {
        const int g = lid / (GM * ntn), r = lid - g * (GM * ntn);
        const int gm = ntm - g * GM < GM ? ntm - g * GM : GM;
        const int n = r / gm, mi = r - n * gm;
        m0 = (g * GM + mi) * 256;
        n0 = n * 256;
    };

    // ---- staging: wave w moves rows [64 w, +64) of both operands' K-tile: 8 + 8 pieces of 8 rows x 128 B (lane -> row lane >> 3, 16-byte
    // chunk lane & 7, XOR-swizzled on the SOURCE side).  Piece pc = 0 .. 15: operand pc & 1 (X, W), row block pc >> 1.
    unsigned so[16];  // per-piece byte offsets from A / W (+ 128 kt through the scalar base)
    auto piece_off = [&](int pc, int m0, int n0) -> unsigned {
        const int r = 64 * wid + 8 * (pc >> 1) + (lane >> 3);
        const int ch = (lane & 7) ^ ((r >> 1) & 7);
        return (unsigned)(((pc & 1) ? n0 : m0) + r) * lda2 + ch * 16;
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_AS char*)smem;
    const unsigned ldsp = lds0 + wid * 8192u;  // this wave's 8 KiB of X buffer 0 (W: + 65536, buffer 1: + 32768, piece row block: + 1024 each)

    // ---- fragment addresses: lane -> row lane & 15 of a 16-row block, 16-byte chunk (4 ks + (lane >> 4)) ^ ((row >> 1) & 7)
    const int fr = lane & 15, kq = lane >> 4, sw = (fr >> 1) & 7;
    unsigned xa[2], wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned ch = (unsigned)(((ks * 4 + kq) ^ sw) << 4);
        xa[ks] = lds0 + (unsigned)((wr * 128 + fr) * 128) + ch;
        wa[ks] = lds0 + 65536u + (unsigned)((wc * 128 + fr) * 128) + ch;
    }

    f32x4 acc[8][8];
    u32x4 Px[8], Pw[8], Qx[8], Qw[8];
#ifdef CPK
    // -DCPK=<n>: n GELU-shaped filler chains per K-tile (13 one- or two-instruction stages each, two elements per chain, every stage pinned
    // into its own MFMA shadow by an empty volatile asm on its result): what a DEFERRED epilogue's arithmetic costs the K loop that carries it.
    // 8 chains per K-tile x 16 K-tiles = the 256 elements per lane of a whole 256 x 256 tile.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f32x2 fa[3], fb[3], fc[3];
    unsigned fsink = 0;
    f32x2 fseed = {(float)(tid & 31) * 0.05f - 0.8f, (float)(tid & 15) * 0.07f - 0.5f};
#endif
    if (VARIANT & 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Px[i] = Pw[i] = Qx[i] = Qw[i] = u32x4{(unsigned)tid, 1u, 2u, 3u};
    }

    const char* const Ab = (const char*)A;
    const char* const Wb = (const char*)W;

    // one piece: pc literal, kt / b wave-uniform
#define PIECE(PC, KT, BUF)                                                                                            \
    if (!(VARIANT & 1)) {                                                                                             \
        if constexpr (((PC) & 1) == 0) GLDS(so[PC], Ab + (size_t)(KT) * 128, ldsp + (BUF) * 32768u + ((PC) >> 1) * 1024u); \
        else GLDS(so[PC], Wb + (size_t)(KT) * 128, ldsp + 65536u + (BUF) * 32768u + ((PC) >> 1) * 1024u);             \
    }

    // One K-tile.  B = its buffer; FIRST: first K-tile of an output tile (accumulators start from zero).  Staging under it:
    //   slots before barrier A: the last NPRE pieces of the K-tile that goes into buffer b ^ 1 (kt_pre; its first pieces went out after A of
    //   the previous K-tile), if pre_on; at A: `at_a()` (switches the piece offsets to the next output tile where the staging crosses over);
    //   slots after A: the first pieces of the K-tile that goes into THIS buffer (kt_post), if post_on.  do_readp: read P of the next K-tile.
    static constexpr int S0 = STAG * WV;                    // this wave's first slot
    static constexpr int NPRE = (PA - S0) / 8 + 1;          // slots at m = S0 + 8 k <= PA
    static constexpr int NPOST = 16 - NPRE;
    static constexpr int NB = (PB - (S0 + 8 * NPRE)) / 8 + 1;  // slots in (PA, PB]: pieces issued between barrier A and barrier B
    static_assert(S0 < 8 && NPRE >= 1 && NPRE < 16 && NB >= 1 && NB <= NPOST, "schedule constants");
    #ifdef CPK
#define FCAP , &fa, &fb, &fc, &fsink, &fseed
#else
#define FCAP
#endif
    auto ktile = [&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsp, Ab, Wb FCAP](auto bc, auto firstc, int kt_pre, bool pre_on, auto&& at_a, int kt_post,
                                                                         bool post_on, bool do_readp) {
        static_for<128>([&acc, &Px, &Pw, &Qx, &Qw, &xa, &wa, &so, &ldsp, Ab, Wb FCAP, &kt_pre, &pre_on, &at_a, &kt_post, &post_on, &do_readp, bc,
                         firstc](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int b = decltype(bc)::value;
            constexpr bool FIRST = decltype(firstc)::value;
            constexpr int ks = m >> 6, j = (m >> 3) & 7, i = m & 7;
            if constexpr (ks == 0) {
                if constexpr (FIRST) MFMA_ZERO(acc[i][j], Pw[j], Px[i]);
                else MFMA_ACC(acc[i][j], Pw[j], Px[i]);
            } else {
                MFMA_ACC(acc[i][j], Qw[j], Qx[i]);
            }
#ifdef CPK
            {
                constexpr int SPACING = 104 / CPK;  // chain c starts at m = 4 + SPACING c and runs 13 consecutive slots
                static_for<CPK>([&fa, &fb, &fc, &fsink, &fseed, mc](auto cc) {
                    constexpr int c = decltype(cc)::value, m = decltype(mc)::value;
                    constexpr int st = m - 4 - SPACING * c, r = c % 3;
                    if constexpr (st == 0) { fa[r] = fseed + f32x2{0.01f * c, -0.02f * c}; asm volatile("" : "+v"(fa[r])); }
                    else if constexpr (st == 1) { f16x2 h = __builtin_convertvector(fa[r], f16x2); fa[r] = __builtin_convertvector(h, f32x2); asm volatile("" : "+v"(fa[r])); }
                    else if constexpr (st == 2) { fb[r] = fa[r] * fa[r]; asm volatile("" : "+v"(fb[r])); }
                    else if constexpr (st == 3) { fb[r] = __builtin_elementwise_fma(fb[r], f32x2{-0.1029432397f, -0.1029432397f}, f32x2{-2.302208199f, -2.302208199f}); asm volatile("" : "+v"(fb[r])); }
                    else if constexpr (st == 4) { fb[r] = fa[r] * fb[r]; asm volatile("" : "+v"(fb[r])); }
                    else if constexpr (st == 5) { fc[r][0] = __builtin_amdgcn_exp2f(fb[r][0]); asm volatile("" : "+v"(fc[r])); }
                    else if constexpr (st == 6) { fc[r][1] = __builtin_amdgcn_exp2f(fb[r][1]); asm volatile("" : "+v"(fc[r])); }
                    else if constexpr (st == 7) { fc[r] = fc[r] + f32x2{1.0f, 1.0f}; asm volatile("" : "+v"(fc[r])); }
                    else if constexpr (st == 8) { fc[r][0] = __builtin_amdgcn_rcpf(fc[r][0]); asm volatile("" : "+v"(fc[r])); }
                    else if constexpr (st == 9) { fc[r][1] = __builtin_amdgcn_rcpf(fc[r][1]); asm volatile("" : "+v"(fc[r])); }
                    else if constexpr (st == 10) { fa[r] = fa[r] * fc[r]; asm volatile("" : "+v"(fa[r])); }
                    else if constexpr (st == 11) { f16x2 h = __builtin_convertvector(fa[r], f16x2); fsink ^= __builtin_bit_cast(unsigned, h); asm volatile("" : "+v"(fsink)); }
                });
            }
#endif
            // Q(t): k-step 1 of this K-tile (W fragments first: their registers have been free longest)
            if constexpr (m % QSTRIDE == 0 && m / QSTRIDE < 16) {
                constexpr int q = m / QSTRIDE;
                if (!(VARIANT & 2)) {
                    if constexpr (q < 8) DSR(Qw[q], wa[1], q * 2048 + b * 32768);
                    else DSR(Qx[q - 8], xa[1], (q - 8) * 2048 + b * 32768);
                }
            }
            static_assert(15 * QSTRIDE < PA, "Q reads are issued before barrier A");
            // staging slots
            if constexpr (m >= S0 && (m - S0) % 8 == 0) {
                constexpr int k = (m - S0) / 8;
                if constexpr (k < NPRE) {
                    if (pre_on) PIECE(NPOST + k, kt_pre, b ^ 1)
                } else {
                    if (post_on) PIECE(k - NPRE, kt_post, b)
                }
            }
            if constexpr (m == PA) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(VARIANT & 4)) asm volatile("s_barrier" ::: "memory");
                at_a();
            }
            if constexpr (m == PB) {
                if (post_on) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(VARIANT & 4)) asm volatile("s_barrier" ::: "memory");
            }
            // P(t + 1): k-step 0 of the next K-tile, from the other buffer
            if constexpr (m > PB && m <= PB + 16) {
                constexpr int q = m - PB - 1;
                if (!(VARIANT & 2) && do_readp) {
                    if constexpr (q < 8) DSR(Pw[q], wa[0], q * 2048 + (b ^ 1) * 32768);
                    else DSR(Px[q - 8], xa[0], (q - 8) * 2048 + (b ^ 1) * 32768);
                }
            }
            static_assert(PB + 16 <= 127, "P reads fit behind barrier B");
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using TT = std::integral_constant<bool, true>;
    using TF = std::integral_constant<bool, false>;
    auto nop = [] {};

    if (bidx < chunkn) {
        int pm0, pn0;
        tile_mn(chunk0 + bidx, pm0, pn0);
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) so[pc] = piece_off(pc, pm0, pn0);
        // K-tile 0 whole, K-tile 1's first NPOST pieces (its last NPRE go out before barrier A of K-tile 0, like everywhere else)
        static_for<16>([&so, &ldsp, Ab, Wb](auto pcc) {
            constexpr int pc = decltype(pcc)::value;
            PIECE(pc, 0, 0)
        });
        static_for<NPOST>([&so, &ldsp, Ab, Wb](auto pcc) {
            constexpr int pc = decltype(pcc)::value;
            PIECE(pc, 1, 1)
        });
        if (!(VARIANT & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPOST) : "memory");
        asm volatile("s_barrier" ::: "memory");
        static_for<16>([&Px, &Pw, &xa, &wa](auto qc) {
            constexpr int q = decltype(qc)::value;
            if (!(VARIANT & 2)) {
                if constexpr (q < 8) DSR(Pw[q], wa[0], q * 2048);
                else DSR(Px[q - 8], xa[0], (q - 8) * 2048);
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int tix = bidx; tix < chunkn; tix += nb_x) {
        int m0, n0;
        tile_mn(chunk0 + tix, m0, n0);
        const bool has_next = tix + nb_x < chunkn;

        // K-tiles 0, 1 (accumulators from zero), the middle, and the last two, under which the staging crosses over to the next output tile
        ktile(T0{}, TT{}, 1, true, nop, 2, true, true);
        ktile(T1{}, TF{}, 2, true, nop, 3, true, true);
#ifdef PROF
        const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int t = 2; t < nk - 2; t += 2) {
            ktile(T0{}, TF{}, t + 1, true, nop, t + 2, true, true);
            ktile(T1{}, TF{}, t + 2, true, nop, t + 3, true, true);
        }
#ifdef PROF
        if (WV == 0 && lane == 0) {
            g_prof[blockIdx.x * 4 + 0] += __builtin_readcyclecounter() - pc0;
            g_prof[blockIdx.x * 4 + 1] += __builtin_amdgcn_s_memrealtime() - pr0;
            g_prof[blockIdx.x * 4 + 2] += (unsigned long long)(nk - 4);
        }
#endif
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_mn(chunk0 + tix + nb_x, nm0, nn0);
        ktile(T0{}, TF{}, nk - 1, true,
              [&] {
                  if (has_next) {
#pragma unroll
                      for (int pc = 0; pc < 16; ++pc) so[pc] = piece_off(pc, nm0, nn0);
                  }
              },
              0, has_next, true);
        ktile(T1{}, TF{}, 0, has_next, nop, 1, has_next, has_next);

        // ---- epilogue: plain f32 stores, acc[i][j][e] = C[m0 + 128 wr + 16 i + (lane & 15)][n0 + 128 wc + 16 j + 4 (lane >> 4) + e]
        static_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int m = m0 + wr * 128 + i * 16 + fr;
                const int n = n0 + wc * 128 + j * 16 + 4 * kq;
                *(f32x4*)(C + (size_t)m * N + n) = acc[i][j];
            });
        });
    }
#ifdef CPK
    if (fsink == 0x12345u) C[0] = 1.0f;  // (keeps the filler chains alive)
#endif
#undef PIECE
}

__global__ __launch_bounds__(256) void gemm4w(const _Float16* __restrict__ A, const _Float16* __restrict__ W, float* __restrict__ C, int M,
                                              int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // four instruction streams, one per wave: they differ only in WHERE their staging slots sit (STAG MFMAs apart)
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: gemm4w_body<0>(A, W, C, M, N, K, smem); break;
        case 1: gemm4w_body<1>(A, W, C, M, N, K, smem); break;
        case 2: gemm4w_body<2>(A, W, C, M, N, K, smem); break;
        default: gemm4w_body<3>(A, W, C, M, N, K, smem); break;
    }
}

// reference: one thread per sampled output
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
    _Float16 *A, *W;
    float* C;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4);
    fill<<<1024, 256>>>(A, (size_t)M * K, 1u); fill<<<1024, 256>>>(W, (size_t)N * K, 2u);
    hipMemset(C, 0xff, (size_t)M * N * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int tiles = (M / 256) * (N / 256);
    const dim3 grid(tiles < 256 ? tiles : 256), block(256);
    hipLaunchKernelGGL(gemm4w, grid, block, 131072, 0, A, W, C, M, N, K);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
    // refcheck on 8192 sampled outputs (transposes / wrong blocks / races show up as O(1) errors)
    const int ns_ = 8192;
    std::vector<int> hm(ns_), hn(ns_);
    for (int i = 0; i < ns_; ++i) { hm[i] = (int)(((unsigned)rand() * 2654435761u) % (unsigned)M); hn[i] = (int)(((unsigned)rand() * 40503u + 17) % (unsigned)N); }
    int *dm, *dn; float* dr;
    hipMalloc(&dm, ns_ * 4); hipMalloc(&dn, ns_ * 4); hipMalloc(&dr, ns_ * 4);
    hipMemcpy(dm, hm.data(), ns_ * 4, hipMemcpyHostToDevice); hipMemcpy(dn, hn.data(), ns_ * 4, hipMemcpyHostToDevice);
    ref_kernel<<<(ns_ + 255) / 256, 256>>>(A, W, dm, dn, dr, K, ns_);
    std::vector<float> href(ns_);
    hipMemcpy(href.data(), dr, ns_ * 4, hipMemcpyDeviceToHost);
    std::vector<float> hc((size_t)M * N > (size_t)64 << 20 ? 0 : (size_t)M * N);
    double worst = 0;
    if (!hc.empty()) hipMemcpy(hc.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < ns_; ++i) {
        float v;
        if (!hc.empty()) v = hc[(size_t)hm[i] * N + hn[i]];
        else hipMemcpy(&v, C + (size_t)hm[i] * N + hn[i], 4, hipMemcpyDeviceToHost);
        const double d = std::fabs((double)v - href[i]);
        worst = std::fmax(worst, d == d ? d : 1e30);
    }
    // race screen: 20 more launches must reproduce the first result bit for bit (small shapes only)
    int diffs = 0;
    if (!hc.empty() && VARIANT == 0) {
        std::vector<float> h2(hc.size());
        for (int r = 0; r < 20; ++r) {
            hipLaunchKernelGGL(gemm4w, grid, block, 131072, 0, A, W, C, M, N, K);
            hipMemcpy(h2.data(), C, h2.size() * 4, hipMemcpyDeviceToHost);
            if (memcmp(h2.data(), hc.data(), h2.size() * 4) != 0) ++diffs;
        }
    }
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(gemm4w, grid, block, 131072, 0, A, W, C, M, N, K);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm4w, grid, block, 131072, 0, A, W, C, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= iters;
        best = std::fmin(best, ms); sum += ms;
    }
    const float ms = sum / 5;
#ifdef PROF
    {
        static unsigned long long h[1024], z[1024];
        hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
        hipLaunchKernelGGL(gemm4w, grid, block, 131072, 0, A, W, C, M, N, K);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof h);
        double cyc = 0, rt = 0, kt = 0;
        for (int b = 0; b < 256; ++b) { cyc += h[b * 4]; rt += h[b * 4 + 1]; kt += h[b * 4 + 2]; }
        if (kt > 0) printf("   prof: %.0f shader cycles per K-tile (MFMA floor 2048 = %.1f %% duty), %.3f us per K-tile, effective clock %.3f GHz\n", cyc / kt,
                           204800.0 / (cyc / kt), rt / kt * 0.01, cyc / rt / 10.0);
    }
#endif
    printf("gemm4w cpk%d v%d A%d B%d Q%d S%d M=%d N=%d K=%d: %.4f ms (best %.4f)  %.1f TFLOP/s  (%.3f us per K-tile-round)  refcheck max|d| = %.3g %s  reruns differing: %d\n",
           
#ifdef CPK
           CPK,
#else
           0,
#endif
           VARIANT, PA, PB, QSTRIDE, STAG, M, N, K, ms, best, 2.0 * M * N * K / ms / 1e9,
           ms * 1e3 / ((double)(K / 64) * ((tiles + 255) / 256)), worst, worst < 2e-2 * std::sqrt((double)K / 1024) ? "OK" : "MISMATCH", diffs);
    hipFree(A); hipFree(W); hipFree(C); hipFree(dm); hipFree(dn); hipFree(dr);
}

int main() {
    run(256, 256, 256, 1);
    run(512, 768, 1024, 10);
    run(2048, 2048, 512, 10);   // 64 tiles, 8 K-tiles
    run(4096, 8192, 256, 10);   // 512 tiles: two per workgroup, 4 K-tiles each (the tile hand-over dominates)
    run(4096, 4096, 4096, 50);
    run(8192, 8192, 8192, 10);
    run(43776, 4096, 1024, 50);  // FFN-in shape rounded down to whole 256-row tiles
    run(43776, 1024, 4096, 50);  // FFN-out
    return 0;
}
