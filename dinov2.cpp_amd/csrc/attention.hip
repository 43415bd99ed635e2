// attention.hip -- fused multi-head self-attention (flash-style, never materialises the T x T scores) for gfx950.
//
// Replaces the whole non-flash branch of `attn` in the reference, /root/reference/dinov2.cpp:479-543:
// the five ggml_cont(permute) copies, KQ = mul_mat(K, Q) (121 MB of f32 scores per layer at ViT-L/518),
// soft_max_ext(scale) and KQV = mul_mat(V, P).  (The reference's opt-in `-fa` path, :499-525, pads keys to a
// multiple of 32 WITHOUT masking them and is documented as less accurate; it is not the parity target.  Here the
// key tail is masked to -inf.)
//
// Numerics: q (pre-scaled by log2(e)/sqrt(64) in the QKV epilogue so that the softmax runs on exp2), k, v and the un-normalised
// probabilities are MFMA inputs in the compute dtype (f16/bf16); scores, running max/sum and the output
// accumulator are f32.  ggml keeps this block in f32 end to end -- the rounding is the documented tolerance source.
//
// Mapping (wave64, MFMA 32x32x16):
//   * one workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries for the whole kernel.
//   * per 64-key tile and wave:  S^T = K Q^T  (A = K rows from LDS, B = Q^T held in registers) so that lane l
//     holds, for ITS query q = l & 31, the scores of 32 of the 64 keys: softmax statistics are lane-local
//     (one cross-half shuffle per tile).
//   * O^T = V^T P^T  (A = V^T via ds_read_b64_tr_b16 from the row-major V tile, B = P^T straight from the score
//     registers): the sum over keys is order-free, so the 8 k-slots of a lane are simply the 8 keys its score
//     registers already hold, and V^T is gathered with the same key permutation -- no P exchange between lanes.
//     O^T keeps q on the lane axis, so the online-softmax rescale and the final 1/l are lane-local too.
//   * K and V tiles are staged HBM -> LDS by global_load_lds_dwordx4, double buffered, 128-byte rows with the same
//     16-byte-chunk XOR swizzle as the GEMM.
//   * 1-D grid, XCD-aware: all query blocks of an (image, head) run on one XCD, so its K/V is fetched from HBM once.
//
// Two kernels with this mapping and identical arithmetic (bit-for-bit equal outputs, tested): attention_kernel (121 VGPRs,
// four workgroups per CU overlap each other -- the throughput kernel) and attention2_kernel (software-pipelined inside
// the wave, 2 waves per SIMD -- wins when there are too few workgroups to overlap, i.e. small batches).  launch_attention
// picks by workgroup count.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "device_types.h"
#include "kernels.h"

namespace dinov2 {

// Clock probe slot of this file's kernels (device_types.h, "clock probe")
__device__ unsigned long long g_clk_att[CLK_SLOTS * 4];
hipError_t attention_clock_probe_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_att), sizeof(unsigned long long) * CLK_SLOTS * 4);
}

// -DDINO_ATT_PROF: per-phase s_memtime sums (tuning builds only; `make variant V=prof VFLAGS=-DDINO_ATT_PROF`)
#ifdef DINO_ATT_PROF
__device__ unsigned long long g_att_prof[32768 * 8];
#define DINO_TS(i) { const unsigned long long t__ = __builtin_readcyclecounter(); prof_acc[i] += t__ - prof_t; prof_t = t__; }
#define DINO_TS_INIT unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long prof_t = __builtin_readcyclecounter();
#define DINO_TS_FLUSH if (lane == 0) { const int w__ = (blockIdx.x * 4 + wid) & 32767; for (int i__ = 0; i__ < 8; ++i__) g_att_prof[w__ * 8 + i__] = prof_acc[i__]; }
#else
#define DINO_TS(i)
#define DINO_TS_INIT
#define DINO_TS_FLUSH
#endif

// max of three without the v_max(x, x) NaN-quieting moves hipcc adds around fmaxf in IEEE mode (scores are finite or -inf)
static __device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max over the two 32-lane halves (lanes l and l + 32 hold the same query): v_permlane32_swap hands every lane both halves'
// values without the LDS round trip of ds_bpermute
static __device__ __forceinline__ float max_halves(float m) {
    // one asm block, padded on both sides: hipcc's hazard recogniser does not look inside the asm statements that produce /
    // consume these registers, and v_permlane32_swap needs wait states after a VALU write and before a VALU read.
    // After the swap: a = lower half's value in every lane, b = upper half's.
    float a = m, b = m;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
    return max3f(a, b, b);
}
// First MFMA of a score chain: accumulator input C (the -m_run tuple) and result D in DIFFERENT registers.  Written as asm
// because hipcc, given the builtin, copies C into D's registers first (8 v_mov_b64 per key tile and chain pair).  The early
// clobber keeps D off the inputs; 16-pass MFMA results are consumed only by further MFMAs (hardware-interlocked) or after
// mfma_settle().
template <typename V8>
static __device__ __forceinline__ f32x16 mfma32_c(V8 a, V8 b, const f32x16& c) {
    f32x16 d;
    if constexpr (sizeof(((V8*)nullptr)[0][0]) == 2 && __is_same(V8, f16x8))
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Score-chain MFMAs of the 512-register kernel, as asm with the register CLASSES spelled out: a kernel that may need AGPRs makes
// hipcc select the accumulator form of every MFMA builtin (D / C in AGPRs), so the scores would come out in AGPRs and every one of
// them would be moved to a VGPR again for the softmax (v_accvgpr_read: ~230 extra VALU operations per key tile, measured in the
// ISA).  Here D / C are architectural VGPRs (D and C share one class bit, so the -m_run tuple is a VGPR too), while the K fragment
// (A) and Q^T (B) sit in AGPRs, which only the matrix core (and, for the fragments, the LDS unit) touches.  Hazards as mfma32_c.
template <typename V8>
static __device__ __forceinline__ f32x16 mfma32_first_av(V8 a, V8 b, const f32x16& c) {
    f32x16 d;
    if constexpr (__is_same(V8, f16x8))
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
    return d;
}
template <typename V8>
static __device__ __forceinline__ void mfma32_acc_av(f32x16& acc, V8 a, V8 b) {
    if constexpr (__is_same(V8, f16x8))
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b));
}

// hipcc's hazard recogniser pads an MFMA result -> VALU read with the required wait states only when it can see the reader;
// the asm v_max3 below is opaque to it, so reading fresh accumulators raced with the matrix pipeline (nondeterministic
// scores, found by the batch-permutation test).  19 wait states cover a 16-pass MFMA; tied operands order the block after
// the MFMAs and before the readers.
static __device__ __forceinline__ void mfma_settle(f32x16 (&s)[2]) {
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));
}
// maximum of the 32 scores a lane holds: 16 v_max3 in four independent chains + 2 to join them
static __device__ __forceinline__ float max32(const f32x16 (&s)[2]) {
    float m0 = max3f(s[0][0], s[0][1], s[0][2]), m1 = max3f(s[0][3], s[0][4], s[0][5]);
    float m2 = max3f(s[0][6], s[0][7], s[0][8]), m3 = max3f(s[0][9], s[0][10], s[0][11]);
    m0 = max3f(m0, s[0][12], s[0][13]);
    m1 = max3f(m1, s[0][14], s[0][15]);
    m2 = max3f(m2, s[1][0], s[1][1]);
    m3 = max3f(m3, s[1][2], s[1][3]);
    m0 = max3f(m0, s[1][4], s[1][5]);
    m1 = max3f(m1, s[1][6], s[1][7]);
    m2 = max3f(m2, s[1][8], s[1][9]);
    m3 = max3f(m3, s[1][10], s[1][11]);
    m0 = max3f(m0, s[1][12], s[1][13]);
    m1 = max3f(m1, s[1][14], s[1][15]);
    return max3f(max3f(m0, m1, m2), m3, m3);
}

#ifndef DINO_PREC
#define DINO_PREC 0  // tuning builds: which operand roundings attention_kernel removes (gemm.hip, "DINO_PREC"); 0 in the product
#endif
#ifndef DINO_ATT_LOADER
#define DINO_ATT_LOADER 0  // attention_kernel, tuning builds (profiles/r05_attention_loader.md): a LOADER wave issues every global_load_lds of the K / V
                           // ring, the compute waves none.  1: 4 compute + 1 loader, registers capped for four such workgroups per CU (96);
                           // 3: the same at 128 registers (three workgroups per CU); 2: 3 compute + 1 loader at 128 registers (96-query blocks)
#endif
#ifndef DINO_ATT_ABL
#define DINO_ATT_ABL 0  // attention2_kernel, timing-only ablations (WRONG results): 1 no exp, 2 no staging, 4 no barrier, 8 no V
                        // reads, 16 no K reads.  The same study of attention_kernel: tools/probes/attention_abl.hip
#endif
#ifndef DINO_ATT3_WAVES
#define DINO_ATT3_WAVES 2  // waves per workgroup of the 64-queries-per-wave kernel (2: 128-query blocks, 4: 256-query blocks)
#endif

// LOG2: scores arrive multiplied by log2(e) (folded into the q scale by the QKV epilogue), so p = exp2(s - m) needs no
// multiply.  launch_bounds(256, 2): allow up to 256 VGPRs -- with the default budget hipcc parked 128 values in AGPRs
// and spent 255 v_accvgpr moves per key tile shuttling them (as many VALU ops as the softmax itself).
// QB = query blocks of 32 per wave.  QB = 1: 121 VGPRs, four waves per SIMD (the round-1 throughput kernel).  QB = 2: every K / V^T
// fragment a wave reads from LDS feeds TWO MFMAs (the kernel moves 16 KiB of LDS reads + 4 KiB of staging per 16 MFMAs at QB = 1 --
// more per MFMA than the GEMM, and like the GEMM it is bound by what the CU can move next to the MFMAs, profiles/r02_gemm_kloop.md):
// half the LDS bytes per MFMA for twice the registers (two waves per SIMD).  Per query the arithmetic is the same instruction
// sequence in the same order, so QB does not change a single bit of the result.
template <typename T, bool LOG2, int NWV, int QB = 1>
#if DINO_ATT_LOADER
__global__ __launch_bounds__((NWV + 1) * 64, DINO_ATT_LOADER == 1 ? 5 : DINO_ATT_LOADER == 2 ? 4 : 3) void attention_kernel(
#else
__global__ __launch_bounds__(NWV * 64, QB == 2 ? 2 : (NWV == 8 ? 4 : 2)) void attention_kernel(
#endif
    const T* __restrict__ qkv, T* __restrict__ out, int Ttok, int H) {
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    using vec4 = typename E::vec4;
    constexpr int KT = 64;          // keys per tile
    constexpr int ROWB = 128;       // bytes per LDS row (64 dims)
    constexpr int TILEB = KT * ROWB;

#if DINO_PREC & 24
    __shared__ __attribute__((aligned(16))) char smem[2 * 4 * TILEB];  // [buf][K|V] + [buf][K_lo|V_lo] at + 4 TILEB
#else
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILEB];  // [buf][K|V]
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    DINO_TS_INIT
    DINO_CLK_BEGIN()
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, which would put the query
    // blocks of one (image, head) on 8 different L2s and fetch its K/V from HBM 8 times (measured: 1.6 GB per launch against
    // 0.36 GB algorithmic, i.e. the kernel ran at HBM speed).  xcd_remap gives each XCD a contiguous range of logical ids,
    // so all query blocks of a head share one L2.
    constexpr int QW = 32 * QB;  // queries per wave
    const int nqb = (Ttok + NWV * QW - 1) / (NWV * QW), nhd = H >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = lid % nqb, h = (lid / nqb) % nhd, b = lid / (nqb * nhd);
#if DINO_PREC & 25
    const int H3 = 6 * H;  // (tuning build) every row carries a second word of q | k | v behind the first: [q k v | q_lo k_lo v_lo]
#else
    const int H3 = 3 * H;
#endif
    const char* base = (const char*)(qkv + (size_t)b * Ttok * H3);

    const int ql = lane & 31, hh = lane >> 5;
    const int qrow0 = qb * (NWV * QW) + wid * QW + ql;  // + 32 u for query block u of this wave

    // Q^T fragments (B operand): lane holds q[qrow][16*ks + 8*hh + 0..7]
    vec8 qf[QB][4];
#if DINO_PREC & 1
    vec8 qfl[QB][4];
#endif
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const int qrc = qrow0 + 32 * u < Ttok ? qrow0 + 32 * u : Ttok - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[u][ks] = *(const vec8*)(base + ((size_t)qrc * H3 + h * 64 + ks * 16 + hh * 8) * 2);
#if DINO_PREC & 1
            qfl[u][ks] = *(const vec8*)(base + ((size_t)qrc * H3 + 3 * H + h * 64 + ks * 16 + hh * 8) * 2);
#endif
        }
    }

    // staging: a wave-instruction covers 8 rows x 128 B; NWV waves x SI instructions = 64 rows, for K and for V.  Per-lane
    // 32-bit byte offsets from the (image, head) K base, advanced by one tile per step and clamped to the last key (tail rows
    // re-read it and are masked below): 2 VALU per instruction instead of a 64-bit multiply-add chain.
    const int srow = lane >> 3;
#if DINO_ATT_LOADER
    constexpr int SI = 8;         // the loader wave (wid == NWV) moves all eight 8-row pieces of K and of V
    constexpr int SW = 1;         // piece j covers rows 8 j .. 8 j + 7
    const int swid = 0;
#else
    constexpr int SI = 8 / NWV;
    constexpr int SW = NWV;
    const int swid = wid;
#endif
    const char* kbase = base + ((size_t)h * 64 + H) * 2;
    const unsigned rowb = (unsigned)H3 * 2u;
    const char* vbase = kbase + (size_t)H * 2;
    unsigned stoff[SI], stmax[SI];
#pragma unroll
    for (int j = 0; j < SI; ++j) {
        const int r = (j * SW + swid) * 8 + srow;
        const unsigned lc = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        stoff[j] = (unsigned)r * rowb + lc;
        stmax[j] = (unsigned)(Ttok - 1) * rowb + lc;
    }
    // The V tile has its own chunk swizzle (see vaddr below): chunk ^ (((row >> 1) & 1) << 2) instead of K's
    // chunk ^ ((row >> 1) & 7).  Both depend on the lane only (row >> 1 = 4 * (j * NWV + wid) + (lane >> 4)), and the chunk
    // index is bits 6:4 of the source offset, so V's source offset is K's with those bits XORed by a per-lane constant.
#if DINO_ATT_LOADER
    unsigned vswzj[SI];  // (row >> 1 = 4 j + (lane >> 4): the piece's parity replaces the wave's)
#pragma unroll
    for (int j = 0; j < SI; ++j) vswzj[j] = (unsigned)((((j & 1) << 2) | ((lane >> 4) & 3)) ^ (((lane >> 4) & 1) << 2)) << 4;
#endif
    const unsigned vswz = (unsigned)((((wid & 1) << 2) | ((lane >> 4) & 3)) ^ (((lane >> 4) & 1) << 2)) << 4;
    auto stage = [&](int buf, int jt) {  // tiles are staged in order: jt only documents which one this call fetches
        char* sK = smem + buf * 2 * TILEB;
        char* sV = sK + TILEB;
#pragma unroll
        for (int j = 0; j < SI; ++j) {
            const unsigned off = stoff[j] < stmax[j] ? stoff[j] : stmax[j];
            stoff[j] += KT * rowb;
#if DINO_ATT_LOADER
            glds16(kbase + off, sK + j * 8 * ROWB);
            glds16(vbase + (off ^ vswzj[j]), sV + j * 8 * ROWB);
#else
            glds16(kbase + off, sK + (j * NWV + wid) * 8 * ROWB);  // uniform base + 32-bit lane offset: scalar-base loads
            glds16(vbase + (off ^ vswz), sV + (j * NWV + wid) * 8 * ROWB);
#if DINO_PREC & 8
            glds16(kbase + (size_t)3 * H * 2 + off, sK + 4 * TILEB + (j * NWV + wid) * 8 * ROWB);
#endif
#if DINO_PREC & 16
            glds16(vbase + (size_t)3 * H * 2 + (off ^ vswz), sV + 4 * TILEB + (j * NWV + wid) * 8 * ROWB);
#endif
#endif
        }
    };

    const int sw = (ql >> 1) & 7;
    // K fragment byte offsets inside a K tile, one per 16-wide k-step (+ kb * 4096 as an immediate)
    int kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = ql * ROWB + (((ks * 2 + hh) ^ sw) << 4);
    // V^T gather for ds_read_b64_tr_b16: within each 16-lane group, lane t supplies the address of
    // V[key0 + (t >> 2)][d0 + 4*(t & 3) .. +3] and receives V[key0 + 0..3][d0 + t].  key0 = 16t + 8*half + 4*hh: the row
    // swizzle term does not depend on t, so four base offsets + t * 2048 as an immediate cover the tile.  The V tile's swizzle
    // is chunk ^ (((row >> 1) & 1) << 2): the 32 lanes of one LDS cycle read 4 consecutive rows x 64 bytes, and this puts
    // rows r, r+1, r+2, r+3 on the four 64-byte quarters of the 256-byte bank row (with K's swizzle rows r and r+2 shared
    // banks: SQ_LDS_BANK_CONFLICT was a third of SQ_LDS_IDX_ACTIVE, every V^T read took two passes).
    const int t16 = lane & 15;
    int vaddr[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int row0 = 8 * half + 4 * hh + (t16 >> 2);
            const int colbyte = db * 64 + (((lane >> 4) & 1) * 16 + (t16 & 3) * 4) * 2;
            vaddr[half][db] = row0 * ROWB + ((((colbyte >> 4) ^ (((row0 >> 1) & 1) << 2)) << 4) | (colbyte & 15));
        }

    f32x16 o[QB][2];
#pragma unroll
    for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[u][0][r] = o[u][1][r] = 0.f;
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) m_run[u] = l_run[u] = 0.f;
    // -m_run broadcast over a 16-register tuple: fed as the C operand of the first MFMA of every score chain, so the
    // accumulators come out as (s - m_run) and the softmax needs no subtraction; rewritten only when m_run moves.
    f32x16 negm[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[u][r] = 0.f;
    // deferred max (online softmax): the reference point m_run only moves when a tile's maximum exceeds it by more than
    // THR, so most tiles skip the O / l rescale.  p <= 2^THR stays far inside f16/bf16 range and keeps full relative
    // precision; the first tile always takes the rescale branch (alpha = 0, whatever its maximum is).
    constexpr float THR = LOG2 ? 8.0f : 5.5f;

    const int ntiles = (Ttok + KT - 1) / KT;
    // wave-uniform: the last query block of a head is ragged (1 374 tokens = 10 blocks + 94 queries -> one idle wave of 44)
    const bool idle_wave = __builtin_amdgcn_readfirstlane(qb * (NWV * QW) + wid * QW) >= Ttok;
    auto tile = [&](int jt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        DINO_TS(0)
        __syncthreads();
        DINO_TS(1)
#if !DINO_ATT_LOADER
        if (!MASKED) stage((jt + 1) & 1, jt + 1);
#endif
        if (idle_wave) return;  // a wave whose 32 queries all lie past the last token only helps with staging and barriers
        const char* sK = smem + (jt & 1) * 2 * TILEB;
        const char* sV = sK + TILEB;

        // ---- S^T = K Q^T : two 32-key blocks; every K fragment is read once and multiplied with all QB query blocks ----
        f32x16 s[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf = *(const vec8*)(sK + kaddr[ks] + kb * 32 * ROWB);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    if (ks == 0) s[u][kb] = mfma32_c(kf, qf[u][0], negm[u]);  // D != C: no copy of the 16 -m_run registers per chain
                    else s[u][kb] = E::mfma32(kf, qf[u][ks], s[u][kb]);
#if DINO_PREC & 1
                    s[u][kb] = E::mfma32(kf, qfl[u][ks], s[u][kb]);
#endif
                }
#if DINO_PREC & 8
                {
                    const vec8 kfl = *(const vec8*)(sK + 4 * TILEB + kaddr[ks] + kb * 32 * ROWB);
#pragma unroll
                    for (int u = 0; u < QB; ++u) s[u][kb] = E::mfma32(kfl, qf[u][ks], s[u][kb]);
                }
#endif
            }
        }
        DINO_TS(2)
        // s[u][kb][r] = score - m_run of key jt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hh; only the last tile has keys >= Ttok
        if constexpr (MASKED) {
            const int kbase = jt * KT + 4 * hh;
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= Ttok) s[u][kb][r] = -INFINITY;
        }
        // ---- online softmax (soft_max_ext semantics: exp(s - max) / sum), statistics per lane = per query ----
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            mfma_settle(s[u]);
            const float mx = max_halves(max32(s[u]));  // tile maximum relative to m_run
            const bool first = jt == 0;             // m_run = 0 is not a real reference yet: take the tile maximum, whatever it is
            const bool need = first || mx > THR;
            if (__any(need)) {  // wave-uniform; lanes that do not need it shift by d = 0 (alpha = 1)
                const float d = need ? mx : 0.f;
                const float alpha = first ? 0.f : (LOG2 ? __builtin_amdgcn_exp2f(-d) : __expf(-d));
                m_run[u] += d;
                l_run[u] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[u][0][r] *= alpha;
                    o[u][1][r] *= alpha;
                    negm[u][r] = -m_run[u];
                    s[u][0][r] -= d;
                    s[u][1][r] -= d;
                }
            }
            // four summation chains (register index mod 4), joined pairwise: the SAME order as attention2_kernel, so that the
            // kernels agree bit for bit and an image's result does not depend on which one its batch size selects
            float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = LOG2 ? __builtin_amdgcn_exp2f(s[u][kb][r]) : __expf(s[u][kb][r]);
                    s[u][kb][r] = pv;
                    ps[r & 3] += pv;
                }
            l_run[u] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        }
        __builtin_amdgcn_s_setprio(1);  // the PV section (converts, V^T gathers, MFMAs) ahead of the other waves' softmax: +1.5 % measured
        DINO_TS(4)
        // ---- O^T += V^T P^T : 4 k-steps of 16 keys; lane's 8 k-slots of step t = score regs (t&1)*8 .. +7 of
        //      block t>>1, i.e. keys 16t + 4hh + {0..3} and 16t + 8 + 4hh + {0..3}.  Every V^T fragment feeds all QB query blocks.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vec8 pf[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[u][j] = E::from_f32(s[u][t >> 1][(t & 1) * 8 + j]);
#if DINO_PREC & 2
            vec8 pfl[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) pfl[u][j] = E::from_f32(s[u][t >> 1][(t & 1) * 8 + j] - E::to_f32(pf[u][j]));
#endif
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                vec8 vf;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const s16x4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (DINO_LDS_AS s16x4*)(sV + vaddr[half][db] + t * 16 * ROWB));
                    const vec4 v4 = __builtin_bit_cast(vec4, raw);
                    vf[half * 4 + 0] = v4[0];
                    vf[half * 4 + 1] = v4[1];
                    vf[half * 4 + 2] = v4[2];
                    vf[half * 4 + 3] = v4[3];
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) o[u][db] = E::mfma32(vf, pf[u], o[u][db]);
#if DINO_PREC & 2
#pragma unroll
                for (int u = 0; u < QB; ++u) o[u][db] = E::mfma32(vf, pfl[u], o[u][db]);
#endif
#if DINO_PREC & 16
                {
                    vec8 vfl;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const s16x4 raw = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (DINO_LDS_AS s16x4*)(sV + 4 * TILEB + vaddr[half][db] + t * 16 * ROWB));
                        const vec4 v4 = __builtin_bit_cast(vec4, raw);
                        vfl[half * 4 + 0] = v4[0];
                        vfl[half * 4 + 1] = v4[1];
                        vfl[half * 4 + 2] = v4[2];
                        vfl[half * 4 + 3] = v4[3];
                    }
#pragma unroll
                    for (int u = 0; u < QB; ++u) o[u][db] = E::mfma32(vfl, pf[u], o[u][db]);
                }
#endif
            }
        }
        __builtin_amdgcn_s_setprio(0);
        DINO_TS(5)
    };
#if DINO_ATT_LOADER
    if (wid == NWV) {  // the loader wave: one tile ahead of the compute waves, the same barriers, nothing else
        stage(0, 0);
        for (int jt = 0; jt < ntiles; ++jt) {
            __syncthreads();
            if (jt + 1 < ntiles) stage((jt + 1) & 1, jt + 1);
        }
        return;
    }
#else
    stage(0, 0);
#endif
    for (int jt = 0; jt + 1 < ntiles; ++jt) tile(jt, std::false_type{});
    tile(ntiles - 1, std::true_type{});
    DINO_TS_FLUSH
    DINO_CLK_END(g_clk_att, CLK_ATTENTION)

    // ---- normalise and store: o[u][db][r] = O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const int qrow = qrow0 + 32 * u;
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32);
        const float inv = 1.0f / l_tot;
        if (qrow < Ttok) {
            T* orow = out + ((size_t)b * Ttok + qrow) * H + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    vec4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = E::from_f32(o[u][db][g * 4 + j] * inv);
                    *(vec4*)(orow + db * 32 + g * 8 + hh * 4) = w;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// attention2: the same mapping, software-pipelined inside each wave.  Measured on attention_kernel: per 64-key tile a wave
// issues 512 cycles of MFMA and ~600 cycles of VALU (32 v_exp, max, sum, convert) strictly one after the other -- S MFMAs ->
// max -> exp -> PV MFMAs is one dependency chain -- so each pipe idles while the other works (MFMA busy 40 %).  Here the
// chain is cut in two: while the softmax of tile j runs on the VALU, the matrix core computes the scores of tile j+1
// (independent work), and the maximum of tile j+1 is taken under the PV MFMAs of tile j.  K is therefore staged one tile
// further ahead than V.  The instruction order is written out in groups (one MFMA + its share of VALU + the LDS reads for
// later groups) and pinned with sched_barrier, so that hipcc neither clusters the MFMAs nor sinks the reads to their uses.
#define DINO_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef DINO_ATT_LSUM
#define DINO_ATT_LSUM 0  // 1: softmax denominators on the matrix core (l += ones x P^T, one extra MFMA per 16 keys).  Sums the
                        // f16-rounded probabilities, so it is NOT bit-identical to attention_kernel: off by default
#endif
// QB = 32-query blocks per wave, NWV = waves per workgroup.  <1, 4>: the batch-1 kernel (2 waves per SIMD).  <2, 2>: 64 queries per
// wave, ONE wave per SIMD with the whole 512-entry register file (two 128-query workgroups per CU): every K / V^T fragment read
// from LDS feeds two MFMAs (half the LDS bytes per MFMA) and the MFMA / softmax overlap happens inside the wave's own instruction
// stream -- the design profiles/r02_attention_anatomy.md section 4 points to.  Per query the arithmetic is the same instruction
// sequence in the same order for every (QB, NWV): results are bit-identical (test_attention_kernels_agree_bit_for_bit).
template <typename T, bool LOG2, int QB = 1, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, QB == 2 ? 1 : 2) void attention2_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Ttok, int H) {
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    using vec4 = typename E::vec4;
    constexpr int KT = 64, ROWB = 128, TILEB = KT * ROWB;
    constexpr bool LSUM = DINO_ATT_LSUM != 0;
    constexpr int QW = 32 * QB, WGQ = NWV * QW;  // queries per wave / per workgroup
    constexpr int SI = (8 + NWV - 1) / NWV;      // 8-row staging pieces per wave, for K and for V (three waves: 3, 3, 2)
    // K/V ring depth.  Two slots (the batch-1 kernel): a tile is staged one step before its use, and the other waves of the SIMD
    // cover what is left of its latency.  Three slots (one wave per SIMD: nobody covers anything): a tile is staged TWO steps ahead
    // and the step begins with a counted wait that leaves the newest tile's loads in flight.
    constexpr int RD = QB == 2 ? 3 : 2;
    static_assert(!LSUM || QB == 1, "the matrix-core row sums exist for the single-block kernel only");
    __shared__ __attribute__((aligned(16))) char smem[RD * 2 * TILEB];  // [slot][K|V]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    DINO_TS_INIT
    DINO_CLK_BEGIN()
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (Ttok + WGQ - 1) / WGQ, nhd = H >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);  // all query blocks of a head on one XCD (see attention_kernel)
    const int qb = lid % nqb, h = (lid / nqb) % nhd, b = lid / (nqb * nhd);
    const int H3 = 3 * H;
    const char* base = (const char*)(qkv + (size_t)b * Ttok * H3);
    const int ql = lane & 31, hh = lane >> 5;
    const int qrow0 = qb * WGQ + wid * QW + ql;  // + 32 u for query block u of this wave

    vec8 qf[QB][4];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const int qrc = qrow0 + 32 * u < Ttok ? qrow0 + 32 * u : Ttok - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[u][ks] = *(const vec8*)(base + ((size_t)qrc * H3 + h * 64 + ks * 16 + hh * 8) * 2);
    }

    // staging offsets as in attention_kernel; K runs one tile ahead of V, so each has its own cursor
    const char* kbase = base + ((size_t)h * 64 + H) * 2;
    const unsigned rowb = (unsigned)H3 * 2u, vdelta = (unsigned)H * 2u;
    unsigned koff[SI], voff[SI], stmax[SI], vswz[SI];
#pragma unroll
    for (int j = 0; j < SI; ++j) {
        const int r = (j * NWV + wid) * 8 + (lane >> 3);
        const unsigned lc = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        koff[j] = voff[j] = (unsigned)r * rowb + lc;
        stmax[j] = (unsigned)(Ttok - 1) * rowb + lc;
        // V's chunk swizzle against K's (see attention_kernel): depends on the piece's parity and the lane
        vswz[j] = (unsigned)(((((j * NWV + wid) & 1) << 2) | ((lane >> 4) & 3)) ^ (((lane >> 4) & 1) << 2)) << 4;
    }
    // (with three waves the last wave has no third piece: piece index 8 does not exist)
    auto stage_k1 = [&](int buf, int j) {
        if (8 % NWV != 0 && j * NWV + wid >= 8) return;
        glds16(kbase + (koff[j] < stmax[j] ? koff[j] : stmax[j]), smem + buf * 2 * TILEB + (j * NWV + wid) * 8 * ROWB);
        koff[j] += KT * rowb;
    };
    auto stage_v1 = [&](int buf, int j) {
        if (8 % NWV != 0 && j * NWV + wid >= 8) return;
        glds16(kbase + ((voff[j] < stmax[j] ? voff[j] : stmax[j]) ^ vswz[j]) + vdelta, smem + buf * 2 * TILEB + TILEB + (j * NWV + wid) * 8 * ROWB);
        voff[j] += KT * rowb;
    };
    auto stage_k = [&](int buf) {
#pragma unroll
        for (int j = 0; j < SI; ++j) stage_k1(buf, j);
    };
    auto stage_v = [&](int buf) {
#pragma unroll
        for (int j = 0; j < SI; ++j) stage_v1(buf, j);
    };

    const int sw = (ql >> 1) & 7;
    int kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = ql * ROWB + (((ks * 2 + hh) ^ sw) << 4);
    const int t16 = lane & 15;
    int vaddr[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int row0 = 8 * half + 4 * hh + (t16 >> 2);
            const int colbyte = db * 64 + (((lane >> 4) & 1) * 16 + (t16 & 3) * 4) * 2;
            vaddr[half][db] = row0 * ROWB + ((((colbyte >> 4) ^ (((row0 >> 1) & 1) << 2)) << 4) | (colbyte & 15));
        }
    auto read_vt = [&](const char* sV, int t, int db) {
        vec8 vf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const s16x4 raw =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((DINO_LDS_AS s16x4*)(sV + vaddr[half][db] + t * 16 * ROWB));
            const vec4 v4 = __builtin_bit_cast(vec4, raw);
            vf[half * 4 + 0] = v4[0];
            vf[half * 4 + 1] = v4[1];
            vf[half * 4 + 2] = v4[2];
            vf[half * 4 + 3] = v4[3];
        }
        return vf;
    };
    auto ex2 = [](float x) { return (DINO_ATT_ABL & 1) ? x * 0.5f : LOG2 ? __builtin_amdgcn_exp2f(x) : __expf(x); };

    f32x16 o[QB][2], negm[QB], lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[u][0][r] = o[u][1][r] = negm[u][r] = 0.f;
    vec8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = E::from_f32(1.0f);
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) m_run[u] = l_run[u] = 0.f;
    constexpr float THR = LOG2 ? 8.0f : 5.5f;
    const int ntiles = (Ttok + KT - 1) / KT;

    auto mask_tail = [&](f32x16(&s)[2], int jt) {
        const int kbase_ = jt * KT + 4 * hh;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase_ + kb * 32 + (r & 3) + 8 * (r >> 2) >= Ttok) s[kb][r] = -INFINITY;
    };
    // move the softmax reference point of query block u when the tile maximum (relative to it) exceeds THR; `s` holds scores - m_run
    auto rescale = [&](int u, f32x16(&s)[2], float mx, bool first) {
        const bool need = first || mx > THR;
        if (__any(need)) {
            const float d = need ? mx : 0.f;
            const float alpha = first ? 0.f : ex2(-d);
            m_run[u] += d;
            l_run[u] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[u][0][r] *= alpha;
                o[u][1][r] *= alpha;
                if (LSUM) lacc[r] *= alpha;
                negm[u][r] = -m_run[u];
                s[0][r] -= d;
                s[1][r] -= d;
            }
        }
    };

    // one steady-state step: softmax(cur = tile jt) + PV(tile jt), scores of tile jt+1 into nxt.  PAR = jt & 1 (compile
    // time: LDS offsets become immediates).  All LDS reads of the step are inline asm with hand-counted lgkmcnt: hipcc puts
    // `s_waitcnt vmcnt(0)` in front of a ds_read that follows an LDS-DMA (it cannot see that the buffers differ), which
    // would park the wave on the loads it has just issued.  DS returns are in order, so each wait names how many younger
    // reads may still be outstanding.  Issue order: K0..K3 | group g < 4: K(g+4), V(2g), V(2g+1) | group g >= 4: V(2g), V(2g+1).
    // Before score MFMA g < 4: (3 - g) + 3g younger reads; g >= 4: 2 + the three groups in between = 11, 10, 9, 8.
    // PV MFMA u (V fragment u, read in group u) runs in group u + 4 for u < 4 (three groups in between: 9, 8, 7, 6 younger
    // reads) and after the groups for u >= 4 (2 * (7 - u)).  The fragment reads do not depend on QB: every fragment feeds the
    // MFMAs of all QB query blocks.
    const unsigned lds0 = (unsigned)(uintptr_t)(DINO_LDS_AS char*)smem;
    unsigned kad[4], vad[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) kad[i] = lds0 + kaddr[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) vad[i >> 1][i & 1] = lds0 + vaddr[i >> 1][i & 1];
#define DINO_KRD(DST, ADDR, OFF)                                                                             \
    if (!(DINO_ATT_ABL & 16)) {                                                                              \
        if constexpr (QB == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(DST) : "v"(ADDR), "n"(OFF)); \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF));                \
    }
#define DINO_VRD(DST, ADDR, OFF) \
    if (!(DINO_ATT_ABL & 8)) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
    auto iter = [&](int jt, f32x16(&cur)[QB][2], f32x16(&nxt)[QB][2], auto mask_tag, auto par_tag) {
        constexpr bool MASKNEXT = decltype(mask_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int KOFF = RD == 2 ? ((PAR + 1) & 1) * 2 * TILEB : 0;  // K_{jt+1}
        constexpr int VOFF = RD == 2 ? PAR * 2 * TILEB + TILEB : 0;      // V_jt
        DINO_TS(0)
        if (!(DINO_ATT_ABL & 4)) {
            if constexpr (RD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SI) : "memory");  // all but the previous step's loads (K_{jt+2}, V_{jt+1})
            __syncthreads();  // K_{jt+1}, V_jt landed; the slots of K_jt, V_{jt-1} are free
        }
        DINO_TS(1)
        // ring slots of this step: K_{jt+1} / V_jt are read, K_{jt+RD} / V_{jt+RD-1} are staged.  Two slots: compile-time parity,
        // LDS offsets are immediates.  Three slots: run-time offsets added to the eight fragment address registers.
        const int kst = RD == 2 ? PAR : jt % 3, vst = RD == 2 ? (PAR + 1) & 1 : (jt + 2) % 3;
        unsigned kA[4], vA[2][2];
        {
            const unsigned ko = RD == 2 ? 0u : (unsigned)(((jt + 1) % 3) * 2 * TILEB), vo = RD == 2 ? 0u : (unsigned)((jt % 3) * 2 * TILEB + TILEB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kA[i] = kad[i] + ko;
                vA[i >> 1][i & 1] = vad[i >> 1][i & 1] + vo;
            }
        }
        if constexpr (QB == 2) {
            // 512-register kernel: the values only the matrix core touches live in the accumulator half of the file for the whole
            // step (O, Q^T and the K fragments: 128 registers), so that the 256 architectural VGPRs hold the two score tiles, -m_run,
            // P and the V^T fragments.  Without the pins hipcc shuttles ~230 values per step through v_accvgpr_read / _write.
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                asm volatile("" : "+a"(o[u][0]), "+a"(o[u][1]));
                asm volatile("" : "+a"(qf[u][0]), "+a"(qf[u][1]), "+a"(qf[u][2]), "+a"(qf[u][3]));
            }
        }
        vec8 kf[8], pf[QB][4];
        s16x4 vl[8], vh[8];
        if (DINO_ATT_ABL & 24) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                kf[i] = qf[0][i & 3];
                vl[i] = vh[i] = __builtin_bit_cast(s16x4, (double)jt);
            }
        }
        DINO_KRD(kf[0], kA[0], KOFF);
        DINO_KRD(kf[1], kA[1], KOFF);
        DINO_KRD(kf[2], kA[2], KOFF);
        DINO_KRD(kf[3], kA[3], KOFF);
        DINO_SB();
        float ps[QB][4];
#pragma unroll
        for (int u = 0; u < QB; ++u) ps[u][0] = ps[u][1] = ps[u][2] = ps[u][3] = 0.f;
// One MFMA per micro-slot, its share of the VALU work behind it, a scheduling fence between slots: with ONE wave per SIMD the
// matrix pipe only stays busy if every MFMA is followed by a few VALU instructions and then the next MFMA -- an in-order wave that
// meets two MFMAs in a row waits out the first one's 32 cycles without issuing anything (measured: the clumped order of the
// four-waves-per-SIMD kernels ran this kernel at 40 % matrix-pipe utilisation with all data movement removed).
#define DINO_PVMMA1(U, UU)                                                                                   \
        {                                                                                                    \
            const vec4 lo = __builtin_bit_cast(vec4, vl[U]), hi = __builtin_bit_cast(vec4, vh[U]);           \
            const vec8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                         \
            o[UU][(U) & 1] = E::mfma32(vf, pf[UU][(U) >> 1], o[UU][(U) & 1]);                                \
        }
#define DINO_PVWAIT(U, WAITN) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vl[U]), "+v"(vh[U]) : "n"(WAITN));
#define DINO_GROUP(G)                                                                                        \
        {                                                                                                    \
            if constexpr (QB == 2) asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(kf[G]) : "n"((G) < 4 ? 3 + 2 * (G) : 15 - (G))); \
            else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kf[G]) : "n"((G) < 4 ? 3 + 2 * (G) : 15 - (G))); \
            _Pragma("unroll") for (int u = 0; u < QB; ++u) {                                                 \
                if constexpr (QB == 1) nxt[u][(G) >> 2] = E::mfma32(kf[G], qf[u][(G) & 3], ((G) & 3) == 0 ? negm[u] : nxt[u][(G) >> 2]); \
                else if (((G) & 3) == 0) nxt[u][(G) >> 2] = mfma32_first_av(kf[G], qf[u][0], negm[u]);       \
                else mfma32_acc_av(nxt[u][(G) >> 2], kf[G], qf[u][(G) & 3]);                                 \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
                    const int idx = (G) * 4 + j;                                                             \
                    const float pv = ex2(cur[u][idx >> 4][idx & 15]);                                        \
                    cur[u][idx >> 4][idx & 15] = pv;                                                         \
                    if (!LSUM) ps[u][j] += pv;                                                               \
                }                                                                                            \
                if (((G) & 1) && (G) < 4) {                                                                  \
                    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
                        pf[u][(G) >> 1][j] = E::from_f32(cur[u][(G) >> 2][(((G) >> 1) & 1) * 8 + j]);        \
                }                                                                                            \
                DINO_SB();                                                                                   \
            }                                                                                                \
            if ((G) >= 4) {                                                                                  \
                DINO_PVWAIT(((G) - 4) & 7, 13 - (G))                                                         \
                _Pragma("unroll") for (int u = 0; u < QB; ++u) {                                             \
                    DINO_PVMMA1(((G) - 4) & 7, u)                                                            \
                    if ((G) & 1) {                                                                           \
                        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                        \
                            pf[u][(G) >> 1][j] = E::from_f32(cur[u][(G) >> 2][(((G) >> 1) & 1) * 8 + j]);    \
                    }                                                                                        \
                    DINO_SB();                                                                               \
                }                                                                                            \
            }                                                                                                \
            if (LSUM && ((G) & 1)) lacc = E::mfma32(ones, pf[0][(G) >> 1], lacc);                            \
            if (!(DINO_ATT_ABL & 2)) {  /* K_{jt+RD} and V_{jt+RD-1}: the loads go out under the MFMAs */     \
                if (SI == 2) {                                                                               \
                    if ((G) == 0) stage_k1(kst, 0);                                                          \
                    if ((G) == 1) stage_k1(kst, 1);                                                          \
                    if ((G) == 2) stage_v1(vst, 0);                                                          \
                    if ((G) == 3) stage_v1(vst, 1);                                                          \
                } else if (SI == 3) {                                                                        \
                    if ((G) < 3) stage_k1(kst, (G));                                                         \
                    else if ((G) < 6) stage_v1(vst, (G) - 3);                                                \
                } else {                                                                                     \
                    if ((G) < 4) stage_k1(kst, (G) & (SI - 1));                                              \
                    else stage_v1(vst, ((G) - 4) & (SI - 1));                                                \
                }                                                                                            \
            }                                                                                                \
            if ((G) < 4) DINO_KRD(kf[((G) + 4) & 7], kA[(G) & 3], KOFF + 4096);                              \
            DINO_VRD(vl[G], vA[0][(G) & 1], VOFF + ((G) >> 1) * 16 * ROWB);                                  \
            DINO_VRD(vh[G], vA[1][(G) & 1], VOFF + ((G) >> 1) * 16 * ROWB);                                  \
            DINO_SB();                                                                                       \
        }
        DINO_GROUP(0) DINO_GROUP(1) DINO_GROUP(2) DINO_GROUP(3) DINO_GROUP(4) DINO_GROUP(5) DINO_GROUP(6) DINO_GROUP(7)
#undef DINO_GROUP
        if (!LSUM) {
#pragma unroll
            for (int u = 0; u < QB; ++u) l_run[u] += (ps[u][0] + ps[u][1]) + (ps[u][2] + ps[u][3]);
        }
        DINO_TS(2)
        if constexpr (MASKNEXT) {
#pragma unroll
            for (int u = 0; u < QB; ++u) mask_tail(nxt[u], jt + 1);
        }
        // second half of PV (keys 32..63 of the tile) with the maximum of the next tile's scores underneath
#pragma unroll
        for (int u = 0; u < QB; ++u) mfma_settle(nxt[u]);
        float ma[QB], mb[QB];
#define DINO_PV(U)                                                                                           \
        {                                                                                                    \
            DINO_PVWAIT(U, 14 - 2 * (U))                                                                     \
            _Pragma("unroll") for (int u = 0; u < QB; ++u) {                                                 \
                DINO_PVMMA1(U, u)                                                                            \
                _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                              \
                    const f32x16& n = nxt[u][((U) - 4) >> 1];                                                \
                    const int i0 = (((U) - 4) & 1) * 8 + c * 4;                                              \
                    ma[u] = ((U) == 4 && c == 0) ? max3f(n[0], n[0], n[1]) : max3f(ma[u], n[i0], n[i0 + 1]); \
                    mb[u] = ((U) == 4 && c == 0) ? max3f(n[2], n[2], n[3]) : max3f(mb[u], n[i0 + 2], n[i0 + 3]); \
                }                                                                                            \
                DINO_SB();                                                                                   \
            }                                                                                                \
        }
        DINO_PV(4) DINO_PV(5) DINO_PV(6) DINO_PV(7)
#undef DINO_PV
#undef DINO_PVMMA1
#undef DINO_PVWAIT
        DINO_TS(3)
#pragma unroll
        for (int u = 0; u < QB; ++u) rescale(u, nxt[u], max_halves(max3f(ma[u], mb[u], mb[u])), false);
        DINO_TS(4)
    };
    auto last = [&](int jt, f32x16(&cur)[QB][2]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* sV = smem + (jt % RD) * 2 * TILEB + TILEB;
        float ps[QB][4];
#pragma unroll
        for (int u = 0; u < QB; ++u) ps[u][0] = ps[u][1] = ps[u][2] = ps[u][3] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vec8 pf[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = ex2(cur[u][t >> 1][(t & 1) * 8 + j]);
                    ps[u][j & 3] += pv;
                    pf[u][j] = E::from_f32(pv);
                }
            if (LSUM) lacc = E::mfma32(ones, pf[0], lacc);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const vec8 vf = read_vt(sV, t, db);
#pragma unroll
                for (int u = 0; u < QB; ++u) o[u][db] = E::mfma32(vf, pf[u], o[u][db]);
            }
        }
        if (!LSUM) {
#pragma unroll
            for (int u = 0; u < QB; ++u) l_run[u] += (ps[u][0] + ps[u][1]) + (ps[u][2] + ps[u][3]);
        }
    };

    // prologue: K_0, V_0, K_1 (and, with three slots, V_1, K_2) in flight; scores of tile 0 (needs K_0 only)
    stage_k(0);
    stage_v(0);
    stage_k(1);
    if constexpr (RD == 3) {
        stage_v(1);
        stage_k(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * SI) : "memory");  // K_0 (and V_0)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    f32x16 sa[QB][2], sb[QB][2];
    {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf0 = *(const vec8*)(smem + kaddr[ks] + kb * 32 * ROWB);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    if constexpr (QB == 1) sa[u][kb] = E::mfma32(kf0, qf[u][ks], ks == 0 ? negm[u] : sa[u][kb]);
                    else if (ks == 0) sa[u][kb] = mfma32_first_av(kf0, qf[u][0], negm[u]);
                    else mfma32_acc_av(sa[u][kb], kf0, qf[u][ks]);
                }
            }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            if (ntiles == 1) mask_tail(sa[u], 0);
            mfma_settle(sa[u]);
            rescale(u, sa[u], max_halves(max32(sa[u])), true);
        }
    }
    // steps jt = 0 .. ntiles-2 (the last of them masks the tail of its next tile), two per trip so that the score
    // registers swap roles without moves
    DINO_TS(5)
    int jt = 0;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    for (; jt + 2 < ntiles - 1; jt += 2) {
        iter(jt, sa, sb, std::false_type{}, P0{});
        iter(jt + 1, sb, sa, std::false_type{}, P1{});
    }
    if (jt + 2 == ntiles - 1) {  // two steps left
        iter(jt, sa, sb, std::false_type{}, P0{});
        iter(jt + 1, sb, sa, std::true_type{}, P1{});
        last(ntiles - 1, sa);
    } else if (jt + 1 == ntiles - 1) {  // one step left
        iter(jt, sa, sb, std::true_type{}, P0{});
        last(ntiles - 1, sb);
    } else {
        last(ntiles - 1, sa);  // ntiles == 1
    }

    DINO_TS(6)
    DINO_TS_FLUSH
    DINO_CLK_END(g_clk_att, CLK_ATTENTION)
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        // with LSUM every element of lacc is the full row sum (both lane halves included)
        const float l_tot = LSUM ? lacc[0] : l_run[u] + __shfl_xor(l_run[u], 32);
        const float inv = 1.0f / l_tot;
        const int qrow = qrow0 + 32 * u;
        if (qrow < Ttok) {
            T* orow = out + ((size_t)b * Ttok + qrow) * H + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    vec4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = E::from_f32(o[u][db][g * 4 + j] * inv);
                    *(vec4*)(orow + db * 32 + g * 8 + hh * 4) = w;
                }
        }
    }
}
#undef DINO_KRD
#undef DINO_VRD

#ifdef DINO_ATT_PROF
static void att_prof_dump(int nwaves) {
    static std::vector<unsigned long long> h(32768 * 8);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_att_prof), h.size() * 8);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int n = nwaves < 32768 ? nwaves : 32768;
    for (int w = 0; w < n; ++w)
        for (int i = 0; i < 8; ++i) acc[i] += (double)h[w * 8 + i];
    fprintf(stderr, "att_prof cycles/wave:");
    for (int i = 0; i < 8; ++i) fprintf(stderr, " [%d] %.0f", i, acc[i] / n);
    fprintf(stderr, "\n");
}
#endif

static hipError_t launch_attention_impl(DType dt, const void* qkv, void* out, int B, int T, int H, int nh, bool log2_scores,
                                        hipStream_t st);
hipError_t launch_attention(DType dt, const void* qkv, void* out, int B, int T, int H, int nh, bool log2_scores,
                            hipStream_t st) {
    const hipError_t e = launch_attention_impl(dt, qkv, out, B, T, H, nh, log2_scores, st);
#ifdef DINO_ATT_PROF
    att_prof_dump(B * nh * ((T + 127) / 128) * 4);
#endif
    return e;
}

static hipError_t launch_attention_impl(DType dt, const void* qkv, void* out, int B, int T, int H, int nh, bool log2_scores,
                                        hipStream_t st) {
    if (H != nh * 64 || T <= 0 || B <= 0) return hipErrorInvalidValue;
    if ((size_t)B * T * 3 * H * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;  // 32-bit staging cursors into qkv
    // 4 waves = 128 queries per workgroup.  Measured alternatives (profiles/r01_gemm_tuning.md): 8 waves halve the K/V staging per
    // query but waste more on the ragged last query block (1 374 tokens: 0.322 vs 0.296 ms); 2 waves double the workgroup count at
    // batch 1 but run slower (28 vs 25 us).
    constexpr int nwv = 4;
    // Two kernels, chosen by how many workgroups there are per CU.  Many (batch 32: 5 632 on 256 CUs): attention_kernel, 121
    // VGPRs, four workgroups per CU hide each other's latencies (0.296 ms vs 0.31-0.32).  Few (batch 1: 176): nothing to
    // overlap with, so the per-wave dependency chain decides and the software-pipelined attention2_kernel wins (22 vs 26 us).
    // DINOV2_HIP_ATTN_V=1|2 forces one (testing aid, include/dinov2_hip.h "Environment": the two kernels must agree bit for bit).
    const int forced_ver = tune_get(TUNE_ATTN_V);  // (read from the environment once; the tests use dinov2_hip_op_set_tuning)
    const long units = (long)((T + 127) / 128) * nh * B;
    const int ver = forced_ver ? forced_ver : units <= 512 ? 2 : 1;
    if (ver == 2) {
        // Workgroup size.  Smaller blocks spread a batch-1 forward over more CUs (T = 1 374, 16 heads: 176 workgroups of 128 queries,
        // 240 of 96, 352 of 64) -- and measure SLOWER there (17.3 / 18.4 / 23.0 us): a wave's serial chain of 22 key tiles is what
        // the kernel takes, and with fewer waves per workgroup each wave issues more of the tile's staging.  Only short sequences
        // gain (T = 261: 6.9 -> 6.2 us with 64-query blocks).  DINOV2_HIP_ATTN_NWV=2|3|4 forces a size (testing aid; all sizes give
        // the same bits).
        int nw = tune_get(TUNE_ATTN_NWV);
        if (nw != 2 && nw != 3 && nw != 4) nw = (T <= 512 && (long)((T + 63) / 64) * nh * B <= 256) ? 2 : 4;
        const dim3 grid2(((T + 32 * nw - 1) / (32 * nw)) * nh * B), block2(64 * nw);
#define DINO_ATT2(TT, LG)                                                                                                        \
        {                                                                                                                        \
            if (nw == 4) hipLaunchKernelGGL((attention2_kernel<TT, LG, 1, 4>), grid2, block2, 0, st, (const TT*)qkv, (TT*)out, T, H); \
            else if (nw == 3) hipLaunchKernelGGL((attention2_kernel<TT, LG, 1, 3>), grid2, block2, 0, st, (const TT*)qkv, (TT*)out, T, H); \
            else hipLaunchKernelGGL((attention2_kernel<TT, LG, 1, 2>), grid2, block2, 0, st, (const TT*)qkv, (TT*)out, T, H); \
        }
        if (dt == DT_F16) { if (log2_scores) DINO_ATT2(_Float16, true) else DINO_ATT2(_Float16, false) }
        else { if (log2_scores) DINO_ATT2(__bf16, true) else DINO_ATT2(__bf16, false) }
#undef DINO_ATT2
        return hipGetLastError();
    }
    if (ver == 4) {  // pipelined, 64 queries per wave, one wave per SIMD, two 128-query workgroups per CU
        const dim3 grid4(((T + 127) / 128) * nh * B), block4(128);
#define DINO_ATT4(TT, LG) hipLaunchKernelGGL((attention2_kernel<TT, LG, 2, 2>), grid4, block4, 0, st, (const TT*)qkv, (TT*)out, T, H)
        if (dt == DT_F16) { if (log2_scores) DINO_ATT4(_Float16, true); else DINO_ATT4(_Float16, false); }
        else { if (log2_scores) DINO_ATT4(__bf16, true); else DINO_ATT4(__bf16, false); }
#undef DINO_ATT4
        return hipGetLastError();
    }
    if (ver == 3) {  // 64 queries per wave, NWQ waves per workgroup
        constexpr int NWQ = DINO_ATT3_WAVES;
        const dim3 grid3(((T + NWQ * 64 - 1) / (NWQ * 64)) * nh * B), block3(NWQ * 64);
#define DINO_ATT3(TT, LG) hipLaunchKernelGGL((attention_kernel<TT, LG, NWQ, 2>), grid3, block3, 0, st, (const TT*)qkv, (TT*)out, T, H)
        if (dt == DT_F16) { if (log2_scores) DINO_ATT3(_Float16, true); else DINO_ATT3(_Float16, false); }
        else { if (log2_scores) DINO_ATT3(__bf16, true); else DINO_ATT3(__bf16, false); }
#undef DINO_ATT3
        return hipGetLastError();
    }
#if DINO_ATT_LOADER
    constexpr int cw = DINO_ATT_LOADER == 2 ? 3 : 4;  // compute waves; one loader wave on top
    const dim3 grid(((T + cw * 32 - 1) / (cw * 32)) * nh * B), block((cw + 1) * 64);
#define DINO_ATT(TT, LG, NW) \
    hipLaunchKernelGGL((attention_kernel<TT, LG, cw>), grid, block, 0, st, (const TT*)qkv, (TT*)out, T, H)
#else
    const dim3 grid(((T + nwv * 32 - 1) / (nwv * 32)) * nh * B), block(nwv * 64);
#define DINO_ATT(TT, LG, NW) \
    hipLaunchKernelGGL((attention_kernel<TT, LG, NW>), grid, block, 0, st, (const TT*)qkv, (TT*)out, T, H)
#endif
#define DINO_ATT_N(TT, LG) { DINO_ATT(TT, LG, 4); }
    if (dt == DT_F16) { if (log2_scores) DINO_ATT_N(_Float16, true) else DINO_ATT_N(_Float16, false) }
    else { if (log2_scores) DINO_ATT_N(__bf16, true) else DINO_ATT_N(__bf16, false) }
#undef DINO_ATT_N
#undef DINO_ATT
    return hipGetLastError();
}

// ---- probe: empirical lane mapping of ds_read_b64_tr_b16 (kept as a regression test of the assumption above) ----
__global__ void probe_tr16_kernel(int16_t* out) {
    __shared__ __attribute__((aligned(16))) int16_t lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (int16_t)i;
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((DINO_LDS_AS s16x4*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

hipError_t launch_probe_tr16(int16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, st, out);
    return hipGetLastError();
}

}  // namespace dinov2
