// Device-side element traits for the two MFMA input types (f16 / bf16) on gfx950.
#pragma once
#include <hip/hip_runtime.h>

namespace dinov2 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define DINO_GLOBAL_AS __attribute__((address_space(1)))
#define DINO_LDS_AS __attribute__((address_space(3)))

template <typename T>
struct Elem;

template <>
struct Elem<_Float16> {
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {  // 16x16x32: D[4 (l >> 4) + e][l & 15], rows = A rows
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ _Float16 from_f32(float x) { return (_Float16)x; }
    static __device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
};

template <>
struct Elem<__bf16> {
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ __bf16 from_f32(float x) { return (__bf16)x; }
    static __device__ __forceinline__ float to_f32(__bf16 x) { return (float)x; }
};

// ---- clock probe (bench.py `effective_clock_ghz`, `kernel_clocks_ghz`).  Thread 0 of workgroup 0 of every launch of the forward's
// heavy kernels stamps the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) at entry and exit; the LAST launch's
// differences stay in the slot of its kind: shader cycles / wall time = the clock the power-limited part sustained under that kernel's load
// (1.6 - 2.0 GHz of a nominal 2.4).  Persistent GEMMs: workgroup 0 lives for the whole launch; attention: for its own tile only (the
// first of ~ 22 waves of workgroups).  Four scalar loads, three loads and four stores per launch.  One array per translation unit (no relocatable
// device code), merged by dinov2_hip_op_clock_slots (ops_testing.cpp): per slot, the unit with the latest end stamp wins.
enum ClockSlot : int { CLK_QKV = 0, CLK_ATTN_OUT = 1, CLK_FFN_IN = 2, CLK_FFN_OUT = 3, CLK_ATTENTION = 4, CLK_OTHER = 5, CLK_SLOTS = 6 };
#define DINO_CLK_BEGIN()                                             \
    const bool ck_on__ = blockIdx.x == 0 && threadIdx.x == 0;        \
    unsigned long long ck_c0__ = 0, ck_r0__ = 0;                     \
    if (ck_on__) {                                                   \
        ck_c0__ = __builtin_readcyclecounter();                      \
        ck_r0__ = __builtin_amdgcn_s_memrealtime();                  \
    }
// (atomic adds: two sessions on concurrent streams of one device -- the group front's lanes -- may end launches of one kind at the same time)
#define DINO_CLK_END(ARR, SLOT)                                                                  \
    if (ck_on__) {                                                                               \
        const unsigned long long ck_r1__ = __builtin_amdgcn_s_memrealtime();                     \
        atomicAdd(&(ARR)[(SLOT) * 4 + 0], (unsigned long long)(__builtin_readcyclecounter() - ck_c0__)); \
        atomicAdd(&(ARR)[(SLOT) * 4 + 1], ck_r1__ - ck_r0__);                                    \
        (ARR)[(SLOT) * 4 + 2] = ck_r1__;                                                         \
        atomicAdd(&(ARR)[(SLOT) * 4 + 3], 1ull);                                                 \
    }
// which slot a GEMM launch belongs to (EPI: kernels.h Epilogue incl. the LN-fold variants; the residual epilogue serves attn-out, K = N, and
// FFN-out, K > N).  Evaluated ONCE per logical launch by launch_gemm on the caller's N and K (GemmArgs::clk_slot): the parts of a split
// launch see a smaller N (ViT-S attn-out, N = 384 -> 256 + 128, was counted as FFN-out).
#define DINO_CLK_GEMM_SLOT(EPI, N, K)                                                                                             \
    (((EPI) == 1 || (EPI) == 7) ? CLK_QKV : ((EPI) == 2 || (EPI) == 6) ? ((K) > (N) ? CLK_FFN_OUT : CLK_ATTN_OUT)                   \
                                          : ((EPI) == 3 || (EPI) == 4 || (EPI) == 8 || (EPI) == 9) ? CLK_FFN_IN : CLK_OTHER)

// 16-byte async global -> LDS copy (global_load_lds_dwordx4): LDS destination = wave-uniform `lds` + lane*16
static __device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
    __builtin_amdgcn_global_load_lds((const DINO_GLOBAL_AS void*)gsrc, (DINO_LDS_AS void*)lds, 16, 0, 0);
}

// XCD-aware, bijective remap of a 1-D block id: the dispatcher places block b on XCD b % 8, so give every XCD a
// contiguous chunk of the logical tile order (neighbouring tiles share operand panels -> same L2).
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- LN fold (kernels.h, EPI_RESID_LN and the *_LN consumers) -------------------------------------------------------------------------
// DPP lane exchange inside rows of 16 lanes: quad_perm [1,0,3,2] / [2,3,0,1] (xor 1 / xor 2), row_half_mirror (7 - i within 8 lanes: the other
// quad once both quads are uniform), row_mirror (15 - i: the other half once both halves are uniform)
template <int CTRL>
static __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// (sum, sum of squares) of four consecutive columns: the leaves of the fixed pairwise tree every producer uses
static __device__ __forceinline__ void ln_leaf4(float a, float b, float c, float d, float& s, float& q) {
#pragma clang fp contract(off)
    s = (a + b) + (c + d);
    q = (a * a + b * b) + (c * c + d * d);
}
// LayerNorm coefficients of one row from its per-64-column partial sums: y = r * (acc - mean * s[n]) + c[n] = fma(r, acc, fma(nrm, s[n], c[n]))
// with r = 1 / sqrt(var + eps), nrm = -mean * r.  A row of the statistics buffer holds 12 or 24 group slots (ln_gs: 12 for hidden <= 768;
// slots past hidden / 64 are zero and stay zero), so a reader needs no bounds: it takes whole halves of twelve slots with immediate
// offsets from one address -- anything computed per group would be hoisted out of the persistent tile loops as dozens of loop invariants.
// In steps, so that a caller can put work between the requests and their use and never holds more than twelve slots in registers:
// load half 0 / add it / (24-slot rows: load half 1 / add it) / finish.  Slots are added in ascending order in f32.
struct LnRaw {  // one HALF of a row's slots: [0, 12) or [12, 24)
    float2 g[12];
};
struct LnAcc {
    float S, Q;
};
template <int HALF>
static __device__ __forceinline__ void ln_row_load(const float* __restrict__ stats, int row, int gs, LnRaw& raw) {
    const float2* p = (const float2*)(stats + (size_t)row * gs * 2) + 12 * HALF;
#pragma unroll
    for (int g = 0; g < 12; ++g) raw.g[g] = p[g];
}
// (f32, slot after slot in ascending order: the slots are f32 sums of 64 elements each, at most 24 of them; double costs a one-wave-per-SIMD
//  epilogue ~ 1.5 us per tile in dependent half-rate adds and buys nothing next to the 2^-11 rounding of the operand)
template <int HALF>
static __device__ __forceinline__ void ln_row_add(const LnRaw& raw, LnAcc& a) {
#pragma clang fp contract(off)
    if (HALF == 0) a.S = a.Q = 0.0f;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
        a.S += raw.g[g].x;
        a.Q += raw.g[g].y;
    }
}
static __device__ __forceinline__ void ln_row_finish(const LnAcc& a, float inv_h, float eps, float& r, float& nrm) {
#pragma clang fp contract(off)
    const float mean = a.S * inv_h;
    float var = a.Q * inv_h - mean * mean;
    var = var > 0.0f ? var : 0.0f;
    const float rs = 1.0f / sqrtf(var + eps);
    r = rs;
    nrm = -mean * rs;
}

// The same from partial sums that already sit in LDS (the small-tile kernel stages its tile's rows there by LDS-DMA): a plain loop, no
// register staging.  Same summation order, same bits.
static __device__ __forceinline__ void ln_row_coeffs_lds(const float2* p, int gs, float inv_h, float eps, float& r, float& nrm) {
#pragma clang fp contract(off)
    // all twelve (or twenty-four) slots requested before the first add: the reads pipeline instead of taking one LDS round trip each
    const float4* q = (const float4*)p;
    float4 v[12];
#pragma unroll
    for (int g = 0; g < 6; ++g) v[g] = q[g];
    if (gs > 12) {
#pragma unroll
        for (int g = 6; g < 12; ++g) v[g] = q[g];
    }
    LnAcc a;
    a.S = a.Q = 0.0f;
#pragma unroll
    for (int g = 0; g < 6; ++g) {  // (slot after slot, the zero ones included: the order and the bits of ln_row_add)
        a.S += v[g].x;
        a.Q += v[g].y;
        a.S += v[g].z;
        a.Q += v[g].w;
    }
    if (gs > 12) {
#pragma unroll
        for (int g = 6; g < 12; ++g) {
            a.S += v[g].x;
            a.Q += v[g].y;
            a.S += v[g].z;
            a.Q += v[g].w;
        }
    }
    ln_row_finish(a, inv_h, eps, r, nrm);
}

// ggml tanh-GELU (ggml_gelu_f32)
static __device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.79788456080286535587989211986876f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * x * (1.0f + k1 * x * x)));
}

}  // namespace dinov2
