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
#define DINO_CLK_END(ARR, SLOT)                                               \
    if (ck_on__) {                                                            \
        const unsigned long long ck_r1__ = __builtin_amdgcn_s_memrealtime();  \
        (ARR)[(SLOT) * 4 + 0] += __builtin_readcyclecounter() - ck_c0__;      \
        (ARR)[(SLOT) * 4 + 1] += ck_r1__ - ck_r0__;                           \
        (ARR)[(SLOT) * 4 + 2] = ck_r1__;                                      \
        (ARR)[(SLOT) * 4 + 3] += 1;                                           \
    }
// which slot a GEMM launch belongs to (EPI: kernels.h Epilogue; the residual epilogue serves attn-out, K = N, and FFN-out, K > N)
#define DINO_CLK_GEMM_SLOT(EPI, N, K) \
    ((EPI) == 1 ? CLK_QKV : (EPI) == 2 ? ((K) > (N) ? CLK_FFN_OUT : CLK_ATTN_OUT) : ((EPI) == 3 || (EPI) == 4) ? CLK_FFN_IN : CLK_OTHER)

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

// ggml tanh-GELU (ggml_gelu_f32)
static __device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.79788456080286535587989211986876f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * x * (1.0f + k1 * x * x)));
}

}  // namespace dinov2
