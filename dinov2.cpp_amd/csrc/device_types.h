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
