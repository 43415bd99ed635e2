// kernels_misc.hip -- the HBM-bound kernels around the GEMMs: LayerNorm, im2col, token init, load-time weight
// conversion / dequantisation, classifier head.  All are wavefront (64-lane) designs with 16-byte accesses.
#include <type_traits>

#include "device_types.h"
#include "gguf_reader.h"
#include "kernels.h"

namespace dinov2 {

// Sum of a double over the 64 lanes of a wave, result in every lane.  DPP moves on the two 32-bit halves (quad swaps, half-row
// and row mirrors, then one readlane per 16-lane row) instead of six ds_bpermute round trips through the LDS pipe: the
// LayerNorm is two such reductions per row, and at batch 1 their latency was a third of the kernel.
static __device__ __forceinline__ double wave_sum_f64(double v) {
    auto dpp = [](double x, auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
        const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, C, 0xF, 0xF, false);
        const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), C, 0xF, 0xF, false);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane of a 16-lane row holds the row's sum
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    double r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 16 * i), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 16 * i);
        r[i] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm: ggml_norm + mul + add  (/root/reference/dinov2.cpp:694-700, 722-728, 756-760)
// one wave per token row, row kept in registers (H <= 1536 -> <= 6 float4 per lane), statistics in double like
// ggml (mean, then centred sum of squares), 1/sqrtf(var + eps) in f32.
// ---------------------------------------------------------------------------------------------------------
template <typename OutT, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bta, OutT* __restrict__ y, int rows,
                                                        int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = H >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * H);
    float4 v[MAXV], gw[MAXV], gb[MAXV];
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {  // all loads of the row first: x, and the affine parameters needed only at the end
        const int i = lane + 64 * j;
        if (i < nv) {
            v[j] = xr[i];
            gw[j] = ((const float4*)w)[i];
            gb[j] = ((const float4*)bta)[i];
        }
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) sum += (double)v[j].x + (double)v[j].y + (double)v[j].z + (double)v[j].w;
    }
    sum = wave_sum_f64(sum);
    const float mean = (float)(sum / H);
    double sq = 0.0;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
            sq += (double)(v[j].x * v[j].x) + (double)(v[j].y * v[j].y) + (double)(v[j].z * v[j].z) +
                  (double)(v[j].w * v[j].w);
        }
    }
    sq = wave_sum_f64(sq);
    const float var = (float)(sq / H);
    const float scale = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;
        if (i < nv) {
            const float4 ww = gw[j], bb = gb[j];
            float r0 = v[j].x * scale * ww.x + bb.x, r1 = v[j].y * scale * ww.y + bb.y;
            float r2 = v[j].z * scale * ww.z + bb.z, r3 = v[j].w * scale * ww.w + bb.w;
            // f32 result first, f16 rounding second (ggml rounds at the NEXT mul_mat): block v_fma_mix*_f16 fusion
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
            if constexpr (sizeof(OutT) == 4) {
                ((float4*)(y + (size_t)row * H))[i] = make_float4(r0, r1, r2, r3);
            } else {
                typedef OutT o4 __attribute__((ext_vector_type(4)));
                o4 pk;
                pk[0] = (OutT)r0; pk[1] = (OutT)r1; pk[2] = (OutT)r2; pk[3] = (OutT)r3;
                ((o4*)(y + (size_t)row * H))[i] = pk;
            }
        }
    }
}

template <typename OutT>
static hipError_t ln_dispatch(const float* x, const float* w, const float* b, OutT* y, int rows, int H, float eps,
                              hipStream_t st) {
    if (H % 4 != 0 || H > 64 * 4 * 8) return hipErrorInvalidValue;
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (H / 4 + 63) / 64;
    if (nv <= 2) hipLaunchKernelGGL((layernorm_kernel<OutT, 2>), grid, block, 0, st, x, w, b, y, rows, H, eps);
    else if (nv <= 4) hipLaunchKernelGGL((layernorm_kernel<OutT, 4>), grid, block, 0, st, x, w, b, y, rows, H, eps);
    else hipLaunchKernelGGL((layernorm_kernel<OutT, 8>), grid, block, 0, st, x, w, b, y, rows, H, eps);
    return hipGetLastError();
}

hipError_t launch_layernorm(DType dt, const float* x, const float* w, const float* b, void* y, int rows, int H, float eps,
                            hipStream_t st) {
    return dt == DT_F16 ? ln_dispatch<_Float16>(x, w, b, (_Float16*)y, rows, H, eps, st)
                        : ln_dispatch<__bf16>(x, w, b, (__bf16*)y, rows, H, eps, st);
}

hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int H, float eps,
                                hipStream_t st) {
    return ln_dispatch<float>(x, w, b, y, rows, H, eps, st);
}

// ---------------------------------------------------------------------------------------------------------
// LN fold (kernels.h, EPI_RESID_LN): the two small kernels beside the GEMM epilogues.
//  * ln_prepare_kernel: what EPI_RESID_LN leaves behind, for a residual stream no GEMM has written yet (the embeddings, before layer 0):
//    xg = T(x gamma) and per row and 64-column group (sum, sum of squares) in the producers' fixed pairwise order (ln_leaf4, then 16 lanes
//    by row-local DPP).  One wave per row.
//  * ln_fold_vectors_kernel (load time): s[n] = sum_k gamma_k W[n, k], c[n] = bias[n] + sum_k beta_k W[n, k] from the CONVERTED weights
//    (the values the MFMA will see), accumulated in double.  One wave per output column.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_prepare_kernel(const float* __restrict__ x, const float* __restrict__ gamma, T* __restrict__ xg,
                                                         float* __restrict__ stats, int gs, int rows, int H) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = H >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * H);
    typedef T o4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int i = lane + 64 * j;  // (H % 64 == 0: a 16-lane row is inside the matrix or outside it as a whole)
        if (i < nv) {
            const float4 v = xr[i], g = ((const float4*)gamma)[i];
            float g0 = v.x * g.x, g1 = v.y * g.y, g2 = v.z * g.z, g3 = v.w * g.w;
            asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3));  // f32 products first, then the rounding
            o4 pk;
            pk[0] = (T)g0; pk[1] = (T)g1; pk[2] = (T)g2; pk[3] = (T)g3;
            ((o4*)(xg + (size_t)row * H))[i] = pk;
            float s4, q4;
            ln_leaf4(v.x, v.y, v.z, v.w, s4, q4);
            s4 += dpp_f32<0xB1>(s4);
            q4 += dpp_f32<0xB1>(q4);
            s4 += dpp_f32<0x4E>(s4);
            q4 += dpp_f32<0x4E>(q4);
            s4 += dpp_f32<0x141>(s4);
            q4 += dpp_f32<0x141>(q4);
            s4 += dpp_f32<0x140>(s4);
            q4 += dpp_f32<0x140>(q4);
            if ((lane & 15) == 0) *(float2*)(stats + ((size_t)row * gs + (i >> 4)) * 2) = make_float2(s4, q4);
        }
    }
}

hipError_t launch_ln_prepare(DType dt, const float* x, const float* gamma, void* xg, float* stats, int gs, int rows, int H, hipStream_t st) {
    if (H % 64 != 0 || H > 64 * 4 * 8) return hipErrorInvalidValue;
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (H / 4 + 63) / 64;
#define DINO_LNP(TT, MV) hipLaunchKernelGGL((ln_prepare_kernel<TT, MV>), grid, block, 0, st, x, gamma, (TT*)xg, stats, gs, rows, H)
    if (dt == DT_F16) {
        if (nv <= 2) DINO_LNP(_Float16, 2); else if (nv <= 4) DINO_LNP(_Float16, 4); else DINO_LNP(_Float16, 8);
    } else {
        if (nv <= 2) DINO_LNP(__bf16, 2); else if (nv <= 4) DINO_LNP(__bf16, 4); else DINO_LNP(__bf16, 8);
    }
#undef DINO_LNP
    return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const T* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ s, float* __restrict__ c, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T* w = W + (size_t)n * K;
    double sg = 0.0, sb = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double wv = (double)(float)w[k];
        sg += (double)gamma[k] * wv;
        sb += (double)beta[k] * wv;
    }
    sg = wave_sum_f64(sg);
    sb = wave_sum_f64(sb);
    if (lane == 0) {
        s[n] = (float)sg;
        c[n] = (float)((bias ? (double)bias[n] : 0.0) + sb);
    }
}

hipError_t launch_ln_fold_vectors(DType dt, const void* W, const float* bias, const float* gamma, const float* beta, float* s, float* c, int N, int K,
                                  hipStream_t st) {
    const dim3 grid((N + 3) / 4), block(256);
    if (dt == DT_F16) hipLaunchKernelGGL((ln_fold_vectors_kernel<_Float16>), grid, block, 0, st, (const _Float16*)W, bias, gamma, beta, s, c, N, K);
    else hipLaunchKernelGGL((ln_fold_vectors_kernel<__bf16>), grid, block, 0, st, (const __bf16*)W, bias, gamma, beta, s, c, N, K);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// im2col of ggml_conv_2d_sk_p0 (/root/reference/dinov2.cpp:636): image -> [B*P, Kpad] in the kernel's type,
// patch vector order (c, ky, kx) with kx fastest == flattening the [H,3,14,14] weight row; c is the RGB index
// (dino_predict repacks BGR-interleaved to RGB-planar first, dinov2.cpp:914-931 -- folded into the gather here).
// ---------------------------------------------------------------------------------------------------------
// One workgroup per (image, patch row, chunk of <= IM2COL_CHUNK patches): the 3 x ps image rows of the chunk are read as contiguous runs
// (coalesced; the first version gathered 8 pixels per thread from 8 addresses and ran at 1.7 TB/s), converted, laid out as the chunk's
// [patches][Kpad] block in LDS, and leave as whole 16-byte pieces of whole rows.  Same values as ever: one (T) rounding of the f32 pixel.
constexpr int IM2COL_CHUNK = 10;
template <typename T>
__global__ __launch_bounds__(1024) void im2col_kernel(const float* __restrict__ img, T* __restrict__ col, int B, int Hh,
                                                     int Ww, int ps, int Kpad, int layout, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char im2col_lds[];
    T* const tile = (T*)im2col_lds;  // [np][Kpad]
    const int w0 = Ww / ps, h0 = Hh / ps, P = w0 * h0;
    const int ck = blockIdx.x % nchunk, py = (blockIdx.x / nchunk) % h0, b = blockIdx.x / (nchunk * h0);
    const int px0 = ck * IM2COL_CHUNK, np = min(IM2COL_CHUNK, w0 - px0);
    const int pp2 = ps * ps, Kreal = 3 * pp2, run = np * ps;
    const int tid = threadIdx.x, nth = blockDim.x;  // 4 or 16 waves (launch_im2col)
    for (int i = tid; i < np * (Kpad - Kreal); i += nth) {  // the K padding of every row
        const int r = i / (Kpad - Kreal), k = Kreal + i - r * (Kpad - Kreal);
        tile[r * Kpad + k] = (T)0.f;
    }
    // One contiguous run of the chunk per wave and step -- planar: 3 * ps runs of np * ps floats (channel c, row ky); interleaved: ps runs of
    // 3 * np * ps -- so that everything that needs a division is wave-uniform; per element only x / ps and e / 3, by multiplication
    // (magic = 65536 / ps + 1: exact for x < 65536 / ps, the launcher checks).  Five independent loads per lane before the first LDS store
    // (a plain strided loop is one dependent global round trip per element).
    const unsigned magic = 65536u / (unsigned)ps + 1u;
    const int nrun = layout == 1 ? 3 * ps : ps, rlen = layout == 1 ? run : 3 * run;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int U = 5;
    for (int rr = wv; rr < nrun; rr += (nth >> 6)) {
        const int c0 = layout == 1 ? rr / ps : 0, ky = layout == 1 ? rr - c0 * ps : rr;
        const float* src = layout == 1 ? img + (((size_t)b * 3 + c0) * Hh + (size_t)py * ps + ky) * Ww + (size_t)px0 * ps
                                       : img + (((size_t)b * Hh + (size_t)py * ps + ky) * Ww + (size_t)px0 * ps) * 3;
        T* const trow = tile + c0 * pp2 + ky * ps;
        for (int e0 = lane; e0 < rlen; e0 += 64 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + 64 * u < rlen) v[u] = src[e0 + 64 * u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 64 * u;
                if (e < rlen) {
                    int xo = e, cc = 0;
                    if (layout != 1) {
                        xo = (int)(((unsigned)e * 21846u) >> 16);  // e / 3 (exact below 32 768)
                        cc = (2 - (e - 3 * xo)) * pp2;             // position 0 / 1 / 2 = B / G / R -> RGB index 2 / 1 / 0
                    }
                    const int pl = (int)(((unsigned)xo * magic) >> 16), kx = xo - pl * ps;
                    trow[pl * Kpad + cc + kx] = (T)v[u];
                }
            }
        }
    }
    __syncthreads();
    const int cpr = Kpad >> 3;  // 16-byte pieces per row
    T* const dst = col + ((size_t)b * P + (size_t)py * w0 + px0) * Kpad;  // the chunk's rows are consecutive in col
    for (int i = tid; i < np * cpr; i += nth) ((uint4*)dst)[i] = ((const uint4*)tile)[i];
}

hipError_t launch_im2col(DType dt, const float* img, void* col, int B, int Hh, int Ww, int patch, int Kpad, int layout,
                         hipStream_t st) {
    if (patch <= 0 || Kpad % 8 != 0 || Kpad < 3 * patch * patch) return hipErrorInvalidValue;
    const int w0 = Ww / patch, h0 = Hh / patch;
    if (w0 <= 0 || h0 <= 0 || B <= 0) return hipErrorInvalidValue;
    const int nchunk = (w0 + IM2COL_CHUNK - 1) / IM2COL_CHUNK;
    const size_t lds = (size_t)IM2COL_CHUNK * Kpad * 2;
    if (lds > 64 * 1024 || 3 * IM2COL_CHUNK * patch >= 32768) return hipErrorInvalidValue;  // (Kpad <= 1 638: patch sizes up to 23)
    for (unsigned x = 0, mg = 65536u / (unsigned)patch + 1u; x < (unsigned)(IM2COL_CHUNK * patch); ++x)
        if (((x * mg) >> 16) != x / (unsigned)patch) return hipErrorInvalidValue;  // (never for patch <= 23; the kernel divides by multiplication)
    // few workgroups (batch 1: 148): 16 waves each, so that a wave's chain of runs is three deep (10 us; 15 with four waves); many (batch 32:
    // 4 736): four waves, more workgroups per CU overlap better (51 us against 69; the gather this replaces took 92 / 10 us)
    const size_t nwg = (size_t)B * h0 * nchunk;
    const dim3 grid((unsigned)nwg), block(nwg >= 1024 ? 256 : 1024);
    if (dt == DT_F16)
        hipLaunchKernelGGL(im2col_kernel<_Float16>, grid, block, lds, st, img, (_Float16*)col, B, Hh, Ww, patch, Kpad, layout, nchunk);
    else
        hipLaunchKernelGGL(im2col_kernel<__bf16>, grid, block, lds, st, img, (__bf16*)col, B, Hh, Ww, patch, Kpad, layout, nchunk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// dino_preprocess / dino_classify_preprocess on the device (/root/reference/dinov2.cpp:106-156), same arithmetic as
// csrc/preprocess.cpp: u8 BGR [B,h,w,3] -> /255 -> bicubic (cv::INTER_CUBIC: A = -0.75, half-pixel centres, clamped
// taps, horizontal pass first) to rh x rw -> crop (y0, x0, oh, ow) -> (c - mean[2-c]) / std[2-c] -> f32 BGR [B,oh,ow,3].
// One thread per output pixel; the 16 source pixels of a thread are 4 runs of <= 4 adjacent BGR triples (L2 resident).
// ---------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void cubic_taps_dev(float t, float w[4]) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B,
                                                            int h, int w, int rh, int rw, int y0, int x0, int oh, int ow) {
#pragma clang fp contract(off)  // same roundings as the host implementation
    const size_t total = (size_t)B * oh * ow;
    const float inv255 = (float)(1.0 / 255.0);
    const float sx = (float)w / (float)rw, sy = (float)h / (float)rh;
    const float mean[3] = {0.406f, 0.456f, 0.485f}, stdv[3] = {0.225f, 0.224f, 0.229f};  // indexed by BGR channel
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % ow), y = (int)((idx / ow) % oh), b = (int)(idx / ((size_t)ow * oh));
        float fx = ((float)(x + x0) + 0.5f) * sx - 0.5f, fy = ((float)(y + y0) + 0.5f) * sy - 0.5f;
        const int ix0 = (int)floorf(fx), iy0 = (int)floorf(fy);
        float wx[4], wy[4];
        cubic_taps_dev(fx - (float)ix0, wx);
        cubic_taps_dev(fy - (float)iy0, wy);
        const uint8_t* img = src + (size_t)b * h * w * 3;
        float acc[3] = {0.f, 0.f, 0.f};
        float rowv[4][3];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int yy = min(max(iy0 - 1 + ky, 0), h - 1);
            const uint8_t* row = img + (size_t)yy * w * 3;
            const int xa = min(max(ix0 - 1, 0), w - 1), xb = min(max(ix0, 0), w - 1), xc = min(max(ix0 + 1, 0), w - 1),
                      xd = min(max(ix0 + 2, 0), w - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                rowv[ky][c] = (float)row[xa * 3 + c] * inv255 * wx[0] + (float)row[xb * 3 + c] * inv255 * wx[1] +
                              (float)row[xc * 3 + c] * inv255 * wx[2] + (float)row[xd * 3 + c] * inv255 * wx[3];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[c] = rowv[0][c] * wy[0] + rowv[1][c] * wy[1] + rowv[2][c] * wy[2] + rowv[3][c] * wy[3];
            dst[idx * 3 + c] = (acc[c] - mean[c]) / stdv[c];
        }
    }
}

hipError_t launch_preprocess_u8(const uint8_t* src, float* dst, int B, int h, int w, int rh, int rw, int y0, int x0, int oh,
                                int ow, hipStream_t st) {
    const size_t total = (size_t)B * oh * ow;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(blocks), dim3(256), 0, st, src, dst, B, h, w, rh, rw, y0, x0, oh, ow);
    return hipGetLastError();
}

// x[b, 0] = cls + pos[0];  x[b, 1 + r] = register_tokens[r]  (no pos-embed on registers)  dinov2.cpp:662-685
__global__ void init_tokens_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                   const float* __restrict__ reg, int T, int R, int H) {
    const int b = blockIdx.y, t = blockIdx.x;  // t in [0, 1+R)
    float* dst = x + ((size_t)b * T + t) * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) dst[i] = t == 0 ? cls[i] + pos[i] : reg[(size_t)(t - 1) * H + i];
}

hipError_t launch_init_tokens(float* x, const float* cls, const float* pos, const float* reg, int B, int T, int R, int H,
                              hipStream_t st) {
    hipLaunchKernelGGL(init_tokens_kernel, dim3(1 + R, B), dim3(256), 0, st, x, cls, pos, reg, T, R, H);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// load-time weight conversion: one GGUF 2-D tensor [N rows, K] of any supported ggml type -> T [N, Kpad].
// Quant block layouts (32 weights per block, d/m are f16): Q4_0 {d, qs[16]}, Q4_1 {d, m, qs[16]},
// Q5_0 {d, qh[4], qs[16]}, Q5_1 {d, m, qh[4], qs[16]}, Q8_0 {d, int8 qs[32]}  (ggml-quants dequantize_row_*;
// the reference leaves dequantisation to ggml's vec_dot, /root/reference/dinov2.cpp:227-236, 355-453).
// ---------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float ld_f16(const uint8_t* p) {
    const uint16_t u = (uint16_t)p[0] | ((uint16_t)p[1] << 8);
    return (float)__builtin_bit_cast(_Float16, u);
}

static __device__ __forceinline__ float dequant_elem(const uint8_t* src, uint32_t type, size_t row, int K, int k) {
    switch (type) {
        case GGML_F32: return ((const float*)src)[row * K + k];
        case GGML_F16: return (float)((const _Float16*)src)[row * K + k];
        case GGML_BF16: return (float)((const __bf16*)src)[row * K + k];
        default: break;
    }
    const size_t blk = row * (size_t)(K >> 5) + (k >> 5);
    const int j = k & 31;
    switch (type) {
        case GGML_Q8_0: {
            const uint8_t* b = src + blk * 34;
            return (float)(int8_t)b[2 + j] * ld_f16(b);
        }
        case GGML_Q4_0: {
            const uint8_t* b = src + blk * 18;
            const int q = j < 16 ? (b[2 + j] & 0xF) : (b[2 + j - 16] >> 4);
            return (float)(q - 8) * ld_f16(b);
        }
        case GGML_Q4_1: {
            const uint8_t* b = src + blk * 20;
            const int q = j < 16 ? (b[4 + j] & 0xF) : (b[4 + j - 16] >> 4);
            return (float)q * ld_f16(b) + ld_f16(b + 2);
        }
        case GGML_Q5_0: {
            const uint8_t* b = src + blk * 22;
            const uint32_t qh = (uint32_t)b[2] | ((uint32_t)b[3] << 8) | ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 24);
            const int lo = j < 16 ? (b[6 + j] & 0xF) : (b[6 + j - 16] >> 4);
            const int q = lo | (int)(((qh >> j) & 1u) << 4);
            return (float)(q - 16) * ld_f16(b);
        }
        case GGML_Q5_1: {
            const uint8_t* b = src + blk * 24;
            const uint32_t qh = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
            const int lo = j < 16 ? (b[8 + j] & 0xF) : (b[8 + j - 16] >> 4);
            const int q = lo | (int)(((qh >> j) & 1u) << 4);
            return (float)q * ld_f16(b) + ld_f16(b + 2);
        }
        default: return 0.f;
    }
}

static __device__ __forceinline__ int interleave_src_row(int n, int F) {
    // destination rows alternate 32-row blocks x1[32q..] | x2[32q..]; source is [x1 (F rows); x2 (F rows)]
    const int blk = n >> 5, within = n & 31;
    return (blk & 1) * F + (blk >> 1) * 32 + within;
}

template <typename T>
__global__ __launch_bounds__(256) void convert_weight_kernel(const uint8_t* __restrict__ src, uint32_t type,
                                                             T* __restrict__ dst, int N, int K, int Kpad, int F) {
    const size_t total = (size_t)N * Kpad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % Kpad);
        const int n = (int)(idx / Kpad);
        const int sr = F > 0 ? interleave_src_row(n, F) : n;
        dst[idx] = k < K ? (T)dequant_elem(src, type, (size_t)sr, K, k) : (T)0.f;
    }
}

hipError_t launch_convert_weight(DType dt, const void* src, uint32_t type, void* dst, int N, int K, int Kpad, int F,
                                 hipStream_t st) {
    const size_t total = (size_t)N * Kpad;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dt == DT_F16)
        hipLaunchKernelGGL(convert_weight_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)src, type,
                           (_Float16*)dst, N, K, Kpad, F);
    else
        hipLaunchKernelGGL(convert_weight_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)src, type,
                           (__bf16*)dst, N, K, Kpad, F);
    return hipGetLastError();
}

__global__ void permute_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int F) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) dst[n] = src[F > 0 ? interleave_src_row(n, F) : n];
}

hipError_t launch_permute_bias(const float* src, float* dst, int N, int F, hipStream_t st) {
    hipLaunchKernelGGL(permute_bias_kernel, dim3((N + 255) / 256), dim3(256), 0, st, src, dst, N, F);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// classifier head = forward_head (/root/reference/dinov2.cpp:792-821)
// ---------------------------------------------------------------------------------------------------------
// feat[b] = [cls ; sum_t fin[b, t, :] * inv_div] rounded to the weight type (ggml mul_mat activation rounding);
// sum_rows accumulates in double like ggml_vec_sum_f32.
template <typename T>
__global__ __launch_bounds__(1024) void head_pool_kernel(const float* __restrict__ fin, float* __restrict__ feat, int T_,
                                                         int H, int first, float inv_div) {
    // 64 columns x 16 token groups per block: lane = column (coalesced 256-byte rows), wave g sums tokens first+g, +16, ...
    // in double, four independent chains per wave so that the loads overlap (a single dependent chain over 1374 tokens
    // took 100 us at batch 1); all partial sums are combined in a fixed order (deterministic, same for every image).
    constexpr int G = 16;
    __shared__ double part[G][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    const float* f = fin + (size_t)b * T_ * H;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (h < H) {
        int t = first + g;
        for (; t + 3 * G < T_; t += 4 * G) {
            const float a0 = f[(size_t)t * H + h], a1 = f[(size_t)(t + G) * H + h];
            const float a2 = f[(size_t)(t + 2 * G) * H + h], a3 = f[(size_t)(t + 3 * G) * H + h];
            s0 += (double)a0;
            s1 += (double)a1;
            s2 += (double)a2;
            s3 += (double)a3;
        }
        for (; t < T_; t += G) s0 += (double)f[(size_t)t * H + h];
    }
    part[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && h < H) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) tot += part[i][lane];
        float c = f[h], pm = (float)tot * inv_div;
        asm volatile("" : "+v"(c), "+v"(pm));  // f32 values first, then the rounding to the weight type
        feat[(size_t)b * 2 * H + h] = (float)(T)c;
        feat[(size_t)b * 2 * H + H + h] = (float)(T)pm;
    }
}

// one wave per class: logits[b, c] = bias[c] + sum_k W[c, k] * feat[b, k]
template <typename T>
__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ feat, const T* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits,
                                                          int K, int C) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    typedef T t8 __attribute__((ext_vector_type(8)));
    const T* wr = W + (size_t)c * K;
    const float* fr = feat + (size_t)b * K;
    float acc = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const t8 wv = *(const t8*)(wr + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)wv[e] * fr[k + e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) logits[(size_t)b * C + c] = acc + bias[c];
}

// ggml_soft_max: max, expf(x - max), sum in double, scale by (float)(1/sum)
__global__ __launch_bounds__(256) void head_softmax_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                           int C) {
    __shared__ float smax[4];
    __shared__ double ssum[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* l = logits + (size_t)b * C;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, l[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) smax[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) s += (double)expf(l[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) ssum[w] = s;
    __syncthreads();
    const float inv = (float)(1.0 / (ssum[0] + ssum[1] + ssum[2] + ssum[3]));
    for (int c = threadIdx.x; c < C; c += 256) probs[(size_t)b * C + c] = expf(l[c] - mx) * inv;
}

hipError_t launch_head(DType dt, const float* fin, const void* W, const float* bias, float* feat, float* logits,
                       float* probs, int B, int T_, int H, int C, int first, float inv_div, hipStream_t st) {
    const dim3 pg((H + 63) / 64, B), lg((C + 3) / 4, B);
    if (dt == DT_F16) {
        hipLaunchKernelGGL(head_pool_kernel<_Float16>, pg, dim3(1024), 0, st, fin, feat, T_, H, first, inv_div);
        hipLaunchKernelGGL(head_logits_kernel<_Float16>, lg, dim3(256), 0, st, feat, (const _Float16*)W, bias, logits,
                           2 * H, C);
    } else {
        hipLaunchKernelGGL(head_pool_kernel<__bf16>, pg, dim3(1024), 0, st, fin, feat, T_, H, first, inv_div);
        hipLaunchKernelGGL(head_logits_kernel<__bf16>, lg, dim3(256), 0, st, feat, (const __bf16*)W, bias, logits, 2 * H, C);
    }
    hipLaunchKernelGGL(head_softmax_kernel, dim3(B), dim3(256), 0, st, logits, probs, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// PCA support (cv::PCA of inference.cpp:76-81): column means of a token matrix and its centred,
// transposed f16 copy Xt[H][Ppad] (zero padded in P), so that the covariance P * C = Xt Xt^T is one plain GEMM on the matrix
// cores (A = W = Xt).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pca_mean_kernel(const float* __restrict__ tok, float* __restrict__ mean, int P, int H) {
    constexpr int G = 16;
    __shared__ double part[G][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int h = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (h < H)
        for (int t = g; t < P; t += G) s += (double)tok[(size_t)t * H + h];
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && h < H) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) tot += part[i][lane];
        mean[h] = (float)(tot / P);
    }
}

__global__ __launch_bounds__(256) void pca_center_transpose_kernel(const float* __restrict__ tok, const float* __restrict__ mean,
                                                                   _Float16* __restrict__ xt, int P, int H, int Ppad) {
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + 8 * i, h = h0 + tx;
        tile[ty + 8 * i][tx] = (p < P && h < H) ? tok[(size_t)p * H + h] - mean[h] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = h0 + ty + 8 * i, p = p0 + tx;
        if (h < H && p < Ppad) xt[(size_t)h * Ppad + p] = (_Float16)tile[tx][ty + 8 * i];
    }
}

hipError_t launch_pca_prepare(const float* tok, float* mean, void* xt, int P, int H, int Ppad, hipStream_t st) {
    hipLaunchKernelGGL(pca_mean_kernel, dim3((H + 63) / 64), dim3(1024), 0, st, tok, mean, P, H);
    hipLaunchKernelGGL(pca_center_transpose_kernel, dim3((Ppad + 31) / 32, (H + 31) / 32), dim3(256), 0, st, tok, mean,
                       (_Float16*)xt, P, H, Ppad);
    return hipGetLastError();
}

// One step of the block (subspace) iteration for the leading eigenvectors of the symmetric cov [H, H]:
//     Q = Y_prev R^-1   (R from the Cholesky factorisation of Y_prev^T Y_prev: CholeskyQR),    Y_next = cov Q
// in ONE launch and without a grid-wide dependency inside it: every workgroup rebuilds the 8 x 8 Gram matrix from the
// per-workgroup partial sums the previous launch left in g_prev (fixed summation order: deterministic), factors it, applies
// R^-1 to its own 16 rows of cov Y_prev (cov (Y R^-1) = (cov Y) R^-1), and leaves its partial Gram sums of Y_next in g_next.
// Y is double [H][8]; cov stays f32 and stays in L2 / MALL across the iterations (4 MB at H = 1024).
__global__ __launch_bounds__(256) void pca_power_kernel(const float* __restrict__ cov, const double* __restrict__ yprev,
                                                        const double* __restrict__ gprev, double* __restrict__ ynext,
                                                        double* __restrict__ gnext, int H) {
    __shared__ double gs[64], rinv[64], ys[PCA_ROWS][PCA_NB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 64) {
        double s = 0.0;
        for (int blk = 0; blk < (int)gridDim.x; ++blk) s += gprev[(size_t)blk * 64 + t];
        gs[t] = s;
    }
    __syncthreads();
    if (t == 0) pca_chol_rinv(gs, rinv);
    // rows of this wave: four, against one pass over Y_prev
    const int i0 = blockIdx.x * PCA_ROWS + w * 4;
    const float* crow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) crow[r] = cov + (size_t)min(i0 + r, H - 1) * H;
    double acc[4][PCA_NB];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < PCA_NB; ++c) acc[r][c] = 0.0;
    for (int j = lane; j < H; j += 64) {
        double y[PCA_NB];
        const double2* yp = (const double2*)(yprev + (size_t)j * PCA_NB);
#pragma unroll
        for (int c = 0; c < PCA_NB / 2; ++c) {
            const double2 v = yp[c];
            y[2 * c] = v.x;
            y[2 * c + 1] = v.y;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double cij = (double)crow[r][j];
#pragma unroll
            for (int c = 0; c < PCA_NB; ++c) acc[r][c] = fma(cij, y[c], acc[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < PCA_NB; ++c) acc[r][c] = wave_sum_f64(acc[r][c]);
    __syncthreads();  // rinv ready
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double q = 0.0;  // lane b < 8 owns column b:  (t R^-1)[b] = sum_{a <= b} t[a] rinv[a][b]
#pragma unroll
        for (int a = 0; a < PCA_NB; ++a) q = fma(acc[r][a], lane < PCA_NB ? rinv[a * PCA_NB + lane] : 0.0, q);
        const bool live = i0 + r < H;
        if (lane < PCA_NB) {
            if (live) ynext[(size_t)(i0 + r) * PCA_NB + lane] = q;
            ys[w * 4 + r][lane] = live ? q : 0.0;
        }
    }
    __syncthreads();
    if (t < 64) {
        const int a = t >> 3, b = t & 7;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < PCA_ROWS; ++r) s = fma(ys[r][a], ys[r][b], s);
        gnext[(size_t)blockIdx.x * 64 + t] = s;
    }
}

hipError_t launch_pca_power(const float* cov, const double* yprev, const double* gprev, double* ynext, double* gnext, int H,
                            hipStream_t st) {
    hipLaunchKernelGGL(pca_power_kernel, dim3(pca_blocks(H)), dim3(256), 0, st, cov, yprev, gprev, ynext, gnext, H);
    return hipGetLastError();
}

// proj[p][c] = sum_j (tok[p][j] - mean[j]) comp[c][j], c < 3: one wave per token row, double accumulation
__global__ __launch_bounds__(256) void pca_project_kernel(const float* __restrict__ tok, const float* __restrict__ mean,
                                                          const float* __restrict__ comp, float* __restrict__ proj, int P, int H) {
    const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int j = lane; j < H; j += 64) {
        const double d = (double)tok[(size_t)p * H + j] - (double)mean[j];
        a0 = fma(d, (double)comp[j], a0);
        a1 = fma(d, (double)comp[H + j], a1);
        a2 = fma(d, (double)comp[2 * H + j], a2);
    }
    a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2);
    if (lane == 0) {
        proj[(size_t)p * 3] = (float)a0;
        proj[(size_t)p * 3 + 1] = (float)a1;
        proj[(size_t)p * 3 + 2] = (float)a2;
    }
}

hipError_t launch_pca_project(const float* tok, const float* mean, const float* comp, float* proj, int P, int H, hipStream_t st) {
    hipLaunchKernelGGL(pca_project_kernel, dim3((P + 3) / 4), dim3(256), 0, st, tok, mean, comp, proj, P, H);
    return hipGetLastError();
}

}  // namespace dinov2
