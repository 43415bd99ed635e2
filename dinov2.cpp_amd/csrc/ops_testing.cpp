// Diagnostic entry points (include/dinov2_hip_ops.h): run ONE kernel on host-provided f32 data so that the parity
// tests can check each hand-written kernel against the oracle / numpy in isolation.  Not used by predict.
#include <chrono>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dinov2_hip.h"
#include "../../include/dinov2_hip_ops.h"
#include "device_types.h"
#include "kernels.h"

using namespace dinov2;

namespace {

template <typename T>
std::vector<T> to_t(const float* src, size_t n) {
    std::vector<T> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (T)src[i];
    return v;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};

#define OP_TRY(x)                          \
    do {                                   \
        if ((x) != hipSuccess) return -1;  \
    } while (0)

hipError_t upload_as(DType dt, const float* src, size_t n, DevBuf& d) {
    hipError_t e = d.alloc(n * 2);
    if (e != hipSuccess) return e;
    if (dt == DT_F16) {
        auto v = to_t<_Float16>(src, n);
        return hipMemcpy(d.p, v.data(), n * 2, hipMemcpyHostToDevice);
    }
    auto v = to_t<__bf16>(src, n);
    return hipMemcpy(d.p, v.data(), n * 2, hipMemcpyHostToDevice);
}

hipError_t download_as(DType dt, const void* dev, size_t n, float* dst) {
    std::vector<uint16_t> raw(n);
    hipError_t e = hipMemcpy(raw.data(), dev, n * 2, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    for (size_t i = 0; i < n; ++i) {
        if (dt == DT_F16) {
            _Float16 h;
            std::memcpy(&h, &raw[i], 2);
            dst[i] = (float)h;
        } else {
            uint32_t u = (uint32_t)raw[i] << 16;
            std::memcpy(&dst[i], &u, 4);
        }
    }
    return hipSuccess;
}

}  // namespace

extern "C" int dinov2_hip_op_gemm(int32_t dtype, int32_t epilogue, const float* A, const float* W, const float* bias,
                                  const float* aux, int64_t aux_count, float* out, int32_t out_rows, int32_t ldo,
                                  int32_t M, int32_t N, int32_t K, int32_t P, int32_t T, int32_t R, int32_t qcols,
                                  float qscale) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    if (gemm_init() != hipSuccess) return -1;
    DevBuf dA, dW, dB, dX, dO;
    OP_TRY(upload_as(dt, A, (size_t)M * K, dA));
    OP_TRY(upload_as(dt, W, (size_t)N * K, dW));
    if (bias) {
        OP_TRY(dB.alloc(sizeof(float) * N));
        OP_TRY(hipMemcpy(dB.p, bias, sizeof(float) * N, hipMemcpyHostToDevice));
    }
    if (aux) {
        OP_TRY(dX.alloc(sizeof(float) * (size_t)aux_count));
        OP_TRY(hipMemcpy(dX.p, aux, sizeof(float) * (size_t)aux_count, hipMemcpyHostToDevice));
    }
    const bool f32out = epilogue == EPI_PATCH || epilogue == EPI_RESID || epilogue == EPI_PLAIN_F32;
    const size_t on = (size_t)out_rows * ldo;
    OP_TRY(dO.alloc(on * 4));
    if (f32out) OP_TRY(hipMemcpy(dO.p, out, on * 4, hipMemcpyHostToDevice));
    else OP_TRY(hipMemset(dO.p, 0, on * 4));
    GemmArgs a{};
    a.A = dA.p; a.W = dW.p; a.bias = (const float*)dB.p; a.out = dO.p; a.aux = (const float*)dX.p;
    a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.P = P; a.T = T; a.R = R; a.qcols = qcols; a.qscale = qscale;
    OP_TRY(launch_gemm(dt, (Epilogue)epilogue, a, nullptr));
    OP_TRY(hipDeviceSynchronize());
    if (f32out) OP_TRY(hipMemcpy(out, dO.p, on * 4, hipMemcpyDeviceToHost));
    else OP_TRY(download_as(dt, dO.p, on, out));
    return 0;
}

// ---- LN fold (kernels.h): the producer epilogue, the consumer epilogues, the two small kernels, each alone ----
extern "C" int dinov2_hip_op_gemm_resid_ln(int32_t dtype, const float* A, const float* W, const float* bias, const float* ls, const float* gamma,
                                           float* x, float* xg, float* stats, int32_t M, int32_t N, int32_t K) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    if (gemm_init() != hipSuccess) return -1;
    const int gs = ln_stat_slots(N);
    DevBuf dA, dW, dV, dX, dG, dS;
    OP_TRY(upload_as(dt, A, (size_t)M * K, dA));
    OP_TRY(upload_as(dt, W, (size_t)N * K, dW));
    OP_TRY(dV.alloc(sizeof(float) * 3 * (size_t)N));
    float* v = (float*)dV.p;
    OP_TRY(hipMemcpy(v, bias, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(v + N, ls, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(v + 2 * N, gamma, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(dX.alloc(sizeof(float) * (size_t)M * N));
    OP_TRY(hipMemcpy(dX.p, x, sizeof(float) * (size_t)M * N, hipMemcpyHostToDevice));
    OP_TRY(dG.alloc(2 * (size_t)M * N));
    OP_TRY(hipMemset(dG.p, 0, 2 * (size_t)M * N));
    OP_TRY(dS.alloc(sizeof(float) * 2 * (size_t)M * gs));
    OP_TRY(hipMemset(dS.p, 0, sizeof(float) * 2 * (size_t)M * gs));
    GemmArgs a{};
    a.A = dA.p; a.W = dW.p; a.bias = v; a.aux = v + N; a.ln_gamma = v + 2 * N; a.out = dX.p; a.xg = dG.p; a.stats = (float*)dS.p; a.ln_gs = gs;
    a.M = M; a.N = N; a.K = K; a.ldo = N;
    OP_TRY(launch_gemm(dt, EPI_RESID_LN, a, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(hipMemcpy(x, dX.p, sizeof(float) * (size_t)M * N, hipMemcpyDeviceToHost));
    OP_TRY(download_as(dt, dG.p, (size_t)M * N, xg));
    OP_TRY(hipMemcpy(stats, dS.p, sizeof(float) * 2 * (size_t)M * gs, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dinov2_hip_op_gemm_ln_consumer(int32_t dtype, int32_t epilogue, const float* A, const float* W, const float* ln_s, const float* ln_c,
                                              const float* stats, float eps, float* out, int32_t ldo, int32_t M, int32_t N, int32_t K,
                                              int32_t qcols, float qscale) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    if (gemm_init() != hipSuccess) return -1;
    if (!epi_ln_consumer((Epilogue)epilogue)) return -1;
    const int gs = ln_stat_slots(K);
    DevBuf dA, dW, dV, dS, dO;
    OP_TRY(upload_as(dt, A, (size_t)M * K, dA));
    OP_TRY(upload_as(dt, W, (size_t)N * K, dW));
    OP_TRY(dV.alloc(sizeof(float) * 2 * (size_t)N));
    float* v = (float*)dV.p;
    OP_TRY(hipMemcpy(v, ln_s, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(v + N, ln_c, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(dS.alloc(sizeof(float) * 2 * (size_t)M * gs));
    OP_TRY(hipMemcpy(dS.p, stats, sizeof(float) * 2 * (size_t)M * gs, hipMemcpyHostToDevice));
    const size_t on = (size_t)M * ldo;
    OP_TRY(dO.alloc(on * 2));
    OP_TRY(hipMemset(dO.p, 0, on * 2));
    GemmArgs a{};
    a.A = dA.p; a.W = dW.p; a.ln_s = v; a.ln_c = v + N; a.stats = (float*)dS.p; a.ln_gs = gs; a.ln_eps = eps; a.out = dO.p;
    a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.qcols = qcols; a.qscale = qscale;
    OP_TRY(launch_gemm(dt, (Epilogue)epilogue, a, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(download_as(dt, dO.p, on, out));
    return 0;
}

extern "C" int dinov2_hip_op_ln_prepare(int32_t dtype, const float* x, const float* gamma, float* xg, float* stats, int32_t rows, int32_t H) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    const int gs = ln_stat_slots(H);
    DevBuf dX, dV, dG, dS;
    OP_TRY(dX.alloc(sizeof(float) * (size_t)rows * H));
    OP_TRY(hipMemcpy(dX.p, x, sizeof(float) * (size_t)rows * H, hipMemcpyHostToDevice));
    OP_TRY(dV.alloc(sizeof(float) * (size_t)H));
    OP_TRY(hipMemcpy(dV.p, gamma, sizeof(float) * (size_t)H, hipMemcpyHostToDevice));
    OP_TRY(dG.alloc(2 * (size_t)rows * H));
    OP_TRY(dS.alloc(sizeof(float) * 2 * (size_t)rows * gs));
    OP_TRY(hipMemset(dS.p, 0, sizeof(float) * 2 * (size_t)rows * gs));
    OP_TRY(launch_ln_prepare(dt, (const float*)dX.p, (const float*)dV.p, dG.p, (float*)dS.p, gs, rows, H, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(download_as(dt, dG.p, (size_t)rows * H, xg));
    OP_TRY(hipMemcpy(stats, dS.p, sizeof(float) * 2 * (size_t)rows * gs, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dinov2_hip_op_im2col(int32_t dtype, const float* img, float* col, int32_t B, int32_t Hh, int32_t Ww, int32_t patch, int32_t Kpad,
                                    int32_t layout) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    if (B <= 0 || Hh <= 0 || Ww <= 0 || patch <= 0) return -1;
    const size_t npix = (size_t)B * 3 * Hh * Ww, rows = (size_t)B * (Hh / patch) * (Ww / patch);
    DevBuf dI, dC;
    OP_TRY(dI.alloc(sizeof(float) * npix));
    OP_TRY(hipMemcpy(dI.p, img, sizeof(float) * npix, hipMemcpyHostToDevice));
    OP_TRY(dC.alloc(2 * rows * (size_t)Kpad));
    OP_TRY(hipMemset(dC.p, 0xff, 2 * rows * (size_t)Kpad));  // (a NaN pattern: every element must be written)
    OP_TRY(launch_im2col(dt, (const float*)dI.p, dC.p, B, Hh, Ww, patch, Kpad, layout, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(download_as(dt, dC.p, rows * (size_t)Kpad, col));
    return 0;
}

extern "C" int dinov2_hip_op_ln_fold_vectors(int32_t dtype, const float* W, const float* bias, const float* gamma, const float* beta, float* s_out,
                                             float* c_out, int32_t N, int32_t K) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    DevBuf dW, dV;
    OP_TRY(upload_as(dt, W, (size_t)N * K, dW));
    OP_TRY(dV.alloc(sizeof(float) * (3 * (size_t)N + 2 * (size_t)K)));
    float* v = (float*)dV.p;
    OP_TRY(hipMemcpy(v, bias, sizeof(float) * N, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(v + 3 * (size_t)N, gamma, sizeof(float) * K, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(v + 3 * (size_t)N + K, beta, sizeof(float) * K, hipMemcpyHostToDevice));
    OP_TRY(launch_ln_fold_vectors(dt, dW.p, v, v + 3 * (size_t)N, v + 3 * (size_t)N + K, v + N, v + 2 * (size_t)N, N, K, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(hipMemcpy(s_out, v + N, sizeof(float) * N, hipMemcpyDeviceToHost));
    OP_TRY(hipMemcpy(c_out, v + 2 * (size_t)N, sizeof(float) * N, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int dinov2_hip_op_attention(int32_t dtype, const float* qkv, float* out, int32_t B, int32_t T, int32_t H,
                                       int32_t nh) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    DevBuf dQ, dO;
    const size_t nq = (size_t)B * T * 3 * H, no = (size_t)B * T * H;
    OP_TRY(upload_as(dt, qkv, nq, dQ));
    OP_TRY(dO.alloc(no * 2));
    OP_TRY(hipMemset(dO.p, 0, no * 2));
    OP_TRY(launch_attention(dt, dQ.p, dO.p, B, T, H, nh, false, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(download_as(dt, dO.p, no, out));
    return 0;
}

extern "C" int dinov2_hip_op_layernorm(int32_t dtype, const float* x, const float* w, const float* b, float* out,
                                       int32_t rows, int32_t H, float eps) {
    DevBuf dX, dW, dB, dO;
    const size_t n = (size_t)rows * H;
    OP_TRY(dX.alloc(n * 4));
    OP_TRY(dW.alloc((size_t)H * 4));
    OP_TRY(dB.alloc((size_t)H * 4));
    OP_TRY(dO.alloc(n * 4));
    OP_TRY(hipMemcpy(dX.p, x, n * 4, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(dW.p, w, (size_t)H * 4, hipMemcpyHostToDevice));
    OP_TRY(hipMemcpy(dB.p, b, (size_t)H * 4, hipMemcpyHostToDevice));
    if (dtype < 0) {
        OP_TRY(launch_layernorm_f32((const float*)dX.p, (const float*)dW.p, (const float*)dB.p, (float*)dO.p, rows, H, eps,
                                    nullptr));
        OP_TRY(hipDeviceSynchronize());
        OP_TRY(hipMemcpy(out, dO.p, n * 4, hipMemcpyDeviceToHost));
    } else {
        const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
        OP_TRY(launch_layernorm(dt, (const float*)dX.p, (const float*)dW.p, (const float*)dB.p, dO.p, rows, H, eps, nullptr));
        OP_TRY(hipDeviceSynchronize());
        OP_TRY(download_as(dt, dO.p, n, out));
    }
    return 0;
}

extern "C" int dinov2_hip_op_convert_weight(int32_t dtype, const void* src, uint64_t src_bytes, uint32_t ggml_type,
                                            float* out, int32_t N, int32_t K, int32_t Kpad, int32_t interleaveF) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    DevBuf dS, dO;
    OP_TRY(dS.alloc(src_bytes));
    OP_TRY(hipMemcpy(dS.p, src, src_bytes, hipMemcpyHostToDevice));
    OP_TRY(dO.alloc((size_t)N * Kpad * 2));
    OP_TRY(launch_convert_weight(dt, dS.p, ggml_type, dO.p, N, K, Kpad, interleaveF, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(download_as(dt, dO.p, (size_t)N * Kpad, out));
    return 0;
}

extern "C" int dinov2_hip_op_probe_tr16(int16_t* out256) {
    DevBuf d;
    OP_TRY(d.alloc(512));
    OP_TRY(launch_probe_tr16((int16_t*)d.p, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(hipMemcpy(out256, d.p, 512, hipMemcpyDeviceToHost));
    return 0;
}

// ---- micro-benchmarks: device-resident random operands, HIP-event timing around `iters` launches ----
namespace {
void fill_random_t(DType dt, void* dev, size_t n, unsigned seed, float scale) {
    std::vector<uint16_t> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = (((float)(s >> 8) / 8388608.0f) - 1.0f) * scale;  // uniform [-scale, scale): full-range random
        if (dt == DT_F16) {
            _Float16 x = (_Float16)v;
            std::memcpy(&h[i], &x, 2);
        } else {
            uint32_t u;
            std::memcpy(&u, &v, 4);
            h[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
    }
    (void)hipMemcpy(dev, h.data(), n * 2, hipMemcpyHostToDevice);
}
}  // namespace

extern "C" float dinov2_hip_op_gemm_bench(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, int32_t iters) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    if (gemm_init() != hipSuccess) return -1.f;
    DevBuf dA, dW, dB, dX, dO;
    // extra elements per row of A / W (row strides K + pad instead of the dense K): 0 in normal use; profiles/r02_gemm_kloop.md
    // measured 64 (= 128 bytes) as neutral, i.e. no power-of-two-stride channel conflict to pad away
    const int padA = 0, padW = 0;
    // EPI_PATCH writes token rows (image b, patch p) -> row b * T + 1 + R + p and reads the position embedding [1 + P, N]: shaped like the model
    // (P = 1369 patches, 4 registers) when M is a multiple of 1369, one "image" of M patches otherwise
    const int pP = epilogue == EPI_PATCH ? (M % 1369 == 0 ? 1369 : M) : 1, pR = epilogue == EPI_PATCH && M % 1369 == 0 ? 4 : 0;
    const int pT = pP + 1 + pR;
    const size_t out_elems = epilogue == EPI_PATCH ? (size_t)(M / pP) * pT * N : (size_t)M * N;
    const size_t aux_elems = epilogue == EPI_PATCH ? (size_t)(pP + 1) * N : (size_t)std::max(N, 4096) * 2;
    if (dA.alloc((size_t)M * (K + padA) * 2) != hipSuccess || dW.alloc((size_t)N * (K + padW) * 2) != hipSuccess ||
        dB.alloc((size_t)N * 4) != hipSuccess || dX.alloc(aux_elems * 4) != hipSuccess || dO.alloc(out_elems * 4) != hipSuccess)
        return -1.f;
    fill_random_t(dt, dA.p, (size_t)M * (K + padA), 1, 1.0f);
    fill_random_t(dt, dW.p, (size_t)N * (K + padW), 2, 0.05f);
    (void)hipMemset(dB.p, 0, (size_t)N * 4);
    (void)hipMemset(dX.p, 0, aux_elems * 4);
    (void)hipMemset(dO.p, 0, out_elems * 4);
    GemmArgs a{};
    a.A = dA.p; a.W = dW.p; a.bias = (const float*)dB.p; a.out = dO.p; a.aux = (const float*)dX.p;
    a.M = M; a.N = N; a.K = K; a.ldo = epilogue == EPI_SWIGLU ? N / 2 : N; a.P = 1; a.T = 2; a.R = 0;
    a.lda = K + padA; a.ldw = K + padW;
    a.qcols = N / 3; a.qscale = 0.125f;
    if (epilogue == EPI_PATCH) { a.P = pP; a.T = pT; a.R = pR; }
    DevBuf dSt, dXg, dV;
    if (epilogue >= EPI_RESID_LN) {  // LN fold: statistics of unit-variance rows, gamma = 1, s = 0, c = 0
        const int hc = epilogue == EPI_RESID_LN ? N : K;
        const int gs = ln_stat_slots(hc), groups = hc / LN_GROUP;
        std::vector<float> st((size_t)M * gs * 2, 0.f), ones((size_t)std::max(N, K), 1.0f);
        for (int m = 0; m < M; ++m)
            for (int g = 0; g < groups; ++g) st[((size_t)m * gs + g) * 2 + 1] = 64.0f;
        if (dSt.alloc(st.size() * 4) != hipSuccess || dXg.alloc((size_t)M * N * 2) != hipSuccess || dV.alloc(ones.size() * 4) != hipSuccess) return -1.f;
        (void)hipMemcpy(dSt.p, st.data(), st.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dV.p, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
        a.stats = (float*)dSt.p; a.ln_gs = gs; a.ln_eps = 1e-6f;
        a.ln_gamma = (const float*)dV.p; a.xg = dXg.p;
        a.ln_s = (const float*)dB.p; a.ln_c = (const float*)dB.p;  // zeros
        if (epilogue == EPI_SWIGLU_LN) a.ldo = N / 2;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (auto t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(100);) {
        for (int i = 0; i < 10; ++i) (void)launch_gemm(dt, (Epilogue)epilogue, a, nullptr);  // ~100 ms warm-up (clock ramp)
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) (void)launch_gemm(dt, (Epilogue)epilogue, a, nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms / iters;
}

extern "C" float dinov2_hip_op_attention_bench(int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t nh, int32_t iters) {
    const DType dt = dtype == 1 ? DT_BF16 : DT_F16;
    DevBuf dQ, dO;
    const size_t nq = (size_t)B * T * 3 * H, no = (size_t)B * T * H;
    if (dQ.alloc(nq * 2) != hipSuccess || dO.alloc(no * 2) != hipSuccess) return -1.f;
    fill_random_t(dt, dQ.p, nq, 3, 1.0f);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // warm up for ~100 ms: a cold GPU runs the first milliseconds at a fraction of its sustained clock
    for (auto t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(100);) {
        for (int i = 0; i < 10; ++i) (void)launch_attention(dt, dQ.p, dO.p, B, T, H, nh, true, nullptr);
        (void)hipDeviceSynchronize();
    }
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) (void)launch_attention(dt, dQ.p, dO.p, B, T, H, nh, true, nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms / iters;
}

// ---- testing aids: the tuning switches (read from the environment once) and the dispatcher's plan for a shape ----
extern "C" int dinov2_hip_op_set_tuning(const char* key, int32_t value) {
    static const char* const names[TUNE_COUNT] = {"gemm_gen", "gemm_tile", "attn_v", "attn_nwv"};
    if (!key) return DINOV2_HIP_ERR_INVALID;
    for (int k = 0; k < TUNE_COUNT; ++k)
        if (std::strcmp(key, names[k]) == 0) {
            tune_set((TuneKey)k, value);
            return DINOV2_HIP_OK;
        }
    return DINOV2_HIP_ERR_INVALID;
}
extern "C" int dinov2_hip_op_get_tuning(const char* key) {
    static const char* const names[TUNE_COUNT] = {"gemm_gen", "gemm_tile", "attn_v", "attn_nwv"};
    if (!key) return -1;
    for (int k = 0; k < TUNE_COUNT; ++k)
        if (std::strcmp(key, names[k]) == 0) return tune_get((TuneKey)k);
    return -1;
}
// no device needed: nothing is launched and no pointer is dereferenced
extern "C" int dinov2_hip_op_gemm_plan(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, char* out, int32_t cap) {
    if (!out || cap <= 0 || epilogue < 0 || epilogue > EPI_SWIGLU_LN) return DINOV2_HIP_ERR_INVALID;
    GemmArgs a{};
    a.M = M; a.N = N; a.K = K; a.ldo = epi_base((Epilogue)epilogue) == EPI_SWIGLU ? N / 2 : N;
    a.P = 1; a.T = 2;
    a.qcols = N / 3;
    return gemm_plan_describe(dtype == 1 ? DT_BF16 : DT_F16, (Epilogue)epilogue, a, out, (size_t)cap) == hipSuccess ? DINOV2_HIP_OK : DINOV2_HIP_ERR_INVALID;
}

namespace dinov2 { void pca_ritz(const double* yprev, const double* ynext, const double* g_parts, int nparts, int H, double* evals, double* comp); }
// host-only: the Rayleigh-Ritz step behind dinov2_hip_pca3, for the CPU test-suite
// Clock probe (csrc/device_types.h): the per-translation-unit slot arrays hold running sums; a kernel kind may be served by more than one
// unit (gemm2.hip / gemm4.hip by shape), so the sums of all units are added.
static int read_clock_slots(unsigned long long out[CLK_SLOTS][4]) {
    unsigned long long a[3][CLK_SLOTS * 4] = {};
    if (hipDeviceSynchronize() != hipSuccess || dinov2::gemm_clock_probe_read(a[0]) != hipSuccess ||
        dinov2::gemm4_clock_probe_read(a[1]) != hipSuccess || dinov2::attention_clock_probe_read(a[2]) != hipSuccess)
        return DINOV2_HIP_ERR_HIP;
    for (int s = 0; s < CLK_SLOTS; ++s) {
        out[s][0] = out[s][1] = out[s][2] = out[s][3] = 0;
        for (int u = 0; u < 3; ++u) {
            out[s][0] += a[u][s * 4 + 0];
            out[s][1] += a[u][s * 4 + 1];
            out[s][2] = a[u][s * 4 + 2] > out[s][2] ? a[u][s * 4 + 2] : out[s][2];
            out[s][3] += a[u][s * 4 + 3];
        }
    }
    return DINOV2_HIP_OK;
}
// Running sums for the FFN-in GEMM (the roofline's dominant kernel): shader cycles and 100 MHz ticks of workgroup 0 over all its launches so
// far on the current device; callers take differences over a window.
extern "C" int dinov2_hip_op_clock_probe(uint64_t* cycles, uint64_t* ticks_100mhz) {
    unsigned long long v[CLK_SLOTS][4];
    const int rc = read_clock_slots(v);
    if (rc != DINOV2_HIP_OK) return rc;
    if (cycles) *cycles = v[CLK_FFN_IN][0];
    if (ticks_100mhz) *ticks_100mhz = v[CLK_FFN_IN][1];
    return DINOV2_HIP_OK;
}
// All slots: out18[3 s] = shader cycles, out18[3 s + 1] = 100 MHz ticks, out18[3 s + 2] = launches, running sums; slots 0 .. 5 = QKV, attn-out,
// FFN-in, FFN-out, attention, other GEMM
extern "C" int dinov2_hip_op_clock_slots(uint64_t* out18) {
    if (!out18) return DINOV2_HIP_ERR_INVALID;
    unsigned long long v[CLK_SLOTS][4];
    const int rc = read_clock_slots(v);
    if (rc != DINOV2_HIP_OK) return rc;
    for (int s = 0; s < CLK_SLOTS; ++s) {
        out18[3 * s] = v[s][0];
        out18[3 * s + 1] = v[s][1];
        out18[3 * s + 2] = v[s][3];
    }
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_op_pca_ritz(const double* yprev, const double* ynext, const double* gram, int32_t H, double* evals, double* comp) {
    if (!yprev || !ynext || !gram || !evals || H < 8) return DINOV2_HIP_ERR_INVALID;
    dinov2::pca_ritz(yprev, ynext, gram, 1, H, evals, comp);
    return DINOV2_HIP_OK;
}

// preprocess_u8_kernel alone (mode 0 = dino_preprocess, 1 = dino_classify_preprocess): raw BGR bytes [B, h, w, 3] in, the
// normalised f32 BGR image [B, oh, ow, 3] the forward would consume out -- so the kernel can be compared with
// oracle/preprocess_np.py directly instead of through a whole forward.
extern "C" int dinov2_hip_op_preprocess_u8(int32_t mode, const uint8_t* bgr, int32_t B, int32_t h, int32_t w, int32_t patch,
                                           float* out) {
    int32_t oh = 0, ow = 0;
    if (!bgr || !out || B <= 0 || dinov2_hip_preprocess_size(mode, h, w, patch, &oh, &ow) != DINOV2_HIP_OK) return -1;
    DevBuf dS, dD;
    const size_t nsrc = (size_t)B * h * w * 3, ndst = (size_t)B * oh * ow * 3;
    OP_TRY(dS.alloc(nsrc));
    OP_TRY(dD.alloc(ndst * sizeof(float)));
    OP_TRY(hipMemcpy(dS.p, bgr, nsrc, hipMemcpyHostToDevice));
    const int rh = mode == 1 ? 256 : oh, rw = mode == 1 ? 256 : ow;
    OP_TRY(launch_preprocess_u8((const uint8_t*)dS.p, (float*)dD.p, B, h, w, rh, rw, (rh - oh) / 2, (rw - ow) / 2, oh, ow, nullptr));
    OP_TRY(hipDeviceSynchronize());
    OP_TRY(hipMemcpy(out, dD.p, ndst * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
