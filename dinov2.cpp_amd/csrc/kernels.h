// Launcher interface of the hand-written gfx950 kernels (gemm.hip, attention.hip, kernels_misc.hip).
// All launchers enqueue on `stream` and return hipGetLastError(); none of them allocates or syncs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dinov2 {

enum DType : int { DT_F16 = 0, DT_BF16 = 1 };

// GEMM epilogues.  C[m,n] = sum_k A[m,k] * W[n,k] (f32 accumulate), then:
enum Epilogue : int {
    EPI_PATCH = 0,   // x[b*T + 1+R+p, n] = acc + bias[n] + pos[1+p, n]           (f32 out; m = b*P + p)
    EPI_QKV = 1,     // out[m, n] = T((acc + bias[n]) * (n < qcols ? qscale : 1))  (T out)
    EPI_RESID = 2,   // x[m, n] += ls[n] * (acc + bias[n])                         (f32 in/out)
    EPI_GELU = 3,    // out[m, n] = T(f16(gelu_tanh(f16(acc + bias[n]))))          (T out; ggml f16-LUT contract)
    EPI_SWIGLU = 4,  // out[m, j] = T(silu(h1) * h2), rows of W interleaved in 32-blocks x1|x2 (T out, width N/2)
    EPI_PLAIN_F32 = 5,  // out[m, n] = acc + bias[n]                                (f32 out; tests / head)
    // ---- LayerNorm folded into the GEMMs on either side of it ("LN fold"; replaces ggml_norm + mul + add,
    // /root/reference/dinov2.cpp:694-700, 722-728, as separate launches).  With LN(x) = gamma (x - mu) r + beta feeding a weight matmul,
    //     sum_k LN(x)[m,k] W[n,k] + b[n]  =  r_m (sum_k (gamma_k x[m,k]) W[n,k]  -  mu_m s[n]) + c[n],
    //     s[n] = sum_k gamma_k W[n,k],   c[n] = b[n] + sum_k beta_k W[n,k]           (both computed once, at load time)
    // so the PRODUCER of x (the residual epilogue) also writes the operand T(gamma x) and per-row partial sums, and the CONSUMER
    // (QKV / FFN-in) applies r_m, mu_m, s, c in its epilogue.
    EPI_RESID_LN = 6,   // EPI_RESID, plus: xg[m, n] = T(x[m, n] * ln_gamma[n]);  stats[m][n / 64] = (sum, sum of squares) of x[m, 64 g .. 64 g + 63]
    EPI_QKV_LN = 7,     // v = r_m * (acc - mu_m * ln_s[n]) + ln_c[n], then as EPI_QKV / EPI_GELU / EPI_SWIGLU with v in place of acc + bias[n]
    EPI_GELU_LN = 8,
    EPI_SWIGLU_LN = 9
};
// the epilogue an LN-fold variant specialises (dispatch decisions depend on this one only)
inline Epilogue epi_base(Epilogue e) {
    return e == EPI_RESID_LN ? EPI_RESID : e == EPI_QKV_LN ? EPI_QKV : e == EPI_GELU_LN ? EPI_GELU : e == EPI_SWIGLU_LN ? EPI_SWIGLU : e;
}
inline bool epi_ln_consumer(Epilogue e) { return e == EPI_QKV_LN || e == EPI_GELU_LN || e == EPI_SWIGLU_LN; }
constexpr int LN_GROUP = 64;      // columns per partial-sum group of EPI_RESID_LN's row statistics
constexpr int LN_MAX_GROUPS = 24; // hidden sizes up to 1 536 (ViT-g); larger models keep the LayerNorm launches
inline int ln_stat_slots(int hidden) { return hidden / LN_GROUP <= 12 ? 12 : 24; }  // slots per row of the statistics buffer (device_types.h, ln_row_load)

struct GemmArgs {
    const void* A;      // [M, K]  T, row-major, K % 64 == 0
    const void* W;      // [N, K]  T, row-major (ggml ne = [K, N])
    const float* bias;  // [N] or nullptr
    void* out;
    const float* aux;   // EPI_PATCH: pos [1+P, N]; EPI_RESID: layer-scale lambda [N]
    int M, N, K;
    int lda, ldw;       // row strides of A and W in elements; 0 = K (dense)
    int ldo;            // leading dimension of out in elements
    int P, T, R;        // EPI_PATCH token mapping
    int qcols;          // EPI_QKV: columns [0, qcols) are multiplied by qscale
    float qscale;
    int small_only;     // launch_gemm internal: this is the tail of a split launch, use the small-tile kernel
    int nt_out;         // launch_gemm internal: 2-byte outputs leave with non-temporal stores (set when the output is larger than the L2s)
    int sub;            // launch_gemm internal: one part of a split launch (inherits nt_out from the whole)
    int clk_slot;       // launch_gemm internal: the clock-probe slot of the LOGICAL launch (device_types.h), decided before any split
    // ---- LN fold (EPI_RESID_LN and the *_LN consumers).  Row statistics: stats[(m * ln_gs + g) * 2 + {0, 1}] = sum / sum of squares
    // of x[m, 64 g .. 64 g + 63] (f32, a fixed pairwise tree over the 64 columns: every kernel produces the same bits); ln_gs = slots per
    // row = ln_stat_slots(hidden): 12 or 24, the slots past hidden / 64 zero (set once, never written)
    int ln_gs;
    const float* ln_gamma;  // EPI_RESID_LN: weight [N] of the LayerNorm that FOLLOWS this residual update
    void* xg;               // EPI_RESID_LN: [M, N] T, T(x * ln_gamma) -- the consumer's A operand
    float* stats;           // EPI_RESID_LN: written ([M][N / 64][2]); consumers: read ([M][K / 64][2])
    const float* ln_s;      // consumers: s[n] = sum_k gamma_k W[n, k]
    const float* ln_c;      // consumers: c[n] = bias[n] + sum_k beta_k W[n, k]   (`bias` is not read)
    float ln_eps;           // consumers: LayerNorm epsilon
};

hipError_t launch_gemm(DType dt, Epilogue epi, const GemmArgs& a, hipStream_t stream);
// must be called once per device before the first launch_gemm (raises the dynamic-LDS limit)
hipError_t gemm_init();
// Which kernel(s) launch_gemm would run for this problem, without launching anything: a ';'-separated list of kernel plans, e.g.
// "gemm4_mixed<256+192>" or "gemm4<256>;small<64x128,w2x2,st3,ks1>" (pointers in `a` are not dereferenced).  Returns hipErrorInvalidValue
// for shapes launch_gemm refuses.  The CPU-side coverage test enumerates the model shapes with it (tests/test_gemm_plans.py).
hipError_t gemm_plan_describe(DType dt, Epilogue epi, const GemmArgs& a, char* out, size_t cap);

// Testing aids that used to be read from the environment on every launch: read ONCE (first use), changed afterwards only through
// tune_set (dinov2_hip_op_set_tuning, include/dinov2_hip_ops.h).  0 = the library's own choice.
enum TuneKey : int {
    TUNE_GEMM_GEN = 0,   // DINOV2_HIP_GEMM_GEN: 2 | 4 = force that generation of the persistent GEMM wherever it can run
    TUNE_GEMM_TILE = 1,  // DINOV2_HIP_GEMM_TILE: 128 | 256 (tuning builds: 129 | 192)
    TUNE_ATTN_V = 2,     // DINOV2_HIP_ATTN_V: 1 .. 4
    TUNE_ATTN_NWV = 3,   // DINOV2_HIP_ATTN_NWV: 2 | 3 | 4
    TUNE_COUNT = 4
};
int tune_get(TuneKey k);
void tune_set(TuneKey k, int v);

// y[r, :] = T(((x - mean) * rsqrt(var + eps)) * w + b)     x f32 [rows, H]; one wave per row
hipError_t launch_layernorm(DType dt, const float* x, const float* w, const float* b, void* y, int rows, int H,
                            float eps, hipStream_t stream);
// same, f32 output (final layernorm)
hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float* y, int rows, int H, float eps,
                                hipStream_t stream);
// LN fold (EPI_RESID_LN): what the residual epilogue leaves behind, computed from a residual stream no GEMM has written (before layer 0):
// xg [rows, H] T = T(x gamma), stats [rows][gs][2] = (sum, sum of squares) per 64 columns in the producers' summation order (gs = ln_stat_slots(H))
hipError_t launch_ln_prepare(DType dt, const float* x, const float* gamma, void* xg, float* stats, int gs, int rows, int H, hipStream_t stream);
// load time: s[n] = sum_k gamma[k] W[n, k], c[n] = bias[n] + sum_k beta[k] W[n, k]   (W [N, K] T, dense; bias may be nullptr)
hipError_t launch_ln_fold_vectors(DType dt, const void* W, const float* bias, const float* gamma, const float* beta, float* s, float* c, int N, int K,
                                  hipStream_t stream);
// fused multi-head attention over token-major qkv [B*T, 3H] (T dtype, q pre-scaled), out [B*T, H]; hd == 64.
// log2_scores: q was scaled by log2(e)/sqrt(hd) instead of 1/sqrt(hd), so softmax uses exp2 directly.
hipError_t launch_attention(DType dt, const void* qkv, void* out, int B, int T, int H, int nh, bool log2_scores,
                            hipStream_t stream);

// im2col of conv_2d_sk_p0: img f32 (layout 0 = BGR HWC interleaved, 1 = RGB CHW planar) -> col [B*P, Kpad] T,
// patch vector order (c_rgb, ky, kx), zero padded to Kpad
hipError_t launch_im2col(DType dt, const float* img, void* col, int B, int Hh, int Ww, int patch, int Kpad, int layout,
                         hipStream_t stream);
// dino_preprocess / dino_classify_preprocess on the device: u8 BGR [B,h,w,3] -> normalised f32 BGR [B,oh,ow,3]
// (bicubic to rh x rw, crop at (y0, x0))
hipError_t launch_preprocess_u8(const uint8_t* src, float* dst, int B, int h, int w, int rh, int rw, int y0, int x0, int oh,
                                int ow, hipStream_t stream);
// x[b*T + 0] = cls + pos[0]; x[b*T + 1 + r] = reg[r]
hipError_t launch_init_tokens(float* x, const float* cls, const float* pos, const float* reg, int B, int T, int R,
                              int H, hipStream_t stream);

// load-time conversion of one GGUF tensor: src (ggml type) rows [N, K] -> dst T [N, Kpad], zero padded.
// interleave32 > 0: destination row order x1|x2 interleaved in 32-row blocks (SwiGLU weights_in), value = F.
hipError_t launch_convert_weight(DType dt, const void* src, uint32_t ggml_type, void* dst, int N, int K, int Kpad,
                                 int interleaveF, hipStream_t stream);
// f32 vector copy with the same optional interleave (bias of weights_in)
hipError_t launch_permute_bias(const float* src, float* dst, int N, int interleaveF, hipStream_t stream);

// classifier head (forward_head): fin = final-LN tokens f32 [B, T, H]
//   pooled[b, h] = sum_{t in [first, T)} fin[b, t, h] * inv_div ; feat = [cls ; pooled] rounded to T
//   logits = W feat + bias ; probs = softmax(logits)
hipError_t launch_head(DType dt, const float* fin, const void* W, const float* bias, float* feat_scratch,
                       float* logits, float* probs, int B, int T, int H, int C, int first, float inv_div,
                       hipStream_t stream);

// PCA support: mean[h] = column mean of tok [P, H]; xt [H, Ppad] f16 = (tok - mean)^T, zero padded in P
hipError_t launch_pca_prepare(const float* tok, float* mean, void* xt, int P, int H, int Ppad, hipStream_t stream);

// Block iteration for the leading eigenvectors of cov [H, H] (see pca_power_kernel): block width, rows per workgroup, grid size
constexpr int PCA_NB = 8, PCA_ROWS = 16;
inline int pca_blocks(int H) { return (H + PCA_ROWS - 1) / PCA_ROWS; }
// g [8][8] = Y^T Y  ->  rinv [8][8] upper triangular with Y rinv orthonormal (g = R^T R).  A direction whose pivot falls below
// 1e-24 of the largest diagonal entry is dropped (its column of rinv is zero), so rank-deficient blocks stay finite.  Shared by
// the kernel and the host-side Rayleigh-Ritz step so that both see the same Q.
__host__ __device__ inline void pca_chol_rinv(const double* g, double* rinv) {
    double R[PCA_NB][PCA_NB];
    bool dead[PCA_NB];
    double big = 0.0;
    for (int a = 0; a < PCA_NB; ++a) big = g[a * PCA_NB + a] > big ? g[a * PCA_NB + a] : big;
    for (int a = 0; a < PCA_NB; ++a) {
        double d = g[a * PCA_NB + a];
        for (int k = 0; k < a; ++k) d -= R[k][a] * R[k][a];
        dead[a] = !(d > 1e-24 * big);
        const double inv = dead[a] ? 0.0 : 1.0 / sqrt(d);
        for (int b = 0; b < PCA_NB; ++b) R[a][b] = 0.0;
        if (dead[a]) continue;
        R[a][a] = d * inv;
        for (int b = a + 1; b < PCA_NB; ++b) {
            double v = g[a * PCA_NB + b];
            for (int k = 0; k < a; ++k) v -= R[k][a] * R[k][b];
            R[a][b] = v * inv;
        }
    }
    for (int a = 0; a < PCA_NB; ++a)
        for (int b = 0; b < PCA_NB; ++b) rinv[a * PCA_NB + b] = 0.0;
    for (int a = 0; a < PCA_NB; ++a) {
        if (dead[a]) continue;
        rinv[a * PCA_NB + a] = 1.0 / R[a][a];
        for (int b = a + 1; b < PCA_NB; ++b) {
            if (dead[b]) continue;
            double v = 0.0;
            for (int k = a; k < b; ++k) v += rinv[a * PCA_NB + k] * R[k][b];
            rinv[a * PCA_NB + b] = -v / R[b][b];
        }
    }
}
hipError_t launch_pca_power(const float* cov, const double* yprev, const double* gprev, double* ynext, double* gnext, int H,
                            hipStream_t stream);
hipError_t launch_pca_project(const float* tok, const float* mean, const float* comp, float* proj, int P, int H,
                              hipStream_t stream);

// clock probe (device_types.h): per translation unit, [CLK_SLOTS][4] = running sums of shader cycles and 100 MHz ticks of workgroup 0 over
// all launches of each kernel kind on the current device, the 100 MHz end stamp of the last one, the launch count
hipError_t gemm_clock_probe_read(unsigned long long* out);
hipError_t gemm4_clock_probe_read(unsigned long long* out);
hipError_t attention_clock_probe_read(unsigned long long* out);

// debugging aid: what ds_read_b64_tr_b16 returns per lane for addr = lane*8 over an LDS image holding its own
// element index (out: [64][4] int16)
hipError_t launch_probe_tr16(int16_t* out, hipStream_t stream);

}  // namespace dinov2
