/*
 * dinov2_hip_ops.h -- diagnostic single-kernel entry points of libdinov2_hip.so.
 *
 * NOT part of the drop-in boundary (that is include/dinov2_hip.h).  These run one hand-written kernel on host f32
 * data (converted to the compute dtype on the way in, back to f32 on the way out) so the parity tests can check each
 * kernel against the oracle in isolation -- the per-op granularity the reference gets from ggml's own op tests and
 * that /root/reference itself never had (it holds no tests at all).  All return 0 on success, -1 on a HIP error.
 */
#ifndef DINOV2_HIP_OPS_H
#define DINOV2_HIP_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* epilogue ids (csrc/kernels.h): 0 patch-embed(+bias+pos, token scatter) 1 qkv(+bias, q scaled) 2 residual
 * (x += ls*(acc+bias)) 3 gelu 4 swiglu 5 plain f32 (6 .. 9: the LN-fold variants, through dinov2_hip_op_gemm_resid_ln / _ln_consumer).  Replaces ggml_mul_mat call sites of dinov2.cpp:471,546,561,570,
 * 582,608,636 with their trailing elementwise nodes.  `out` is [out_rows, ldo] f32, read first for epilogues 0/2/5. */
int dinov2_hip_op_gemm(int32_t dtype, int32_t epilogue, const float *A, const float *W, const float *bias,
                       const float *aux, int64_t aux_count, float *out, int32_t out_rows, int32_t ldo, int32_t M,
                       int32_t N, int32_t K, int32_t P, int32_t T, int32_t R, int32_t qcols, float qscale);

/* LN fold (csrc/kernels.h, epilogues 6 .. 9), each piece alone.  Statistics rows hold `gs` = 12 (hidden <= 768) or 24
 * slots of (sum, sum of squares) per 64 columns, the slots past hidden / 64 zero.
 *   gemm_resid_ln: x [M, N] f32 in/out += ls * (A W^T + bias); xg [M, N] = T(x gamma) (returned as f32); stats [M][gs][2]
 *   gemm_ln_consumer: epilogue 7 (qkv) | 8 (gelu) | 9 (swiglu) on v = r_m (acc - mean_m s[n]) + c[n], mean / r from `stats` [M][gs][2]
 *   ln_prepare: xg and stats of a residual stream no GEMM has written;  ln_fold_vectors: s[n] = sum_k gamma_k W[n,k], c[n] = bias[n] + sum_k beta_k W[n,k] */
int dinov2_hip_op_gemm_resid_ln(int32_t dtype, const float *A, const float *W, const float *bias, const float *ls, const float *gamma, float *x,
                                float *xg, float *stats, int32_t M, int32_t N, int32_t K);
int dinov2_hip_op_gemm_ln_consumer(int32_t dtype, int32_t epilogue, const float *A, const float *W, const float *ln_s, const float *ln_c,
                                   const float *stats, float eps, float *out, int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t qcols,
                                   float qscale);
/* im2col of the patch embedding (ggml_conv_2d_sk_p0, /root/reference/dinov2.cpp:636): img f32, layout 1 = RGB planar [B,3,H,W], 0 = BGR interleaved
 * [B,H,W,3]; col [B * (H/patch) * (W/patch), Kpad] as f32 values of the compute type, k = c * patch^2 + ky * patch + kx, zero beyond 3 * patch^2 */
int dinov2_hip_op_im2col(int32_t dtype, const float *img, float *col, int32_t B, int32_t Hh, int32_t Ww, int32_t patch, int32_t Kpad, int32_t layout);
int dinov2_hip_op_ln_prepare(int32_t dtype, const float *x, const float *gamma, float *xg, float *stats, int32_t rows, int32_t H);
int dinov2_hip_op_ln_fold_vectors(int32_t dtype, const float *W, const float *bias, const float *gamma, const float *beta, float *s_out,
                                  float *c_out, int32_t N, int32_t K);

/* fused attention over token-major qkv [B*T, 3H] (q already scaled) -> [B*T, H]; replaces dinov2.cpp:479-543 */
int dinov2_hip_op_attention(int32_t dtype, const float *qkv, float *out, int32_t B, int32_t T, int32_t H, int32_t nh);

/* ggml_norm * w + b (dinov2.cpp:694-700); dtype -1 = f32 output (final layernorm), 0/1 = f16/bf16 output */
int dinov2_hip_op_layernorm(int32_t dtype, const float *x, const float *w, const float *b, float *out, int32_t rows,
                            int32_t H, float eps);

/* load-time tensor conversion / dequantisation (F32,F16,BF16,Q4_0,Q4_1,Q5_0,Q5_1,Q8_0 -> compute dtype) */
int dinov2_hip_op_convert_weight(int32_t dtype, const void *src, uint64_t src_bytes, uint32_t ggml_type, float *out,
                                 int32_t N, int32_t K, int32_t Kpad, int32_t interleaveF);

/* what ds_read_b64_tr_b16 hands each lane for addr = lane*8 over an LDS image of its own indices: out[64][4] */
int dinov2_hip_op_probe_tr16(int16_t *out256);

/* micro-benchmarks on device-resident uniform-random operands: average ms per launch over `iters` launches (HIP events),
 * negative on error.  Used by tools/kernel_bench.py to price one kernel against its roofline. */
float dinov2_hip_op_gemm_bench(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, int32_t iters);
float dinov2_hip_op_attention_bench(int32_t dtype, int32_t B, int32_t T, int32_t H, int32_t nh, int32_t iters);

/* preprocess_u8_kernel alone: raw 8-bit BGR [B, h, w, 3] -> normalised f32 BGR [B, oh, ow, 3] (mode 0 dino_preprocess,
 * 1 dino_classify_preprocess; sizes from dinov2_hip_preprocess_size).  Replaces dinov2.cpp:106-156 on the device. */
int dinov2_hip_op_preprocess_u8(int32_t mode, const uint8_t *bgr, int32_t B, int32_t h, int32_t w, int32_t patch, float *out);

/* Clock probe.  Workgroup 0 of every launch of the forward's five heavy kernel kinds adds the shader cycles (s_memtime) and the 100 MHz
 * wall-clock ticks (s_memrealtime) it spent in the kernel to running sums on the device: cycles / (ticks * 10 ns) over a window = the clock
 * the power-limited part sustained under that kernel's load.  The sums only grow; take differences.
 * dinov2_hip_op_clock_probe: the FFN-in GEMM (the roofline's dominant kernel).
 * dinov2_hip_op_clock_slots: out18[3 s] = cycles, out18[3 s + 1] = ticks, out18[3 s + 2] = launches, s = 0 QKV GEMM, 1 attn-out GEMM, 2 FFN-in
 * GEMM, 3 FFN-out GEMM, 4 attention (its first workgroup's own lifetime), 5 any other GEMM.  bench.py weights the kinds by their share of the
 * step (`effective_clock_ghz`, `kernel_clocks_ghz`). */
int dinov2_hip_op_clock_probe(uint64_t *cycles, uint64_t *ticks_100mhz);
int dinov2_hip_op_clock_slots(uint64_t *out18);

/* Testing aids.  The switches the library used to read from the environment on every launch (DINOV2_HIP_GEMM_GEN, DINOV2_HIP_GEMM_TILE,
 * DINOV2_HIP_ATTN_V, DINOV2_HIP_ATTN_NWV; include/dinov2_hip.h, "Environment") are read ONCE, on first use; a test that wants to flip one
 * inside a process calls the setter: key = "gemm_gen" | "gemm_tile" | "attn_v" | "attn_nwv", value 0 = the library's own choice,
 * a negative value = back to what the environment said when the library first looked (so a test leaves a `DINOV2_HIP_GEMM_GEN=2 pytest`
 * run as it found it).
 * Not thread-safe against concurrent forwards on other threads in the sense that they may see either value. */
int dinov2_hip_op_set_tuning(const char *key, int32_t value);
int dinov2_hip_op_get_tuning(const char *key); /* -1: unknown key */

/* Which kernel plan launch_gemm picks for a shape, as text: ';'-separated leaves such as "gemm4_mixed<256+192>", "gemm2<128>",
 * "gemm4<256>;small<64x128,w2x2,st3,ks1>".  Needs no device (nothing is launched).  0 on success. */
int dinov2_hip_op_gemm_plan(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, char *out, int32_t cap);

/* host-only: the Rayleigh-Ritz step behind dinov2_hip_pca3.  yprev [H][8] (any full-rank block), gram [8][8] = yprev^T yprev,
 * ynext [H][8] = cov * (yprev R^-1) with gram = R^T R  ->  evals [3] largest Ritz values of cov on span(yprev), comp [3][H] their
 * unit Ritz vectors, each with its largest loading positive (H >= 8) */
int dinov2_hip_op_pca_ritz(const double *yprev, const double *ynext, const double *gram, int32_t H, double *evals, double *comp);

#ifdef __cplusplus
}
#endif
#endif /* DINOV2_HIP_OPS_H */
