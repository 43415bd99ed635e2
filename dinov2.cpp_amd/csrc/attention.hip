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
#include <cstdlib>
#include <type_traits>

#include "device_types.h"
#include "kernels.h"

namespace dinov2 {

// LOG2: scores arrive multiplied by log2(e) (folded into the q scale by the QKV epilogue), so p = exp2(s - m) needs no
// multiply.  launch_bounds(256, 2): allow up to 256 VGPRs -- with the default budget hipcc parked 128 values in AGPRs
// and spent 255 v_accvgpr moves per key tile shuttling them (as many VALU ops as the softmax itself).
template <typename T, bool LOG2, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 4) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Ttok, int H) {
    using E = Elem<T>;
    using vec8 = typename E::vec8;
    using vec4 = typename E::vec4;
    constexpr int KT = 64;          // keys per tile
    constexpr int ROWB = 128;       // bytes per LDS row (64 dims)
    constexpr int TILEB = KT * ROWB;

    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILEB];  // [buf][K|V]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int H3 = 3 * H;
    const char* base = (const char*)(qkv + (size_t)b * Ttok * H3);

    const int ql = lane & 31, hh = lane >> 5;
    const int qrow = blockIdx.x * (NWV * 32) + wid * 32 + ql;
    const int qrc = qrow < Ttok ? qrow : Ttok - 1;

    // Q^T fragments (B operand): lane holds q[qrow][16*ks + 8*hh + 0..7]
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *(const vec8*)(base + ((size_t)qrc * H3 + h * 64 + ks * 16 + hh * 8) * 2);

    // staging: a wave-instruction covers 8 rows x 128 B; 4 waves x 2 instructions = 64 rows, for K and for V
    const int srow = lane >> 3;
    constexpr int SI = 8 / NWV;  // staging wave-instructions per wave per matrix (8 rows each, 64 rows per tile)
    int strow[SI];
    int stlc[SI];
#pragma unroll
    for (int j = 0; j < SI; ++j) {
        strow[j] = (j * NWV + wid) * 8 + srow;
        stlc[j] = ((lane & 7) ^ ((strow[j] >> 1) & 7)) * 16;
    }
    auto stage = [&](int buf, int jt) {
        char* sK = smem + buf * 2 * TILEB;
        char* sV = sK + TILEB;
#pragma unroll
        for (int j = 0; j < SI; ++j) {
            int key = jt * KT + strow[j];
            key = key < Ttok ? key : Ttok - 1;  // tail rows re-read the last key; masked below
            const char* g = base + ((size_t)key * H3 + h * 64) * 2 + stlc[j];
            glds16(g + (size_t)H * 2, sK + (j * NWV + wid) * 8 * ROWB);
            glds16(g + (size_t)H * 4, sV + (j * NWV + wid) * 8 * ROWB);
        }
    };

    const int sw = (ql >> 1) & 7;
    // K fragment byte offsets inside a K tile, one per 16-wide k-step (+ kb * 4096 as an immediate)
    int kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = ql * ROWB + (((ks * 2 + hh) ^ sw) << 4);
    // V^T gather for ds_read_b64_tr_b16: within each 16-lane group, lane t supplies the address of
    // V[key0 + (t >> 2)][d0 + 4*(t & 3) .. +3] and receives V[key0 + 0..3][d0 + t].  key0 = 16t + 8*half + 4*hh: the row
    // swizzle term ((row >> 1) & 7) does not depend on t, so four base offsets + t * 2048 as an immediate cover the tile.
    const int t16 = lane & 15;
    int vaddr[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int row0 = 8 * half + 4 * hh + (t16 >> 2);
            const int colbyte = db * 64 + (((lane >> 4) & 1) * 16 + (t16 & 3) * 4) * 2;
            vaddr[half][db] = row0 * ROWB + ((((colbyte >> 4) ^ ((row0 >> 1) & 7)) << 4) | (colbyte & 15));
        }

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;
    // -m_run broadcast over a 16-register tuple: fed as the C operand of the first MFMA of every score chain, so the
    // accumulators come out as (s - m_run) and the softmax needs no subtraction; rewritten only when m_run moves.
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    // deferred max (online softmax): the reference point m_run only moves when a tile's maximum exceeds it by more than
    // THR, so most tiles skip the O / l rescale.  p <= 2^THR stays far inside f16/bf16 range and keeps full relative
    // precision; the first tile always takes the rescale branch (alpha = 0, whatever its maximum is).
    constexpr float THR = LOG2 ? 8.0f : 5.5f;

    const int ntiles = (Ttok + KT - 1) / KT;
    auto tile = [&](int jt, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        __syncthreads();
        if (!MASKED) stage((jt + 1) & 1, jt + 1);
        const char* sK = smem + (jt & 1) * 2 * TILEB;
        const char* sV = sK + TILEB;

        // ---- S^T = K Q^T : two 32-key blocks ----
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf = *(const vec8*)(sK + kaddr[ks] + kb * 32 * ROWB);
                s[kb] = E::mfma32(kf, qf[ks], ks == 0 ? negm : s[kb]);
            }
        }
        // s[kb][r] = score - m_run of key jt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hh; only the last tile has keys >= Ttok
        if constexpr (MASKED) {
            const int kbase = jt * KT + 4 * hh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= Ttok) s[kb][r] = -INFINITY;
        }
        // ---- online softmax (soft_max_ext semantics: exp(s - max) / sum), statistics per lane = per query ----
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));  // tile maximum relative to m_run
        const bool first = jt == 0;          // m_run = 0 is not a real reference yet: take the tile maximum, whatever it is
        const bool need = first || mx > THR;
        if (__any(need)) {  // wave-uniform; lanes that do not need it shift by d = 0 (alpha = 1)
            const float d = need ? mx : 0.f;
            const float alpha = first ? 0.f : (LOG2 ? __builtin_amdgcn_exp2f(-d) : __expf(-d));
            m_run += d;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[0][r] *= alpha;
                o[1][r] *= alpha;
                negm[r] = -m_run;
                s[0][r] -= d;
                s[1][r] -= d;
            }
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = LOG2 ? __builtin_amdgcn_exp2f(s[kb][r]) : __expf(s[kb][r]);
                s[kb][r] = pv;
                psum += pv;
            }
        l_run += psum;
        // ---- O^T += V^T P^T : 4 k-steps of 16 keys; lane's 8 k-slots of step t = score regs (t&1)*8 .. +7 of
        //      block t>>1, i.e. keys 16t + 4hh + {0..3} and 16t + 8 + 4hh + {0..3}
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            vec8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = E::from_f32(s[t >> 1][(t & 1) * 8 + j]);
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
                o[db] = E::mfma32(vf, pf, o[db]);
            }
        }
    };
    stage(0, 0);
    for (int jt = 0; jt + 1 < ntiles; ++jt) tile(jt, std::false_type{});
    tile(ntiles - 1, std::true_type{});

    // ---- normalise and store: o[db][r] = O[q][d], d = db*32 + (r&3) + 8*(r>>2) + 4*hh ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qrow < Ttok) {
        T* orow = out + ((size_t)b * Ttok + qrow) * H + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = E::from_f32(o[db][g * 4 + j] * inv);
                *(vec4*)(orow + db * 32 + g * 8 + hh * 4) = w;
            }
    }
}

hipError_t launch_attention(DType dt, const void* qkv, void* out, int B, int T, int H, int nh, bool log2_scores,
                            hipStream_t st) {
    if (H != nh * 64 || T <= 0 || B <= 0) return hipErrorInvalidValue;
    // 8 waves (256 queries) per workgroup halve the K/V staging per query; 4 waves waste less on the ragged last query
    // block.  DINOV2_HIP_ATTN_WAVES=4|8 overrides (tuning aid).
    static const int forced = [] {
        const char* e = getenv("DINOV2_HIP_ATTN_WAVES");
        return e ? atoi(e) : 0;
    }();
    const int nwv = forced == 4 || forced == 8 ? forced : 4;  // A/B on MI355X: equal within noise (0.41-0.43 ms)
    const dim3 grid((T + nwv * 32 - 1) / (nwv * 32), nh, B), block(nwv * 64);
#define DINO_ATT(TT, LG, NW) \
    hipLaunchKernelGGL((attention_kernel<TT, LG, NW>), grid, block, 0, st, (const TT*)qkv, (TT*)out, T, H)
    if (dt == DT_F16) {
        if (nwv == 8) { if (log2_scores) DINO_ATT(_Float16, true, 8); else DINO_ATT(_Float16, false, 8); }
        else { if (log2_scores) DINO_ATT(_Float16, true, 4); else DINO_ATT(_Float16, false, 4); }
    } else {
        if (nwv == 8) { if (log2_scores) DINO_ATT(__bf16, true, 8); else DINO_ATT(__bf16, false, 8); }
        else { if (log2_scores) DINO_ATT(__bf16, true, 4); else DINO_ATT(__bf16, false, 4); }
    }
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
