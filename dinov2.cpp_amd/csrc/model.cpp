// model.cpp -- loader, session, forward orchestration and the C-ABI of libdinov2_hip.so.
//
// Replaces, from the reference (lavaman131/dinov2.cpp):
//   dino_model_load        /root/reference/dinov2.cpp:239-352   -> dinov2_hip_model_load
//   interpolate_pos_embed  /root/reference/dinov2.cpp:159-225   -> interpolate_pos_embed() below (no OpenCV)
//   build_graph + dino_predict  :823-838, :900-999              -> forward() + dinov2_hip_predict
// There is no graph builder / allocator / backend scheduler here: the forward is a fixed sequence of ~8 fused
// kernel launches per layer on one HIP stream over a pre-carved workspace.
#include "model.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "gguf_reader.h"

using namespace dinov2;

namespace {

enum Kind : int {
    K_IM2COL = 0, K_INIT, K_PATCH_GEMM, K_LAYERNORM, K_QKV_GEMM, K_ATTENTION, K_OPROJ_GEMM, K_FC1_GEMM, K_FC2_GEMM,
    K_FINAL_LN, K_HEAD, K_COUNT
};
const char* const kKindNames[K_COUNT] = {"im2col", "init_tokens", "gemm_patch_embed", "layernorm", "gemm_qkv",
                                         "attention", "gemm_attn_out", "gemm_ffn_in", "gemm_ffn_out", "final_layernorm",
                                         "head"};

void set_err(char* err, size_t n, const char* fmt, ...) {
    if (!err || n == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, n, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            set_err(err, errlen, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return DINOV2_HIP_ERR_HIP;                                                             \
        }                                                                                          \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- cv::resize(INTER_CUBIC) for CV_32F, restated without OpenCV: separable cubic convolution, A = -0.75,
// source coordinate (d + 0.5) * (src/dst) - 0.5, four taps floor-1..floor+2 clamped to the border, no antialias.
void cubic_taps(float t, float w[4]) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

struct Axis {
    std::vector<int> idx;    // 4 per destination coordinate
    std::vector<float> wgt;  // 4 per destination coordinate
};

Axis make_axis(int src, int dst) {
    Axis a;
    a.idx.resize(4 * (size_t)dst);
    a.wgt.resize(4 * (size_t)dst);
    const float scale = (float)src / (float)dst;
    for (int d = 0; d < dst; ++d) {
        float f = ((float)d + 0.5f) * scale - 0.5f;
        const int s = (int)std::floor(f);
        f -= (float)s;
        cubic_taps(f, &a.wgt[4 * (size_t)d]);
        for (int k = 0; k < 4; ++k) a.idx[4 * (size_t)d + k] = std::min(std::max(s - 1 + k, 0), src - 1);
    }
    return a;
}

// interpolate_pos_embed (dinov2.cpp:159-225).  pos: [1 + M*M, H]; out: [1 + h*w, H].  Identity when the patch
// COUNT matches (the reference compares counts, not shapes: dinov2.cpp:176-179).
void interpolate_pos_embed(const float* pos, int M, int H, int h_new, int w_new, float* out) {
    std::memcpy(out, pos, sizeof(float) * (size_t)H);
    if (h_new * w_new == M * M) {
        std::memcpy(out + H, pos + H, sizeof(float) * (size_t)M * M * H);
        return;
    }
    const Axis ax = make_axis(M, w_new), ay = make_axis(M, h_new);
    std::vector<float> rowbuf((size_t)4 * H);
    for (int dy = 0; dy < h_new; ++dy) {
        const int* iy = &ay.idx[4 * (size_t)dy];
        const float* wy = &ay.wgt[4 * (size_t)dy];
        for (int dx = 0; dx < w_new; ++dx) {
            const int* ix = &ax.idx[4 * (size_t)dx];
            const float* wx = &ax.wgt[4 * (size_t)dx];
            float* o = out + (size_t)(1 + dy * w_new + dx) * H;
            for (int ky = 0; ky < 4; ++ky) {  // horizontal pass per source row, then vertical blend
                float* rb = &rowbuf[(size_t)ky * H];
                const float* r0 = pos + (size_t)(1 + iy[ky] * M + ix[0]) * H;
                const float* r1 = pos + (size_t)(1 + iy[ky] * M + ix[1]) * H;
                const float* r2 = pos + (size_t)(1 + iy[ky] * M + ix[2]) * H;
                const float* r3 = pos + (size_t)(1 + iy[ky] * M + ix[3]) * H;
                for (int c = 0; c < H; ++c) rb[c] = r0[c] * wx[0] + r1[c] * wx[1] + r2[c] * wx[2] + r3[c] * wx[3];
            }
            for (int c = 0; c < H; ++c)
                o[c] = rowbuf[c] * wy[0] + rowbuf[(size_t)H + c] * wy[1] + rowbuf[(size_t)2 * H + c] * wy[2] +
                       rowbuf[(size_t)3 * H + c] * wy[3];
        }
    }
}

// ---- arena planning --------------------------------------------------------------------------------------
struct Plan {
    struct Item {
        std::string name;   // GGUF tensor name
        void** slot;        // where the device pointer goes
        bool matrix;        // 2-D weight converted to the compute dtype, else f32 vector copied as is
        int N, K, Kpad;     // matrix dims (rows, cols, padded cols)
        int interleaveF;    // SwiGLU weights_in row interleave (0 = off)
        size_t offset, bytes;
        bool derived;       // no GGUF tensor behind it: computed on the device after the upload (LN-fold vectors)
    };
    std::vector<Item> items;
    size_t total = 0;
    void add(const std::string& name, void** slot, bool matrix, int N, int K, int Kpad, int F, size_t bytes) {
        Item it{name, slot, matrix, N, K, Kpad, F, total, bytes, false};
        total += align_up(bytes, 256);
        items.push_back(it);
    }
    void add_derived(const std::string& name, float** slot, int count) {
        Item it{name, (void**)slot, false, count, 1, 1, 0, total, sizeof(float) * (size_t)count, true};
        total += align_up(it.bytes, 256);
        items.push_back(it);
    }
};

struct Dims {
    int P, T, M;
};

// dinov2_hip_load_opts.ln_fold == 0: what the library picks (profiles/r06_ln_fold.md)
constexpr bool kLnFoldDefault = false;

}  // namespace

// =============================================================================================================
// load
// =============================================================================================================
int dinov2_check_input(const dinov2_hip_model* m, const dinov2_hip_input* in, char* err, size_t errlen) {
    if (!m || !in || !in->data) {
        set_err(err, errlen, "null session / input");
        return DINOV2_HIP_ERR_INVALID;
    }
    const int ps = (int)m->hp.patch_size;
    if (in->layout == DINOV2_HIP_U8_BGR_HWC) {  // raw images: any size, preprocessed on the device
        if (in->batch <= 0 || in->height <= 0 || in->width <= 0) {
            set_err(err, errlen, "raw image input must have batch, height, width >= 1");
            return DINOV2_HIP_ERR_INVALID;
        }
        return DINOV2_HIP_OK;
    }
    if (in->layout != DINOV2_HIP_BGR_HWC && in->layout != DINOV2_HIP_RGB_CHW) {  // (raw u8 returned above)
        set_err(err, errlen, "unknown input layout %d", in->layout);
        return DINOV2_HIP_ERR_INVALID;
    }
    if (in->batch <= 0 || in->height < ps || in->width < ps || in->height % ps || in->width % ps) {
        set_err(err, errlen, "input must be batch >= 1 and height/width positive multiples of patch_size %d (got %d x %d x %d)",
                ps, in->batch, in->height, in->width);
        return DINOV2_HIP_ERR_INVALID;
    }
    return DINOV2_HIP_OK;
}

extern "C" void dinov2_hip_default_load_opts(dinov2_hip_load_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->device = 0;
    o->compute_dtype = DINOV2_HIP_F16;
    o->classify = 1;
    o->skip_tensor_data = 0;
    o->quirk_pool_const_divisor = 1;
    o->quirk_pool_includes_registers = 1;
    o->batch_invariant = 1;
    o->ln_fold = 0;
}

extern "C" int dinov2_hip_abi_version(void) { return DINOV2_HIP_ABI_VERSION; }

extern "C" void* dinov2_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

extern "C" void dinov2_hip_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int dinov2_hip_model_load(const char* path, const dinov2_hip_load_opts* opts_in, dinov2_hip_model** out,
                                     char* err, size_t errlen) {
    if (!path || !out) {
        set_err(err, errlen, "null argument");
        return DINOV2_HIP_ERR_INVALID;
    }
    *out = nullptr;
    dinov2_hip_load_opts opts;
    if (opts_in) opts = *opts_in; else dinov2_hip_default_load_opts(&opts);
    if (opts.compute_dtype != DINOV2_HIP_F16 && opts.compute_dtype != DINOV2_HIP_BF16) {
        set_err(err, errlen, "compute_dtype must be F16 or BF16");
        return DINOV2_HIP_ERR_INVALID;
    }

    GgufFile gg;
    std::string msg;
    if (!gg.open(path, &msg)) {
        set_err(err, errlen, "%s", msg.c_str());
        const bool io = msg.rfind("failed to open", 0) == 0 || msg.rfind("mmap", 0) == 0;
        return io ? DINOV2_HIP_ERR_IO : DINOV2_HIP_ERR_FORMAT;
    }

    std::unique_ptr<dinov2_hip_model> m(new dinov2_hip_model());
    auto& hp = m->hp;
    // hparams: u32 KVs, every one required (the reference asserts on a missing key, dinov2.cpp:58)
    struct { const char* key; uint32_t* dst; bool required; } keys[] = {
        {"hidden_size", &hp.hidden_size, true},           {"num_hidden_layers", &hp.num_hidden_layers, true},
        {"num_attention_heads", &hp.num_attention_heads, true}, {"patch_size", &hp.patch_size, true},
        {"img_size", &hp.img_size, true},                 {"ftype", &hp.ftype, true},
        {"num_register_tokens", &hp.num_register_tokens, false}, {"num_classes", &hp.num_classes, false}};
    for (auto& k : keys) {
        *k.dst = 0;
        if (!gg.get_u32(k.key, k.dst) && k.required) {
            set_err(err, errlen, "GGUF key '%s' is missing", k.key);
            return DINOV2_HIP_ERR_FORMAT;
        }
    }
    hp.eps = 1e-6f;
    hp.compute_dtype = (uint32_t)opts.compute_dtype;
    m->dt = opts.compute_dtype == DINOV2_HIP_BF16 ? DT_BF16 : DT_F16;
    m->device = opts.device;
    m->quirk_const_div = opts.quirk_pool_const_divisor != 0;
    m->quirk_pool_regs = opts.quirk_pool_includes_registers != 0;
    {
        // LN fold: on request (ln_fold = 1, or DINOV2_HIP_LN_FOLD=1 while the option says "library's choice"), where the model allows it.
        // The library's own choice is in kLnFoldDefault.
        int want = opts.ln_fold;
        if (want == 0)
            if (const char* e = getenv("DINOV2_HIP_LN_FOLD")) want = atoi(e) != 0 ? 1 : -1;
        if (want == 0) want = kLnFoldDefault ? 1 : -1;
        const int Hh = (int)hp.hidden_size;
        m->ln_fold = want > 0 && Hh % 128 == 0 && Hh / LN_GROUP <= LN_MAX_GROUPS;
    }

    const int H = (int)hp.hidden_size, L = (int)hp.num_hidden_layers, nh = (int)hp.num_attention_heads;
    const int ps = (int)hp.patch_size, R = (int)hp.num_register_tokens;
    if (H <= 0 || L <= 0 || nh <= 0 || ps <= 0 || hp.img_size < hp.patch_size) {
        set_err(err, errlen, "invalid hparams in '%s'", path);
        return DINOV2_HIP_ERR_FORMAT;
    }
    if (H != nh * 64) {
        set_err(err, errlen, "unsupported head dim %d (the DINOv2 family and this build use 64)", H / std::max(nh, 1));
        return DINOV2_HIP_ERR_UNSUPPORTED;
    }
    if (H % 64 != 0) {
        set_err(err, errlen, "hidden_size %d is not a multiple of 64", H);
        return DINOV2_HIP_ERR_UNSUPPORTED;
    }
    const int Mgrid = (int)(hp.img_size / hp.patch_size);

    auto need = [&](const std::string& name, const GgufTensor** t) -> bool {
        *t = gg.tensor(name);
        if (!*t) set_err(err, errlen, "GGUF tensor '%s' is missing", name.c_str());
        return *t != nullptr;
    };

    // FFN flavour: by tensor presence (equivalent to the reference's `num_hidden_layers == 40`, dinov2.cpp:740)
    const bool swiglu = gg.tensor("encoder.layer.0.mlp.weights_in.weight") != nullptr;
    hp.swiglu = swiglu;
    const GgufTensor* t = nullptr;
    if (!need(swiglu ? "encoder.layer.0.mlp.weights_out.weight" : "encoder.layer.0.mlp.fc1.weight", &t))
        return DINOV2_HIP_ERR_FORMAT;
    if (t->ne.size() < 2) {
        set_err(err, errlen, "tensor '%s' is not 2-D", t->name.c_str());
        return DINOV2_HIP_ERR_FORMAT;
    }
    const int F = swiglu ? (int)t->ne[0] : (int)t->ne[1];
    hp.ffn_hidden = (uint32_t)F;
    if (F % 64 != 0) {
        set_err(err, errlen, "FFN hidden size %d is not a multiple of 64", F);
        return DINOV2_HIP_ERR_UNSUPPORTED;
    }
    if (!need("encoder.layer.0.attention.attention.qkv.weight", &t)) return DINOV2_HIP_ERR_FORMAT;
    hp.weight_type = t->type;

    const GgufTensor* head = gg.tensor("classifier.weight");
    const bool want_head = opts.classify != 0 && head != nullptr;
    hp.has_classifier = want_head;
    int C = 0;
    if (want_head) {
        C = (int)(head->ne.size() >= 2 ? head->ne[1] : 0);
        if (C <= 0 || (int)head->ne[0] != 2 * H) {
            set_err(err, errlen, "classifier.weight has unexpected shape");
            return DINOV2_HIP_ERR_FORMAT;
        }
        hp.num_classes = (uint32_t)C;
        m->labels.resize((size_t)C);
        for (int i = 0; i < C; ++i) {  // id2label string KVs "0".."C-1" (dinov2.cpp:301-305)
            const GgufValue* v = gg.find(std::to_string(i));
            m->labels[(size_t)i] = v ? v->s : std::string();
        }
    }

    // ---- plan the arena ----
    const size_t esz = 2;
    m->kpe = 3 * ps * ps;
    m->kpe_pad = (int)align_up((size_t)m->kpe, 64);
    m->layers.resize((size_t)L);
    Plan plan;
    auto vec = [&](const std::string& n, float** slot, int count, int F_il = 0) {
        plan.add(n, (void**)slot, false, count, 1, 1, F_il, sizeof(float) * (size_t)count);
    };
    auto mat = [&](const std::string& n, void** slot, int N, int K, int Kpad, int F_il = 0) {
        plan.add(n, slot, true, N, K, Kpad, F_il, esz * (size_t)N * Kpad);
    };
    vec("embeddings.cls_token", &m->cls, H);
    vec("embeddings.position_embeddings", &m->pos, (1 + Mgrid * Mgrid) * H);
    if (R > 0) vec("embeddings.register_tokens", &m->reg, R * H);
    mat("embeddings.patch_embeddings.projection.weight", &m->patch_w, H, m->kpe, m->kpe_pad);
    vec("embeddings.patch_embeddings.projection.bias", &m->patch_b, H);
    for (int i = 0; i < L; ++i) {
        const std::string b = "encoder.layer." + std::to_string(i) + ".";
        LayerWeights& ly = m->layers[(size_t)i];
        vec(b + "norm1.weight", &ly.norm1_w, H);
        vec(b + "norm1.bias", &ly.norm1_b, H);
        mat(b + "attention.attention.qkv.weight", &ly.qkv_w, 3 * H, H, H);
        vec(b + "attention.attention.qkv.bias", &ly.qkv_b, 3 * H);
        mat(b + "attention.output.dense.weight", &ly.o_w, H, H, H);
        vec(b + "attention.output.dense.bias", &ly.o_b, H);
        vec(b + "layer_scale1.lambda1", &ly.ls1, H);
        vec(b + "norm2.weight", &ly.norm2_w, H);
        vec(b + "norm2.bias", &ly.norm2_b, H);
        if (swiglu) {
            mat(b + "mlp.weights_in.weight", &ly.fc1_w, 2 * F, H, H, F);
            vec(b + "mlp.weights_in.bias", &ly.fc1_b, 2 * F, F);
            mat(b + "mlp.weights_out.weight", &ly.fc2_w, H, F, F);
            vec(b + "mlp.weights_out.bias", &ly.fc2_b, H);
        } else {
            mat(b + "mlp.fc1.weight", &ly.fc1_w, F, H, H);
            vec(b + "mlp.fc1.bias", &ly.fc1_b, F);
            mat(b + "mlp.fc2.weight", &ly.fc2_w, H, F, F);
            vec(b + "mlp.fc2.bias", &ly.fc2_b, H);
        }
        vec(b + "layer_scale2.lambda1", &ly.ls2, H);
        if (m->ln_fold) {
            const int nfc1 = swiglu ? 2 * F : F;
            plan.add_derived(b + "ln_fold.qkv_s", &ly.qkv_s, 3 * H);
            plan.add_derived(b + "ln_fold.qkv_c", &ly.qkv_c, 3 * H);
            plan.add_derived(b + "ln_fold.fc1_s", &ly.fc1_s, nfc1);
            plan.add_derived(b + "ln_fold.fc1_c", &ly.fc1_c, nfc1);
        }
    }
    vec("layernorm.weight", &m->ln_w, H);
    vec("layernorm.bias", &m->ln_b, H);
    if (want_head) {
        mat("classifier.weight", &m->head_w, C, 2 * H, 2 * H);
        vec("classifier.bias", &m->head_b, C);
    }

    // validate every tensor against the plan before touching the device
    size_t max_raw = 0;
    for (auto& it : plan.items) {
        if (it.derived) continue;
        const GgufTensor* gt = nullptr;
        if (!need(it.name, &gt)) return DINOV2_HIP_ERR_FORMAT;
        const uint64_t want = it.matrix ? (uint64_t)it.N * it.K : (uint64_t)it.N;
        if (gt->nelements() != want) {
            set_err(err, errlen, "tensor '%s' has %llu elements, expected %llu", it.name.c_str(),
                    (unsigned long long)gt->nelements(), (unsigned long long)want);
            return DINOV2_HIP_ERR_FORMAT;
        }
        if (it.matrix && (int)gt->ne[0] != it.K && it.name.find("patch_embeddings") == std::string::npos) {
            set_err(err, errlen, "tensor '%s' has row length %llu, expected %d", it.name.c_str(),
                    (unsigned long long)gt->ne[0], it.K);
            return DINOV2_HIP_ERR_FORMAT;
        }
        if (!it.matrix && gt->type != GGML_F32) {
            set_err(err, errlen, "tensor '%s' must be F32 (the converter writes 1-D / embedding tensors as F32)",
                    it.name.c_str());
            return DINOV2_HIP_ERR_UNSUPPORTED;
        }
        if (it.matrix && it.name.find("patch_embeddings") != std::string::npos && gt->type != GGML_F16 &&
            gt->type != GGML_F32 && gt->type != GGML_BF16) {
            set_err(err, errlen, "patch-embedding kernel must be F16/F32/BF16");
            return DINOV2_HIP_ERR_UNSUPPORTED;
        }
        max_raw = std::max(max_raw, (size_t)gt->nbytes);
    }

    // ---- device side ----
    HIP_TRY(hipSetDevice(opts.device));
    HIP_TRY(gemm_init());
    m->arena_bytes = plan.total;
    HIP_TRY(hipMalloc((void**)&m->arena, plan.total));
    // every early return below (HIP_TRY included) must give the arena back: the model struct has no destructor of its own
    struct ArenaGuard {
        dinov2_hip_model* m;
        ~ArenaGuard() {
            if (m && m->arena) {
                (void)hipFree(m->arena);
                m->arena = nullptr;
            }
        }
    } arena_guard{m.get()};
    for (auto& it : plan.items) *it.slot = m->arena + it.offset;

    // host copy of the position embeddings for per-resolution interpolation
    {
        const GgufTensor* pt = gg.tensor("embeddings.position_embeddings");
        m->pos_host.assign((const float*)pt->data, (const float*)pt->data + pt->nelements());
    }

    if (!opts.skip_tensor_data) {
        char* staging = nullptr;
        HIP_TRY(hipMalloc((void**)&staging, align_up(max_raw, 256)));
        int rc = DINOV2_HIP_OK;
        for (auto& it : plan.items) {
            if (it.derived) continue;
            const GgufTensor* gt = gg.tensor(it.name);
            hipError_t e = hipSuccess;
            if (!it.matrix && it.interleaveF == 0) {
                e = hipMemcpy(*it.slot, gt->data, gt->nbytes, hipMemcpyHostToDevice);
            } else {
                e = hipMemcpy(staging, gt->data, gt->nbytes, hipMemcpyHostToDevice);
                if (e == hipSuccess) {
                    if (it.matrix)
                        e = launch_convert_weight(m->dt, staging, gt->type, *it.slot, it.N, it.K, it.Kpad, it.interleaveF,
                                                  nullptr);
                    else
                        e = launch_permute_bias((const float*)staging, (float*)*it.slot, it.N, it.interleaveF, nullptr);
                }
                if (e == hipSuccess) e = hipDeviceSynchronize();  // staging is reused by the next tensor
            }
            if (e != hipSuccess) {
                set_err(err, errlen, "uploading '%s' failed: %s", it.name.c_str(), hipGetErrorString(e));
                rc = DINOV2_HIP_ERR_HIP;
                break;
            }
        }
        (void)hipFree(staging);
        if (rc != DINOV2_HIP_OK) return rc;
        if (m->ln_fold) {  // s / c of the QKV and FFN-in weights under the LayerNorm in front of them, from the converted weights
            for (int i = 0; i < L; ++i) {
                const LayerWeights& ly = m->layers[(size_t)i];
                HIP_TRY(launch_ln_fold_vectors(m->dt, ly.qkv_w, ly.qkv_b, ly.norm1_w, ly.norm1_b, ly.qkv_s, ly.qkv_c, 3 * H, H, nullptr));
                HIP_TRY(launch_ln_fold_vectors(m->dt, ly.fc1_w, ly.fc1_b, ly.norm2_w, ly.norm2_b, ly.fc1_s, ly.fc1_c, swiglu ? 2 * F : F, H, nullptr));
            }
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    arena_guard.m = nullptr;  // success: the arena now belongs to the model (dinov2_hip_model_free)
    *out = m.release();
    return DINOV2_HIP_OK;
}

extern "C" void dinov2_hip_model_free(dinov2_hip_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->arena) (void)hipFree(m->arena);
    delete m;
}

extern "C" int dinov2_hip_model_hparams(const dinov2_hip_model* m, dinov2_hip_hparams* out) {
    if (!m || !out) return DINOV2_HIP_ERR_INVALID;
    *out = m->hp;
    return DINOV2_HIP_OK;
}

extern "C" const char* dinov2_hip_model_label(const dinov2_hip_model* m, int32_t id) {
    if (!m || id < 0 || (size_t)id >= m->labels.size()) return nullptr;
    return m->labels[(size_t)id].c_str();
}

extern "C" int dinov2_hip_model_arena(dinov2_hip_model* m, void** ptr, size_t* bytes) {
    if (!m || !ptr || !bytes) return DINOV2_HIP_ERR_INVALID;
    *ptr = m->arena;
    *bytes = m->arena_bytes;
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_interpolate_pos_embed(const dinov2_hip_model* m, int32_t h_new, int32_t w_new, float* out) {
    if (!m || !out || h_new <= 0 || w_new <= 0) return DINOV2_HIP_ERR_INVALID;
    interpolate_pos_embed(m->pos_host.data(), (int)(m->hp.img_size / m->hp.patch_size), (int)m->hp.hidden_size, h_new,
                          w_new, out);
    return DINOV2_HIP_OK;
}

// =============================================================================================================
// session
// =============================================================================================================
namespace {

Dims dims_of(const dinov2_hip_model* m, int B, int h, int w) {
    Dims d;
    d.P = (h / (int)m->hp.patch_size) * (w / (int)m->hp.patch_size);
    d.T = 1 + (int)m->hp.num_register_tokens + d.P;
    d.M = B * d.T;
    return d;
}

struct Carve {
    size_t img, col, x, ln, qkv, att, hid, fin, feat, logits, probs, pos, stats, stats_bytes, total;
};

Carve carve_of(const dinov2_hip_model* m, int B, int h, int w) {
    const Dims d = dims_of(m, B, h, w);
    const size_t H = m->hp.hidden_size, F = m->hp.ffn_hidden, C = std::max<size_t>(m->hp.num_classes, 1);
    Carve c{};
    size_t off = 0;
    auto put = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    c.img = put(sizeof(float) * 3 * (size_t)B * h * w);
    c.col = put(2 * (size_t)B * d.P * m->kpe_pad);
    c.x = put(sizeof(float) * (size_t)d.M * H);
    c.ln = put(2 * (size_t)d.M * H);
#if defined(DINO_PREC) && (DINO_PREC & 25)
    c.qkv = put(2 * (size_t)d.M * 6 * H);  // (tuning build, profiles/r05_parity_attribution.md) second f16 word of q | k | v behind the first
#else
    c.qkv = put(2 * (size_t)d.M * 3 * H);
#endif
    c.att = put(2 * (size_t)d.M * H);
    c.hid = put(2 * (size_t)d.M * F);
    c.fin = put(sizeof(float) * (size_t)d.M * H);
    c.feat = put(sizeof(float) * (size_t)B * 2 * H);
    c.logits = put(sizeof(float) * (size_t)B * C);
    c.probs = put(sizeof(float) * (size_t)B * C);
    c.pos = put(sizeof(float) * (size_t)(1 + d.P) * H);
    c.stats_bytes = m->ln_fold ? sizeof(float) * 2 * (size_t)d.M * ln_stat_slots((int)H) : 0;
    c.stats = put(c.stats_bytes);
    c.total = off;
    return c;
}

int ensure_workspace(dinov2_hip_session* s, int B, int h, int w, char* err, size_t errlen) {
    if (s->cur_b == B && s->cur_h == h && s->cur_w == w && s->ws) return DINOV2_HIP_OK;
    const Carve c = carve_of(s->model, B, h, w);
    if (c.total > s->ws_bytes) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        for (auto& g : s->graphs)  // captured graphs point into the old workspace
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        s->graphs.clear();
        if (s->ws) HIP_TRY(hipFree(s->ws));
        s->ws = nullptr;
        s->ws_bytes = 0;
        HIP_TRY(hipMalloc((void**)&s->ws, c.total));
        s->ws_bytes = c.total;
    }
    s->img = (float*)(s->ws + c.img);
    s->col = s->ws + c.col;
    s->x = (float*)(s->ws + c.x);
    s->ln = s->ws + c.ln;
    s->qkv = s->ws + c.qkv;
    s->att = s->ws + c.att;
    s->hid = s->ws + c.hid;
    s->fin = (float*)(s->ws + c.fin);
    s->feat = (float*)(s->ws + c.feat);
    s->logits = (float*)(s->ws + c.logits);
    s->probs = (float*)(s->ws + c.probs);
    s->pos = (float*)(s->ws + c.pos);
    s->stats = s->model->ln_fold ? (float*)(s->ws + c.stats) : nullptr;
    // (the slots past hidden / 64 of every statistics row are read by the consumers and written by nobody: zero them with the carve)
    if (c.stats_bytes) HIP_TRY(hipMemsetAsync(s->stats, 0, c.stats_bytes, s->stream));
    s->pos_h = s->pos_w = -1;  // the carve moved: re-upload the pos-embed
    s->cur_b = B;
    s->cur_h = h;
    s->cur_w = w;
    return DINOV2_HIP_OK;
}

struct Scope {  // optional per-launch event pair
    dinov2_hip_session* s;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    Scope(dinov2_hip_session* s_, int kind_) : s(s_), kind(kind_) {
        if (!s->profiling) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!s->free_events.empty()) {
                e = s->free_events.back();
                s->free_events.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, s->stream);
    }
    ~Scope() {
        if (!s->profiling) return;
        (void)hipEventRecord(b, s->stream);
        s->records.push_back(ProfRecord{kind, a, b});
    }
};

void drain_profile(dinov2_hip_session* s) {
    if (s->prof_ms.empty()) {
        s->prof_ms.assign(K_COUNT, 0.0);
        s->prof_n.assign(K_COUNT, 0);
    }
    for (auto& r : s->records) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            s->prof_ms[(size_t)r.kind] += ms;
            s->prof_n[(size_t)r.kind] += 1;
        }
        s->free_events.push_back(r.a);
        s->free_events.push_back(r.b);
    }
    s->records.clear();
}

// The forward pass: forward_features (dinov2.cpp:616-790) [+ forward_head :792-821], `nlayers` <= L layers.
// pos-embed for this grid, cached per (h0, w0): the reference recomputes it on every call (dinov2.cpp:937).  Host work +
// synchronisation: must run BEFORE a stream capture, never inside one.
int prepare_pos(dinov2_hip_session* s, int B, int h, int w, char* err, size_t errlen) {
    const dinov2_hip_model* m = s->model;
    const int H = (int)m->hp.hidden_size, ps = (int)m->hp.patch_size;
    const int h0 = h / ps, w0 = w / ps;
    if (s->pos_h == h0 && s->pos_w == w0) return DINOV2_HIP_OK;
    const Dims d = dims_of(m, B, h, w);
    hipStream_t st = s->stream;
    HIP_TRY(hipStreamSynchronize(st));  // pos_stage may still be in flight from a previous shape
    s->pos_stage.resize((size_t)(1 + d.P) * H);
    interpolate_pos_embed(m->pos_host.data(), (int)(m->hp.img_size / m->hp.patch_size), H, h0, w0, s->pos_stage.data());
    HIP_TRY(hipMemcpyAsync(s->pos, s->pos_stage.data(), sizeof(float) * s->pos_stage.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    s->pos_h = h0;
    s->pos_w = w0;
    return DINOV2_HIP_OK;
}

// `img` is a DEVICE pointer.  Leaves final-LN tokens in s->fin, logits/probs in s->logits/s->probs.
int forward(dinov2_hip_session* s, const float* img, int B, int h, int w, int layout, bool classify, int nlayers,
            bool finalize, char* err, size_t errlen) {
    const dinov2_hip_model* m = s->model;
    const int H = (int)m->hp.hidden_size, F = (int)m->hp.ffn_hidden, R = (int)m->hp.num_register_tokens;
    const int nh = (int)m->hp.num_attention_heads, ps = (int)m->hp.patch_size;
    const Dims d = dims_of(m, B, h, w);
    hipStream_t st = s->stream;
    const DType dt = m->dt;

    {
        const int rcp = prepare_pos(s, B, h, w, err, errlen);  // no-op when predict already did it (graph capture relies on that)
        if (rcp != DINOV2_HIP_OK) return rcp;
    }

    {
        Scope sc(s, K_IM2COL);
        HIP_TRY(launch_im2col(dt, img, s->col, B, h, w, ps, m->kpe_pad, layout, st));
    }
    {
        Scope sc(s, K_INIT);
        HIP_TRY(launch_init_tokens(s->x, m->cls, s->pos, m->reg, B, d.T, R, H, st));
    }
    {
        Scope sc(s, K_PATCH_GEMM);
        GemmArgs a{};
        a.A = s->col; a.W = m->patch_w; a.bias = m->patch_b; a.out = s->x; a.aux = s->pos;
        a.M = B * d.P; a.N = H; a.K = m->kpe_pad; a.ldo = H; a.P = d.P; a.T = d.T; a.R = R;
        HIP_TRY(launch_gemm(dt, EPI_PATCH, a, st));
    }
    const float eps = m->hp.eps;
    // LN fold (dinov2_hip_load_opts.ln_fold; kernels.h EPI_RESID_LN): no LayerNorm launches inside the layers.  `ln` holds T(gamma x) for the
    // NEXT LayerNorm and `stats` the row sums behind it, both written by whoever wrote x last: ln_prepare before layer 0, the residual
    // epilogues afterwards; the QKV / FFN-in epilogues apply mean, rstd and beta.
    const bool fold = m->ln_fold;
    const int gs = ln_stat_slots(H);
    if (fold && nlayers > 0) {
        Scope sc(s, K_LAYERNORM);
        HIP_TRY(launch_ln_prepare(dt, s->x, m->layers[0].norm1_w, s->ln, s->stats, gs, d.M, H, st));
    }
    for (int il = 0; il < nlayers; ++il) {
        const LayerWeights& ly = m->layers[(size_t)il];
        if (!fold) {
            Scope sc(s, K_LAYERNORM);
            HIP_TRY(launch_layernorm(dt, s->x, ly.norm1_w, ly.norm1_b, s->ln, d.M, H, eps, st));
        }
        {
            Scope sc(s, K_QKV_GEMM);
            GemmArgs a{};
            a.A = s->ln; a.W = ly.qkv_w; a.bias = ly.qkv_b; a.out = s->qkv;
            a.M = d.M; a.N = 3 * H; a.K = H; a.ldo = 3 * H; a.qcols = H;
#if defined(DINO_PREC) && (DINO_PREC & 25)
            a.ldo = 6 * H;
#endif
            a.qscale = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) (dinov2.cpp:626) x log2(e): softmax runs on exp2
            if (fold) {
                a.stats = s->stats; a.ln_gs = gs; a.ln_s = ly.qkv_s; a.ln_c = ly.qkv_c; a.ln_eps = eps;
            }
            HIP_TRY(launch_gemm(dt, fold ? EPI_QKV_LN : EPI_QKV, a, st));
        }
        {
            Scope sc(s, K_ATTENTION);
            HIP_TRY(launch_attention(dt, s->qkv, s->att, B, d.T, H, nh, true, st));
        }
        {
            Scope sc(s, K_OPROJ_GEMM);
            GemmArgs a{};
            a.A = s->att; a.W = ly.o_w; a.bias = ly.o_b; a.out = s->x; a.aux = ly.ls1;
            a.M = d.M; a.N = H; a.K = H; a.ldo = H;
            if (fold) {
                a.ln_gamma = ly.norm2_w; a.xg = s->ln; a.stats = s->stats; a.ln_gs = gs;
            }
            HIP_TRY(launch_gemm(dt, fold ? EPI_RESID_LN : EPI_RESID, a, st));
        }
        if (!fold) {
            Scope sc(s, K_LAYERNORM);
            HIP_TRY(launch_layernorm(dt, s->x, ly.norm2_w, ly.norm2_b, s->ln, d.M, H, eps, st));
        }
        {
            Scope sc(s, K_FC1_GEMM);
            GemmArgs a{};
            a.A = s->ln; a.W = ly.fc1_w; a.bias = ly.fc1_b; a.out = s->hid;
            a.M = d.M; a.N = m->hp.swiglu ? 2 * F : F; a.K = H; a.ldo = F;
            if (fold) {
                a.stats = s->stats; a.ln_gs = gs; a.ln_s = ly.fc1_s; a.ln_c = ly.fc1_c; a.ln_eps = eps;
            }
            HIP_TRY(launch_gemm(dt, m->hp.swiglu ? (fold ? EPI_SWIGLU_LN : EPI_SWIGLU) : (fold ? EPI_GELU_LN : EPI_GELU), a, st));
        }
        {
            Scope sc(s, K_FC2_GEMM);
            GemmArgs a{};
            a.A = s->hid; a.W = ly.fc2_w; a.bias = ly.fc2_b; a.out = s->x; a.aux = ly.ls2;
            a.M = d.M; a.N = H; a.K = F; a.ldo = H;
            const bool feeds_ln1 = fold && il + 1 < nlayers;  // (the last layer's x goes to the final LayerNorm kernel)
            if (feeds_ln1) {
                a.ln_gamma = m->layers[(size_t)il + 1].norm1_w; a.xg = s->ln; a.stats = s->stats; a.ln_gs = gs;
            }
            HIP_TRY(launch_gemm(dt, feeds_ln1 ? EPI_RESID_LN : EPI_RESID, a, st));
        }
    }
    if (!finalize) return DINOV2_HIP_OK;
    {
        Scope sc(s, K_FINAL_LN);
        HIP_TRY(launch_layernorm_f32(s->x, m->ln_w, m->ln_b, s->fin, d.M, H, eps, st));
    }
    if (classify) {
        Scope sc(s, K_HEAD);
        const int first = m->quirk_pool_regs ? 1 : 1 + R;
        const int Mg = (int)(m->hp.img_size / m->hp.patch_size);
        const float div = m->quirk_const_div ? (float)(Mg * Mg) : (float)(d.T - first);
        HIP_TRY(launch_head(dt, s->fin, m->head_w, m->head_b, s->feat, s->logits, s->probs, B, d.T, H,
                            (int)m->hp.num_classes, first, 1.0f / div, st));
    }
    return DINOV2_HIP_OK;
}

int check_input(const dinov2_hip_session* s, const dinov2_hip_input* in, char* err, size_t errlen) {
    if (!s) {
        set_err(err, errlen, "null session / input");
        return DINOV2_HIP_ERR_INVALID;
    }
    return dinov2_check_input(s->model, in, err, errlen);
}

// Opt-in (DINOV2_HIP_GRAPHS=1): second and later forwards with the same (workspace, input pointer, shape, flags) replay a
// captured hipGraph; the first occurrence runs eagerly (one-off shapes never pay for a capture), the second is captured.
// Off by default because it buys nothing on an idle host: the forward is kernel-bound (178 launches, mean gap 1.0 us in the
// rocprofv3 trace at batch 1), measured p50 3.07 ms with graphs vs 3.06 ms without.  It is there for hosts whose launch
// thread is contended.
int forward_maybe_graph(dinov2_hip_session* s, const float* img, int B, int h, int w, int layout, bool classify, char* err,
                        size_t errlen) {
    static const bool enabled = [] {
        const char* e = getenv("DINOV2_HIP_GRAPHS");
        return e && atoi(e) != 0;
    }();
    const int nl = (int)s->model->hp.num_hidden_layers;
    if (!enabled || s->profiling) return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    hipStream_t st = s->stream;
    dinov2_hip_session::GraphEntry* hit = nullptr;
    for (auto& g : s->graphs)
        if (g.ws == s->ws && g.img == img && g.b == B && g.h == h && g.w == w && g.layout == layout && g.classify == (int)classify)
            hit = &g;
    if (hit && hit->exec) {
        ++hit->uses;
        HIP_TRY(hipGraphLaunch(hit->exec, st));
        return DINOV2_HIP_OK;
    }
    if (hit && hit->uses < 0) return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);  // capture failed before
    if (!hit) {  // first sighting: remember it, run eagerly
        if (s->graphs.size() >= 8) {  // evict the least used entry
            size_t v = 0;
            for (size_t i = 1; i < s->graphs.size(); ++i)
                if (s->graphs[i].uses < s->graphs[v].uses) v = i;
            if (s->graphs[v].exec) (void)hipGraphExecDestroy(s->graphs[v].exec);
            s->graphs.erase(s->graphs.begin() + (long)v);
        }
        s->graphs.push_back({s->ws, img, B, h, w, layout, (int)classify, 0, nullptr});
        return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    }
    // second sighting: capture.  Nothing in forward() synchronises or allocates once prepare_pos has run.
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    }
    const int rc = forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &graph);
    if (rc != DINOV2_HIP_OK || ec != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        hit->uses = -1000000;  // do not try again for this key
        if (rc != DINOV2_HIP_OK) return rc;
        return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) {
        (void)hipGetLastError();
        hit->uses = -1000000;
        return forward(s, img, B, h, w, layout, classify, nl, true, err, errlen);
    }
    hit->exec = exec;
    hit->uses = 1;
    HIP_TRY(hipGraphLaunch(exec, st));
    return DINOV2_HIP_OK;
}

// Copy-out of the session's last forward (shape in s->last_*): the tail of dino_predict (dinov2.cpp:950-999).
int fetch_outputs(dinov2_hip_session* s, dinov2_hip_output* out, char* err, size_t errlen) {
    const dinov2_hip_model* m = s->model;
    const int B = s->last_b, h = s->last_h, w = s->last_w;
    const bool classify = s->last_classify;
    hipStream_t st = s->stream;
    const Dims d = dims_of(m, B, h, w);
    const size_t H = m->hp.hidden_size, C = m->hp.num_classes;
    const int R = (int)m->hp.num_register_tokens;
    const hipMemcpyKind kind = out->on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t pitch = sizeof(float) * (size_t)d.T * H;
    if (out->cls)  // "cls_token" = final-LN row 0 (dinov2.cpp:764-768)
        HIP_TRY(hipMemcpy2DAsync(out->cls, sizeof(float) * H, s->fin, pitch, sizeof(float) * H, (size_t)B, kind, st));
    if (out->patch_tokens) {  // rows [1+R, T) for features, [1, T) when classifying (dinov2.cpp:770-789)
        const int first = classify ? 1 : 1 + R;
        const size_t wbytes = sizeof(float) * (size_t)(d.T - first) * H;
        HIP_TRY(hipMemcpy2DAsync(out->patch_tokens, wbytes, s->fin + (size_t)first * H, pitch, wbytes, (size_t)B, kind, st));
    }
    if (classify) {
        if (out->logits) HIP_TRY(hipMemcpyAsync(out->logits, s->logits, sizeof(float) * B * C, kind, st));
        if (out->probs) HIP_TRY(hipMemcpyAsync(out->probs, s->probs, sizeof(float) * B * C, kind, st));
    }
    if (out->on_device) return DINOV2_HIP_OK;

    std::vector<float> probs_host;
    const bool want_topk = classify && out->topk > 0 && (out->topk_ids || out->topk_probs);
    if (want_topk) {
        probs_host.resize((size_t)B * C);
        HIP_TRY(hipMemcpyAsync(probs_host.data(), s->probs, sizeof(float) * B * C, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (want_topk) {  // descending sort of all classes like dinov2.cpp:961-965, ids and probabilities returned
        const int k = std::min<int>(out->topk, (int)C);
        std::vector<int> idx(C);
        for (int b = 0; b < B; ++b) {
            const float* p = probs_host.data() + (size_t)b * C;
            std::iota(idx.begin(), idx.end(), 0);
            std::partial_sort(idx.begin(), idx.begin() + k, idx.end(),
                              [&](int a, int c2) { return p[a] > p[c2] || (p[a] == p[c2] && a < c2); });
            for (int i = 0; i < out->topk; ++i) {
                if (out->topk_ids) out->topk_ids[(size_t)b * out->topk + i] = i < k ? idx[(size_t)i] : -1;
                if (out->topk_probs) out->topk_probs[(size_t)b * out->topk + i] = i < k ? p[idx[(size_t)i]] : 0.f;
            }
        }
    }
    return DINOV2_HIP_OK;
}

}  // namespace

extern "C" int dinov2_hip_session_create(dinov2_hip_model* m, void* stream, dinov2_hip_session** out, char* err,
                                         size_t errlen) {
    if (!m || !out) {
        set_err(err, errlen, "null argument");
        return DINOV2_HIP_ERR_INVALID;
    }
    *out = nullptr;
    HIP_TRY(hipSetDevice(m->device));
    std::unique_ptr<dinov2_hip_session> s(new dinov2_hip_session());
    s->model = m;
    if (stream) {
        s->stream = (hipStream_t)stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        s->own_stream = true;
    }
    *out = s.release();
    return DINOV2_HIP_OK;
}

extern "C" void dinov2_hip_session_free(dinov2_hip_session* s) {
    if (!s) return;
    (void)hipSetDevice(s->model->device);
    (void)hipStreamSynchronize(s->stream);
    drain_profile(s);
    for (auto e : s->free_events) (void)hipEventDestroy(e);
    for (auto& g : s->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (s->ws) (void)hipFree(s->ws);
    if (s->raw) (void)hipFree(s->raw);
    if (s->pca_buf) (void)hipFree(s->pca_buf);
    if (s->own_stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" size_t dinov2_hip_workspace_bytes(const dinov2_hip_model* m, int32_t batch, int32_t height, int32_t width) {
    if (!m || batch <= 0 || height <= 0 || width <= 0) return 0;
    return carve_of(m, batch, height, width).total;
}

extern "C" int dinov2_hip_session_sync(dinov2_hip_session* s) {
    if (!s) return DINOV2_HIP_ERR_INVALID;
    return hipStreamSynchronize(s->stream) == hipSuccess ? DINOV2_HIP_OK : DINOV2_HIP_ERR_HIP;
}

extern "C" void* dinov2_hip_session_stream(dinov2_hip_session* s) { return s ? (void*)s->stream : nullptr; }

extern "C" int dinov2_hip_session_profile(dinov2_hip_session* s, int32_t enable) {
    if (!s) return DINOV2_HIP_ERR_INVALID;
    (void)hipStreamSynchronize(s->stream);
    drain_profile(s);
    s->profiling = enable != 0;
    if (enable) {
        s->prof_ms.assign(K_COUNT, 0.0);
        s->prof_n.assign(K_COUNT, 0);
    }
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_session_profile_read(dinov2_hip_session* s, int32_t max, const char** names, float* total_ms,
                                               int32_t* launches) {
    if (!s) return 0;
    (void)hipStreamSynchronize(s->stream);
    drain_profile(s);
    const int n = std::min<int>(max, K_COUNT);
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = kKindNames[i];
        if (total_ms) total_ms[i] = (float)s->prof_ms[(size_t)i];
        if (launches) launches[i] = s->prof_n[(size_t)i];
    }
    return n;
}

size_t dinov2_max_pass_batch(const dinov2_hip_model* m, int h, int w) {
    // The kernels address activations with 32-bit offsets (staging cursors of the GEMMs and of the attention): the widest
    // activation buffer of one forward must stay below 2^31 bytes (ViT-L @518: 190 images per pass).
    const Dims d1 = dims_of(m, 1, h, w);
    const size_t widest = std::max<size_t>({3 * (size_t)m->hp.hidden_size, (size_t)m->hp.ffn_hidden, (size_t)m->kpe_pad});
    size_t bmax = std::max<size_t>(1, ((size_t)1 << 31) / ((size_t)d1.T * widest * 2));
    if (const char* e = getenv("DINOV2_HIP_MAX_CHUNK"))  // testing aid: force the split at small sizes
        if (atoi(e) > 0) bmax = std::min<size_t>(bmax, (size_t)atoi(e));
    return bmax;
}

// =============================================================================================================
// predict
// =============================================================================================================
extern "C" int dinov2_hip_predict(dinov2_hip_session* s, const dinov2_hip_input* in, dinov2_hip_output* out,
                                  uint32_t flags, char* err, size_t errlen) {
    int rc = check_input(s, in, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    const dinov2_hip_model* m = s->model;
    const bool classify = (flags & DINOV2_HIP_CLASSIFY) != 0;
    if (classify && !m->hp.has_classifier) {
        set_err(err, errlen, "classify requested but the model was loaded without a classifier head");
        return DINOV2_HIP_ERR_NO_HEAD;
    }
    if (out && out->on_device && (out->topk_ids || out->topk_probs)) {
        set_err(err, errlen, "top-k outputs are host-only");
        return DINOV2_HIP_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    s->last_b = 0;  // nothing to fetch until this forward has succeeded (a failed or re-carving call must not leave the old shape behind)
    const int B = in->batch;
    int h = in->height, w = in->width, layout = in->layout;
    const bool raw_u8 = in->layout == DINOV2_HIP_U8_BGR_HWC;
    if (raw_u8) {  // dino_classify_preprocess | dino_preprocess decide the network input size (dinov2.cpp:106-156)
        int32_t oh, ow;
        dinov2_hip_preprocess_size(classify ? 1 : 0, in->height, in->width, (int32_t)m->hp.patch_size, &oh, &ow);
        h = oh;
        w = ow;
        layout = DINOV2_HIP_BGR_HWC;
    }
    {
        // Batches longer than one pass takes (dinov2_max_pass_batch) are split here, transparently -- B images are B independent
        // forwards, so the results do not change.  Passes run last chunk first, so the session ends up holding chunk 0
        // (dinov2_hip_pca3's "image 0 of the last predict").
        const Dims d1 = dims_of(m, 1, h, w);
        const size_t bmax = dinov2_max_pass_batch(m, h, w);
        if ((size_t)B > bmax) {
            const size_t H = m->hp.hidden_size, C = m->hp.num_classes;
            const size_t tok_rows = (size_t)(d1.T - (classify ? 1 : 1 + (int)m->hp.num_register_tokens));
            const size_t in_stride = raw_u8 ? (size_t)in->height * in->width * 3  /* bytes */
                                            : (size_t)3 * h * w * sizeof(float);
            const int nchunks = (int)(((size_t)B + bmax - 1) / bmax);
            for (int c = nchunks - 1; c >= 0; --c) {
                const size_t b0 = (size_t)c * bmax, bn = std::min<size_t>(bmax, (size_t)B - b0);
                dinov2_hip_input ci = *in;
                ci.data = reinterpret_cast<const float*>(reinterpret_cast<const char*>(in->data) + b0 * in_stride);
                ci.batch = (int32_t)bn;
                dinov2_hip_output co{};
                if (out) {
                    co = *out;
                    if (out->cls) co.cls = out->cls + b0 * H;
                    if (out->patch_tokens) co.patch_tokens = out->patch_tokens + b0 * tok_rows * H;
                    if (out->logits) co.logits = out->logits + b0 * C;
                    if (out->probs) co.probs = out->probs + b0 * C;
                    if (out->topk_ids) co.topk_ids = out->topk_ids + b0 * (size_t)out->topk;
                    if (out->topk_probs) co.topk_probs = out->topk_probs + b0 * (size_t)out->topk;
                }
                rc = dinov2_hip_predict(s, &ci, out ? &co : nullptr, flags, err, errlen);
                if (rc != DINOV2_HIP_OK) return rc;
            }
            s->last_b = 0;  // the workspace holds chunk 0 only: nothing for dinov2_hip_fetch
            return DINOV2_HIP_OK;
        }
    }
    rc = ensure_workspace(s, B, h, w, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    hipStream_t st = s->stream;

    const float* img = in->data;
    if (raw_u8) {
        const size_t nraw = (size_t)B * in->height * in->width * 3;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(in->data);
        if (!in->on_device) {
            if (nraw > s->raw_bytes) {
                HIP_TRY(hipStreamSynchronize(st));
                if (s->raw) HIP_TRY(hipFree(s->raw));
                s->raw = nullptr;
                s->raw_bytes = 0;
                HIP_TRY(hipMalloc((void**)&s->raw, nraw));
                s->raw_bytes = nraw;
            }
            HIP_TRY(hipMemcpyAsync(s->raw, src, nraw, hipMemcpyHostToDevice, st));
            src = s->raw;
        }
        const int rh = classify ? 256 : h, rw = classify ? 256 : w;
        HIP_TRY(launch_preprocess_u8(src, s->img, B, in->height, in->width, rh, rw, (rh - h) / 2, (rw - w) / 2, h, w, st));
        img = s->img;
    } else if (!in->on_device) {
        HIP_TRY(hipMemcpyAsync(s->img, in->data, sizeof(float) * 3 * (size_t)B * h * w, hipMemcpyHostToDevice, st));
        img = s->img;
    }
    rc = prepare_pos(s, B, h, w, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    rc = forward_maybe_graph(s, img, B, h, w, layout, classify, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    {  // what dinov2_hip_pca3(tokens = NULL) works on: the patch rows of image 0 in `fin`
        const Dims dd = dims_of(m, B, h, w);
        s->last_first = classify ? 1 : 1 + (int)m->hp.num_register_tokens;
        s->last_patches = dd.T - s->last_first;
    }
    s->last_b = B;
    s->last_h = h;
    s->last_w = w;
    s->last_classify = classify;
    if (!out) return DINOV2_HIP_OK;
    return fetch_outputs(s, out, err, errlen);
}

extern "C" int dinov2_hip_fetch(dinov2_hip_session* s, dinov2_hip_output* out, char* err, size_t errlen) {
    if (!s || !out) {
        set_err(err, errlen, "null session / output");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (s->last_b <= 0) {
        set_err(err, errlen, "no forward to fetch from (no predict yet, or the last one was split into passes: pass outputs to predict)");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (out->on_device && (out->topk_ids || out->topk_probs)) {
        set_err(err, errlen, "top-k outputs are host-only");
        return DINOV2_HIP_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(s->model->device));
    return fetch_outputs(s, out, err, errlen);
}

extern "C" int dinov2_hip_debug_hidden(dinov2_hip_session* s, const dinov2_hip_input* in, int32_t layer, float* out,
                                       char* err, size_t errlen) {
    int rc = check_input(s, in, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    const dinov2_hip_model* m = s->model;
    if (!out || layer < 0 || layer > (int)m->hp.num_hidden_layers) {
        set_err(err, errlen, "layer out of range");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (in->layout == DINOV2_HIP_U8_BGR_HWC) {  // check_input accepts raw images of any size; this entry point has no preprocess step
        set_err(err, errlen, "debug_hidden takes preprocessed f32 images (BGR_HWC or RGB_CHW), not raw 8-bit input");
        return DINOV2_HIP_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m->device));
    s->last_b = 0;  // this call overwrites the workspace: the previous predict's results are gone for dinov2_hip_fetch
    const int B = in->batch, h = in->height, w = in->width;
    rc = ensure_workspace(s, B, h, w, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    const float* img = in->data;
    if (!in->on_device) {
        HIP_TRY(hipMemcpyAsync(s->img, in->data, sizeof(float) * 3 * (size_t)B * h * w, hipMemcpyHostToDevice, s->stream));
        img = s->img;
    }
    rc = forward(s, img, B, h, w, in->layout, false, layer, false, err, errlen);
    if (rc != DINOV2_HIP_OK) return rc;
    const Dims d = dims_of(m, B, h, w);
    HIP_TRY(hipMemcpyAsync(out, s->x, sizeof(float) * (size_t)d.M * m->hp.hidden_size, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return DINOV2_HIP_OK;
}

// =============================================================================================================
// PCA of patch tokens (cv::PCA(tokens, noArray(), DATA_AS_ROW, 3) + project, inference.cpp:76-81)
// =============================================================================================================
namespace dinov2 {
// Rayleigh-Ritz step on the host for the device-side block iteration (pca_power_kernel): from Y_prev [H][8], the per-workgroup
// Gram partials of Y_prev (g_parts [nparts][64]) and Y_next = cov Q [H][8] with Q = Y_prev R^-1, the eigen-decomposition of
// the 8 x 8 matrix Q^T cov Q.  evals [3]: the three largest Ritz values; comp [3][H] (may be null): their Ritz vectors Q v,
// unit length, largest loading positive.
void pca_ritz(const double* yprev, const double* ynext, const double* g_parts, int nparts, int H, double* evals, double* comp) {
    constexpr int NB = PCA_NB;
    double G[NB * NB], rinv[NB * NB], Bm[NB * NB] = {0}, V[NB * NB] = {0};
    for (int t = 0; t < NB * NB; ++t) {
        double s = 0.0;
        for (int blk = 0; blk < nparts; ++blk) s += g_parts[(size_t)blk * NB * NB + t];
        G[t] = s;
    }
    pca_chol_rinv(G, rinv);
    std::vector<double> Q((size_t)H * NB);
    for (int j = 0; j < H; ++j)
        for (int b = 0; b < NB; ++b) {
            double q = 0.0;
            for (int a2 = 0; a2 <= b; ++a2) q += yprev[(size_t)j * NB + a2] * rinv[a2 * NB + b];
            Q[(size_t)j * NB + b] = q;
        }
    for (int j = 0; j < H; ++j)
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c < NB; ++c) Bm[r * NB + c] += Q[(size_t)j * NB + r] * ynext[(size_t)j * NB + c];
    for (int r = 0; r < NB; ++r)
        for (int c = r + 1; c < NB; ++c) Bm[r * NB + c] = Bm[c * NB + r] = 0.5 * (Bm[r * NB + c] + Bm[c * NB + r]);
    for (int k = 0; k < NB; ++k) V[k * NB + k] = 1.0;
    for (int sweep = 0; sweep < 50; ++sweep) {  // cyclic Jacobi: eigenvalues on Bm's diagonal, eigenvectors in V's columns
        double off = 0, diag = 0;
        for (int r = 0; r < NB; ++r)
            for (int c = 0; c < NB; ++c) (r == c ? diag : off) += Bm[r * NB + c] * Bm[r * NB + c];
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < NB - 1; ++p)
            for (int q = p + 1; q < NB; ++q) {
                const double apq = Bm[p * NB + q];
                if (apq == 0.0) continue;
                const double th = 0.5 * std::atan2(2 * apq, Bm[q * NB + q] - Bm[p * NB + p]);
                const double c = std::cos(th), sn = std::sin(th);
                for (int k = 0; k < NB; ++k) {
                    const double x = Bm[k * NB + p], y = Bm[k * NB + q];
                    Bm[k * NB + p] = c * x - sn * y; Bm[k * NB + q] = sn * x + c * y;
                }
                for (int k = 0; k < NB; ++k) {
                    const double x = Bm[p * NB + k], y = Bm[q * NB + k];
                    Bm[p * NB + k] = c * x - sn * y; Bm[q * NB + k] = sn * x + c * y;
                }
                for (int k = 0; k < NB; ++k) {
                    const double x = V[k * NB + p], y = V[k * NB + q];
                    V[k * NB + p] = c * x - sn * y; V[k * NB + q] = sn * x + c * y;
                }
            }
    }
    int order[NB];
    for (int k = 0; k < NB; ++k) order[k] = k;
    std::sort(order, order + NB, [&](int x, int y) { return Bm[x * NB + x] > Bm[y * NB + y]; });
    for (int c = 0; c < 3; ++c) {
        const int o = order[c];
        evals[c] = Bm[o * NB + o];
        if (!comp) continue;
        int big = 0;
        double nrm = 0;
        for (int j = 0; j < H; ++j) {
            double v = 0;
            for (int k = 0; k < NB; ++k) v += Q[(size_t)j * NB + k] * V[k * NB + o];
            comp[(size_t)c * H + j] = v;
            nrm += v * v;
            if (std::fabs(v) > std::fabs(comp[(size_t)c * H + big])) big = j;
        }
        const double sc = nrm > 0 ? (comp[(size_t)c * H + big] < 0 ? -1.0 : 1.0) / std::sqrt(nrm) : 0.0;
        for (int j = 0; j < H; ++j) comp[(size_t)c * H + j] *= sc;
    }
}
}  // namespace dinov2

extern "C" int dinov2_hip_pca3(dinov2_hip_session* s, const float* tokens, int32_t P, int32_t H, int32_t on_device,
                               float* components, float* mean, float* projection, char* err, size_t errlen) {
    if (!s || P < 4 || H < 8 || H > 4096) {
        set_err(err, errlen, "pca3: need tokens [P >= 4, 8 <= H <= 4096]");
        return DINOV2_HIP_ERR_INVALID;
    }
    if (!tokens && (s->last_patches != P || (int)s->model->hp.hidden_size != H || !s->fin)) {
        set_err(err, errlen, "pca3: tokens == NULL means the last forward's patch tokens of image 0, which are [%d, %d]",
                s->last_patches, (int)s->model->hp.hidden_size);
        return DINOV2_HIP_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(s->model->device));
    hipStream_t st = s->stream;
    constexpr bool trace = false;  // flip to print the means / iteration / total times of a call to stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_setup = 0, t_iter = 0;
    int n_steps = 0;
    const int Ppad = (P + 127) / 128 * 128;  // K of the covariance GEMM: multiple of 64 with an even K / 64
    const int nb = pca_blocks(H);
    const size_t n_tok = (size_t)P * H * 4, n_xt = (size_t)H * Ppad * 2, n_cov = (size_t)H * H * 4, n_mean = (size_t)H * 4;
    const size_t n_y = (size_t)H * PCA_NB * 8, n_g = (size_t)nb * 64 * 8, n_comp = (size_t)3 * H * 4, n_proj = (size_t)P * 3 * 4;
    size_t need = 0;
    auto take = [&](size_t bytes) { const size_t off = need; need += align_up(bytes, 256); return off; };
    const size_t o_tok = take(tokens && !on_device ? n_tok : 0), o_xt = take(n_xt), o_cov = take(n_cov), o_mean = take(n_mean);
    const size_t o_y[3] = {take(n_y), take(n_y), take(n_y)}, o_g[3] = {take(n_g), take(n_g), take(n_g)};
    const size_t o_comp = take(n_comp), o_proj = take(n_proj);
    if (need > s->pca_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (s->pca_buf) HIP_TRY(hipFree(s->pca_buf));
        s->pca_buf = nullptr;
        s->pca_bytes = 0;
        HIP_TRY(hipMalloc((void**)&s->pca_buf, need));
        s->pca_bytes = need;
    }
    char* buf = s->pca_buf;
    float* d_cov = (float*)(buf + o_cov);
    float* d_mean = (float*)(buf + o_mean);
    float* d_comp = (float*)(buf + o_comp);
    float* d_proj = (float*)(buf + o_proj);
    double* d_y[3] = {(double*)(buf + o_y[0]), (double*)(buf + o_y[1]), (double*)(buf + o_y[2])};
    double* d_g[3] = {(double*)(buf + o_g[0]), (double*)(buf + o_g[1]), (double*)(buf + o_g[2])};
    const float* tok = tokens;
    if (!tokens) {
        tok = s->fin + (size_t)s->last_first * H;
    } else if (!on_device) {
        HIP_TRY(hipMemcpyAsync(buf + o_tok, tokens, n_tok, hipMemcpyHostToDevice, st));
        tok = (const float*)(buf + o_tok);
    }
    HIP_TRY(launch_pca_prepare(tok, d_mean, buf + o_xt, P, H, Ppad, st));
    GemmArgs a{};  // P * C = Xt Xt^T: both operands are the same [H, Ppad] matrix
    a.A = buf + o_xt; a.W = buf + o_xt; a.out = d_cov; a.M = H; a.N = H; a.K = Ppad; a.ldo = H;
    HIP_TRY(launch_gemm(DT_F16, EPI_PLAIN_F32, a, st));

    // start block (slot 2): a fixed, well-conditioned pattern; its Gram matrix goes into workgroup 0's partial slot
    std::vector<double> y0((size_t)H * PCA_NB), g0((size_t)nb * 64, 0.0);
    for (int j = 0; j < H; ++j)
        for (int c = 0; c < PCA_NB; ++c) y0[(size_t)j * PCA_NB + c] = std::sin(0.37 * (j + 1) * (c + 1)) + (c == j % PCA_NB ? 0.5 : 0.0);
    for (int j = 0; j < H; ++j)
        for (int r = 0; r < PCA_NB; ++r)
            for (int c = 0; c < PCA_NB; ++c) g0[(size_t)r * PCA_NB + c] += y0[(size_t)j * PCA_NB + r] * y0[(size_t)j * PCA_NB + c];
    HIP_TRY(hipMemcpyAsync(d_y[2], y0.data(), n_y, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_g[2], g0.data(), n_g, hipMemcpyHostToDevice, st));

    if (trace) { (void)hipStreamSynchronize(st); t_setup = since(); }
    // block iteration on the device, Rayleigh-Ritz + convergence test on the host every CHECK steps
    // (steep spectra -- real images -- converge within the first two or three checks; a flat one needs a few hundred steps)
    constexpr int MAX_CHECKS = 28;
    std::vector<double> yp((size_t)H * PCA_NB), yn((size_t)H * PCA_NB), gp((size_t)nb * 64), comp((size_t)3 * H);
    double ev[3] = {0, 0, 0}, prev[3] = {0, 0, 0};
    int src = 2;  // slot holding Y_prev / its Gram partials
    for (int chk = 0; chk < MAX_CHECKS; ++chk) {
        const int CHECK = chk < 4 ? 8 : 16;
        int dst = 0;
        for (int it = 0; it < CHECK; ++it) {
            dst = src == 0 ? 1 : 0;
            HIP_TRY(launch_pca_power(d_cov, d_y[src], d_g[src], d_y[dst], d_g[dst], H, st));
            if (it + 1 < CHECK) src = dst;
        }
        // here: src = Y_prev of the last step, dst = Y_next
        HIP_TRY(hipMemcpyAsync(yp.data(), d_y[src], n_y, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(gp.data(), d_g[src], n_g, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(yn.data(), d_y[dst], n_y, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        dinov2::pca_ritz(yp.data(), yn.data(), gp.data(), nb, H, ev, nullptr);
        n_steps += CHECK;
        bool done = chk > 0;
        for (int c = 0; c < 3; ++c) {
            if (!(std::fabs(ev[c] - prev[c]) <= 1e-8 * std::fabs(ev[0]))) done = false;
            prev[c] = ev[c];
        }
        if (done || !(ev[0] > 0.0)) break;  // converged, or a zero / non-finite covariance: nothing to iterate on
        src = dst;
    }
    t_iter = since();
    if (!std::isfinite(ev[0]) || !std::isfinite(ev[2])) {
        set_err(err, errlen, "pca3: non-finite covariance (tokens beyond the f16 range?)");
        return DINOV2_HIP_ERR_INVALID;
    }
    dinov2::pca_ritz(yp.data(), yn.data(), gp.data(), nb, H, ev, comp.data());
    std::vector<float> compf(comp.begin(), comp.end());
    if (projection) {
        HIP_TRY(hipMemcpyAsync(d_comp, compf.data(), n_comp, hipMemcpyHostToDevice, st));
        HIP_TRY(launch_pca_project(tok, d_mean, d_comp, d_proj, P, H, st));
        HIP_TRY(hipMemcpyAsync(projection, d_proj, n_proj, hipMemcpyDeviceToHost, st));
    }
    if (mean) HIP_TRY(hipMemcpyAsync(mean, d_mean, n_mean, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (components) std::memcpy(components, compf.data(), n_comp);
    if (trace)
        fprintf(stderr, "pca3: P %d H %d: means + covariance %.3f ms, %d steps %.3f ms, total %.3f ms (eigenvalues %.4g %.4g %.4g)\n", P, H,
                t_setup, n_steps, t_iter - t_setup, since(), ev[0], ev[1], ev[2]);
    return DINOV2_HIP_OK;
}
