// Host-side model / session objects behind the C-ABI (include/dinov2_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dinov2_hip.h"
#include "kernels.h"

namespace dinov2 {

struct LayerWeights {
    float *norm1_w, *norm1_b, *qkv_b, *o_b, *ls1, *norm2_w, *norm2_b, *fc1_b, *fc2_b, *ls2;
    void *qkv_w, *o_w, *fc1_w, *fc2_w;  // compute dtype, [N, K] row-major (ggml ne = [K, N])
    // LN fold (kernels.h): s[n] = sum_k gamma_k W[n, k], c[n] = bias[n] + sum_k beta_k W[n, k] of the QKV / FFN-in weights with the
    // LayerNorm in front of them; derived at load, part of the arena (so they travel with a weight broadcast)
    float *qkv_s = nullptr, *qkv_c = nullptr, *fc1_s = nullptr, *fc1_c = nullptr;
};

}  // namespace dinov2

// replaces `struct dino_model` (/root/reference/dinov2.h:49-55): hparams + one device buffer + name->tensor map
struct dinov2_hip_model {
    dinov2_hip_hparams hp{};
    dinov2::DType dt = dinov2::DT_F16;
    int device = 0;
    bool quirk_const_div = true, quirk_pool_regs = true;
    bool ln_fold = false;      // LayerNorm 1 / 2 of every layer folded into the neighbouring GEMM epilogues (dinov2_hip_load_opts.ln_fold)
    int kpe = 0, kpe_pad = 0;  // patch-embed K (3*p*p) and its padding to a multiple of 64
    char* arena = nullptr;     // ONE allocation, like model.buffer (dinov2.cpp:341)
    size_t arena_bytes = 0;
    std::vector<dinov2::LayerWeights> layers;
    void* patch_w = nullptr;
    float *patch_b = nullptr, *cls = nullptr, *pos = nullptr, *reg = nullptr, *ln_w = nullptr, *ln_b = nullptr;
    void* head_w = nullptr;
    float* head_b = nullptr;
    std::vector<float> pos_host;  // the reference reads position_embeddings on the host every call (dinov2.cpp:935-938)
    std::vector<std::string> labels;
};

struct ProfRecord {
    int kind;
    hipEvent_t a, b;
};

// replaces the caller-owned ggml_gallocr_t (/root/reference/dinov2.h:111-112): stream + workspace + cached pos-embed
struct dinov2_hip_session {
    dinov2_hip_model* model = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    uint8_t* raw = nullptr;  // raw 8-bit images for DINOV2_HIP_U8_BGR_HWC inputs
    size_t raw_bytes = 0;
    char* pca_buf = nullptr;  // dinov2_hip_pca3's device scratch, grown on demand
    size_t pca_bytes = 0;
    int last_b = 0, last_h = 0, last_w = 0;  // shape of the last un-split forward (0: none): what dinov2_hip_fetch copies out
    bool last_classify = false;
    int last_first = 0, last_patches = 0;  // rows [last_first, last_first + last_patches) of image 0 in `fin`: its patch tokens
    // carved views (valid for cur_* shape)
    int cur_b = 0, cur_h = 0, cur_w = 0;
    float *img = nullptr, *x = nullptr, *fin = nullptr, *feat = nullptr, *logits = nullptr, *probs = nullptr,
          *pos = nullptr;
    void *col = nullptr, *ln = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr;
    float* stats = nullptr;  // LN fold: [M][ln_stat_slots(H)][2] row statistics of the residual stream (slots past H / 64 stay zero)
    int pos_h = -1, pos_w = -1;  // grid the cached interpolated pos-embed in `pos` belongs to
    std::vector<float> pos_stage;
    // hipGraph cache (the "allocr reuse" of the reference taken one step further): a forward that repeats with the same
    // shape, input pointer and workspace is captured once and replayed; 178 launches become one graph launch
    struct GraphEntry {
        const void* ws;
        const void* img;
        int b, h, w, layout, classify, uses;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
    // profiling
    bool profiling = false;
    std::vector<ProfRecord> records;
    std::vector<hipEvent_t> free_events;
    std::vector<double> prof_ms;
    std::vector<int> prof_n;
};

// Internal (not C-ABI) helpers shared by model.cpp and group.cpp.
// Argument checks of dinov2_hip_predict that need no session: layout, batch, height / width against the patch size.
int dinov2_check_input(const dinov2_hip_model* m, const dinov2_hip_input* in, char* err, size_t errlen);
// Largest batch ONE pass of the forward takes at network input size h x w (32-bit activation offsets; DINOV2_HIP_MAX_CHUNK lowers it for
// tests): dinov2_hip_predict cuts longer batches into passes and leaves nothing for dinov2_hip_fetch.
size_t dinov2_max_pass_batch(const dinov2_hip_model* m, int h, int w);
