// dinov2_compat.hpp -- header-only C++ shim with the reference's host API shape, on top of the C-ABI.
//
// The reference's own header (/root/reference/dinov2.h) cannot be reused verbatim: it includes ggml and OpenCV
// headers, declares `constexpr std::string PATTERN` at namespace scope (dinov2.h:18, ill-formed before C++20
// library support) and its `attn` declaration does not match the definition (dinov2.h:70 vs dinov2.cpp:458).
// This shim keeps the names, argument order and error behaviour of the two entry points callers use
// (inference.cpp:45,65; realtime.cpp:45,70) so that swapping the include and the link line is the whole port:
//
//   reference                                                    here
//   -----------------------------------------------------------  ------------------------------------------------
//   struct dino_params            dinov2.h:57-68                  struct dino_params   (same fields/defaults)
//   struct dino_hparams           dinov2.h:25-47                  struct dino_hparams
//   struct dino_model             dinov2.h:49-55                  struct dino_model    (owns the C-ABI handles)
//   struct dino_output            dinov2.h:85-88                  struct dino_output   (Mat32f instead of cv::Mat)
//   bool dino_model_load(cv::Size, const std::string&,            bool dino_model_load(Size2i, const std::string&,
//        dino_model&, const dino_params&)     dinov2.h:98-99           dino_model&, const dino_params&)
//   std::unique_ptr<dino_output> dino_predict(const dino_model&,  std::unique_ptr<dino_output> dino_predict(
//        const cv::Mat&, const dino_params&, ggml_gallocr_t)           const dino_model&, const Mat32f&,
//                                             dinov2.h:111-112         const dino_params&, dinov2_hip_session*)
//
//   dino_preprocess / dino_classify_preprocess dinov2.h:93-96      same names, Mat8u in / Mat32f out (no OpenCV)
//   interpolate_pos_embed         dinov2.h:101-103                same name, takes the model instead of the raw table
//   print_usage / dino_params_parse  dinov2.h:114-116             same
//   dino_model_quantize           dinov2.h:118                    same
//
// `Mat32f` is layout-compatible with a continuous CV_32FC3 cv::Mat.
//
// Define DINOV2_WITH_OPENCV before including to get the reference's OpenCV-typed surface on top (dinov2.h:85-112):
// dino_output::patch_tokens becomes a cv::Mat (fed to cv::PCA at inference.cpp:77-78), and dino_preprocess /
// dino_classify_preprocess / dino_model_load / dino_predict take cv::Mat / cv::Size exactly as dinov2.h:93-112 declare them.
// Define DINOV2_COMPAT_GGML_NAMES as well to get the handful of ggml identifiers the reference's mains touch around the two
// calls (ggml_time_init/ms, ggml_backend_synchronize(model.backend), ggml_gallocr_new/free, the three frees of
// inference.cpp:70-73), so that inference.cpp / realtime.cpp compile with only their #include lines changed.
// This image has no OpenCV: the branch is compile-checked against a minimal stand-in of the few cv:: types it touches
// (tests/cpp/opencv_stub/, test infrastructure) and has never been built against the real library.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <thread>
#include <vector>

#include "dinov2_hip.h"

#ifdef DINOV2_WITH_OPENCV
#include <opencv2/core/mat.hpp>
#endif
#ifdef DINOV2_COMPAT_GGML_NAMES
#include <chrono>
#endif

struct Size2i {
    int width = 0, height = 0;
};

// Row-major H x W x C float image / matrix view (step in bytes), as cv::Mat(rows, cols, CV_32FC(channels)).
struct Mat32f {
    int rows = 0, cols = 0, channels = 1;
    float* data = nullptr;
    size_t step = 0;                      // bytes per row; 0 = continuous
    std::shared_ptr<std::vector<float>> owner;  // set when the Mat owns its memory (dino_output::patch_tokens)
    Size2i size() const { return Size2i{cols, rows}; }
    bool isContinuous() const { return step == 0 || step == sizeof(float) * (size_t)cols * channels; }
};

struct dino_hparams {  // dinov2.h:25-47
    uint32_t hidden_size = 768, num_hidden_layers = 12, num_attention_heads = 12, num_classes = 1000;
    uint32_t num_register_tokens = 0, patch_size = 8, img_size = 224, ftype = 1;
    float eps = 1e-6f;
    std::string interpolation = "bicubic";
    std::map<int, std::string> id2label;
    uint32_t n_enc_head_dim() const { return hidden_size / num_attention_heads; }
    uint32_t n_img_size() const { return img_size; }
    uint32_t n_patch_size() const { return patch_size; }
    uint32_t n_img_embd() const { return img_size / patch_size; }
};

struct dino_params {  // dinov2.h:57-68
    uint32_t seed = 42;
    uint32_t topk = 5;
    bool enable_flash_attn = false;  // accepted, ignored: attention is always the fused, exactly-masked kernel
    uint8_t camera_id = 0;
    uint32_t n_threads = std::min(4u, std::thread::hardware_concurrency());  // unused on the GPU path
    bool classify = false;
    std::string model = "../ggml-model-f16.gguf";
    std::string fname_inp = "../assets/tench.jpg";
    std::string image_out = "pca_visual.jpg";
    float eps = 1e-6f;
    int device = 0;                       // extension: HIP device ordinal
    int compute_dtype = DINOV2_HIP_F16;   // extension: DINOV2_HIP_F16 | DINOV2_HIP_BF16
};

struct dino_model;
// what `model.backend` / `model.ctx` / `model.buffer` (dinov2.h:50-53) are here: tokens that lead back to the model, so the
// reference's calls on them (DINOV2_COMPAT_GGML_NAMES below) have something to work on
struct dinov2_compat_backend { dino_model* model = nullptr; };
struct dinov2_compat_ctx { dino_model* model = nullptr; };

struct dino_model {  // dinov2.h:49-55 (ctx/backend/buffer/tensors collapse into one opaque handle)
    dino_hparams hparams;
    dinov2_hip_model* handle = nullptr;
    dinov2_hip_session* default_session = nullptr;
    dinov2_compat_backend backend{this};  // dinov2.h:50
    dinov2_compat_ctx ctx{this}, buffer{this};  // dinov2.h:51-52
    dino_model() = default;
    dino_model(const dino_model&) = delete;
    dino_model& operator=(const dino_model&) = delete;
    ~dino_model() {
        if (default_session) dinov2_hip_session_free(default_session);
        if (handle) dinov2_hip_model_free(handle);
    }
};

#ifdef DINOV2_WITH_OPENCV
using dino_mat = cv::Mat;  // dinov2.h:87: std::optional<cv::Mat> patch_tokens
#else
using dino_mat = Mat32f;
#endif

struct dino_output {  // dinov2.h:85-88
    std::optional<std::vector<uint32_t>> preds;  // top-k class ids (the reference stores uint32(prob): dinov2.cpp:975)
    std::optional<std::vector<float>> probs;     // their probabilities (extension)
    std::optional<dino_mat> patch_tokens;        // P x H CV_32F, row = y*w0 + x, owning (dinov2.cpp:979-992)
};

// dinov2.h:98-99 / dinov2.cpp:239-352.  img_size is unused, exactly as in the reference.
inline bool dino_model_load(Size2i /*img_size*/, const std::string& fname, dino_model& model, const dino_params& params) {
    printf("%s: loading model from '%s' - please wait\n", __func__, fname.c_str());
    dinov2_hip_load_opts o;
    dinov2_hip_default_load_opts(&o);
    o.device = params.device;
    o.compute_dtype = params.compute_dtype;
    o.classify = params.classify ? 1 : 0;
    char err[512] = {0};
    if (dinov2_hip_model_load(fname.c_str(), &o, &model.handle, err, sizeof err) != DINOV2_HIP_OK) {
        fprintf(stderr, "%s: %s\n", __func__, err);  // reference: "failed to open" + return false (dinov2.cpp:269-272)
        return false;
    }
    dinov2_hip_hparams hp;
    dinov2_hip_model_hparams(model.handle, &hp);
    auto& h = model.hparams;
    h.hidden_size = hp.hidden_size; h.num_hidden_layers = hp.num_hidden_layers;
    h.num_attention_heads = hp.num_attention_heads; h.num_classes = hp.num_classes;
    h.num_register_tokens = hp.num_register_tokens; h.patch_size = hp.patch_size; h.img_size = hp.img_size;
    h.ftype = hp.ftype; h.eps = hp.eps;
    // same echo as dinov2.cpp:288-299
    printf("%s: hidden_size            = %u\n", __func__, h.hidden_size);
    printf("%s: num_hidden_layers      = %u\n", __func__, h.num_hidden_layers);
    printf("%s: num_register_tokens    = %u\n", __func__, h.num_register_tokens);
    printf("%s: num_attention_heads    = %u\n", __func__, h.num_attention_heads);
    printf("%s: patch_size             = %u\n", __func__, h.patch_size);
    printf("%s: img_size               = %u\n", __func__, h.img_size);
    printf("%s: ftype                  = %u\n", __func__, h.ftype);
    printf("%s: qntvr                  = %u\n", __func__, h.ftype / 1000u);  // GGML_QNT_VERSION_FACTOR (dinov2.cpp:286,295)
    if (params.classify && hp.has_classifier) {
        printf("%s: num_classes            = %u\n", __func__, h.num_classes);
        for (uint32_t i = 0; i < h.num_classes; ++i) {
            const char* s = dinov2_hip_model_label(model.handle, (int32_t)i);
            h.id2label[(int)i] = s ? s : "";
        }
    }
    if (dinov2_hip_session_create(model.handle, nullptr, &model.default_session, err, sizeof err) != DINOV2_HIP_OK) {
        fprintf(stderr, "%s: %s\n", __func__, err);
        return false;
    }
    return true;
}

// dinov2.h:111-112 / dinov2.cpp:900-999.  `img`: preprocessed CV_32FC3-compatible BGR-interleaved image whose size is a
// multiple of patch_size; `allocr`: a reusable session (nullptr = the model's default one).  Returns {} on failure.
inline std::unique_ptr<dino_output> dino_predict(const dino_model& model, const Mat32f& img, const dino_params& params,
                                                 dinov2_hip_session* allocr = nullptr) {
    dinov2_hip_session* s = allocr ? allocr : model.default_session;
    if (!s || !img.data || img.channels != 3 || !img.isContinuous()) {
        fprintf(stderr, "%s: need a continuous 3-channel float image\n", __func__);
        return {};
    }
    dinov2_hip_input in{img.data, 1, img.rows, img.cols, DINOV2_HIP_BGR_HWC, 0};
    dinov2_hip_output out{};
    auto output = std::make_unique<dino_output>();
    char err[512] = {0};
    if (params.classify) {
        std::vector<int32_t> ids(params.topk);
        std::vector<float> pr(params.topk);
        out.topk_ids = ids.data(); out.topk_probs = pr.data(); out.topk = (int32_t)params.topk;
        if (dinov2_hip_predict(s, &in, &out, DINOV2_HIP_CLASSIFY, err, sizeof err) != DINOV2_HIP_OK) {
            fprintf(stderr, "%s: %s\n", __func__, err);
            return {};
        }
        fprintf(stderr, "\n");
        std::vector<uint32_t> preds;
        for (uint32_t i = 0; i < params.topk && ids[i] >= 0; ++i) {  // same lines as dinov2.cpp:972-974
            auto it = model.hparams.id2label.find(ids[i]);
            printf(" > %s : %.2f\n", it == model.hparams.id2label.end() ? "?" : it->second.c_str(), pr[i]);
            preds.push_back((uint32_t)ids[i]);
        }
        output->preds = preds;
        output->probs = pr;
    } else {
        const int ps = (int)model.hparams.patch_size;
        const int P = (img.rows / ps) * (img.cols / ps), H = (int)model.hparams.hidden_size;
#ifdef DINOV2_WITH_OPENCV
        cv::Mat m(P, H, CV_32F);  // owning, continuous: what dinov2.cpp:979-992 returns
        out.patch_tokens = reinterpret_cast<float*>(m.data);
#else
        Mat32f m;
        m.rows = P; m.cols = H; m.channels = 1;
        m.owner = std::make_shared<std::vector<float>>((size_t)P * H);
        m.data = m.owner->data();
        out.patch_tokens = m.data;
#endif
        if (dinov2_hip_predict(s, &in, &out, 0, err, sizeof err) != DINOV2_HIP_OK) {
            fprintf(stderr, "%s: %s\n", __func__, err);
            return {};
        }
        output->patch_tokens = m;
    }
    return output;
}

// 8-bit BGR interleaved image view, as cv::imread returns it (CV_8UC3, continuous)
struct Mat8u {
    int rows = 0, cols = 0;
    const uint8_t* data = nullptr;
    Size2i size() const { return Size2i{cols, rows}; }
};

namespace dinov2_compat_detail {
inline Mat32f preprocess(int mode, const Mat8u& img, const dino_hparams& hp) {
    Mat32f out;
    int32_t oh = 0, ow = 0;
    if (!img.data || dinov2_hip_preprocess_size(mode, img.rows, img.cols, (int32_t)hp.patch_size, &oh, &ow) != DINOV2_HIP_OK) return out;
    out.rows = oh; out.cols = ow; out.channels = 3;
    out.owner = std::make_shared<std::vector<float>>((size_t)oh * ow * 3);
    out.data = out.owner->data();
    if (dinov2_hip_preprocess(mode, img.data, img.rows, img.cols, (int32_t)hp.patch_size, out.data) != DINOV2_HIP_OK) out = Mat32f{};
    return out;
}
}  // namespace dinov2_compat_detail

// dinov2.h:93-96 / dinov2.cpp:106-156 (img_size is unused there as well): 256 x 256 squash + centre crop 224 | resize to the
// next multiple of the patch size plus one patch; /255, bicubic, (c - mean) / std with the reference's BGR <-> mean indexing
inline Mat32f dino_classify_preprocess(const Mat8u& img, Size2i /*img_size*/, const dino_hparams& params) {
    return dinov2_compat_detail::preprocess(1, img, params);
}
inline Mat32f dino_preprocess(const Mat8u& img, Size2i /*img_size*/, const dino_hparams& params) {
    return dinov2_compat_detail::preprocess(0, img, params);
}

// dinov2.h:101-103 / dinov2.cpp:159-225.  The reference passes the raw table pointer; here the model owns it.
// Returns [(1 + h*w), hidden] for an image of img_size (multiples of the patch size).
inline std::vector<float> interpolate_pos_embed(Size2i img_size, const dino_model& model) {
    const int ps = (int)model.hparams.patch_size;
    const int h = img_size.height / ps, w = img_size.width / ps;
    std::vector<float> out((size_t)(1 + h * w) * model.hparams.hidden_size);
    if (dinov2_hip_interpolate_pos_embed(model.handle, h, w, out.data()) != DINOV2_HIP_OK) out.clear();
    return out;
}

// dinov2.h:114 / dinov2.cpp:840-863
inline void print_usage(int /*argc*/, char** argv, const dino_params& params) {
    fprintf(stderr, "usage: %s [options]\n\noptions:\n", argv[0]);
    fprintf(stderr, "  -h, --help              show this help message and exit\n");
    fprintf(stderr, "  -m FNAME, --model       model path (default: %s)\n", params.model.c_str());
    fprintf(stderr, "  -i FNAME, --inp         input file (default: %s)\n", params.fname_inp.c_str());
    fprintf(stderr, "  -o FNAME, --out         output file for backbone PCA features (default: %s)\n", params.image_out.c_str());
    fprintf(stderr, "  -k N, --topk            top k classes to print (default: %u)\n", params.topk);
    fprintf(stderr, "  -t N, --threads         number of threads to use during computation (default: %u)\n", params.n_threads);
    fprintf(stderr, "  -c, --classify          whether to classify the image or get backbone PCA features (default: %d)\n", (int)params.classify);
    fprintf(stderr, "  -fa, --flash_attn          whether to enable flash_attn, less accurate (default: %d)\n\n", (int)params.enable_flash_attn);
}

// dinov2.h:116 / dinov2.cpp:865-898.  One deliberate difference: -o sets image_out (the reference overwrites fname_inp, :875).
inline bool dino_params_parse(int argc, char** argv, dino_params& params) {
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-s" || a == "--seed") params.seed = (uint32_t)atoi(next());
        else if (a == "-m" || a == "--model") params.model = next();
        else if (a == "-i" || a == "--inp") params.fname_inp = next();
        else if (a == "-o" || a == "--out") params.image_out = next();
        else if (a == "-t" || a == "--threads") params.n_threads = (uint32_t)atoi(next());
        else if (a == "-k" || a == "--topk") params.topk = (uint32_t)atoi(next());
        else if (a == "-cid" || a == "--camera_id") params.camera_id = (uint8_t)atoi(next());
        else if (a == "-fa" || a == "--flash_attn") params.enable_flash_attn = true;
        else if (a == "-c" || a == "--classify") params.classify = true;
        else {
            if (a != "-h" && a != "--help") fprintf(stderr, "error: unknown argument: %s\n", a.c_str());
            print_usage(argc, argv, params);
            exit(0);
        }
    }
    return true;
}

// dinov2.h:118 / dinov2.cpp:355-453
inline bool dino_model_quantize(const std::string& fname_inp, const std::string& fname_out, int itype) {
    char err[512] = {0};
    if (dinov2_hip_quantize(fname_inp.c_str(), fname_out.c_str(), itype, err, sizeof err) != DINOV2_HIP_OK) {
        fprintf(stderr, "%s: %s\n", __func__, err);
        return false;
    }
    return true;
}

#ifdef DINOV2_WITH_OPENCV
// ---- the reference's OpenCV-typed declarations (dinov2.h:93-112), as thin adapters over the functions above ----
inline bool dino_model_load(cv::Size sz, const std::string& fname, dino_model& model, const dino_params& params) {
    return dino_model_load(Size2i{sz.width, sz.height}, fname, model, params);
}
inline std::unique_ptr<dino_output> dino_predict(const dino_model& model, const cv::Mat& img, const dino_params& params,
                                                 dinov2_hip_session* allocr = nullptr) {
    if (img.type() != CV_32FC3) {
        fprintf(stderr, "%s: need a CV_32FC3 image (dino_preprocess output)\n", __func__);
        return {};
    }
    cv::Mat c = img.isContinuous() ? img : img.clone();
    Mat32f v;
    v.rows = c.rows; v.cols = c.cols; v.channels = 3; v.data = reinterpret_cast<float*>(c.data);
    return dino_predict(model, v, params, allocr);
}
namespace dinov2_compat_detail {
inline cv::Mat preprocess_cv(int mode, cv::Mat& img, const dino_hparams& hp) {
    cv::Mat out;
    if (img.empty() || img.type() != CV_8UC3) {
        fprintf(stderr, "dino_preprocess: need a CV_8UC3 image (cv::imread(..., cv::IMREAD_COLOR))\n");
        return out;
    }
    cv::Mat c = img.isContinuous() ? img : img.clone();
    int32_t oh = 0, ow = 0;
    if (dinov2_hip_preprocess_size(mode, c.rows, c.cols, (int32_t)hp.patch_size, &oh, &ow) != DINOV2_HIP_OK) return out;
    out.create(oh, ow, CV_32FC3);
    if (dinov2_hip_preprocess(mode, c.data, c.rows, c.cols, (int32_t)hp.patch_size, reinterpret_cast<float*>(out.data)) != DINOV2_HIP_OK)
        out.release();
    return out;
}
}  // namespace dinov2_compat_detail
// dinov2.h:94, 96 (non-const reference and unused img_size, as declared there)
inline cv::Mat dino_classify_preprocess(cv::Mat& img, cv::Size /*img_size*/, const dino_hparams& params) {
    return dinov2_compat_detail::preprocess_cv(1, img, params);
}
inline cv::Mat dino_preprocess(cv::Mat& img, cv::Size /*img_size*/, const dino_hparams& params) {
    return dinov2_compat_detail::preprocess_cv(0, img, params);
}
#endif  // DINOV2_WITH_OPENCV

#ifdef DINOV2_COMPAT_GGML_NAMES
// ---- the ggml identifiers the reference's mains use AROUND dino_model_load / dino_predict (inference.cpp:25,60-73;
// realtime.cpp:56,62,68-72), mapped onto the C-ABI so those files need no edits beyond their #include lines.  Nothing here
// is ggml: `ggml_gallocr_t` is the session handle that plays the allocator's role. ----
using ggml_gallocr_t = dinov2_hip_session*;
struct dinov2_compat_buft { dino_model* model; };
inline void ggml_time_init() {}
inline int64_t ggml_time_ms() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline void ggml_backend_synchronize(const dinov2_compat_backend& b) {  // inference.cpp:61,66: the timed region's fences
    if (b.model && b.model->default_session) dinov2_hip_session_sync(b.model->default_session);
}
inline dinov2_compat_buft ggml_backend_get_default_buffer_type(const dinov2_compat_backend& b) { return {b.model}; }
inline ggml_gallocr_t ggml_gallocr_new(dinov2_compat_buft t) {  // inference.cpp:62: a reusable session (stream + workspace)
    dinov2_hip_session* s = nullptr;
    char err[256] = {0};
    if (!t.model || dinov2_hip_session_create(t.model->handle, nullptr, &s, err, sizeof err) != DINOV2_HIP_OK)
        fprintf(stderr, "%s: %s\n", __func__, err);
    return s;
}
inline void ggml_gallocr_free(ggml_gallocr_t a) { dinov2_hip_session_free(a); }          // inference.cpp:71
inline void ggml_free(dinov2_compat_ctx&) {}                                             // inference.cpp:70: owned by dino_model
inline void ggml_backend_buffer_free(dinov2_compat_ctx& b) {                             // inference.cpp:72: the weight arena
    if (!b.model) return;
    if (b.model->default_session) { dinov2_hip_session_free(b.model->default_session); b.model->default_session = nullptr; }
    if (b.model->handle) { dinov2_hip_model_free(b.model->handle); b.model->handle = nullptr; }
}
inline void ggml_backend_free(dinov2_compat_backend&) {}                                 // inference.cpp:73
#endif  // DINOV2_COMPAT_GGML_NAMES
