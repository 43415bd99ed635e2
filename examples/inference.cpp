// inference.cpp -- the reference's `inference` program (/root/reference/inference.cpp:24-104) written against the C++ shim
// (include/dinov2_compat.hpp) and the C-ABI, without OpenCV: same flags (dino_params_parse, dinov2.cpp:865-898), same stderr /
// stdout lines ("main: graph computation took N ms" is what scripts/benchmark.sh:73-77 scrapes), same flow: read the image ->
// dino_model_load -> preprocess -> timed dino_predict -> top-k lines, or a 3-component PCA of the patch tokens -> min-max to
// 0..255 -> patch grid -> nearest-neighbour resize to the preprocessed size -> image file.
// Images: JPEG (baseline and progressive, decoded to the bytes libjpeg / cv::imread produce -- examples/jpeg_codec.hpp) or binary PPM in;
// JPEG (.jpg) or PPM out.  Defaults as the reference's: -i ../assets/tench.jpg, -o pca_visual.jpg (dinov2.h:65-66), so that
// scripts/benchmark.sh:66-77 (`./bin/inference -c -m ../ggml-model.gguf -i ../assets/tench.jpg -t N`, scraping the line above) runs as is.
//
//   make -C dinov2.cpp_amd examples        ->  build/bin/inference, build/bin/quantize, build/bin/realtime
//   ./build/bin/inference -m model.gguf -i image.jpg [-c] [-k 5] [-o pca_visual.jpg]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dinov2_compat.hpp"
#include "jpeg_codec.hpp"

int main(int argc, char** argv) {
    dino_params params;
    if (!dino_params_parse(argc, argv, params)) return 1;
    fprintf(stderr, "%s: seed = %u\n", __func__, params.seed);
    std::vector<uint8_t> bgr;
    int h = 0, w = 0;
    if (!dinojpeg::imread_bgr(params.fname_inp, bgr, h, w)) {  // cv::imread(IMREAD_COLOR): JPEG (baseline / progressive) or binary PPM
        fprintf(stderr, "%s: failed to load image from '%s'\n", __func__, params.fname_inp.c_str());
        return 1;
    }
    fprintf(stderr, "%s: loaded image '%s' (%d x %d)\n", __func__, params.fname_inp.c_str(), h, w);
    dino_model model;
    if (!dino_model_load(Size2i{w, h}, params.model, model, params)) {
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }
    // dino_classify_preprocess | dino_preprocess (dinov2.cpp:106-156) without OpenCV
    const int ps = (int)model.hparams.patch_size;
    Mat8u raw;
    raw.rows = h; raw.cols = w; raw.data = bgr.data();
    Mat32f img = params.classify ? dino_classify_preprocess(raw, Size2i{w, h}, model.hparams) : dino_preprocess(raw, Size2i{w, h}, model.hparams);
    if (!img.data) return 1;
    const int oh = img.rows, ow = img.cols;
    fprintf(stderr, "%s: preprocessed image (%d x %d)\n", __func__, oh, ow);

    const auto t0 = std::chrono::steady_clock::now();
    std::unique_ptr<dino_output> output = dino_predict(model, img, params);
    dinov2_hip_session_sync(model.default_session);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "%s: graph computation took %lld ms\n", __func__,
            (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
    if (!output) return 1;
    if (output->patch_tokens) {
        const Mat32f& tok = *output->patch_tokens;
        // cv::PCA(tokens, noArray(), DATA_AS_ROW, 3) + project (inference.cpp:76-81): covariance on the device, see dinov2_hip.h
        std::vector<float> proj((size_t)tok.rows * 3);
        char err[256] = {0};
        if (dinov2_hip_pca3(model.default_session, tok.data, tok.rows, tok.cols, 0, nullptr, nullptr, proj.data(), err, sizeof err) !=
            DINOV2_HIP_OK) {
            fprintf(stderr, "%s: PCA failed: %s\n", __func__, err);
            return 1;
        }
        float lo = proj[0], hi = proj[0];
        for (float v : proj) { lo = std::min(lo, v); hi = std::max(hi, v); }
        const int gr = oh / ps, gc = ow / ps;
        std::vector<uint8_t> small((size_t)gr * gc * 3);
        for (size_t i = 0; i < small.size(); ++i)
            small[i] = (uint8_t)std::min(255.0f, std::max(0.0f, std::nearbyint(hi == lo ? 0.f : (proj[i] - lo) * (255.0f / (hi - lo)))));
        std::vector<uint8_t> big((size_t)oh * ow * 3);
        // cv::resize(INTER_NEAREST): source index = min(floor(dst * ifx), src - 1) with ifx = 1 / (dst_size / src_size)
        const double ify = 1.0 / ((double)oh / gr), ifx = 1.0 / ((double)ow / gc);
        for (int y = 0; y < oh; ++y) {
            const int sy = std::min((int)std::floor(y * ify), gr - 1);
            for (int x = 0; x < ow; ++x) {
                const int sx = std::min((int)std::floor(x * ifx), gc - 1);
                memcpy(&big[((size_t)y * ow + x) * 3], &small[((size_t)sy * gc + sx) * 3], 3);
            }
        }
        if (dinojpeg::imwrite_bgr(params.image_out, big.data(), oh, ow)) fprintf(stderr, "%s: Saved image to: %s\n", __func__, params.image_out.c_str());
        else fprintf(stderr, "%s: failed to save image to '%s'\n", __func__, params.image_out.c_str());
    }
    return 0;
}
