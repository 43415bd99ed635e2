// inference.cpp -- the reference's `inference` program (/root/reference/inference.cpp:24-104) written against the C++ shim
// (include/dinov2_compat.hpp) and the C-ABI, without OpenCV: same flags (dino_params_parse, dinov2.cpp:865-898), same stderr /
// stdout lines ("main: graph computation took N ms" is what scripts/benchmark.sh:73-77 scrapes), same flow: read the image ->
// dino_model_load -> preprocess -> timed dino_predict -> top-k lines, or a 3-component PCA of the patch tokens -> min-max to
// 0..255 -> patch grid -> nearest-neighbour resize to the preprocessed size -> image file.
// Images are binary PPM (P6): the reference's decoders live in OpenCV, which this build does not link.
//
//   g++ -O2 -std=c++17 -I include examples/inference.cpp -o inference dinov2.cpp_amd/libdinov2_hip.so -Wl,-rpath,$PWD/dinov2.cpp_amd
//   ./inference -m model.gguf -i image.ppm [-c] [-k 5] [-o pca_visual.ppm]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dinov2_compat.hpp"

namespace {

// binary PPM (P6, maxval 255) -> BGR interleaved like cv::imread
bool read_ppm_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& h, int& w) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    auto token = [&](std::string& t) {
        t.clear();
        int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} continue; }
            if (!isspace(c)) { t.push_back((char)c); break; }
        }
        while ((c = fgetc(f)) != EOF && !isspace(c)) t.push_back((char)c);
        return !t.empty();
    };
    std::string t;
    bool ok = token(t) && t == "P6" && token(t);
    if (ok) { w = atoi(t.c_str()); ok = token(t); }
    if (ok) { h = atoi(t.c_str()); ok = token(t) && atoi(t.c_str()) == 255 && w > 0 && h > 0; }
    if (ok) {
        std::vector<uint8_t> rgb((size_t)h * w * 3);
        ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
        bgr.resize(rgb.size());
        for (size_t i = 0; ok && i < rgb.size(); i += 3) { bgr[i] = rgb[i + 2]; bgr[i + 1] = rgb[i + 1]; bgr[i + 2] = rgb[i]; }
    }
    fclose(f);
    return ok;
}

bool write_ppm_from_bgr(const std::string& path, const std::vector<uint8_t>& bgr, int h, int w) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<uint8_t> rgb(bgr.size());
    for (size_t i = 0; i < bgr.size(); i += 3) { rgb[i] = bgr[i + 2]; rgb[i + 1] = bgr[i + 1]; rgb[i + 2] = bgr[i]; }
    const bool ok = fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    fclose(f);
    return ok;
}

}  // namespace

int main(int argc, char** argv) {
    dino_params params;
    params.fname_inp = "../assets/tench.ppm";
    params.image_out = "pca_visual.ppm";
    if (!dino_params_parse(argc, argv, params)) return 1;
    fprintf(stderr, "%s: seed = %u\n", __func__, params.seed);
    std::vector<uint8_t> bgr;
    int h = 0, w = 0;
    if (!read_ppm_bgr(params.fname_inp, bgr, h, w)) {
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
        if (write_ppm_from_bgr(params.image_out, big, oh, ow)) fprintf(stderr, "%s: Saved image to: %s\n", __func__, params.image_out.c_str());
        else fprintf(stderr, "%s: failed to save image to '%s'\n", __func__, params.image_out.c_str());
    }
    return 0;
}
