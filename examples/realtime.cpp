// realtime.cpp -- the loop of the reference's `realtime` program (/root/reference/realtime.cpp:56-108) on the C++ shim
// (include/dinov2_compat.hpp) and the C-ABI, HEADLESS: no camera (cv::VideoCapture) and no window (cv::imshow) -- frames are
// synthesised (a moving pattern, 854 x 480 like realtime.h) or read from one PPM file, and the last combined frame
// (input | PCA map, as hconcat builds it) can be written to a PPM.  Per frame, as in the reference:
//   frame (854 x 480, 8-bit BGR) -> dino_preprocess (868 x 490 = 62 x 35 = 2 170 patches) -> timed dino_predict ->
//   PCA(3) of the patch tokens + project -> min-max to 0..255 -> 35 x 62 x 3 map -> nearest-neighbour resize to the frame size.
// What changes underneath: the raw 8-bit frame goes to the device as it is (DINOV2_HIP_U8_BGR_HWC: preprocessing runs there),
// the patch tokens never leave the device (predict with no token output, then dinov2_hip_pca3(tokens = NULL) works on what the
// forward left in the session) and only the [P, 3] projection crosses PCIe.
//
//   g++ -O2 -std=c++17 -I include examples/realtime.cpp -o realtime dinov2.cpp_amd/libdinov2_hip.so -Wl,-rpath,$PWD/dinov2.cpp_amd
//   ./realtime -m model.gguf [-n frames (default 30)] [-i frame.ppm] [-o last_combined.ppm]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dinov2_compat.hpp"
#include "jpeg_codec.hpp"

constexpr int FRAME_WIDTH = 854;  // realtime.h:4-5
constexpr int FRAME_HEIGHT = 480;

namespace {

// cv::resize(src, dst, size, 0, 0, INTER_NEAREST) for 3-channel 8-bit images: source index = min(floor(dst * (1 / (dst/src))), src - 1)
void resize_nearest(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    const double ify = 1.0 / ((double)dh / sh), ifx = 1.0 / ((double)dw / sw);
    for (int y = 0; y < dh; ++y) {
        const int sy = std::min((int)std::floor(y * ify), sh - 1);
        for (int x = 0; x < dw; ++x) {
            const int sx = std::min((int)std::floor(x * ifx), sw - 1);
            memcpy(dst + ((size_t)y * dw + x) * 3, src + ((size_t)sy * sw + sx) * 3, 3);
        }
    }
}

// what the camera would deliver: three soft blobs drifting over a gradient (gives the tokens a few dominant directions)
void synth_frame(int t, std::vector<uint8_t>& bgr, int h, int w) {
    bgr.resize((size_t)h * w * 3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float c[3] = {40.f + 60.f * x / w, 50.f + 50.f * y / h, 70.f};
            for (int k = 0; k < 3; ++k) {
                const float cx = w * (0.25f + 0.25f * k) + 60.f * std::sin(0.21f * t + 2.1f * k);
                const float cy = h * (0.35f + 0.15f * k) + 40.f * std::cos(0.17f * t + 1.3f * k);
                const float d2 = ((x - cx) * (x - cx) + (y - cy) * (y - cy)) / (70.f * 70.f);
                c[k] += 150.f * std::exp(-d2);
            }
            uint8_t* p = &bgr[((size_t)y * w + x) * 3];
            for (int k = 0; k < 3; ++k) p[k] = (uint8_t)std::min(255.f, c[k]);
        }
}

}  // namespace

int main(int argc, char** argv) {
    dino_params params;
    params.fname_inp = "";
    params.image_out = "";
    int frames = 30;
    std::vector<char*> rest{argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) frames = atoi(argv[++i]);
        else rest.push_back(argv[i]);
    }
    if (!dino_params_parse((int)rest.size(), rest.data(), params)) return 1;
    fprintf(stderr, "%s: seed = %u\n", __func__, params.seed);
    dino_model model;
    if (!dino_model_load(Size2i{FRAME_WIDTH, FRAME_HEIGHT}, params.model, model, params)) {
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }
    const int ps = (int)model.hparams.patch_size, H = (int)model.hparams.hidden_size;
    const int new_w = (FRAME_WIDTH / ps + 1) * ps, new_h = (FRAME_HEIGHT / ps + 1) * ps;  // realtime.cpp:50-51
    const int gr = new_h / ps, gc = new_w / ps, P = gr * gc;

    std::vector<uint8_t> file_frame, frame, small((size_t)P * 3), pca_image((size_t)FRAME_HEIGHT * FRAME_WIDTH * 3), combined;
    int fh = 0, fw = 0;
    if (!params.fname_inp.empty() && !dinojpeg::imread_bgr(params.fname_inp, file_frame, fh, fw)) {
        fprintf(stderr, "%s: failed to load image from '%s'\n", __func__, params.fname_inp.c_str());
        return 1;
    }
    std::vector<float> proj((size_t)P * 3);
    dinov2_hip_session* sess = model.default_session;  // the reference's allocr, reused by every frame (realtime.cpp:53)
    char err[512] = {0};
    double sum_predict = 0.0, sum_loop = 0.0;
    for (int t = 0; t < frames; ++t) {
        const auto l0 = std::chrono::steady_clock::now();
        if (file_frame.empty()) {
            synth_frame(t, frame, FRAME_HEIGHT, FRAME_WIDTH);
        } else {  // cv::resize(frame, frame, size, 0, 0, INTER_NEAREST) (realtime.cpp:63)
            frame.resize((size_t)FRAME_HEIGHT * FRAME_WIDTH * 3);
            resize_nearest(file_frame.data(), fh, fw, frame.data(), FRAME_HEIGHT, FRAME_WIDTH);
        }
        // dino_preprocess + dino_predict (realtime.cpp:66-74): the raw frame in, nothing out -- tokens stay in the session
        dinov2_hip_input in{reinterpret_cast<const float*>(frame.data()), 1, FRAME_HEIGHT, FRAME_WIDTH, DINOV2_HIP_U8_BGR_HWC, 0};
        dinov2_hip_session_sync(sess);
        const auto t0 = std::chrono::steady_clock::now();
        if (dinov2_hip_predict(sess, &in, nullptr, 0, err, sizeof err) != DINOV2_HIP_OK) {
            fprintf(stderr, "%s: %s\n", __func__, err);
            return 1;
        }
        dinov2_hip_session_sync(sess);
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        fprintf(stderr, "%s: graph computation took %lld ms\n", __func__, (long long)std::llround(ms));
        // cv::PCA(patch_tokens, Mat(), DATA_AS_ROW, 3) + project (realtime.cpp:77-82) on the device-resident tokens
        if (dinov2_hip_pca3(sess, nullptr, P, H, 0, nullptr, nullptr, proj.data(), err, sizeof err) != DINOV2_HIP_OK) {
            fprintf(stderr, "%s: PCA failed: %s\n", __func__, err);
            return 1;
        }
        // cv::normalize(projected, projected_norm, 0, 255, NORM_MINMAX, CV_8U); reshape(3, rows) (realtime.cpp:84-87)
        float lo = proj[0], hi = proj[0];
        for (float v : proj) { lo = std::min(lo, v); hi = std::max(hi, v); }
        for (size_t i = 0; i < small.size(); ++i)
            small[i] = (uint8_t)std::min(255.0f, std::max(0.0f, std::nearbyint(hi == lo ? 0.f : (proj[i] - lo) * (255.0f / (hi - lo)))));
        resize_nearest(small.data(), gr, gc, pca_image.data(), FRAME_HEIGHT, FRAME_WIDTH);  // realtime.cpp:89
        combined.resize((size_t)FRAME_HEIGHT * 2 * FRAME_WIDTH * 3);                        // hconcat {frame, pca_image} (:91-93)
        for (int y = 0; y < FRAME_HEIGHT; ++y) {
            memcpy(&combined[(size_t)y * 2 * FRAME_WIDTH * 3], &frame[(size_t)y * FRAME_WIDTH * 3], (size_t)FRAME_WIDTH * 3);
            memcpy(&combined[((size_t)y * 2 + 1) * FRAME_WIDTH * 3], &pca_image[(size_t)y * FRAME_WIDTH * 3], (size_t)FRAME_WIDTH * 3);
        }
        const auto l1 = std::chrono::steady_clock::now();
        if (t > 0) {  // (frame 0 pays for the workspace allocation and the pos-embed interpolation)
            sum_predict += ms;
            sum_loop += std::chrono::duration<double, std::milli>(l1 - l0).count();
        }
    }
    if (frames > 1)
        printf("%s: %d frames, %d x %d -> %d patches: predict %.2f ms/frame, whole loop %.2f ms/frame (%.1f frames/s)\n", __func__, frames,
               FRAME_WIDTH, FRAME_HEIGHT, P, sum_predict / (frames - 1), sum_loop / (frames - 1), 1e3 * (frames - 1) / sum_loop);
    if (!params.image_out.empty()) {
        if (dinojpeg::imwrite_bgr(params.image_out, combined.data(), FRAME_HEIGHT, 2 * FRAME_WIDTH))
            fprintf(stderr, "%s: Saved image to: %s\n", __func__, params.image_out.c_str());
        else
            fprintf(stderr, "%s: failed to save image to '%s'\n", __func__, params.image_out.c_str());
    }
    printf("REALTIME_OK\n");
    return 0;
}
