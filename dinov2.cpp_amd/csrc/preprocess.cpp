// preprocess.cpp -- host image preprocessing without OpenCV (SURVEY.md section 8(f) "next-1": the step immediately
// before the hot path).
//
// Replaces  dino_preprocess           /root/reference/dinov2.cpp:135-156  (resize to (dim/patch + 1)*patch, normalise)
//           dino_classify_preprocess  /root/reference/dinov2.cpp:106-132  (resize to 256x256 ignoring aspect, centre
//                                                                          crop 224, normalise)
// for 8-bit BGR interleaved images (what cv::imread hands the reference, inference.cpp:36).  Steps, in the reference's
// order: convertTo(CV_32FC3, 1/255) -> cv::resize(INTER_CUBIC) on the float image -> (crop) -> per channel
// (c - mean[2-i]) / std[2-i] on the B,G,R planes.  The output is the continuous CV_32FC3-compatible BGR image that
// dino_predict (DINOV2_HIP_BGR_HWC) takes.  Both quirks are kept: the resize ALWAYS grows by one patch, even when the
// side is already a multiple of the patch size (dinov2.cpp:140-141), and the classify resize ignores aspect ratio.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/dinov2_hip.h"

namespace {

const float kMean[3] = {0.485f, 0.456f, 0.406f};  // IMAGENET_DEFAULT_MEAN, RGB order (dinov2.h:16)
const float kStd[3] = {0.229f, 0.224f, 0.225f};   // IMAGENET_DEFAULT_STD            (dinov2.h:17)

void cubic_taps(float t, float w[4]) {  // cv::resize INTER_CUBIC coefficients, A = -0.75
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

struct Axis {
    std::vector<int> idx;
    std::vector<float> wgt;
    Axis(int src, int dst) : idx(4 * (size_t)dst), wgt(4 * (size_t)dst) {
        const float scale = (float)src / (float)dst;
        for (int d = 0; d < dst; ++d) {
            float f = ((float)d + 0.5f) * scale - 0.5f;
            const int s = (int)std::floor(f);
            f -= (float)s;
            cubic_taps(f, &wgt[4 * (size_t)d]);
            for (int k = 0; k < 4; ++k) idx[4 * (size_t)d + k] = std::min(std::max(s - 1 + k, 0), src - 1);
        }
    }
};

}  // namespace

extern "C" int dinov2_hip_preprocess_size(int32_t mode, int32_t height, int32_t width, int32_t patch, int32_t* out_h,
                                          int32_t* out_w) {
    if (!out_h || !out_w || height <= 0 || width <= 0 || patch <= 0 || (mode != 0 && mode != 1)) return DINOV2_HIP_ERR_INVALID;
    if (mode == 1) {
        *out_h = *out_w = 224;  // crop_size, dinov2.cpp:116
    } else {
        *out_w = (width / patch + 1) * patch;  // dinov2.cpp:140-141
        *out_h = (height / patch + 1) * patch;
    }
    return DINOV2_HIP_OK;
}

extern "C" int dinov2_hip_preprocess(int32_t mode, const uint8_t* bgr, int32_t height, int32_t width, int32_t patch,
                                     float* out) {
    int32_t oh, ow;
    if (!bgr || !out || dinov2_hip_preprocess_size(mode, height, width, patch, &oh, &ow) != DINOV2_HIP_OK)
        return DINOV2_HIP_ERR_INVALID;
    const int rh = mode == 1 ? 256 : oh, rw = mode == 1 ? 256 : ow;  // resize target (dinov2.cpp:111, 140)
    const int y0 = (rh - oh) / 2, x0 = (rw - ow) / 2;                // crop offsets (dinov2.cpp:117-118); 0 for features
    const float inv255 = (float)(1.0 / 255.0);                       // convertTo(CV_32FC3, 1.0 / 255.0)
    const Axis ax(width, rw), ay(height, rh);

    // horizontal pass for the source rows the vertical pass needs, then the vertical blend: the order cv::resize uses
    std::vector<float> hbuf((size_t)height * ow * 3);
    for (int y = 0; y < height; ++y) {
        const uint8_t* srow = bgr + (size_t)y * width * 3;
        float* hrow = &hbuf[(size_t)y * ow * 3];
        for (int x = 0; x < ow; ++x) {
            const int* ix = &ax.idx[4 * (size_t)(x + x0)];
            const float* wx = &ax.wgt[4 * (size_t)(x + x0)];
            for (int c = 0; c < 3; ++c)
                hrow[x * 3 + c] = (float)srow[ix[0] * 3 + c] * inv255 * wx[0] + (float)srow[ix[1] * 3 + c] * inv255 * wx[1] +
                                  (float)srow[ix[2] * 3 + c] * inv255 * wx[2] + (float)srow[ix[3] * 3 + c] * inv255 * wx[3];
        }
    }
    for (int y = 0; y < oh; ++y) {
        const int* iy = &ay.idx[4 * (size_t)(y + y0)];
        const float* wy = &ay.wgt[4 * (size_t)(y + y0)];
        const float* r0 = &hbuf[(size_t)iy[0] * ow * 3];
        const float* r1 = &hbuf[(size_t)iy[1] * ow * 3];
        const float* r2 = &hbuf[(size_t)iy[2] * ow * 3];
        const float* r3 = &hbuf[(size_t)iy[3] * ow * 3];
        float* orow = out + (size_t)y * ow * 3;
        for (int x = 0; x < ow; ++x)
            for (int c = 0; c < 3; ++c) {  // c: 0 = B, 1 = G, 2 = R  ->  mean/std index 2 - c (dinov2.cpp:124-127, 148-151)
                const int i = x * 3 + c;
                const float v = r0[i] * wy[0] + r1[i] * wy[1] + r2[i] * wy[2] + r3[i] * wy[3];
                orow[i] = (v - kMean[2 - c]) / kStd[2 - c];
            }
    }
    return DINOV2_HIP_OK;
}
