// Sanitizer harness for csrc/preprocess.cpp (dino_preprocess / dino_classify_preprocess of /root/reference/dinov2.cpp:106-156 on the host): both modes over
// a sweep of image sizes -- tiny (1 x 1 ... smaller than one patch), odd, extreme aspect ratios, a few random ones -- with the output buffer sized exactly
// as dinov2_hip_preprocess_size says, under AddressSanitizer + UndefinedBehaviorSanitizer (built by tests/test_preprocess.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/dinov2_hip.h"

int main() {
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    int ok = 0, refused = 0;
    std::vector<std::pair<int, int>> sizes = {{1, 1}, {1, 7}, {7, 1}, {13, 13}, {14, 14}, {15, 29}, {224, 224}, {518, 518}, {480, 854}, {3, 2000}, {2000, 3}, {257, 255}};
    for (int i = 0; i < 40; ++i) sizes.push_back({1 + (int)(rnd() % 700), 1 + (int)(rnd() % 700)});
    for (int mode = 0; mode < 2; ++mode)
        for (auto [h, w] : sizes)
            for (int patch : {14, 16}) {
                int32_t oh = -1, ow = -1;
                const int rc = dinov2_hip_preprocess_size(mode, h, w, patch, &oh, &ow);
                if (rc != 0 || oh <= 0 || ow <= 0) { ++refused; continue; }
                if (oh % patch || ow % patch) { fprintf(stderr, "size %d x %d mode %d -> %d x %d is not a multiple of the patch\n", h, w, mode, oh, ow); return 1; }
                std::vector<uint8_t> img((size_t)h * w * 3);
                for (auto& b : img) b = (uint8_t)rnd();
                std::vector<float> out((size_t)oh * ow * 3);  // exactly what the size call promised: one element more written = an ASan report
                if (dinov2_hip_preprocess(mode, img.data(), h, w, patch, out.data()) != 0) { ++refused; continue; }
                for (float v : out)
                    if (!std::isfinite(v)) { fprintf(stderr, "non-finite output for %d x %d mode %d\n", h, w, mode); return 1; }
                ++ok;
            }
    // invalid arguments are statuses
    int32_t oh, ow;
    if (dinov2_hip_preprocess_size(0, 0, 10, 14, &oh, &ow) == 0 || dinov2_hip_preprocess_size(0, 10, -1, 14, &oh, &ow) == 0 || dinov2_hip_preprocess_size(0, 10, 10, 0, &oh, &ow) == 0) {
        fprintf(stderr, "an invalid size was accepted\n");
        return 1;
    }
    printf("preprocessed %d, refused %d\n", ok, refused);
    return 0;
}
