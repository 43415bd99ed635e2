// Reads like the body of /root/reference/inference.cpp:24-104 with the include and the image type swapped:
// load -> (synthetic preprocessed image) -> dino_predict -> print.  Usage: compat_smoke model.gguf [classify]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dinov2_compat.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s model.gguf [classify]\n", argv[0]);
        return 2;
    }
    dino_params params;
    params.model = argv[1];
    params.classify = argc > 2;
    params.topk = 3;
    dino_model model;
    if (!dino_model_load(Size2i{70, 70}, params.model, model, params)) {
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }
    const int H = 70, W = 84;
    std::vector<float> pix((size_t)H * W * 3);
    unsigned s = params.seed;
    for (auto& p : pix) {
        s = s * 1664525u + 1013904223u;
        p = ((float)(s >> 8) / 8388608.0f) - 1.0f;
    }
    Mat32f img;
    img.rows = H; img.cols = W; img.channels = 3; img.data = pix.data();
    std::unique_ptr<dino_output> out = dino_predict(model, img, params);
    if (!out) return 1;
    if (params.classify) {
        printf("preds:");
        for (auto p : *out->preds) printf(" %u", p);
        printf("\n");
    } else {
        printf("patch_tokens: %d x %d, first = %.6f\n", out->patch_tokens->rows, out->patch_tokens->cols,
               out->patch_tokens->data[0]);
    }
    return 0;
}
