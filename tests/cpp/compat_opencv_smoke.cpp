// The reference's `inference` main (/root/reference/inference.cpp:24-104), call for call, on include/dinov2_compat.hpp with
// DINOV2_WITH_OPENCV + DINOV2_COMPAT_GGML_NAMES: every line that touches the API -- dino_params_parse, dino_model_load(img.size(),
// ...), dino_[classify_]preprocess(img, img.size(), model.hparams), the ggml_* lines around dino_predict(model, img, params,
// allocr), output->patch_tokens.value() as a cv::Mat -- is written exactly as the reference writes it.  What is NOT here is what
// needs the real OpenCV (imread / PCA / normalize / resize / imwrite): the image is synthetic and the PCA is skipped.
// Compiled against tests/cpp/opencv_stub (this image has no OpenCV); run by tests/test_gguf_and_abi.py.
#define DINOV2_WITH_OPENCV
#define DINOV2_COMPAT_GGML_NAMES
#include <cmath>

#include "dinov2_compat.hpp"

int main(int argc, char** argv) {
    ggml_time_init();
    dino_params params;
    dino_model model;

    if (dino_params_parse(argc, argv, params) == false) {
        return 1;
    }

    fprintf(stderr, "%s: seed = %d\n", __func__, params.seed);

    // load the image (synthetic stand-in for cv::imread(params.fname_inp, cv::IMREAD_COLOR))
    cv::Mat img(90, 123, CV_8UC3);
    for (int i = 0; i < 90 * 123 * 3; ++i) img.data[i] = (unsigned char)((i * 2654435761u) >> 24);
    fprintf(stderr, "%s: loaded image '%s' (%d x %d)\n", __func__, params.fname_inp.c_str(), img.size[0], img.size[1]);

    // load the model
    if (!dino_model_load(img.size(), params.model, model, params)) {
        fprintf(stderr, "%s: failed to load model from '%s'\n", __func__, params.model.c_str());
        return 1;
    }

    if (params.classify)
        img = dino_classify_preprocess(img, img.size(), model.hparams);
    else
        img = dino_preprocess(img, img.size(), model.hparams);

    cv::Size original_size = img.size();

    fprintf(stderr, "%s: preprocessed image (%d x %d)\n", __func__, img.size[0], img.size[1]);

    // prepare for graph computation, memory allocation and results processing
    {
        ggml_backend_synchronize(model.backend);
        ggml_gallocr_t allocr = ggml_gallocr_new(ggml_backend_get_default_buffer_type(model.backend));
        int64_t start_time = ggml_time_ms();
        std::unique_ptr<dino_output> output = dino_predict(model, img, params, allocr);
        ggml_backend_synchronize(model.backend);
        int64_t end_time = ggml_time_ms();
        fprintf(stderr, "%s: graph computation took %lld ms\n", __func__, (long long)(end_time - start_time));

        ggml_free(model.ctx);
        ggml_gallocr_free(allocr);
        ggml_backend_buffer_free(model.buffer);
        ggml_backend_free(model.backend);

        if (!output) return 1;
        if (!params.classify) {
            const cv::Mat& patch_tokens = output->patch_tokens.value();
            printf("patch_tokens: %d x %d type %d (input %d x %d)\n", patch_tokens.rows, patch_tokens.cols, patch_tokens.type(),
                   original_size.height, original_size.width);
            const float* t = reinterpret_cast<const float*>(patch_tokens.data);
            double s = 0;
            for (int i = 0; i < patch_tokens.rows * patch_tokens.cols; ++i) s += (double)t[i] * t[i];
            printf("patch_tokens rms %.4f\n", std::sqrt(s / ((double)patch_tokens.rows * patch_tokens.cols)));
        }
    }
    return 0;
}
