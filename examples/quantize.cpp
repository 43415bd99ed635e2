// quantize.cpp -- the reference's `quantize` program (/root/reference/quantize.cpp:24-36) on the C++ shim:
//   ./quantize model-f16.gguf model-q4_0.gguf 2        type = 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0
//   g++ -O2 -std=c++17 -I include examples/quantize.cpp -o quantize dinov2.cpp_amd/libdinov2_hip.so -Wl,-rpath,$PWD/dinov2.cpp_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "dinov2_compat.hpp"

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s /path/to/model-f16.gguf /path/to/model-quant.gguf type\n", argv[0]);
        fprintf(stderr, "  type = 2 - q4_0\n  type = 3 - q4_1\n  type = 6 - q5_0\n  type = 7 - q5_1\n  type = 8 - q8_0\n");
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (!dino_model_quantize(argv[1], argv[2], atoi(argv[3]))) {
        fprintf(stderr, "%s: failed to quantize model from '%s'\n", __func__, argv[1]);
        return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    printf("\n%s: quantize time = %8.2f ms\n", __func__, std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
