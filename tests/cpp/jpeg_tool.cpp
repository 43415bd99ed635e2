// Test driver of examples/jpeg_codec.hpp (tests/test_jpeg_codec.py):  jpeg_tool decode in.jpg out.ppm  |  jpeg_tool encode in.ppm out.jpg
#include <cstdio>
#include <cstring>

#include "../../examples/jpeg_codec.hpp"

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    std::vector<uint8_t> bgr;
    int h = 0, w = 0;
    std::string err;
    if (!dinojpeg::imread_bgr(argv[2], bgr, h, w, &err)) {
        fprintf(stderr, "read failed: %s\n", err.c_str());
        return 1;
    }
    if (!strcmp(argv[1], "decode") || !strcmp(argv[1], "encode")) return dinojpeg::imwrite_bgr(argv[3], bgr.data(), h, w) ? 0 : 1;
    return 2;
}
