// Test driver of examples/jpeg_codec.hpp (tests/test_jpeg_codec.py):  jpeg_tool decode in.jpg out.ppm  |  jpeg_tool encode in.ppm out.jpg
//                                                                     jpeg_tool fuzz in.jpg "<seed> <iterations>"  (mutated copies, in process)
#include <cstdio>
#include <cstring>
#include <random>

#include "../../examples/jpeg_codec.hpp"

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    if (!strcmp(argv[1], "fuzz")) {  // byte flips and truncations of a valid file: the decoder may refuse, it must not misbehave (run under ASan / UBSan)
        std::vector<uint8_t> base;
        if (!dinojpeg::read_file(argv[2], base)) return 2;
        unsigned seed = 0;
        int iters = 0;
        if (sscanf(argv[3], "%u %d", &seed, &iters) != 2) return 2;
        std::mt19937 rng(seed);
        int ok = 0;
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> f = base;
            const int nflip = 1 + (int)(rng() % 8);
            for (int k = 0; k < nflip; ++k) f[rng() % f.size()] = (uint8_t)rng();
            if (rng() % 4 == 0) f.resize(rng() % f.size() + 4);
            std::vector<uint8_t> out;
            int fh = 0, fw = 0;
            ok += dinojpeg::Decoder().decode(f.data(), f.size(), out, fh, fw) ? 1 : 0;
        }
        printf("decoded %d of %d\n", ok, iters);
        return 0;
    }
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
