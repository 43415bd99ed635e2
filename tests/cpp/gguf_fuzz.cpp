// Sanitizer harness for the host-side GGUF code (csrc/gguf_reader.cpp, csrc/quantize.cpp): mutate a valid file N times (byte flips, 8-byte field
// overwrites with extreme values, truncations), open each mutant, walk every key and tensor the way the loader does (touching the first and last byte
// of every tensor's data) and run the quantiser on it.  Built with -fsanitize=address,undefined by tests/test_gguf_and_abi.py: the reader may refuse
// a file, it must not read out of bounds, overflow or hang.   gguf_fuzz <file.gguf> <seed> <count> <scratch dir>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../dinov2.cpp_amd/csrc/gguf_reader.h"
#include "../../include/dinov2_hip.h"

static uint64_t rng_state;
static uint64_t rnd() {  // xorshift64*
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return rng_state * 2685821657736338717ull;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (good.size() < 64) return 2;
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)atoll(argv[2]);
    const int count = atoi(argv[3]);
    const std::string tmp = std::string(argv[4]) + "/mutant.gguf", qout = std::string(argv[4]) + "/mutant_q.gguf";
    const uint64_t extremes[] = {0, 1, 31, 32, 33, 0x7fffffffull, 0xffffffffull, 0x100000000ull, 1ull << 40, 1ull << 62, ~0ull, ~0ull - 31};
    const size_t meta = good.size() < 32768 ? good.size() : 32768;  // header + KV + tensor infos live here
    int opened = 0, refused = 0, quantised = 0;
    volatile unsigned sink = 0;
    for (int it = 0; it < count; ++it) {
        std::vector<uint8_t> b = good;
        const int kind = (int)(rnd() % 4);
        if (kind == 0) {
            const int n = 1 + (int)(rnd() % 4);
            for (int i = 0; i < n; ++i) b[4 + rnd() % (meta - 4)] ^= (uint8_t)(1 + rnd() % 255);
        } else if (kind == 1) {  // an aligned-or-not 8-byte field becomes an extreme value
            const size_t pos = 4 + rnd() % (meta - 12);
            const uint64_t v = extremes[rnd() % (sizeof extremes / sizeof extremes[0])];
            for (int i = 0; i < 8; ++i) b[pos + i] = (uint8_t)(v >> (8 * i));
        } else if (kind == 2) {  // a 4-byte field (types, dimension counts, string lengths' low words)
            const size_t pos = 4 + rnd() % (meta - 8);
            const uint32_t v = (uint32_t)extremes[rnd() % 8];
            for (int i = 0; i < 4; ++i) b[pos + i] = (uint8_t)(v >> (8 * i));
        } else {
            b.resize((size_t)(rnd() % b.size()));
        }
        {
            std::ofstream o(tmp, std::ios::binary | std::ios::trunc);
            o.write((const char*)b.data(), (std::streamsize)b.size());
        }
        dinov2::GgufFile g;
        std::string err;
        if (!g.open(tmp, &err)) {
            if (err.empty()) { fprintf(stderr, "refused without a message (iteration %d)\n", it); return 1; }
            ++refused;
        } else {
            ++opened;
            uint32_t v = 0;
            (void)g.get_u32("hidden_size", &v);
            (void)g.find("general.alignment");
            for (const auto& t : g.tensors()) {
                uint32_t be = 0, bb = 0;
                if (!dinov2::ggml_type_layout(t.type, &be, &bb)) continue;
                if (t.nbytes && t.data) sink += t.data[0] + t.data[t.nbytes - 1];  // the reader vouches for [data, data + nbytes)
                (void)t.nelements();
            }
        }
        if (it % 8 == 0) {  // the quantiser parses the same file again and re-encodes every 2-D *weight tensor
            char e2[256] = {0};
            const int types[] = {2, 3, 6, 7, 8};
            if (dinov2_hip_quantize(tmp.c_str(), qout.c_str(), types[rnd() % 5], e2, sizeof e2) == 0) ++quantised;
        }
    }
    printf("mutants %d: opened %d, refused %d, quantised %d (sink %u)\n", count, opened, refused, quantised, (unsigned)sink);
    return 0;
}
