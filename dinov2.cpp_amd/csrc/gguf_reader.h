// Dependency-free GGUF v2/v3 reader (mmap).  Replaces gguf_init_from_file + the gguf_get_* calls of
// /root/reference/dinov2.cpp:263-339 (ggml's gguf.h is an un-vendored submodule of the reference).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace dinov2 {

enum GgmlType : uint32_t {
    GGML_F32 = 0, GGML_F16 = 1, GGML_Q4_0 = 2, GGML_Q4_1 = 3, GGML_Q5_0 = 6, GGML_Q5_1 = 7, GGML_Q8_0 = 8,
    GGML_BF16 = 30
};

// elements per block / bytes per block; returns false for types outside the set the reference can produce
// (F32/F16 from the converter, Q4_0..Q8_0 from quantize.cpp: /root/reference/README.md:342-346)
bool ggml_type_layout(uint32_t type, uint32_t* block_elems, uint32_t* block_bytes);

struct GgufTensor {
    std::string name;
    std::vector<uint64_t> ne;  // ne[0] fastest
    uint32_t type = 0;
    uint64_t offset = 0;       // from the start of the data section
    uint64_t nbytes = 0;
    const uint8_t* data = nullptr;  // into the mapping
    // product of the dimensions; UINT64_MAX when it does not fit (a crafted file): callers treat that as "too large"
    uint64_t nelements() const {
        uint64_t n = 1;
        for (auto v : ne) {
            if (v != 0 && n > UINT64_MAX / v) return UINT64_MAX;
            n *= v;
        }
        return n;
    }
};

struct GgufValue {
    uint32_t type = 0;  // gguf value type id
    uint64_t u = 0;     // integer / bool payload
    double f = 0;       // float payload
    std::string s;      // string payload
};

class GgufFile {
public:
    GgufFile() = default;
    ~GgufFile();
    GgufFile(const GgufFile&) = delete;
    GgufFile& operator=(const GgufFile&) = delete;

    // returns false and fills err on failure; never throws
    bool open(const std::string& path, std::string* err);

    const GgufValue* find(const std::string& key) const;
    bool get_u32(const std::string& key, uint32_t* out) const;  // get_val_u32, dinov2.cpp:55-61 (asserts there)
    const GgufTensor* tensor(const std::string& name) const;
    const std::vector<GgufTensor>& tensors() const { return tensors_; }
    uint32_t version() const { return version_; }

private:
    void* map_ = nullptr;
    size_t map_len_ = 0;
    int fd_ = -1;
    uint32_t version_ = 0;
    std::map<std::string, GgufValue> kv_;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, size_t> index_;
};

// general.alignment as the loader AND the quantiser accept it (one rule): a power of two in [1, 2^20]; 0 / absent = 32
inline bool gguf_alignment_ok(uint64_t a) { return a >= 1 && a <= ((uint64_t)1 << 20) && (a & (a - 1)) == 0; }

}  // namespace dinov2
