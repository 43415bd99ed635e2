// GGUF v2/v3 reader.  Format (little endian): "GGUF", u32 version, u64 n_tensors, u64 n_kv, KVs,
// tensor infos, padding to general.alignment (default 32), tensor data.
#include "gguf_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>

namespace dinov2 {

bool ggml_type_layout(uint32_t type, uint32_t* be, uint32_t* bb) {
    switch (type) {
        case GGML_F32: *be = 1; *bb = 4; return true;
        case GGML_F16: *be = 1; *bb = 2; return true;
        case GGML_BF16: *be = 1; *bb = 2; return true;
        case GGML_Q4_0: *be = 32; *bb = 18; return true;
        case GGML_Q4_1: *be = 32; *bb = 20; return true;
        case GGML_Q5_0: *be = 32; *bb = 22; return true;
        case GGML_Q5_1: *be = 32; *bb = 24; return true;
        case GGML_Q8_0: *be = 32; *bb = 34; return true;
        default: return false;
    }
}

namespace {

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    template <typename T>
    T rd() {
        T v{};
        if (!ok || (size_t)(end - p) < sizeof(T)) {
            ok = false;
            return v;
        }
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string rstr() {
        uint64_t n = rd<uint64_t>();
        if (!ok || (uint64_t)(end - p) < n) {
            ok = false;
            return {};
        }
        std::string s(reinterpret_cast<const char*>(p), (size_t)n);
        p += n;
        return s;
    }
};

enum : uint32_t { T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 };

bool read_value(Cursor& c, uint32_t type, GgufValue* out, int depth) {
    out->type = type;
    switch (type) {
        case T_U8: out->u = c.rd<uint8_t>(); break;
        case T_I8: out->u = (uint64_t)(int64_t)c.rd<int8_t>(); break;
        case T_U16: out->u = c.rd<uint16_t>(); break;
        case T_I16: out->u = (uint64_t)(int64_t)c.rd<int16_t>(); break;
        case T_U32: out->u = c.rd<uint32_t>(); break;
        case T_I32: out->u = (uint64_t)(int64_t)c.rd<int32_t>(); break;
        case T_F32: out->f = c.rd<float>(); break;
        case T_BOOL: out->u = c.rd<uint8_t>(); break;
        case T_STR: out->s = c.rstr(); break;
        case T_U64: out->u = c.rd<uint64_t>(); break;
        case T_I64: out->u = (uint64_t)c.rd<int64_t>(); break;
        case T_F64: out->f = c.rd<double>(); break;
        case T_ARR: {
            if (depth > 2) return false;
            uint32_t et = c.rd<uint32_t>();
            uint64_t n = c.rd<uint64_t>();
            GgufValue tmp;  // arrays are parsed past, not kept: the DINOv2 schema has none
            for (uint64_t i = 0; i < n && c.ok; ++i)
                if (!read_value(c, et, &tmp, depth + 1)) return false;
            out->u = n;
            break;
        }
        default: return false;
    }
    return c.ok;
}

}  // namespace

GgufFile::~GgufFile() {
    if (map_) munmap(map_, map_len_);
    if (fd_ >= 0) close(fd_);
}

bool GgufFile::open(const std::string& path, std::string* err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) {
        *err = "failed to open '" + path + "': " + std::strerror(errno);
        return false;
    }
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 24) {
        *err = "'" + path + "' is too small to be a GGUF file";
        return false;
    }
    map_len_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (map_ == MAP_FAILED) {
        map_ = nullptr;
        *err = "mmap failed for '" + path + "'";
        return false;
    }
    const uint8_t* base = static_cast<const uint8_t*>(map_);
    Cursor c{base, base + map_len_};
    if (std::memcmp(base, "GGUF", 4) != 0) {
        *err = "'" + path + "' is not a GGUF file (bad magic)";
        return false;
    }
    c.p += 4;
    version_ = c.rd<uint32_t>();
    if (version_ != 2 && version_ != 3) {
        *err = "unsupported GGUF version " + std::to_string(version_);
        return false;
    }
    const uint64_t n_tensors = c.rd<uint64_t>();
    const uint64_t n_kv = c.rd<uint64_t>();
    if (!c.ok || n_tensors > (1u << 20) || n_kv > (1u << 20)) {
        *err = "corrupt GGUF header";
        return false;
    }
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string key = c.rstr();
        uint32_t type = c.rd<uint32_t>();
        GgufValue v;
        if (!c.ok || !read_value(c, type, &v, 0)) {
            *err = "corrupt GGUF key/value section near key '" + key + "'";
            return false;
        }
        kv_[key] = std::move(v);
    }
    tensors_.resize((size_t)n_tensors);
    for (auto& t : tensors_) {
        t.name = c.rstr();
        uint32_t nd = c.rd<uint32_t>();
        if (!c.ok || nd > 8) {
            *err = "corrupt GGUF tensor info";
            return false;
        }
        t.ne.resize(nd);
        for (auto& d : t.ne) d = c.rd<uint64_t>();
        t.type = c.rd<uint32_t>();
        t.offset = c.rd<uint64_t>();
        if (!c.ok) {
            *err = "corrupt GGUF tensor info";
            return false;
        }
    }
    uint64_t align = 32;
    if (const GgufValue* a = find("general.alignment")) align = a->u ? a->u : 32;
    if (!gguf_alignment_ok(align)) {
        *err = "implausible general.alignment";
        return false;
    }
    const uint64_t data0 = ((uint64_t)(c.p - base) + align - 1) / align * align;
    if (data0 > map_len_ && !tensors_.empty()) {
        *err = "GGUF data section starts past the end of the file";
        return false;
    }
    for (size_t i = 0; i < tensors_.size(); ++i) {
        auto& t = tensors_[i];
        uint32_t be, bb;
        if (!ggml_type_layout(t.type, &be, &bb)) {
            *err = "tensor '" + t.name + "' has unsupported ggml type " + std::to_string(t.type);
            return false;
        }
        const uint64_t n = t.nelements();
        if (t.ne.empty() || t.ne[0] % be != 0) {
            *err = "tensor '" + t.name + "' row length is not a multiple of its block size";
            return false;
        }
        if (n == UINT64_MAX || n / be > UINT64_MAX / bb) {
            *err = "tensor '" + t.name + "' has an implausible element count";
            return false;
        }
        t.nbytes = n / be * bb;
        // overflow-safe form of data0 + offset + nbytes <= map_len_ (a crafted offset near 2^64 wrapped the plain sum)
        if (t.offset > map_len_ - data0 || t.nbytes > map_len_ - data0 - t.offset) {
            *err = "tensor '" + t.name + "' extends past the end of the file";
            return false;
        }
        t.data = base + data0 + t.offset;
        index_[t.name] = i;
    }
    return true;
}

const GgufValue* GgufFile::find(const std::string& key) const {
    auto it = kv_.find(key);
    return it == kv_.end() ? nullptr : &it->second;
}

bool GgufFile::get_u32(const std::string& key, uint32_t* out) const {
    const GgufValue* v = find(key);
    if (!v || v->type == T_STR || v->type == T_ARR || v->type == T_F32 || v->type == T_F64) return false;
    *out = (uint32_t)v->u;
    return true;
}

const GgufTensor* GgufFile::tensor(const std::string& name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors_[it->second];
}

}  // namespace dinov2
