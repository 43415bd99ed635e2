// quantize.cpp -- GGUF -> quantised GGUF re-writer (host only; SURVEY.md section 8(f) "next-3").
//
// Replaces dino_model_quantize (/root/reference/dinov2.h:118, dinov2.cpp:355-453) and the tensor selection rule of
// do_quantize (dinov2.cpp:227-236): every tensor whose name matches `.*weight` and that is 2-D (trailing ones ignored) is
// re-encoded to ggml type `itype` (2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0 -- /root/reference/README.md:342-346); everything
// else is copied; all KVs are copied and `ftype` is overwritten with `itype` (dinov2.cpp:377).  The block encoders restate
// ggml's reference quantisers (quantize_row_q*_ref): 32 weights per block, f16 scale (and minimum), round-to-nearest with the
// upstream's exact arithmetic -- bit-identical to the numpy quantisers in dinov2.cpp_amd/gguf_writer.py that the oracle's
// dequantisers are tested against.  No device code: this lives in the library so that the C++ shim has the whole dinov2.h API.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dinov2_hip.h"
#include "gguf_reader.h"

namespace {

constexpr int QK = 32;

uint16_t f32_to_f16(float f) {  // round to nearest even, like ggml's GGML_FP32_TO_FP16 / numpy astype(float16)
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
float f16_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}

size_t block_bytes(int t) { return t == 2 ? 18 : t == 3 ? 20 : t == 6 ? 22 : t == 7 ? 24 : t == 8 ? 34 : 0; }

void quant_block(const float* x, int t, uint8_t* o) {
    if (t == 8) {  // q8_0: d = amax / 127, q = roundf(x / d)
        float amax = 0.f;
        for (int j = 0; j < QK; ++j) amax = std::fmax(amax, std::fabs(x[j]));
        const float d = amax / 127.0f, id = d != 0.f ? 1.0f / d : 0.f;
        const uint16_t dh = f32_to_f16(d);
        std::memcpy(o, &dh, 2);
        for (int j = 0; j < QK; ++j) {
            const float v = x[j] * id;
            o[2 + j] = (uint8_t)(int8_t)(v < 0 ? -std::floor(-v + 0.5f) : std::floor(v + 0.5f));
        }
        return;
    }
    if (t == 2 || t == 6) {  // q4_0 / q5_0: d = (value of largest magnitude) / -8 (-16), q = (int8)(x / d + 8.5 (16.5)), clamped
        float amax = 0.f, mx = 0.f;
        for (int j = 0; j < QK; ++j)
            if (std::fabs(x[j]) > amax) { amax = std::fabs(x[j]); mx = x[j]; }
        const float d = mx / (t == 2 ? -8.0f : -16.0f), id = d != 0.f ? 1.0f / d : 0.f;
        const float off = t == 2 ? 8.5f : 16.5f;
        const int top = t == 2 ? 15 : 31;
        const uint16_t dh = f32_to_f16(d);
        std::memcpy(o, &dh, 2);
        uint32_t qh = 0;
        uint8_t* qs = o + (t == 2 ? 2 : 6);
        for (int j = 0; j < 16; ++j) {
            const int q0 = std::min(top, (int)(int8_t)(x[j] * id + off)), q1 = std::min(top, (int)(int8_t)(x[j + 16] * id + off));
            qs[j] = (uint8_t)((q0 & 0xF) | ((q1 & 0xF) << 4));
            qh |= ((uint32_t)(q0 & 0x10) >> 4) << j;
            qh |= ((uint32_t)(q1 & 0x10) >> 4) << (j + 16);
        }
        if (t == 6) std::memcpy(o + 2, &qh, 4);
        return;
    }
    // q4_1 / q5_1: d = (max - min) / 15 (31), m = min, q = (int)((x - min) / d + 0.5)
    float mn = x[0], mx = x[0];
    for (int j = 1; j < QK; ++j) { mn = std::fmin(mn, x[j]); mx = std::fmax(mx, x[j]); }
    const float d = (mx - mn) / (t == 3 ? 15.0f : 31.0f), id = d != 0.f ? 1.0f / d : 0.f;
    const uint16_t dh = f32_to_f16(d), mh = f32_to_f16(mn);
    std::memcpy(o, &dh, 2);
    std::memcpy(o + 2, &mh, 2);
    uint32_t qh = 0;
    uint8_t* qs = o + (t == 3 ? 4 : 8);
    for (int j = 0; j < 16; ++j) {
        const float x0 = (x[j] - mn) * id + 0.5f, x1 = (x[j + 16] - mn) * id + 0.5f;
        int q0, q1;
        if (t == 3) { q0 = std::min(15, (int)(int8_t)x0); q1 = std::min(15, (int)(int8_t)x1); }
        else { q0 = (uint8_t)x0; q1 = (uint8_t)x1; }
        qs[j] = (uint8_t)((q0 & 0xF) | ((q1 & 0xF) << 4));
        qh |= ((uint32_t)(q0 & 0x10) >> 4) << j;
        qh |= ((uint32_t)(q1 & 0x10) >> 4) << (j + 16);
    }
    if (t == 7) std::memcpy(o + 4, &qh, 4);
}

struct Rd {
    const std::vector<uint8_t>& b;
    size_t p = 0;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (p + sizeof(T) > b.size()) { ok = false; return v; }
        std::memcpy(&v, &b[p], sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (!ok || n > b.size() - p) { ok = false; return {}; }
        std::string s((const char*)&b[p], (size_t)n);
        p += (size_t)n;
        return s;
    }
};

void put(std::vector<uint8_t>& o, const void* p, size_t n) { o.insert(o.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
void put_str(std::vector<uint8_t>& o, const std::string& s) {
    const uint64_t n = s.size();
    put(o, &n, 8);
    put(o, s.data(), s.size());
}
void fail(char* err, size_t n, const char* msg, const std::string& a = "") {
    if (err && n) snprintf(err, n, "%s%s", msg, a.c_str());
}
size_t scalar_size(uint32_t t) {  // gguf value types
    switch (t) {
        case 0: case 1: case 7: return 1;
        case 2: case 3: return 2;
        case 4: case 5: case 6: return 4;
        case 10: case 11: case 12: return 8;
        default: return 0;
    }
}

}  // namespace

extern "C" int dinov2_hip_quantize(const char* fname_inp, const char* fname_out, int32_t itype, char* err, size_t errlen) {
    if (!fname_inp || !fname_out) return DINOV2_HIP_ERR_INVALID;
    if (!block_bytes(itype)) {
        fail(err, errlen, "invalid quantization type");  // dinov2.cpp:365-373
        return DINOV2_HIP_ERR_INVALID;
    }
    std::vector<uint8_t> buf;
    {
        FILE* f = fopen(fname_inp, "rb");
        if (!f) { fail(err, errlen, "failed to open ", fname_inp); return DINOV2_HIP_ERR_IO; }
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(n > 0 ? (size_t)n : 0);
        const bool ok = n >= 0 && fread(buf.data(), 1, buf.size(), f) == buf.size();
        fclose(f);
        if (!ok) { fail(err, errlen, "short read: ", fname_inp); return DINOV2_HIP_ERR_IO; }
    }
    if (buf.size() < 24 || std::memcmp(buf.data(), "GGUF", 4) != 0) { fail(err, errlen, "not a GGUF file: ", fname_inp); return DINOV2_HIP_ERR_FORMAT; }
    Rd r{buf, 4};
    const uint32_t version = r.get<uint32_t>();
    const uint64_t n_tensors = r.get<uint64_t>(), n_kv = r.get<uint64_t>();
    if (version < 2 || version > 3 || n_kv > (1u << 20) || n_tensors > (1u << 20)) { fail(err, errlen, "unsupported GGUF header"); return DINOV2_HIP_ERR_FORMAT; }
    struct KV { std::string key; uint32_t type; std::vector<uint8_t> raw; std::string s; };
    std::vector<KV> kvs;
    uint64_t align = 32;
    std::string arch = "dinov2";
    for (uint64_t i = 0; i < n_kv && r.ok; ++i) {
        KV kv;
        kv.key = r.str();
        kv.type = r.get<uint32_t>();
        if (kv.type == 8) {
            kv.s = r.str();
        } else {
            const size_t n = scalar_size(kv.type);
            if (!n || r.p + n > buf.size()) { fail(err, errlen, "unsupported KV type for key ", kv.key); return DINOV2_HIP_ERR_UNSUPPORTED; }
            kv.raw.assign(&buf[r.p], &buf[r.p] + n);
            r.p += n;
        }
        if (kv.key == "general.alignment" && kv.type == 4) std::memcpy(&align, kv.raw.data(), 4);
        if (kv.key == "general.architecture" && kv.type == 8) arch = kv.s;
        kvs.push_back(std::move(kv));
    }
    struct TI { std::string name; std::vector<uint64_t> ne; uint32_t type; uint64_t off; };
    std::vector<TI> tis;
    for (uint64_t i = 0; i < n_tensors && r.ok; ++i) {
        TI t;
        t.name = r.str();
        const uint32_t nd = r.get<uint32_t>();
        if (nd > 8) { r.ok = false; break; }
        for (uint32_t d = 0; d < nd; ++d) t.ne.push_back(r.get<uint64_t>());
        t.type = r.get<uint32_t>();
        t.off = r.get<uint64_t>();
        tis.push_back(std::move(t));
    }
    if (align == 0) align = 32;  // same reading as the loader (gguf_reader.cpp)
    if (!r.ok || !dinov2::gguf_alignment_ok(align)) { fail(err, errlen, "truncated or corrupt GGUF metadata"); return DINOV2_HIP_ERR_FORMAT; }
    const size_t data0 = (r.p + align - 1) / align * align;
    if (data0 > buf.size() && !tis.empty()) { fail(err, errlen, "GGUF data section starts past the end of the file"); return DINOV2_HIP_ERR_FORMAT; }

    // ---- re-encode ----
    std::vector<std::vector<uint8_t>> datas(tis.size());
    std::vector<uint32_t> types(tis.size());
    double total_in = 0, total_out = 0;
    for (size_t i = 0; i < tis.size(); ++i) {
        const TI& t = tis[i];
        uint64_t n = 1;
        bool huge = false;
        for (auto v : t.ne) {
            if (v != 0 && n > (UINT64_MAX / 8) / v) huge = true;  // n * (bytes per element <= 4) must not wrap either
            else n *= v;
        }
        if (huge) { fail(err, errlen, "implausible element count: ", t.name); return DINOV2_HIP_ERR_FORMAT; }
        const size_t esz = t.type == 0 ? 4 : t.type == 1 ? 2 : 0;
        size_t nd = t.ne.size();
        while (nd > 1 && t.ne[nd - 1] == 1) --nd;
        const bool is_weight = t.name.size() >= 6 && t.name.compare(t.name.size() - 6, 6, "weight") == 0;  // regex ".*weight" (full match)
        const bool quant = is_weight && nd == 2;
        size_t nbytes;
        if (esz) nbytes = (size_t)n * esz;
        else if (block_bytes((int)t.type)) nbytes = (size_t)n / QK * block_bytes((int)t.type);
        else { fail(err, errlen, "unsupported tensor type for ", t.name); return DINOV2_HIP_ERR_UNSUPPORTED; }
        // overflow-safe form of data0 + off + nbytes <= size (a crafted offset near 2^64 wrapped the plain sum)
        if (t.off > buf.size() - data0 || nbytes > buf.size() - data0 - t.off) { fail(err, errlen, "tensor data out of bounds: ", t.name); return DINOV2_HIP_ERR_FORMAT; }
        const uint8_t* src = &buf[data0 + t.off];
        total_in += (double)nbytes;
        if (quant) {
            if (!esz) { fail(err, errlen, "unsupported tensor type for ", t.name); return DINOV2_HIP_ERR_UNSUPPORTED; }  // dinov2.cpp:425
            if (t.ne[0] % QK) { fail(err, errlen, "row length not a multiple of 32: ", t.name); return DINOV2_HIP_ERR_UNSUPPORTED; }
            std::vector<uint8_t>& o = datas[i];
            o.resize((size_t)n / QK * block_bytes(itype));
            float x[QK];
            for (uint64_t b = 0; b < n / QK; ++b) {
                for (int j = 0; j < QK; ++j) {
                    if (t.type == 0) std::memcpy(&x[j], src + (b * QK + j) * 4, 4);
                    else { uint16_t h; std::memcpy(&h, src + (b * QK + j) * 2, 2); x[j] = f16_to_f32(h); }
                    // (a NaN / Inf weight has no block encoding -- the float -> int casts below would be undefined behaviour -- and means a damaged
                    //  file: refuse, as ggml's quantiser validates its rows, rather than write something that loads)
                    if (!std::isfinite(x[j])) { fail(err, errlen, "non-finite value in tensor ", t.name); return DINOV2_HIP_ERR_FORMAT; }
                }
                quant_block(x, itype, &o[(size_t)b * block_bytes(itype)]);
            }
            types[i] = (uint32_t)itype;
        } else {
            datas[i].assign(src, src + nbytes);
            types[i] = t.type;
        }
        total_out += (double)datas[i].size();
    }

    // ---- write (general.architecture first, the other KVs in file order, `ftype` replaced; tensors aligned to the INPUT's
    //      general.alignment, whose KV is copied through unchanged -- writing 32 regardless corrupted files with another value) ----
    std::vector<uint8_t> out;
    const uint32_t v3 = 3;
    uint64_t nkv_out = 1;
    for (const KV& kv : kvs) nkv_out += kv.key != "general.architecture";
    const uint64_t nt = tis.size(), a = align;
    put(out, "GGUF", 4);
    put(out, &v3, 4);
    put(out, &nt, 8);
    put(out, &nkv_out, 8);
    const uint32_t T_STR = 8;
    put_str(out, "general.architecture");
    put(out, &T_STR, 4);
    put_str(out, arch);
    for (const KV& kv : kvs) {
        if (kv.key == "general.architecture") continue;
        put_str(out, kv.key);
        put(out, &kv.type, 4);
        if (kv.type == 8) put_str(out, kv.s);
        else if (kv.key == "ftype" && kv.raw.size() == 4) { const uint32_t v = (uint32_t)itype; put(out, &v, 4); }
        else put(out, kv.raw.data(), kv.raw.size());
    }
    uint64_t off = 0;
    for (size_t i = 0; i < tis.size(); ++i) {
        put_str(out, tis[i].name);
        const uint32_t nd = (uint32_t)tis[i].ne.size();
        put(out, &nd, 4);
        put(out, tis[i].ne.data(), 8 * tis[i].ne.size());
        put(out, &types[i], 4);
        put(out, &off, 8);
        off += (datas[i].size() + a - 1) / a * a;
    }
    out.resize((out.size() + a - 1) / a * a, 0);
    for (auto& d : datas) {
        put(out, d.data(), d.size());
        out.resize((out.size() + a - 1) / a * a, 0);
    }
    FILE* f = fopen(fname_out, "wb");
    if (!f) { fail(err, errlen, "failed to open for writing: ", fname_out); return DINOV2_HIP_ERR_IO; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) { fail(err, errlen, "short write: ", fname_out); return DINOV2_HIP_ERR_IO; }
    static const char* names[9] = {"", "", "q4_0", "q4_1", "", "", "q5_0", "q5_1", "q8_0"};
    printf("dino_model_quantize: model size = %8.2f MB -> quant size = %8.2f MB (%s)\n", total_in / 1048576.0, total_out / 1048576.0, names[itype]);
    return DINOV2_HIP_OK;
}
