// jpeg_codec.hpp -- what the reference's programs get from OpenCV's codecs (cv::imread / cv::imwrite, /root/reference/inference.cpp:36, :95;
// realtime.cpp frames), written out for the example programs of this repository, which link no image library:
//   * a JPEG decoder for 8-bit Huffman-coded files, baseline / extended sequential (SOF0 / SOF1) AND progressive (SOF2 -- the reference's
//     own default input, assets/tench.jpg, is progressive), grey or YCbCr, sampling factors 1 or 2 per axis, restart intervals; output
//     BGR interleaved like cv::imread(IMREAD_COLOR).  The arithmetic follows what libjpeg(-turbo) -- the decoder behind cv::imread and
//     PIL -- computes, so the bytes handed to the preprocessing are the bytes the reference sees: the 13-bit fixed-point
//     Loeffler-Ligtenberg-Moschytz inverse DCT ("islow"), triangle-filter ("fancy") chroma upsampling, 16-bit fixed-point YCbCr -> RGB.
//     tests/test_jpeg_codec.py holds it to PIL byte for byte (tench.jpg and generated baseline / progressive / 4:2:0 / 4:2:2 / grey /
//     restart-interval files).
//   * a baseline JPEG encoder (4:4:4, Annex K Huffman tables, libjpeg's quality scaling, default 95 like cv::imwrite) for the PCA picture.
// ITU-T T.81 is the specification followed (section / figure numbers in the comments).  Header-only, no dependencies.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace dinojpeg {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

class Decoder {
public:
    // BGR interleaved, rows top to bottom.  false + message on anything unsupported or corrupt.
    bool decode(const uint8_t* data, size_t size, std::vector<uint8_t>& bgr, int& height, int& width, std::string* err = nullptr) {
        d_ = data;
        n_ = size;
        pos_ = 0;
        msg_.clear();
        const bool ok = run(bgr, height, width);
        if (!ok && err) *err = msg_.empty() ? "corrupt JPEG" : msg_;
        return ok;
    }

private:
    struct Huff {
        bool present = false;
        int maxcode[18];   // largest code of each length, -1 if none (F.2.2.3, figure F.15)
        int valptr[17], mincode[17];
        uint8_t vals[256];
        int lookup[512];   // 9-bit prefix -> (length << 8) | value, 0 = longer code
    };
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
        int bw = 0, bh = 0;     // blocks per row / column of the stored coefficient array (whole MCUs)
        int cw = 0, ch = 0;     // true size in samples: ceil(W h / hmax), ceil(H v / vmax)
        int dcpred = 0;
        std::vector<int16_t> coef;  // bw * bh blocks of 64, natural (row-major) order inside a block
    };

    const uint8_t* d_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    std::string msg_;
    uint16_t qt_[4][64];
    bool qt_ok_[4] = {false, false, false, false};
    Huff hd_[4], ha_[4];
    std::vector<Comp> comps_;
    int W_ = 0, H_ = 0, hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0, restart_ = 0;
    bool progressive_ = false;
    // bit reader
    uint32_t bitbuf_ = 0;
    int bitcnt_ = 0;
    bool hit_marker_ = false;
    int eobrun_ = 0;

    bool fail(const char* m) {
        if (msg_.empty()) msg_ = m;
        return false;
    }
    int u8() { return pos_ < n_ ? d_[pos_++] : -1; }
    int u16() {
        const int a = u8(), b = u8();
        return a < 0 || b < 0 ? -1 : (a << 8) | b;
    }

    // ---- entropy-coded segment: bits MSB first, 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment (B.1.1.5) ----
    void fill() {
        while (bitcnt_ <= 24) {
            int b = 0;
            if (!hit_marker_ && pos_ < n_) {
                b = d_[pos_];
                if (b == 0xFF) {
                    const int b2 = pos_ + 1 < n_ ? d_[pos_ + 1] : 0xD9;
                    if (b2 == 0) {
                        pos_ += 2;
                    } else {  // a marker: leave it for the caller, feed zeros
                        hit_marker_ = true;
                        b = 0;
                    }
                } else {
                    ++pos_;
                }
            } else {
                hit_marker_ = true;
            }
            bitbuf_ |= (uint32_t)b << (24 - bitcnt_);
            bitcnt_ += 8;
        }
    }
    int getbits(int n) {
        if (n == 0) return 0;
        if (bitcnt_ < n) fill();
        const int v = (int)(bitbuf_ >> (32 - n));
        bitbuf_ <<= n;
        bitcnt_ -= n;
        return v;
    }
    int getbit() { return getbits(1); }
    static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }  // figure F.12
    int decode_huff(const Huff& h) {
        if (bitcnt_ < 16) fill();
        const int look = h.lookup[bitbuf_ >> 23];
        if (look) {
            const int len = look >> 8;
            bitbuf_ <<= len;
            bitcnt_ -= len;
            return look & 255;
        }
        int code = (int)(bitbuf_ >> 23), len = 9;  // figure F.16
        for (;;) {
            ++len;
            if (len > 16) return -1;
            code = (int)(bitbuf_ >> (32 - len));
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len]) break;
        }
        bitbuf_ <<= len;
        bitcnt_ -= len;
        return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    void reset_bits() {
        bitbuf_ = 0;
        bitcnt_ = 0;
        hit_marker_ = false;
    }

    bool read_dht(int len) {
        const size_t end = pos_ + (size_t)len - 2;
        while (pos_ < end) {
            const int tc_th = u8();
            const int tc = tc_th >> 4, th = tc_th & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table id");
            Huff& h = tc ? ha_[th] : hd_[th];
            int counts[17], total = 0;
            for (int i = 1; i <= 16; ++i) {
                counts[i] = u8();
                total += counts[i];
            }
            if (total > 256 || pos_ + (size_t)total > n_) return fail("bad Huffman table");
            for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8();
            int code = 0, k = 0;  // figures C.1 - C.3, F.15
            for (int i = 0; i < 512; ++i) h.lookup[i] = 0;
            for (int l = 1; l <= 16; ++l) {
                if (code + counts[l] > (1 << l)) return fail("over-subscribed Huffman table");
                h.valptr[l] = k;
                h.mincode[l] = code;
                for (int i = 0; i < counts[l]; ++i, ++k, ++code)
                    if (l <= 9)
                        for (int f = 0; f < (1 << (9 - l)); ++f) h.lookup[(code << (9 - l)) | f] = (l << 8) | h.vals[k];
                h.maxcode[l] = counts[l] ? code - 1 : -1;
                code <<= 1;
            }
            h.maxcode[17] = 0x7fffffff;
            h.present = true;
        }
        return true;
    }
    bool read_dqt(int len) {
        const size_t end = pos_ + (size_t)len - 2;
        while (pos_ < end) {
            const int pq_tq = u8();
            const int pq = pq_tq >> 4, tq = pq_tq & 15;
            if (tq > 3 || pq > 1) return fail("bad quantisation table");
            for (int i = 0; i < 64; ++i) {
                const int v = pq ? u16() : u8();
                if (v < 0) return fail("truncated quantisation table");
                qt_[tq][kZigzag[i]] = (uint16_t)v;
            }
            qt_ok_[tq] = true;
        }
        return true;
    }
    bool read_sof(int len) {
        (void)len;
        const int prec = u8();
        H_ = u16();
        W_ = u16();
        const int nc = u8();
        if (prec != 8) return fail("only 8-bit JPEG is supported");
        if (H_ <= 0 || W_ <= 0) return fail("bad image size");
        if (nc != 1 && nc != 3) return fail("only grey and 3-component (YCbCr) JPEG is supported");
        comps_.assign((size_t)nc, Comp());
        hmax_ = vmax_ = 1;
        for (auto& c : comps_) {
            c.id = u8();
            const int hv = u8();
            c.h = hv >> 4;
            c.v = hv & 15;
            c.tq = u8();
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("unsupported sampling factors");
            hmax_ = c.h > hmax_ ? c.h : hmax_;
            vmax_ = c.v > vmax_ ? c.v : vmax_;
        }
        if (nc == 1) comps_[0].h = comps_[0].v = hmax_ = vmax_ = 1;  // (a single component is never interleaved: A.2.2)
        mcux_ = (W_ + 8 * hmax_ - 1) / (8 * hmax_);
        mcuy_ = (H_ + 8 * vmax_ - 1) / (8 * vmax_);
        for (auto& c : comps_) {
            c.bw = mcux_ * c.h;
            c.bh = mcuy_ * c.v;
            c.cw = (W_ * c.h + hmax_ - 1) / hmax_;
            c.ch = (H_ * c.v + vmax_ - 1) / vmax_;
            if ((size_t)c.bw * c.bh > ((size_t)1 << 24)) return fail("image too large");
            c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        return true;
    }

    // ---- one block of one scan ----
    bool block_baseline(Comp& c, int16_t* b) {
        const int t = decode_huff(hd_[c.td]);
        if (t < 0 || t > 11) return fail("bad DC code");
        c.dcpred += t ? extend(getbits(t), t) : 0;
        b[0] = (int16_t)c.dcpred;
        for (int k = 1; k < 64;) {
            const int rs = decode_huff(ha_[c.ta]);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;  // EOB
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return fail("AC run past the block");
            b[kZigzag[k]] = (int16_t)extend(getbits(s), s);
            ++k;
        }
        return true;
    }
    bool block_dc_prog(Comp& c, int16_t* b, int ah, int al) {  // G.1.2.1
        if (ah == 0) {
            const int t = decode_huff(hd_[c.td]);
            if (t < 0 || t > 11) return fail("bad DC code");
            c.dcpred += t ? extend(getbits(t), t) : 0;
            b[0] = (int16_t)(c.dcpred * (1 << al));
        } else if (getbit()) {
            b[0] = (int16_t)(b[0] | (1 << al));
        }
        return true;
    }
    bool block_ac_first(Comp& c, int16_t* b, int ss, int se, int al) {  // G.1.2.2
        if (eobrun_ > 0) {
            --eobrun_;
            return true;
        }
        for (int k = ss; k <= se;) {
            const int rs = decode_huff(ha_[c.ta]);
            if (rs < 0) return fail("bad AC code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun_ = (1 << r) - 1;
                    if (r) eobrun_ += getbits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return fail("AC run past the block");
            b[kZigzag[k]] = (int16_t)(extend(getbits(s), s) * (1 << al));
            ++k;
        }
        return true;
    }
    bool block_ac_refine(Comp& c, int16_t* b, int ss, int se, int al) {  // G.1.2.3, figure G.7
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        auto refine = [&](int16_t& coef) {
            if (getbit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
        };
        if (eobrun_ == 0) {
            for (; k <= se; ++k) {
                const int rs = decode_huff(ha_[c.ta]);
                if (rs < 0) return fail("bad AC code");
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s) {
                    if (s != 1) return fail("bad AC refinement code");
                    value = getbit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun_ = 1 << r;
                    if (r) eobrun_ += getbits(r);
                    break;
                }
                // advance over coefficients that are already non-zero (each takes a correction bit) and over r zero ones
                while (k <= se) {
                    int16_t& coef = b[kZigzag[k]];
                    if (coef != 0) {
                        refine(coef);
                    } else if (--r < 0) {
                        break;
                    }
                    ++k;
                }
                if (s && k <= se) b[kZigzag[k]] = (int16_t)value;
            }
        }
        if (eobrun_ > 0) {  // the rest of the band: only correction bits
            for (; k <= se; ++k) {
                int16_t& coef = b[kZigzag[k]];
                if (coef != 0) refine(coef);
            }
            --eobrun_;
        }
        return true;
    }

    bool restart_marker() {  // between restart intervals: byte-align, expect RSTn, reset the predictions
        reset_bits();
        while (pos_ + 1 < n_ && !(d_[pos_] == 0xFF && d_[pos_ + 1] >= 0xD0 && d_[pos_ + 1] <= 0xD7)) {
            if (d_[pos_] == 0xFF && d_[pos_ + 1] != 0 && d_[pos_ + 1] != 0xFF) return fail("missing restart marker");
            ++pos_;
        }
        if (pos_ + 1 >= n_) return fail("missing restart marker");
        pos_ += 2;
        for (auto& c : comps_) c.dcpred = 0;
        eobrun_ = 0;
        return true;
    }

    bool read_scan(int len) {
        (void)len;
        const int ns = u8();
        if (ns < 1 || ns > (int)comps_.size()) return fail("bad scan header");
        Comp* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = u8(), tt = u8();
            sc[i] = nullptr;
            for (auto& c : comps_)
                if (c.id == id) sc[i] = &c;
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = tt >> 4;
            sc[i]->ta = tt & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        const int ss = u8(), se = u8(), ahal = u8();
        const int ah = ahal >> 4, al = ahal & 15;
        if (progressive_) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("bad progressive scan parameters");
        }
        for (int i = 0; i < ns; ++i) {
            const bool need_dc = !progressive_ || (ss == 0 && ah == 0), need_ac = !progressive_ || ss > 0;
            if ((need_dc && !hd_[sc[i]->td].present) || (need_ac && !ha_[sc[i]->ta].present)) return fail("scan uses an undefined Huffman table");
            sc[i]->dcpred = 0;
        }
        eobrun_ = 0;
        reset_bits();
        auto do_block = [&](Comp& c, int bx, int by) -> bool {
            int16_t* b = &c.coef[((size_t)by * c.bw + bx) * 64];
            if (!progressive_) return block_baseline(c, b);
            if (ss == 0) return block_dc_prog(c, b, ah, al);
            return ah == 0 ? block_ac_first(c, b, ss, se, al) : block_ac_refine(c, b, ss, se, al);
        };
        int count = 0;
        if (ns == 1) {  // non-interleaved: the component's own blocks in raster order, only those that hold image samples (A.2.2)
            Comp& c = *sc[0];
            const int nbx = (c.cw + 7) / 8, nby = (c.ch + 7) / 8;
            for (int by = 0; by < nby; ++by)
                for (int bx = 0; bx < nbx; ++bx) {
                    if (restart_ && count == restart_) {
                        if (!restart_marker()) return false;
                        count = 0;
                    }
                    if (!do_block(c, bx, by)) return false;
                    ++count;
                }
        } else {
            for (int my = 0; my < mcuy_; ++my)
                for (int mx = 0; mx < mcux_; ++mx) {
                    if (restart_ && count == restart_) {
                        if (!restart_marker()) return false;
                        count = 0;
                    }
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h)
                                if (!do_block(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v)) return false;
                    ++count;
                }
        }
        reset_bits();  // (the byte position now sits at the marker that ended the segment, or inside trailing fill)
        return true;
    }

    // ---- inverse DCT: the 13-bit fixed-point LL&M algorithm with libjpeg's scaling and rounding ("islow") ----
    static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                      F2053 = 16819, F2562 = 20995, F3072 = 25172;
        auto descale = [](int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); };
        int ws[64];
        for (int c = 0; c < 8; ++c) {
            const int16_t* p = in + c;
            const uint16_t* qq = q + c;
            if (!(p[8] | p[16] | p[24] | p[32] | p[40] | p[48] | p[56])) {
                const int dc = (p[0] * qq[0]) * (1 << P1);
                for (int r = 0; r < 8; ++r) ws[r * 8 + c] = dc;
                continue;
            }
            int64_t z2 = p[16] * qq[16], z3 = p[48] * qq[48];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            z2 = p[0] * qq[0];
            z3 = p[32] * qq[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = p[56] * qq[56];
            tmp1 = p[40] * qq[40];
            tmp2 = p[24] * qq[24];
            tmp3 = p[8] * qq[8];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5;
            z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1);
            ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1);
            ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1);
            ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1);
            ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const int* w = ws + r * 8;
            uint8_t* o = out + (size_t)r * stride;
            if (!(w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7])) {
                const uint8_t dc = clamp8(descale(w[0], P1 + 3) + 128);
                for (int c = 0; c < 8; ++c) o[c] = dc;
                continue;
            }
            int64_t z2 = w[2], z3 = w[6];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7];
            tmp1 = w[5];
            tmp2 = w[3];
            tmp3 = w[1];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5;
            z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int S = CB + P1 + 3;
            o[0] = clamp8(descale(tmp10 + tmp3, S) + 128);
            o[7] = clamp8(descale(tmp10 - tmp3, S) + 128);
            o[1] = clamp8(descale(tmp11 + tmp2, S) + 128);
            o[6] = clamp8(descale(tmp11 - tmp2, S) + 128);
            o[2] = clamp8(descale(tmp12 + tmp1, S) + 128);
            o[5] = clamp8(descale(tmp12 - tmp1, S) + 128);
            o[3] = clamp8(descale(tmp13 + tmp0, S) + 128);
            o[4] = clamp8(descale(tmp13 - tmp0, S) + 128);
        }
    }

    // ---- chroma upsampling to full resolution: libjpeg's triangle filters, edges replicated ----
    static void upsample(const std::vector<uint8_t>& in, int istride, int cw, int ch, int h, int v, int hmax, int vmax, std::vector<uint8_t>& out, int W, int H) {
        out.assign((size_t)W * H, 0);
        const int fx = hmax / h, fy = vmax / v;
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], &in[(size_t)y * istride], (size_t)W);
            return;
        }
        auto row = [&](int y) { return &in[(size_t)(y < 0 ? 0 : y >= ch ? ch - 1 : y) * istride]; };
        std::vector<int> cur((size_t)cw);
        std::vector<uint8_t> line((size_t)2 * cw + 2);
        for (int y = 0; y < H; ++y) {
            if (fy == 2 && fx == 2) {  // h2v2 "fancy": 3/4 nearer row + 1/4 farther row, then the same horizontally, rounding 8 / 7 alternately
                const int sy = y >> 1;
                const uint8_t *r0 = row(sy), *r1 = row((y & 1) ? sy + 1 : sy - 1);
                for (int x = 0; x < cw; ++x) cur[(size_t)x] = 3 * r0[x] + r1[x];
                for (int x = 0; x < cw; ++x) {
                    const int t = cur[(size_t)x], l = x > 0 ? cur[(size_t)x - 1] : t, n = x + 1 < cw ? cur[(size_t)x + 1] : t;
                    line[(size_t)2 * x] = (uint8_t)(x == 0 ? (t * 4 + 8) >> 4 : (t * 3 + l + 8) >> 4);
                    line[(size_t)2 * x + 1] = (uint8_t)(x + 1 == cw ? (t * 4 + 7) >> 4 : (t * 3 + n + 7) >> 4);
                }
            } else if (fx == 2) {  // h2v1 "fancy": 3/4 nearer + 1/4 farther sample, rounding 1 / 2 alternately
                const uint8_t* r0 = row(y);
                for (int x = 0; x < cw; ++x) {
                    const int t = r0[x], l = x > 0 ? r0[x - 1] : t, n = x + 1 < cw ? r0[x + 1] : t;
                    line[(size_t)2 * x] = (uint8_t)(x == 0 ? t : (t * 3 + l + 1) >> 2);
                    line[(size_t)2 * x + 1] = (uint8_t)(x + 1 == cw ? t : (t * 3 + n + 2) >> 2);
                }
            } else {  // h1v2: rows are replicated (libjpeg has no triangle filter for this layout)
                memcpy(line.data(), row(y >> 1), (size_t)cw);
            }
            memcpy(&out[(size_t)y * W], line.data(), (size_t)W);
        }
    }

    bool run(std::vector<uint8_t>& bgr, int& height, int& width) {
        if (n_ < 4 || d_[0] != 0xFF || d_[1] != 0xD8) return fail("not a JPEG file");
        pos_ = 2;
        bool have_sof = false, done = false;
        while (!done) {
            // next marker (skip fill bytes and whatever an entropy-coded segment left behind)
            while (pos_ < n_ && d_[pos_] != 0xFF) ++pos_;
            while (pos_ < n_ && d_[pos_] == 0xFF) ++pos_;
            if (pos_ >= n_) break;
            const int m = d_[pos_++];
            if (m == 0xD9) break;                          // EOI
            if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;  // stuffed byte / stray RSTn / TEM
            const int len = u16();
            if (len < 2 || pos_ + (size_t)len - 2 > n_) return fail("truncated segment");
            const size_t next = pos_ + (size_t)len - 2;
            switch (m) {
                case 0xC0: case 0xC1: case 0xC2:
                    if (have_sof) return fail("more than one frame");
                    progressive_ = m == 0xC2;
                    if (!read_sof(len)) return false;
                    have_sof = true;
                    break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
                case 0xC4:
                    if (!read_dht(len)) return false;
                    break;
                case 0xDB:
                    if (!read_dqt(len)) return false;
                    break;
                case 0xDD:
                    restart_ = u16();
                    break;
                case 0xDA:
                    if (!have_sof) return fail("scan before frame header");
                    if (!read_scan(len)) return false;
                    continue;  // (pos_ already stands behind the entropy-coded data)
                default:
                    break;  // APPn, COM, ...: skipped
            }
            pos_ = next;
        }
        if (!have_sof) return fail("no frame header");
        // ---- reconstruction ----
        std::vector<std::vector<uint8_t>> plane(comps_.size()), full(comps_.size());
        for (size_t ci = 0; ci < comps_.size(); ++ci) {
            Comp& c = comps_[ci];
            if (!qt_ok_[c.tq]) return fail("missing quantisation table");
            const int stride = c.bw * 8;
            plane[ci].assign((size_t)stride * c.bh * 8, 0);
            for (int by = 0; by < c.bh; ++by)
                for (int bx = 0; bx < c.bw; ++bx)
                    idct(&c.coef[((size_t)by * c.bw + bx) * 64], qt_[c.tq], &plane[ci][((size_t)by * 8) * stride + (size_t)bx * 8], stride);
            upsample(plane[ci], stride, c.cw, c.ch, c.h, c.v, hmax_, vmax_, full[ci], W_, H_);
        }
        height = H_;
        width = W_;
        bgr.resize((size_t)W_ * H_ * 3);
        if (comps_.size() == 1) {
            for (size_t i = 0; i < (size_t)W_ * H_; ++i) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = full[0][i];
            return true;
        }
        // YCbCr -> RGB in 16-bit fixed point with libjpeg's rounding: R = Y + 1.402 Cr, G = Y - 0.34414 Cb - 0.71414 Cr, B = Y + 1.772 Cb
        constexpr int SB = 16, HALF = 1 << 15;
        auto FIXc = [](double x) { return (int)(x * 65536.0 + 0.5); };
        int crr[256], cbb[256], crg[256], cbg[256];
        for (int i = 0; i < 256; ++i) {
            const int x = i - 128;
            crr[i] = (FIXc(1.40200) * x + HALF) >> SB;
            cbb[i] = (FIXc(1.77200) * x + HALF) >> SB;
            crg[i] = -FIXc(0.71414) * x;
            cbg[i] = -FIXc(0.34414) * x + HALF;
        }
        for (size_t i = 0; i < (size_t)W_ * H_; ++i) {
            const int y = full[0][i], cb = full[1][i], cr = full[2][i];
            bgr[3 * i + 2] = clamp8(y + crr[cr]);
            bgr[3 * i + 1] = clamp8(y + ((cbg[cb] + crg[cr]) >> SB));
            bgr[3 * i] = clamp8(y + cbb[cb]);
        }
        return true;
    }
};

inline bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// ---- baseline encoder: 4:4:4 YCbCr, the Annex K tables, libjpeg's quality -> scale rule (quality 95 is cv::imwrite's default) ----
class Encoder {
public:
    static bool write(const std::string& path, const uint8_t* bgr, int h, int w, int quality = 95) {
        std::vector<uint8_t> out;
        encode(bgr, h, w, quality, out);
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
        fclose(f);
        return ok;
    }
    static void encode(const uint8_t* bgr, int h, int w, int quality, std::vector<uint8_t>& out) {
        static const uint8_t ql[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                       18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
        static const uint8_t qc[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
        static const uint8_t dcl_n[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, dcc_n[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
        static const uint8_t dc_v[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
        static const uint8_t acl_n[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, acc_n[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
        static const uint8_t acl_v[162] = {
            0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
            0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
            0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
            0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
            0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
            0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
        static const uint8_t acc_v[162] = {
            0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
            0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
            0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
            0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
            0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
            0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
        quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
        const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
        uint8_t q[2][64];
        for (int i = 0; i < 64; ++i) {
            int a = (ql[i] * scale + 50) / 100, b = (qc[i] * scale + 50) / 100;
            q[0][i] = (uint8_t)(a < 1 ? 1 : a > 255 ? 255 : a);
            q[1][i] = (uint8_t)(b < 1 ? 1 : b > 255 ? 255 : b);
        }
        out.clear();
        auto put = [&](int b) { out.push_back((uint8_t)b); };
        auto put16 = [&](int v) { put(v >> 8); put(v & 255); };
        put(0xFF); put(0xD8);
        put(0xFF); put(0xE0); put16(16); put('J'); put('F'); put('I'); put('F'); put(0); put(1); put(1); put(0); put16(1); put16(1); put(0); put(0);
        for (int t = 0; t < 2; ++t) {
            put(0xFF); put(0xDB); put16(67); put(t);
            for (int i = 0; i < 64; ++i) put(q[t][kZigzag[i]]);
        }
        put(0xFF); put(0xC0); put16(17); put(8); put16(h); put16(w); put(3);
        for (int c = 0; c < 3; ++c) { put(c + 1); put(0x11); put(c ? 1 : 0); }
        struct Tab { uint16_t code[256]; uint8_t len[256]; } tabs[4];
        auto dht = [&](int tc_th, const uint8_t* n, const uint8_t* v, int nv, Tab& t) {
            put(0xFF); put(0xC4); put16(19 + nv); put(tc_th);
            for (int i = 1; i <= 16; ++i) put(n[i]);
            for (int i = 0; i < nv; ++i) put(v[i]);
            memset(&t, 0, sizeof t);
            int code = 0, k = 0;
            for (int l = 1; l <= 16; ++l) {
                for (int i = 0; i < n[l]; ++i, ++k, ++code) { t.code[v[k]] = (uint16_t)code; t.len[v[k]] = (uint8_t)l; }
                code <<= 1;
            }
        };
        dht(0x00, dcl_n, dc_v, 12, tabs[0]);
        dht(0x10, acl_n, acl_v, 162, tabs[1]);
        dht(0x01, dcc_n, dc_v, 12, tabs[2]);
        dht(0x11, acc_n, acc_v, 162, tabs[3]);
        put(0xFF); put(0xDA); put16(12); put(3); put(1); put(0x00); put(2); put(0x11); put(3); put(0x11); put(0); put(63); put(0);
        uint32_t acc = 0;
        int nb = 0;
        auto bits = [&](unsigned v, int n) {
            acc = (acc << n) | (v & ((1u << n) - 1));
            nb += n;
            while (nb >= 8) {
                const int b = (acc >> (nb - 8)) & 255;
                put(b);
                if (b == 255) put(0);
                nb -= 8;
            }
        };
        static double cosv[8][8];
        static bool cos_ready = false;
        if (!cos_ready) {
            for (int u = 0; u < 8; ++u)
                for (int x = 0; x < 8; ++x) cosv[u][x] = std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * (u ? 0.5 : 0.35355339059327376);
            cos_ready = true;
        }
        int pred[3] = {0, 0, 0};
        for (int by = 0; by < (h + 7) / 8; ++by)
            for (int bx = 0; bx < (w + 7) / 8; ++bx)
                for (int c = 0; c < 3; ++c) {
                    double blk[64], tmp[64];
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) {
                            const int yy = by * 8 + y < h ? by * 8 + y : h - 1, xx = bx * 8 + x < w ? bx * 8 + x : w - 1;
                            const uint8_t* px = bgr + ((size_t)yy * w + xx) * 3;
                            const double B = px[0], G = px[1], R = px[2];
                            blk[y * 8 + x] = c == 0 ? 0.299 * R + 0.587 * G + 0.114 * B - 128.0
                                                    : c == 1 ? -0.168735892 * R - 0.331264108 * G + 0.5 * B : 0.5 * R - 0.418687589 * G - 0.081312411 * B;
                        }
                    for (int y = 0; y < 8; ++y)
                        for (int u = 0; u < 8; ++u) {
                            double s = 0;
                            for (int x = 0; x < 8; ++x) s += blk[y * 8 + x] * cosv[u][x];
                            tmp[y * 8 + u] = s;
                        }
                    int zz[64];
                    for (int v = 0; v < 8; ++v)
                        for (int u = 0; u < 8; ++u) {
                            double s = 0;
                            for (int y = 0; y < 8; ++y) s += tmp[y * 8 + u] * cosv[v][y];
                            const int qq = q[c ? 1 : 0][v * 8 + u];
                            blk[v * 8 + u] = std::nearbyint(s / qq);
                        }
                    for (int i = 0; i < 64; ++i) zz[i] = (int)blk[kZigzag[i]];
                    const Tab &td = tabs[c ? 2 : 0], &ta = tabs[c ? 3 : 1];
                    auto magnitude = [](int v, int& s, unsigned& b) {
                        int a = v < 0 ? -v : v;
                        s = 0;
                        while (a) { ++s; a >>= 1; }
                        b = (unsigned)(v < 0 ? v - 1 : v);
                    };
                    int s;
                    unsigned b;
                    magnitude(zz[0] - pred[c], s, b);
                    pred[c] = zz[0];
                    bits(td.code[s], td.len[s]);
                    if (s) bits(b, s);
                    int run = 0;
                    for (int k = 1; k < 64; ++k) {
                        if (zz[k] == 0) { ++run; continue; }
                        while (run > 15) { bits(ta.code[0xF0], ta.len[0xF0]); run -= 16; }
                        magnitude(zz[k], s, b);
                        bits(ta.code[(run << 4) | s], ta.len[(run << 4) | s]);
                        bits(b, s);
                        run = 0;
                    }
                    if (run) bits(ta.code[0], ta.len[0]);
                }
        if (nb) bits(0x7F, 8 - nb);
        put(0xFF); put(0xD9);
    }
};

// cv::imread(IMREAD_COLOR) for the example programs: JPEG (by its SOI marker) or binary PPM (P6, maxval 255) -> BGR interleaved
inline bool imread_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& h, int& w, std::string* err = nullptr) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) {
        if (err) *err = "cannot read file";
        return false;
    }
    if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) return Decoder().decode(file.data(), file.size(), bgr, h, w, err);
    size_t p = 0;
    auto token = [&](std::string& t) {
        t.clear();
        while (p < file.size()) {
            if (file[p] == '#') { while (p < file.size() && file[p] != '\n') ++p; continue; }
            if (!isspace(file[p])) break;
            ++p;
        }
        while (p < file.size() && !isspace(file[p])) t.push_back((char)file[p++]);
        return !t.empty();
    };
    std::string t;
    bool ok = token(t) && t == "P6" && token(t);
    if (ok) { w = atoi(t.c_str()); ok = token(t); }
    if (ok) { h = atoi(t.c_str()); ok = token(t) && atoi(t.c_str()) == 255 && w > 0 && h > 0; }
    ++p;  // the single whitespace byte behind maxval
    ok = ok && p + (size_t)h * w * 3 <= file.size();
    if (!ok) {
        if (err) *err = "neither a JPEG nor a binary PPM (P6, maxval 255)";
        return false;
    }
    bgr.resize((size_t)h * w * 3);
    for (size_t i = 0; i < (size_t)h * w; ++i) {
        bgr[3 * i] = file[p + 3 * i + 2];
        bgr[3 * i + 1] = file[p + 3 * i + 1];
        bgr[3 * i + 2] = file[p + 3 * i];
    }
    return true;
}

// cv::imwrite for the example programs: by extension, .jpg / .jpeg -> baseline JPEG (quality 95), anything else -> binary PPM
inline bool imwrite_bgr(const std::string& path, const uint8_t* bgr, int h, int w) {
    const size_t dot = path.rfind('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (auto& c : ext) c = (char)tolower(c);
    if (ext == "jpg" || ext == "jpeg") return Encoder::write(path, bgr, h, w, 95);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<uint8_t> rgb((size_t)h * w * 3);
    for (size_t i = 0; i < (size_t)h * w; ++i) { rgb[3 * i] = bgr[3 * i + 2]; rgb[3 * i + 1] = bgr[3 * i + 1]; rgb[3 * i + 2] = bgr[3 * i]; }
    const bool ok = fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
    fclose(f);
    return ok;
}

}  // namespace dinojpeg
