// image_codec.h — native decoders for the image files Vision scenes reference: PNG (8-bit, non-interlaced) and baseline JPEG — and the
// encoders of the save path (PNG, Radiance HDR; OpenEXR lives in exr.h).
// Replaces ocarina Image::load for these two containers (base/mgr/image_pool.cpp:23-28 -> stb / FreeImage in ocarina, absent
// from the checkout), so a C / C++ host can load a textured scene without handing decoded pixels in (vmk_host_register_image
// still takes precedence when a host already holds them).  Own implementation of the published formats:
//   PNG  — RFC 2083 + DEFLATE RFC 1951: stored / fixed / dynamic Huffman blocks, the five scanline filters, colour types
//          0 / 2 / 3 / 4 / 6 at 8 bits per sample.  Lossless, so the pixels are the file's pixels.
//   JPEG — ITU T.81 baseline sequential DCT (SOF0 / SOF1 8-bit), Huffman coding, restart intervals, any 1x/2x sampling.
//          The arithmetic follows the reference decoder everyone's pixels come from (IJG libjpeg "islow" integer IDCT, triangle
//          "fancy" chroma upsampling, 16-bit fixed-point YCbCr->RGB) so that a texture decodes to the same bytes here and in Pillow;
//          tests/test_host.py compares both on every JPEG the scenes ship.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace vmk_img {

struct Decoded { uint32_t w{0}, h{0}, channels{0}; std::vector<uint8_t> px; std::string error; };

// ---------------------------------------------------------------------------------------------------------
// DEFLATE (RFC 1951) inside a zlib stream (RFC 1950)
// ---------------------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p, *end; uint32_t bits{0}; int n{0}; bool overrun{false}; // overrun: a read had to invent bytes past the end of the input
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    int get(int k) { while (n < k) { uint32_t byte = 0u; if (p < end) byte = *p++; else overrun = true; bits |= byte << n; n += 8; } int v = (int) (bits & ((1u << k) - 1u)); bits >>= k; n -= k; return v; }
    bool eof() const { return p >= end && n <= 0; }
};
struct Huff { // canonical Huffman table: counts per length + symbols in code order
    uint16_t count[16]{}, symbol[320]{};
    void build(const uint8_t *len, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t) (offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t) i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= br.get(1);
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};
// `max_out`: the caller's bound on the decoded size (a PNG's h * (stride + 1), an EXR block's byte count); a stream that decodes to more is
// refused instead of growing `out` without limit, and a stream that runs past its last byte (truncated / corrupt input) is an error as soon
// as the reader has to invent bits, not only when the bit counter happens to be empty.
inline bool inflate_zlib(const std::vector<uint8_t> &in, std::vector<uint8_t> &out, std::string &err, size_t max_out = (size_t) -1) {
    if (in.size() < 6 || (in[0] & 0x0f) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) { err = "bad zlib header"; return false; }
    BitReader br(in.data() + 2, in.data() + in.size());
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int last = br.get(1), type = br.get(2);
        if (type == 0) { // stored
            br.bits = 0; br.n = 0;
            if (br.end - br.p < 4) { err = "truncated stored block"; return false; }
            uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8); br.p += 4;
            if ((len ^ 0xffffu) != nlen || (size_t) (br.end - br.p) < len) { err = "bad stored block"; return false; }
            if (out.size() + len > max_out) { err = "deflate stream decodes to more than the container allows"; return false; }
            out.insert(out.end(), br.p, br.p + len); br.p += len;
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lens[i] = 8;
                for (; i < 256; ++i) lens[i] = 9;
                for (; i < 280; ++i) lens[i] = 7;
                for (; i < 288; ++i) lens[i] = 8;
                hl.build(lens, 288);
                for (i = 0; i < 30; ++i) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                int nlen = br.get(5) + 257, ndist = br.get(5) + 1, ncode = br.get(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t) br.get(3);
                Huff hc; hc.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = hc.decode(br);
                    if (sym < 0) { err = "bad code-length code"; return false; }
                    if (sym < 16) lens[idx++] = (uint8_t) sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!idx) { err = "repeat without previous length"; return false; } val = lens[idx - 1]; rep = 3 + br.get(2); }
                        else if (sym == 17) rep = 3 + br.get(3);
                        else rep = 11 + br.get(7);
                        if (idx + rep > nlen + ndist) { err = "code lengths overrun"; return false; }
                        while (rep--) lens[idx++] = (uint8_t) val;
                    }
                }
                hl.build(lens, nlen); hd.build(lens + nlen, ndist);
            }
            for (;;) {
                int sym = hl.decode(br);
                if (sym < 0) { err = "bad literal/length code"; return false; }
                if (br.overrun) { err = "truncated deflate stream"; return false; }
                if (sym < 256) { if (out.size() >= max_out) { err = "deflate stream decodes to more than the container allows"; return false; } out.push_back((uint8_t) sym); }
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) { err = "bad length symbol"; return false; }
                    int len = lbase[sym] + br.get(lext[sym]);
                    int ds = hd.decode(br);
                    if (ds < 0 || ds >= 30) { err = "bad distance symbol"; return false; }
                    size_t dist = dbase[ds] + (size_t) br.get(dext[ds]);
                    if (dist > out.size()) { err = "distance beyond output"; return false; }
                    if (out.size() + (size_t) len > max_out) { err = "deflate stream decodes to more than the container allows"; return false; }
                    size_t from = out.size() - dist;
                    for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
                }
                if (br.p >= br.end && br.n <= 0 && !last) { err = "truncated deflate stream"; return false; }
            }
        } else { err = "bad block type"; return false; }
        if (br.overrun) { err = "truncated deflate stream"; return false; }
        if (last) break;
        if (br.p >= br.end && br.n <= 0) { err = "truncated deflate stream"; return false; }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }
inline Decoded decode_png(const std::vector<uint8_t> &f) {
    Decoded d;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 13, 10, 26, 10};
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8) != 0) { d.error = "not a PNG file"; return d; }
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= f.size()) {
        uint32_t len = be32(&f[pos]);
        if (pos + 12 + (size_t) len > f.size()) { d.error = "truncated PNG chunk"; return d; }
        const uint8_t *type = &f[pos + 4], *data = &f[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) { d.error = "bad IHDR"; return d; }
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t) len;
    }
    if (!w || !h || w > 32768 || h > 32768) { d.error = "bad PNG dimensions"; return d; }
    if (depth != 8 || interlace != 0) { d.error = "PNG: only 8-bit non-interlaced files are decoded natively"; return d; }
    int spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!spp || (ctype == 3 && plte.size() < 3)) { d.error = "bad PNG colour type"; return d; }
    std::vector<uint8_t> raw;
    const size_t expect = (size_t) h * ((size_t) w * spp + 1);
    raw.reserve(std::min(expect, idat.size() * 1032 + 64)); // (deflate expands by at most 1032x: nothing is reserved on the IHDR's word alone)
    if (!inflate_zlib(idat, raw, d.error, expect)) return d;
    const size_t stride = (size_t) w * spp;
    if (raw.size() < (stride + 1) * h) { d.error = "PNG data too short"; return d; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? cur - stride : nullptr;
        const int ft = src[0];
        ++src;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t) spp ? cur[x - spp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t) spp) ? up[x - spp] : 0, v = src[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: { int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p; v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: d.error = "bad PNG filter"; return d;
            }
            cur[x] = (uint8_t) v;
        }
    }
    d.w = w; d.h = h;
    if (ctype == 0) { d.channels = 1; d.px.swap(img); }
    else if (ctype == 2) { d.channels = 3; d.px.swap(img); }
    else if (ctype == 6) { d.channels = 4; d.px.swap(img); }
    else if (ctype == 4) { d.channels = 4; d.px.resize((size_t) w * h * 4); for (size_t i = 0; i < (size_t) w * h; ++i) { d.px[i * 4] = d.px[i * 4 + 1] = d.px[i * 4 + 2] = img[i * 2]; d.px[i * 4 + 3] = img[i * 2 + 1]; } }
    else { // palette -> RGB (RGBA with a tRNS chunk)
        const bool alpha = !trns.empty();
        d.channels = alpha ? 4 : 3; d.px.resize((size_t) w * h * d.channels);
        for (size_t i = 0; i < (size_t) w * h; ++i) {
            size_t k = img[i];
            for (int c = 0; c < 3; ++c) d.px[i * d.channels + c] = k * 3 + c < plte.size() ? plte[k * 3 + c] : 0;
            if (alpha) d.px[i * 4 + 3] = k < trns.size() ? trns[k] : 255;
        }
    }
    return d;
}

// ---------------------------------------------------------------------------------------------------------
// baseline JPEG
// ---------------------------------------------------------------------------------------------------------
struct JHuff { uint8_t bits[17]{}, vals[256]{}; int mincode[17]{}, maxcode[18]{}, valptr[17]{}; bool set{false};
    void build() { int code = 0, k = 0; for (int l = 1; l <= 16; ++l) { valptr[l] = k; mincode[l] = code; code += bits[l]; k += bits[l]; maxcode[l] = bits[l] ? code - 1 : -1; code <<= 1; } maxcode[17] = 0x7fffffff; set = true; } };
struct JBits {
    const uint8_t *p, *end; uint32_t buf{0}; int n{0}; bool marker{false};
    JBits(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() { while (n <= 24) { int byte = 0; if (!marker && p < end) { byte = *p++; if (byte == 0xff) { int b2 = p < end ? *p : 0; if (b2 == 0) ++p; else { marker = true; --p; byte = 0; } } } buf |= (uint32_t) byte << (24 - n); n += 8; } }
    int bit() { if (n < 1) fill(); int v = (int) (buf >> 31); buf <<= 1; --n; return v; }
    int get(int k) { if (!k) return 0; if (n < k) fill(); int v = (int) (buf >> (32 - k)); buf <<= k; n -= k; return v; }
    void reset() { buf = 0; n = 0; marker = false; }
};
inline int jdecode(JBits &br, const JHuff &h) { int code = 0; for (int l = 1; l <= 16; ++l) { code = (code << 1) | br.bit(); if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]]; } return -1; }
inline int jextend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
// IJG jidctint.c ("islow"): 13-bit fixed-point constants, two passes, DESCALE with rounding; output level-shifted and clamped
inline void idct_islow(const int *in /* dequantised, natural order */, uint8_t *out, int out_stride) {
    const int CB = 13, P1 = 2;
    const long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    long ws[64];
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int c = 0; c < 8; ++c) {
        const int *ip = in + c; long *wp = ws + c;
        long z2 = ip[16], z3 = ip[48];
        long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = ip[0]; z3 = ip[32];
        long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = ip[56]; tmp1 = ip[40]; tmp2 = ip[24]; tmp3 = ip[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long *wp = ws + r * 8; uint8_t *op = out + r * out_stride;
        long z2 = wp[2], z3 = wp[6];
        long z1 = (z2 + z3) * F0541, tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        long tmp0 = (wp[0] + wp[4]) << CB, tmp1 = (wp[0] - wp[4]) << CB;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3, z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        auto px = [&](long v) { long s = descale(v, CB + P1 + 3) + 128; return (uint8_t) (s < 0 ? 0 : (s > 255 ? 255 : s)); };
        op[0] = px(tmp10 + tmp3); op[7] = px(tmp10 - tmp3); op[1] = px(tmp11 + tmp2); op[6] = px(tmp11 - tmp2);
        op[2] = px(tmp12 + tmp1); op[5] = px(tmp12 - tmp1); op[3] = px(tmp13 + tmp0); op[4] = px(tmp13 - tmp0);
    }
}
inline Decoded decode_jpeg(const std::vector<uint8_t> &f) {
    Decoded d;
    if (f.size() < 4 || f[0] != 0xff || f[1] != 0xd8) { d.error = "not a JPEG file"; return d; }
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int qt[4][64] = {}; bool qset[4] = {};
    JHuff hdc[4], hac[4];
    struct Comp { int id, h, v, tq, td, ta, pred; uint32_t bw, bh; std::vector<uint8_t> plane; uint32_t dw, dh; } comp[4];
    int ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    uint32_t W = 0, H = 0;
    size_t pos = 2;
    bool got_sof = false;
    while (pos + 4 <= f.size()) {
        if (f[pos] != 0xff) { ++pos; continue; }
        int m = f[pos + 1];
        if (m == 0xff) { ++pos; continue; }
        pos += 2;
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
        if (m == 0xd9) break;
        size_t len = ((size_t) f[pos] << 8) | f[pos + 1];
        if (len < 2 || pos + len > f.size()) { d.error = "truncated JPEG segment"; return d; }
        const uint8_t *s = &f[pos + 2]; size_t n = len - 2;
        if (m == 0xdb) { // DQT
            while (n) { int pq = s[0] >> 4, tq = s[0] & 15; size_t need = 1 + (pq ? 128 : 64); if (tq > 3 || n < need) { d.error = "bad DQT"; return d; }
                for (int i = 0; i < 64; ++i) qt[tq][zz[i]] = pq ? ((s[1 + 2 * i] << 8) | s[2 + 2 * i]) : s[1 + i];
                qset[tq] = true; s += need; n -= need; }
        } else if (m == 0xc4) { // DHT
            while (n) { if (n < 17) { d.error = "bad DHT"; return d; } int tc = s[0] >> 4, th = s[0] & 15; if (tc > 1 || th > 3) { d.error = "bad DHT"; return d; }
                JHuff &h = tc ? hac[th] : hdc[th]; int total = 0; h.bits[0] = 0; for (int i = 1; i <= 16; ++i) { h.bits[i] = s[i]; total += s[i]; }
                if (total > 256 || n < (size_t) 17 + total) { d.error = "bad DHT"; return d; }
                std::memcpy(h.vals, s + 17, total); h.build(); s += 17 + total; n -= 17 + total; }
        } else if (m == 0xc0 || m == 0xc1) { // SOF0 / SOF1
            if (n < 6 || s[0] != 8) { d.error = "JPEG: only 8-bit baseline files are decoded natively"; return d; }
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4]; ncomp = s[5];
            if (!W || !H || (ncomp != 1 && ncomp != 3) || n < (size_t) 6 + 3 * ncomp) { d.error = "bad SOF"; return d; }
            for (int i = 0; i < ncomp; ++i) { comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) { d.error = "JPEG: unsupported sampling factors"; return d; }
                hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax; }
            got_sof = true;
        } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) { d.error = "JPEG: progressive / lossless / arithmetic files are not decoded natively"; return d; }
        else if (m == 0xdd) { if (n >= 2) restart = (s[0] << 8) | s[1]; }
        else if (m == 0xda) { // SOS: decode the single interleaved (or grayscale) scan
            if (!got_sof || n < 1 || s[0] != ncomp || n < (size_t) 1 + 2 * ncomp + 3) { d.error = "JPEG: multi-scan files are not decoded natively"; return d; }
            for (int i = 0; i < ncomp; ++i) { int cid = s[1 + 2 * i], k = -1; for (int j = 0; j < ncomp; ++j) if (comp[j].id == cid) k = j; if (k < 0) { d.error = "bad SOS"; return d; }
                comp[k].td = s[2 + 2 * i] >> 4; comp[k].ta = s[2 + 2 * i] & 15; if (comp[k].td > 3 || comp[k].ta > 3 || !hdc[comp[k].td].set || !hac[comp[k].ta].set || !qset[comp[k].tq]) { d.error = "JPEG: missing table"; return d; } }
            const uint32_t mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int i = 0; i < ncomp; ++i) { Comp &c = comp[i]; c.bw = mcux * c.h * 8; c.bh = mcuy * c.v * 8; c.plane.assign((size_t) c.bw * c.bh, 0); c.pred = 0;
                c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax; }
            JBits br(&f[pos + len], f.data() + f.size());
            int coef[64], todo = restart;
            for (uint32_t my = 0; my < mcuy; ++my) for (uint32_t mx = 0; mx < mcux; ++mx) {
                if (restart && todo == 0) { // RSTn: byte-align, skip the marker, reset predictors
                    br.reset();
                    while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) ++br.p;
                    if (br.p + 1 < br.end) br.p += 2;
                    for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
                    todo = restart;
                }
                for (int i = 0; i < ncomp; ++i) { Comp &c = comp[i];
                    for (int by = 0; by < c.v; ++by) for (int bx = 0; bx < c.h; ++bx) {
                        std::memset(coef, 0, sizeof coef);
                        int t = jdecode(br, hdc[c.td]);
                        if (t < 0 || t > 11) { d.error = "JPEG: bad DC code"; return d; }
                        int diff = t ? jextend(br.get(t), t) : 0;
                        c.pred += diff; coef[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            int rs = jdecode(br, hac[c.ta]);
                            if (rs < 0) { d.error = "JPEG: bad AC code"; return d; }
                            int r = rs >> 4, sz = rs & 15;
                            if (!sz) { if (r == 15) { k += 16; continue; } break; }
                            k += r; if (k > 63) { d.error = "JPEG: coefficient overrun"; return d; }
                            coef[zz[k]] = jextend(br.get(sz), sz) * qt[c.tq][zz[k]]; ++k;
                        }
                        idct_islow(coef, &c.plane[((size_t) (my * c.v + by) * 8) * c.bw + (size_t) (mx * c.h + bx) * 8], (int) c.bw);
                    } }
                if (restart) --todo;
            }
            break;
        }
        pos += len;
    }
    if (!got_sof || comp[0].plane.empty()) { d.error = "JPEG: no image data"; return d; }
    d.w = W; d.h = H; d.channels = (uint32_t) ncomp; d.px.resize((size_t) W * H * ncomp);
    if (ncomp == 1) { for (uint32_t y = 0; y < H; ++y) std::memcpy(&d.px[(size_t) y * W], &comp[0].plane[(size_t) y * comp[0].bw], W); return d; }
    // chroma upsampling as IJG jdsample.c does with do_fancy_upsampling (its default): triangle filters for 2:1 horizontal (h2v1) and
    // 2:1 both ways (h2v2), box replication otherwise; edges use the REAL sample rows / columns (downsampled_width / height)
    std::vector<uint8_t> up[3];
    for (int i = 0; i < 3; ++i) {
        Comp &c = comp[i];
        const int hx = hmax / c.h, vx = vmax / c.v;
        std::vector<uint8_t> &o = up[i]; o.resize((size_t) W * H);
        auto at = [&](long x, long y) -> int { x = x < 0 ? 0 : (x >= (long) c.dw ? (long) c.dw - 1 : x); y = y < 0 ? 0 : (y >= (long) c.dh ? (long) c.dh - 1 : y); return c.plane[(size_t) y * c.bw + x]; };
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) {
            int v;
            if (hx == 1 && vx == 1) v = at(x, y);
            else if (hx == 2 && vx == 1) { long sx = x >> 1; int cur = at(sx, y); v = c.dw == 1 ? cur : ((x & 1) ? ((sx + 1 < (long) c.dw) ? (cur * 3 + at(sx + 1, y) + 2) >> 2 : cur) : (sx > 0 ? (cur * 3 + at(sx - 1, y) + 1) >> 2 : cur)); }
            else if (hx == 2 && vx == 2) { // h2v2_fancy_upsample: vertical 3:1 blend with the nearer neighbour row, then horizontal 3:1
                long sx = x >> 1, sy = y >> 1, ny = (y & 1) ? sy + 1 : sy - 1;
                auto colsum = [&](long cx) { return 3 * at(cx, sy) + at(cx, ny); };
                int cs = colsum(sx);
                if (c.dw == 1) v = (cs * 4 + 8) >> 4;
                else if (x & 1) v = (sx + 1 < (long) c.dw) ? (cs * 3 + colsum(sx + 1) + 7) >> 4 : (cs * 4 + 7) >> 4;
                else v = sx > 0 ? (cs * 3 + colsum(sx - 1) + 8) >> 4 : (cs * 4 + 8) >> 4;
            } else if (hx == 1 && vx == 2) { long sy = y >> 1, ny = (y & 1) ? sy + 1 : sy - 1; v = (3 * at(x, sy) + at(x, ny) + ((y & 1) ? 2 : 1)) >> 2; }
            else v = at(x / hx, y / vx);
            o[(size_t) y * W + x] = (uint8_t) v;
        }
    }
    // jdcolor.c ycc_rgb_convert: 16-bit fixed point tables
    auto FIX = [](double x) { return (long) (x * 65536.0 + 0.5); };
    const long HALF = 1L << 15;
    for (size_t i = 0; i < (size_t) W * H; ++i) {
        int y = up[0][i], cb = up[1][i] - 128, cr = up[2][i] - 128;
        long r = y + ((FIX(1.40200) * cr + HALF) >> 16);
        long g = y + ((-FIX(0.34414) * cb + HALF - FIX(0.71414) * cr) >> 16);
        long b = y + ((FIX(1.77200) * cb + HALF) >> 16);
        d.px[i * 3] = (uint8_t) (r < 0 ? 0 : r > 255 ? 255 : r); d.px[i * 3 + 1] = (uint8_t) (g < 0 ? 0 : g > 255 ? 255 : g); d.px[i * 3 + 2] = (uint8_t) (b < 0 ? 0 : b > 255 ? 255 : b);
    }
    return d;
}


// ---------------------------------------------------------------------------------------------------------
// encoders (Image::save_image of the reference goes through ocarina / stb, absent): DEFLATE, PNG, Radiance HDR
// ---------------------------------------------------------------------------------------------------------
inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}
inline uint32_t adler32(const uint8_t *p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}
// zlib stream with ONE fixed-Huffman block: greedy LZ77 over a 32 KiB window with a hash of 3-byte prefixes (chains of 32)
inline std::vector<uint8_t> deflate_zlib(const uint8_t *in, size_t n) {
    std::vector<uint8_t> out = {0x78, 0x01};
    uint32_t acc = 0; int nb = 0;
    auto put = [&](uint32_t v, int k) { acc |= v << nb; nb += k; while (nb >= 8) { out.push_back((uint8_t) acc); acc >>= 8; nb -= 8; } };
    auto put_rev = [&](uint32_t code, int k) { uint32_t r = 0; for (int i = 0; i < k; ++i) r |= ((code >> i) & 1u) << (k - 1 - i); put(r, k); }; // Huffman codes go MSB first
    auto lit = [&](int s) { if (s < 144) put_rev(0x30 + s, 8); else if (s < 256) put_rev(0x190 + s - 144, 9); else if (s < 280) put_rev(s - 256, 7); else put_rev(0xc0 + s - 280, 8); };
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    put(1, 1); put(1, 2); // last block, fixed Huffman
    constexpr int HB = 15;
    std::vector<int32_t> head((size_t) 1 << HB, -1), prev(n, -1);
    auto hash = [&](size_t i) { return (uint32_t) ((in[i] * 2654435761u + in[i + 1] * 40503u + in[i + 2] * 2246822519u) >> (32 - HB)); };
    size_t i = 0;
    while (i < n) {
        int best_len = 0; size_t best_dist = 0;
        if (i + 3 <= n) {
            uint32_t h = hash(i);
            int32_t c = head[h]; int chain = 32;
            while (c >= 0 && chain-- && i - (size_t) c <= 32768) {
                int l = 0; const size_t lim = std::min<size_t>(258, n - i);
                while ((size_t) l < lim && in[c + l] == in[i + l]) ++l;
                if (l > best_len) { best_len = l; best_dist = i - (size_t) c; if (l == 258) break; }
                c = prev[(size_t) c];
            }
        }
        if (best_len >= 3) {
            int ls = 28; while (lbase[ls] > best_len) --ls;
            lit(257 + ls); put((uint32_t) (best_len - lbase[ls]), lext[ls]);
            int ds = 29; while (dbase[ds] > best_dist) --ds;
            put_rev((uint32_t) ds, 5); put((uint32_t) (best_dist - dbase[ds]), dext[ds]);
            for (int k = 0; k < best_len; ++k, ++i) if (i + 3 <= n) { uint32_t h = hash(i); prev[i] = head[h]; head[h] = (int32_t) i; }
        } else {
            lit(in[i]);
            if (i + 3 <= n) { uint32_t h = hash(i); prev[i] = head[h]; head[h] = (int32_t) i; }
            ++i;
        }
    }
    lit(256);
    if (nb) put(0, 8 - nb);
    const uint32_t ad = adler32(in, n);
    for (int k = 3; k >= 0; --k) out.push_back((uint8_t) (ad >> (8 * k)));
    return out;
}
// 8-bit PNG, colour type 2 (RGB) or 6 (RGBA); every row takes the filter (none / sub / up / paeth) with the smallest sum of |residual|
inline std::vector<uint8_t> encode_png(uint32_t w, uint32_t h, int channels, const uint8_t *px) {
    const size_t stride = (size_t) w * channels;
    std::vector<uint8_t> raw; raw.reserve((stride + 1) * h);
    std::vector<uint8_t> cand[4]; for (auto &c : cand) c.resize(stride);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *cur = px + stride * y, *up = y ? cur - stride : nullptr;
        long best = -1; int bi = 0;
        for (int ft = 0; ft < 4; ++ft) {
            long sum = 0;
            for (size_t x = 0; x < stride; ++x) {
                int a = x >= (size_t) channels ? cur[x - channels] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t) channels) ? up[x - channels] : 0, pr = 0;
                if (ft == 1) pr = a; else if (ft == 2) pr = b;
                else if (ft == 3) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
                uint8_t r = (uint8_t) (cur[x] - pr); cand[ft][x] = r; sum += r < 128 ? r : 256 - r;
            }
            if (best < 0 || sum < best) { best = sum; bi = ft; }
        }
        raw.push_back((uint8_t) (bi == 3 ? 4 : bi));
        raw.insert(raw.end(), cand[bi].begin(), cand[bi].end());
    }
    std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 13, 10, 26, 10};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        uint32_t len = (uint32_t) data.size();
        for (int k = 3; k >= 0; --k) f.push_back((uint8_t) (len >> (8 * k)));
        size_t at = f.size();
        f.insert(f.end(), type, type + 4); f.insert(f.end(), data.begin(), data.end());
        uint32_t crc = crc32(f.data() + at, 4 + data.size());
        for (int k = 3; k >= 0; --k) f.push_back((uint8_t) (crc >> (8 * k)));
    };
    std::vector<uint8_t> ihdr(13);
    for (int k = 0; k < 4; ++k) { ihdr[k] = (uint8_t) (w >> (8 * (3 - k))); ihdr[4 + k] = (uint8_t) (h >> (8 * (3 - k))); }
    ihdr[8] = 8; ihdr[9] = channels == 4 ? 6 : 2; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr);
    chunk("IDAT", deflate_zlib(raw.data(), raw.size()));
    chunk("IEND", {});
    return f;
}
// Radiance RGBE, flat (no run-length) scanlines
inline std::vector<uint8_t> encode_hdr(uint32_t w, uint32_t h, const float *rgba) {
    std::string head = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(h) + " +X " + std::to_string(w) + "\n";
    std::vector<uint8_t> f(head.begin(), head.end());
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const float r = rgba[i * 4], g = rgba[i * 4 + 1], b = rgba[i * 4 + 2];
        float m = std::max(r, std::max(g, b));
        if (!(m > 1e-32f)) { f.insert(f.end(), {0, 0, 0, 0}); continue; }
        int e; float s = std::frexp(m, &e) * 256.f / m;
        f.push_back((uint8_t) std::max(0.f, r * s)); f.push_back((uint8_t) std::max(0.f, g * s)); f.push_back((uint8_t) std::max(0.f, b * s)); f.push_back((uint8_t) (e + 128));
    }
    return f;
}

}// namespace vmk_img
