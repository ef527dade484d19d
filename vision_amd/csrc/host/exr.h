// exr.h — OpenEXR reader / writer of the C++ host (single-part scanline images: what Vision's scenes name as environment maps
// and outputs, e.g. classroom "textures/spaichingen_hill_2k.exr", cbox-prism "dispersion-hero-2000.exr").
// Replaces ocarina's Image::load / Image::save_image for ".exr" (src/base/mgr/image_pool.cpp:13-35, src/base/mgr/pipeline.cpp:190-198;
// ocarina wraps tinyexr, absent from the checkout).  The file format is public (openexr.com/en/latest/OpenEXRFileLayout.html); the
// decoders below follow the published algorithms of the format's compression methods:
//   NONE, RLE, ZIPS, ZIP (zlib + byte predictor + even/odd interleave), PIZ (bitmap LUT + 2-D Haar-like wavelet + canonical Huffman
//   with run-length symbol); HALF / FLOAT / UINT channels; increasing or decreasing line order; any data-window origin.
// Not handled (reported, never guessed): tiles, multi-part, deep data, subsampled channels, PXR24 / B44 / DWA compression.
// Writer: scanline, NONE or ZIP, HALF or FLOAT, channels A? B G R.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "image_codec.h"

namespace vmk_exr {

struct ImageF { uint32_t w{0}, h{0}, channels{0}; std::vector<float> px; std::string error; }; // interleaved; channels = 1 (Y), 3 (RGB) or 4 (RGBA)

inline float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t) (h >> 15) << 31, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else { // subnormal: renormalise
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            bits = s | ((uint32_t) (127 - 15 - sh + 1) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) bits = s | 0x7f800000u | (m << 13);
    else bits = s | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}
inline uint16_t float_to_half(float f) { // round to nearest even, overflow to infinity
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t s = (x >> 16) & 0x8000u, e = (x >> 23) & 0xffu, m = x & 0x7fffffu;
    if (e == 255) return (uint16_t) (s | 0x7c00u | (m ? 0x200u | (m >> 13) : 0u));
    int ne = (int) e - 127 + 15;
    if (ne >= 31) return (uint16_t) (s | 0x7c00u);
    if (ne <= 0) {
        if (ne < -10) return (uint16_t) s;
        m |= 0x800000u;
        int shift = 14 - ne;
        uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) ++hm;
        return (uint16_t) (s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fffu;
    uint16_t h = (uint16_t) (s | ((uint32_t) ne << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) ++h; // may carry into the exponent: still the right value
    return h;
}

namespace detail {
inline uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t) p[3] << 24); }
inline uint64_t le64(const uint8_t *p) { return (uint64_t) le32(p) | ((uint64_t) le32(p + 4) << 32); }

// undo "predictor + interleave" of the RLE / ZIP family
inline void unpredict(std::vector<uint8_t> &t, std::vector<uint8_t> &out) {
    for (size_t i = 1; i < t.size(); ++i) t[i] = (uint8_t) (t[i - 1] + t[i] - 128);
    out.resize(t.size());
    size_t half = (t.size() + 1) / 2;
    for (size_t i = 0; i < t.size(); ++i) out[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
}
inline bool unrle(const uint8_t *in, size_t n, size_t expect, std::vector<uint8_t> &out) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        int c = (int8_t) in[i++];
        if (c < 0) { size_t k = (size_t) -c; if (i + k > n) return false; out.insert(out.end(), in + i, in + i + k); i += k; }
        else { if (i >= n) return false; out.insert(out.end(), (size_t) c + 1, in[i++]); }
        if (out.size() > expect) return false;
    }
    return out.size() == expect;
}

// ---- PIZ ----
struct MsbBits { // the Huffman stream of PIZ is read most significant bit first
    const uint8_t *p, *end; uint64_t c{0}; int lc{0};
    int bit() { if (lc == 0) { c = p < end ? *p++ : 0u; lc = 8; } --lc; return (int) ((c >> lc) & 1u); }
    uint32_t get(int n) { uint32_t v = 0; while (n--) v = (v << 1) | (uint32_t) bit(); return v; }
};
inline bool huf_uncompress(const uint8_t *in, size_t n_in, uint16_t *out, size_t n_out, std::string &err) {
    if (n_out == 0) return true;
    if (n_in < 20) { err = "PIZ: truncated Huffman header"; return false; }
    const uint32_t im = le32(in), iM = le32(in + 4), n_bits = le32(in + 12);
    constexpr uint32_t ENC = (1u << 16) + 1u;
    if (im >= ENC || iM >= ENC || im > iM) { err = "PIZ: bad Huffman symbol range"; return false; }
    std::vector<uint8_t> len(ENC, 0);
    MsbBits br{in + 20, in + n_in};
    for (uint32_t i = im; i <= iM; ++i) { // packed code lengths: 6 bits each, 59..62 = short zero runs, 63 = long zero run
        uint32_t l = br.get(6);
        if (l == 63) { uint32_t run = br.get(8) + 6; if (i + run > iM + 1) { err = "PIZ: code-length run overflows the table"; return false; } i += run - 1; }
        else if (l >= 59) { uint32_t run = l - 59 + 2; if (i + run > iM + 1) { err = "PIZ: code-length run overflows the table"; return false; } i += run - 1; }
        else len[i] = (uint8_t) l;
    }
    // the table is followed by the data at the next byte boundary
    const uint8_t *data = br.p;
    if ((size_t) (br.end - data) * 8 < n_bits) { err = "PIZ: truncated Huffman data"; return false; }
    // canonical codes: assigned from the longest length up, consecutive within a length in symbol order
    uint64_t cnt[59] = {0}, base[59];
    for (uint32_t i = im; i <= iM; ++i) cnt[len[i]]++;
    cnt[0] = 0;
    uint64_t c = 0;
    for (int l = 58; l > 0; --l) { uint64_t nc = (c + cnt[l]) >> 1; base[l] = c; c = nc; }
    uint32_t offs[60]; offs[1] = 0;
    for (int l = 1; l < 59; ++l) offs[l + 1] = offs[l] + (uint32_t) cnt[l];
    std::vector<uint32_t> sym(offs[59] ? offs[59] : 1);
    { uint32_t fill[60]; std::memcpy(fill, offs, sizeof fill); for (uint32_t i = im; i <= iM; ++i) if (len[i]) sym[fill[len[i]]++] = i; }
    MsbBits d{data, data + (n_bits + 7) / 8};
    size_t o = 0; uint64_t used = 0;
    while (o < n_out && used < n_bits) {
        uint64_t code = 0; int l = 0; uint32_t s = 0xffffffffu;
        while (l < 58 && used < n_bits) {
            code = (code << 1) | (uint64_t) d.bit(); ++l; ++used;
            if (cnt[l] && code >= base[l] && code - base[l] < cnt[l]) { s = sym[offs[l] + (uint32_t) (code - base[l])]; break; }
        }
        if (s == 0xffffffffu) { err = "PIZ: invalid Huffman code"; return false; }
        if (s == iM) { // run-length symbol: repeat the previous output value
            if (used + 8 > n_bits) { err = "PIZ: truncated run"; return false; }
            uint32_t run = d.get(8); used += 8;
            if (o == 0 || o + run > n_out) { err = "PIZ: bad run"; return false; }
            for (uint32_t k = 0; k < run; ++k) { out[o] = out[o - 1]; ++o; }
        } else out[o++] = (uint16_t) s;
    }
    if (o != n_out) { err = "PIZ: Huffman data ends early"; return false; }
    return true;
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    int16_t ls = (int16_t) l, hs = (int16_t) h;
    int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t) (int16_t) ai; b = (uint16_t) (int16_t) (ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    int m = l, d = h;
    int bb = (m - (d >> 1)) & 0xffff;
    int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t) bb; a = (uint16_t) aa;
}
inline void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    int n = nx > ny ? ny : nx, p = 1;
    while (p <= n) p <<= 1;
    p >>= 1; int p2 = p; p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t *py = in, *ey = in + (ptrdiff_t) oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t) oy * p, oy2 = (ptrdiff_t) oy * p2, ox1 = (ptrdiff_t) ox * p, ox2 = (ptrdiff_t) ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                dec(*px, *p10, i00, i10); dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01); dec(i10, i11, *p10, *p11);
            }
            if (nx & p) { uint16_t *p10 = px + oy1; dec(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t *px = py, *ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t *p01 = px + ox1; dec(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}
struct Channel { std::string name; int type; int size; }; // type 0 uint, 1 half, 2 float; size in bytes
inline bool unpiz(const uint8_t *in, size_t n, const std::vector<Channel> &chans, int nx, int ny, std::vector<uint8_t> &out, std::string &err) {
    size_t words = 0;
    for (auto &c : chans) words += (size_t) nx * ny * (c.size / 2);
    if (n < 4) { err = "PIZ: truncated block"; return false; }
    const uint32_t min_nz = in[0] | (in[1] << 8), max_nz = in[2] | (in[3] << 8);
    std::vector<uint8_t> bitmap(8192, 0);
    size_t p = 4;
    if (min_nz <= max_nz) {
        if (max_nz >= 8192 || p + (max_nz - min_nz + 1) > n) { err = "PIZ: bad bitmap range"; return false; }
        std::memcpy(bitmap.data() + min_nz, in + p, max_nz - min_nz + 1); p += max_nz - min_nz + 1;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t) i;
    const uint16_t max_value = (uint16_t) (k - 1);
    if (p + 4 > n) { err = "PIZ: truncated block"; return false; }
    const uint32_t hlen = le32(in + p); p += 4;
    if (p + hlen > n) { err = "PIZ: truncated Huffman stream"; return false; }
    std::vector<uint16_t> tmp(words);
    if (!huf_uncompress(in + p, hlen, tmp.data(), words, err)) return false;
    { // wavelet: per channel, per 16-bit sub-plane of the sample
        uint16_t *q = tmp.data();
        for (auto &c : chans) {
            const int size = c.size / 2;
            for (int j = 0; j < size; ++j) wav2_decode(q + j, nx, size, ny, nx * size, max_value);
            q += (size_t) nx * ny * size;
        }
    }
    for (auto &v : tmp) v = lut[v];
    // channel planes -> scanline-interleaved bytes (little endian)
    out.resize(words * 2);
    std::vector<const uint16_t *> cur(chans.size());
    { const uint16_t *q = tmp.data(); for (size_t c = 0; c < chans.size(); ++c) { cur[c] = q; q += (size_t) nx * ny * (chans[c].size / 2); } }
    uint8_t *o = out.data();
    for (int y = 0; y < ny; ++y)
        for (size_t c = 0; c < chans.size(); ++c) {
            const size_t cnt = (size_t) nx * (chans[c].size / 2);
            for (size_t i = 0; i < cnt; ++i) { uint16_t v = cur[c][i]; *o++ = (uint8_t) v; *o++ = (uint8_t) (v >> 8); }
            cur[c] += cnt;
        }
    return true;
}
}// namespace detail

inline ImageF decode(const std::vector<uint8_t> &f) {
    using namespace detail;
    ImageF img;
    auto fail = [&](const std::string &m) { img.error = "exr: " + m; img.px.clear(); return img; };
    if (f.size() < 16 || le32(f.data()) != 20000630u) return fail("bad magic number");
    const uint32_t version = le32(f.data() + 4);
    if ((version & 0xffu) != 2) return fail("unsupported file version");
    if (version & 0x200u) return fail("tiled images are not supported (scanline only)");
    if (version & 0x1800u) return fail("deep / multi-part files are not supported");
    size_t p = 8;
    std::vector<Channel> chans;
    int compression = -1, line_order = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    bool have_dw = false;
    auto cstr = [&](std::string &s) { size_t e = p; while (e < f.size() && f[e]) ++e; if (e >= f.size()) return false; s.assign((const char *) f.data() + p, e - p); p = e + 1; return true; };
    for (;;) {
        std::string name, type;
        if (p >= f.size()) return fail("truncated header");
        if (f[p] == 0) { ++p; break; }
        if (!cstr(name) || !cstr(type) || p + 4 > f.size()) return fail("truncated header");
        const uint32_t size = le32(f.data() + p); p += 4;
        if (p + size > f.size()) return fail("truncated attribute '" + name + "'");
        const uint8_t *v = f.data() + p;
        if (name == "channels") {
            size_t q = 0;
            while (q < size && v[q]) {
                Channel c; size_t e = q; while (e < size && v[e]) ++e;
                if (e + 17 > size) return fail("truncated channel list");
                c.name.assign((const char *) v + q, e - q); q = e + 1;
                c.type = (int) le32(v + q);
                if (le32(v + q + 8) != 1 || le32(v + q + 12) != 1) return fail("subsampled channel '" + c.name + "' is not supported");
                if (c.type < 0 || c.type > 2) return fail("bad pixel type");
                c.size = c.type == 1 ? 2 : 4;
                q += 16; chans.push_back(c);
            }
        } else if (name == "compression" && size >= 1) compression = v[0];
        else if (name == "dataWindow" && size >= 16) { for (int i = 0; i < 4; ++i) dw[i] = (int32_t) le32(v + 4 * i); have_dw = true; }
        else if (name == "lineOrder" && size >= 1) line_order = v[0];
        p += size;
    }
    if (chans.empty() || !have_dw || compression < 0) return fail("header lacks channels / dataWindow / compression");
    if (compression > 4) return fail("compression method " + std::to_string(compression) + " (PXR24 / B44 / DWA) is not supported");
    if (line_order > 1) return fail("random line order is a tiled-file feature");
    const int64_t w = (int64_t) dw[2] - dw[0] + 1, h = (int64_t) dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) return fail("bad data window");
    const int lines_per_block = compression == 4 ? 32 : (compression == 3 ? 16 : 1);
    const size_t n_blocks = (size_t) ((h + lines_per_block - 1) / lines_per_block);
    if (p + n_blocks * 8 > f.size()) return fail("truncated offset table");
    size_t line_bytes = 0;
    for (auto &c : chans) line_bytes += (size_t) w * c.size;
    // channel -> output slot: R G B A (or Y as grey)
    int slot[4] = {-1, -1, -1, -1};
    for (size_t i = 0; i < chans.size(); ++i) {
        const std::string &n = chans[i].name;
        if (n == "R") slot[0] = (int) i; else if (n == "G") slot[1] = (int) i; else if (n == "B") slot[2] = (int) i; else if (n == "A") slot[3] = (int) i;
    }
    int out_ch;
    if (slot[0] >= 0 && slot[1] >= 0 && slot[2] >= 0) out_ch = slot[3] >= 0 ? 4 : 3;
    else {
        int y = -1;
        for (size_t i = 0; i < chans.size(); ++i) if (chans[i].name == "Y") y = (int) i;
        if (y < 0) return fail("neither R, G, B nor Y channels");
        slot[0] = y; out_ch = 1;
    }
    img.w = (uint32_t) w; img.h = (uint32_t) h; img.channels = (uint32_t) out_ch;
    img.px.assign((size_t) w * h * out_ch, 0.f);
    std::vector<size_t> chan_off(chans.size());
    { size_t o = 0; for (size_t i = 0; i < chans.size(); ++i) { chan_off[i] = o; o += (size_t) w * chans[i].size; } }
    std::vector<uint8_t> raw, tmp;
    for (size_t b = 0; b < n_blocks; ++b) {
        const uint64_t off = le64(f.data() + p + b * 8);
        if (off + 8 > f.size()) return fail("block offset beyond the file");
        const int32_t y0 = (int32_t) le32(f.data() + off);
        const uint32_t csize = le32(f.data() + off + 4);
        if (off + 8 + csize > f.size()) return fail("truncated block");
        if (y0 < dw[1] || y0 > dw[3]) return fail("block outside the data window");
        const int ny = (int) std::min<int64_t>(lines_per_block, (int64_t) dw[3] - y0 + 1);
        const size_t expect = line_bytes * (size_t) ny;
        const uint8_t *src = f.data() + off + 8;
        if (csize == expect || compression == 0) { if (csize != expect) return fail("bad uncompressed block size"); raw.assign(src, src + csize); }
        else if (compression == 1) { if (!unrle(src, csize, expect, tmp)) return fail("bad RLE block"); unpredict(tmp, raw); }
        else if (compression == 2 || compression == 3) {
            std::vector<uint8_t> z(src, src + csize); std::string e;
            tmp.clear();
            if (!vmk_img::inflate_zlib(z, tmp, e, expect) || tmp.size() != expect) return fail("bad ZIP block" + (e.empty() ? std::string() : ": " + e));
            unpredict(tmp, raw);
        } else { std::string e; if (!unpiz(src, csize, chans, (int) w, ny, raw, e)) return fail(e); }
        for (int ly = 0; ly < ny; ++ly) {
            const size_t row = (size_t) (y0 - dw[1] + ly);
            const uint8_t *line = raw.data() + (size_t) ly * line_bytes;
            for (int k = 0; k < (out_ch == 1 ? 1 : out_ch); ++k) {
                const int ci = slot[k];
                const uint8_t *q = line + chan_off[(size_t) ci];
                float *dst = img.px.data() + row * (size_t) w * out_ch + k;
                for (int64_t x = 0; x < w; ++x) {
                    float v;
                    if (chans[(size_t) ci].type == 1) v = half_to_float((uint16_t) (q[2 * x] | (q[2 * x + 1] << 8)));
                    else if (chans[(size_t) ci].type == 2) { uint32_t u = le32(q + 4 * x); std::memcpy(&v, &u, 4); }
                    else v = (float) le32(q + 4 * x);
                    dst[(size_t) x * out_ch] = v;
                }
            }
        }
    }
    return img;
}

// ---- writer: scanline, HALF or FLOAT, NONE or ZIP; `rgba` has 4 floats per pixel, `channels` = 3 (B G R) or 4 (A B G R) are stored ----
inline std::vector<uint8_t> encode(uint32_t w, uint32_t h, const float *rgba, int channels, bool half, bool zip) {
    std::vector<uint8_t> f;
    auto u8 = [&](uint8_t v) { f.push_back(v); };
    auto u32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) f.push_back((uint8_t) (v >> (8 * i))); };
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; ++i) f.push_back((uint8_t) (v >> (8 * i))); };
    auto str = [&](const char *s) { while (*s) f.push_back((uint8_t) *s++); f.push_back(0); };
    auto f32 = [&](float v) { uint32_t u; std::memcpy(&u, &v, 4); u32(u); };
    u32(20000630u); u32(2u);
    const char *names[4] = {"A", "B", "G", "R"}; const int src[4] = {3, 2, 1, 0};
    const int first = channels == 4 ? 0 : 1, nch = channels == 4 ? 4 : 3;
    str("channels"); str("chlist"); u32((uint32_t) (nch * 18 + 1));
    for (int c = first; c < 4; ++c) { str(names[c]); u32(half ? 1u : 2u); u8(0); u8(0); u8(0); u8(0); u32(1); u32(1); }
    u8(0);
    str("compression"); str("compression"); u32(1); u8(zip ? 3 : 0);
    str("dataWindow"); str("box2i"); u32(16); u32(0); u32(0); u32(w - 1); u32(h - 1);
    str("displayWindow"); str("box2i"); u32(16); u32(0); u32(0); u32(w - 1); u32(h - 1);
    str("lineOrder"); str("lineOrder"); u32(1); u8(0);
    str("pixelAspectRatio"); str("float"); u32(4); f32(1.f);
    str("screenWindowCenter"); str("v2f"); u32(8); f32(0.f); f32(0.f);
    str("screenWindowWidth"); str("float"); u32(4); f32(1.f);
    u8(0);
    const uint32_t lines = zip ? 16u : 1u, n_blocks = (h + lines - 1) / lines;
    const size_t table = f.size();
    for (uint32_t b = 0; b < n_blocks; ++b) u64(0);
    const size_t px_size = half ? 2 : 4, line_bytes = (size_t) w * nch * px_size;
    std::vector<uint8_t> raw, t;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t y0 = b * lines, ny = std::min(lines, h - y0);
        raw.resize(line_bytes * ny);
        uint8_t *o = raw.data();
        for (uint32_t y = y0; y < y0 + ny; ++y)
            for (int c = first; c < 4; ++c)
                for (uint32_t x = 0; x < w; ++x) {
                    const float v = rgba[((size_t) y * w + x) * 4 + src[c]];
                    if (half) { uint16_t hv = float_to_half(v); *o++ = (uint8_t) hv; *o++ = (uint8_t) (hv >> 8); }
                    else { uint32_t u; std::memcpy(&u, &v, 4); for (int i = 0; i < 4; ++i) *o++ = (uint8_t) (u >> (8 * i)); }
                }
        const uint64_t off = f.size();
        for (int i = 0; i < 8; ++i) f[table + b * 8 + i] = (uint8_t) (off >> (8 * i));
        u32(y0);
        if (zip) { // interleave + predictor, then zlib; a block that does not shrink is stored raw (as the format prescribes)
            t.resize(raw.size());
            const size_t half_n = (raw.size() + 1) / 2;
            for (size_t i = 0; i < raw.size(); ++i) { if (i & 1) t[half_n + i / 2] = raw[i]; else t[i / 2] = raw[i]; }
            for (size_t i = t.size(); i-- > 1;) t[i] = (uint8_t) (t[i] - t[i - 1] + 128);
            std::vector<uint8_t> z = vmk_img::deflate_zlib(t.data(), t.size());
            if (z.size() < raw.size()) { u32((uint32_t) z.size()); f.insert(f.end(), z.begin(), z.end()); continue; }
        }
        u32((uint32_t) raw.size()); f.insert(f.end(), raw.begin(), raw.end());
    }
    return f;
}

}// namespace vmk_exr
