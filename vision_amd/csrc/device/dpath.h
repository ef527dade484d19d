// dpath.h — device functions of the path-tracing hot path, hand-written for gfx950.
// Each block cites the Vision source (`src/`-relative file:line) whose arithmetic it reproduces.
#pragma once
#include "dscene.h"

// VMK_HERO = 1 compiles the path code for the hero-wavelength spectrum (render_core/spectrum/hero.cpp) in its own
// translation unit (vmk_hero.hip, namespace vmkd_hero): every colour that enters the path is uplifted to a spectrum
// sampled at the path's wavelengths (three, or VMK_SPEC_DIM = 4 of them in vmk_hero4.hip).  With VMK_HERO = 0 (srgb.cpp) the colour helpers below are the identity and
// the SWL_P / SWL_A parameter macros expand to nothing, so the sRGB megakernel is the same code as before.
#ifndef VMK_HERO
#define VMK_HERO 0
#endif
#if VMK_HERO
#define SWL_P , const Swl &swl
#define SWL_A , swl
#else
#define SWL_P
#define SWL_A
#endif

namespace vmkd {

// =====================================================================================================
// a1. sampler: TEA-seeded LCG — render_core/sampler/independent.cpp:24-42, math/util.h:13-33
// =====================================================================================================
VD uint32_t tea(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
struct Sampler {
    uint32_t state;
    VD void start(uint32_t px, uint32_t py, uint32_t sample_index, uint32_t dim) { state = tea(tea(px, py), tea(sample_index, dim)); }
    VD float next_1d() {
        state = 1664525u * state + 1013904223u;
        return ((float) (state & 0x00ffffffu) * 1.f) * (1.f / 16777216.f);
    }
    VD V2 next_2d() { float x = next_1d(); float y = next_1d(); return {x, y}; }
};

// =====================================================================================================
// warps / MIS — math/warp.h
// =====================================================================================================
VD V2 square_to_disk(V2 u) { // warp.h:26-31
    float r = sqrt_(u.x);
    float theta = _2Pi * u.y;
    float s, c; sincos_(theta, &s, &c);
    return {r * c, r * s};
}
VD V3 square_to_cosine_hemisphere(V2 u) { // warp.h:38-43
    V2 d = square_to_disk(u);
    float z = sqrt_(fmax_(0.f, 1.f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
VD float cosine_hemisphere_PDF(float cos_t) { return cos_t * InvPi; }
VD V2 square_to_triangle(V2 u) { float su0 = sqrt_(u.x); return {1.f - su0, u.y * su0}; } // warp.h:65-69
VD float sample_linear(float u, float a, float b) { // warp.h:122-128
    float x = u * (a + b) / (a + sqrt_(lerp_(u, sqr(a), sqr(b))));
    float ret = fmin_(x, OneMinusEpsilon);
    return (u == 0.f && a == 0.f) ? 0.f : ret;
}
VD float sample_tent(float u, float r) { // warp.h:131-146
    return u < 0.5f ? -r * sample_linear((0.5f - u) * 2.f, 1.f, 0.f) : r * sample_linear((u - 0.5f) * 2.f, 1.f, 0.f);
}
VD float MIS_weight(float f_pdf, float g_pdf) { return (1.f * f_pdf) / (1.f * f_pdf + 1.f * g_pdf); } // warp.h:149-199
VD float PDF_wi(float pdf_point, V3 normal, V3 wo_un) { // warp.h:89-94
    float cos_t = abs_(dot(normal, normalize(wo_un)));
    return pdf_point * length_squared(wo_un) / cos_t;
}
VD float remapping(float a, float low, float high) { return (a - low) / (high - low); }

// =====================================================================================================
// textures / LUTs: manual bilinear / trilinear on plain HBM arrays (no texture objects).
// tex.sample semantics: normalised coords, texel centres at (i+0.5)/N, repeat wrap (images), clamp (LUTs).
// =====================================================================================================
// Textured slots are the rare case (3 of 45 materials in classroom): the bilinear fetch is kept OUT of line so the ~40
// slot-evaluation sites of the material code stay a compare + 3 moves (code size / VGPR pressure of the megakernel).
VD V4 fetch_texel(const uint8_t *base, uint32_t format, uint32_t width, const float *srgb_lut, int x, int y) {
    size_t i = (size_t) y * width + (size_t) x;
    if (format == VMK_TEX_RGBA32F) {
        float4 p = ldg(reinterpret_cast<const float4 *>(base + i * 16));
        return {p.x, p.y, p.z, p.w};
    }
    uint32_t p = ldg(reinterpret_cast<const uint32_t *>(base + i * 4));
    uint32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu, a = p >> 24;
    if (format == VMK_TEX_RGBA8_SRGB) return {ldg(srgb_lut + r), ldg(srgb_lut + g), ldg(srgb_lut + b), (float) a * (1.f / 255.f)};
    return {(float) r * (1.f / 255.f), (float) g * (1.f / 255.f), (float) b * (1.f / 255.f), (float) a * (1.f / 255.f)};
}
VD int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
__device__ __noinline__ float4 sample_image_ool(const vmk_texture *textures, const uint8_t *tex_data, const float *srgb_lut, uint32_t tex_id, float u, float v) {
    vmk_texture t; // (the pointers are generic inside this out-of-line routine: say where they point)
    t.offset = ldg(&textures[tex_id].offset); t.width = ldg(&textures[tex_id].width); t.height = ldg(&textures[tex_id].height); t.format = ldg(&textures[tex_id].format);
    const uint8_t *base = tex_data + t.offset;
    float x = u * (float) t.width - 0.5f, y = v * (float) t.height - 0.5f;
    float fx0 = floor_(x), fy0 = floor_(y);
    float tx = x - fx0, ty = y - fy0;
    int x0 = wrap_repeat((int) fx0, (int) t.width), y0 = wrap_repeat((int) fy0, (int) t.height);
    int x1 = wrap_repeat((int) fx0 + 1, (int) t.width), y1 = wrap_repeat((int) fy0 + 1, (int) t.height);
    V4 c00 = fetch_texel(base, t.format, t.width, srgb_lut, x0, y0), c10 = fetch_texel(base, t.format, t.width, srgb_lut, x1, y0);
    V4 c01 = fetch_texel(base, t.format, t.width, srgb_lut, x0, y1), c11 = fetch_texel(base, t.format, t.width, srgb_lut, x1, y1);
    V4 r = lerp4(ty, lerp4(tx, c00, c10), lerp4(tx, c01, c11));
    return make_float4(r.x, r.y, r.z, r.w);
}
VD V4 sample_image(const DScene &S, uint32_t tex_id, V2 uv, DCounters &cnt) {
    cnt.tex++;
    float4 r = sample_image_ool(S.textures, S.tex_data, S.srgb_lut, tex_id, uv.x, uv.y);
    return {r.x, r.y, r.z, r.w};
}
VD int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
template<int NC>
VD void sample_lut2d(const float *lut, float u, float v, float *out) {
    const int N = VMK_LUT_RES;
    float x = u * (float) N - 0.5f, y = v * (float) N - 0.5f;
    float fx0 = floor_(x), fy0 = floor_(y);
    float tx = x - fx0, ty = y - fy0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float c00 = ldg(lut + (y0 * N + x0) * NC + c), c10 = ldg(lut + (y0 * N + x1) * NC + c);
        float c01 = ldg(lut + (y1 * N + x0) * NC + c), c11 = ldg(lut + (y1 * N + x1) * NC + c);
        out[c] = lerp_(ty, lerp_(tx, c00, c10), lerp_(tx, c01, c11));
    }
}
template<int NC>
VD void sample_lut3d(const float *lut, V3 uvw, float *out) {
    const int N = VMK_LUT_RES;
    float x = uvw.x * (float) N - 0.5f, y = uvw.y * (float) N - 0.5f, z = uvw.z * (float) N - 0.5f;
    float fx0 = floor_(x), fy0 = floor_(y), fz0 = floor_(z);
    float tx = x - fx0, ty = y - fy0, tz = z - fz0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
    int z0 = clampi((int) fz0, 0, N - 1), z1 = clampi((int) fz0 + 1, 0, N - 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float v000 = ldg(lut + ((z0 * N + y0) * N + x0) * NC + c), v100 = ldg(lut + ((z0 * N + y0) * N + x1) * NC + c);
        float v010 = ldg(lut + ((z0 * N + y1) * N + x0) * NC + c), v110 = ldg(lut + ((z0 * N + y1) * N + x1) * NC + c);
        float v001 = ldg(lut + ((z1 * N + y0) * N + x0) * NC + c), v101 = ldg(lut + ((z1 * N + y0) * N + x1) * NC + c);
        float v011 = ldg(lut + ((z1 * N + y1) * N + x0) * NC + c), v111 = ldg(lut + ((z1 * N + y1) * N + x1) * NC + c);
        float a = lerp_(ty, lerp_(tx, v000, v100), lerp_(tx, v010, v110));
        float b = lerp_(ty, lerp_(tx, v001, v101), lerp_(tx, v011, v111));
        out[c] = lerp_(tz, a, b);
    }
}
// ShaderNodeSlot::evaluate (shader_node.cpp:242-273): constant or image*scale, swizzled
VD V3 eval_slot3(const DScene &S, const vmk_slot &sl, V2 uv, DCounters &cnt) {
    if (sl.tex == VMK_INVALID) return {sl.v[0], sl.v[1], sl.v[2]};
    V4 t = sample_image(S, sl.tex & 0xffffu, uv, cnt);
    const bool tinted = (sl.tex & VMK_SLOT_TINTED) != 0u; // image x constant ("multiply" node, math.cpp:78-90)
    const float scale = tinted ? 1.f : sl.v[0];
    float c[4] = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
    uint32_t sw = sl.tex >> 16;
    auto pick = [&](uint32_t k) { return k == 0 ? c[0] : (k == 1 ? c[1] : (k == 2 ? c[2] : c[3])); };
    V3 r = {pick(sw & 3u), pick((sw >> 2) & 3u), pick((sw >> 4) & 3u)};
    if (tinted) r = r * V3{sl.v[0], sl.v[1], sl.v[2]};
    return r;
}
VD float eval_slot1(const DScene &S, const vmk_slot &sl, V2 uv, DCounters &cnt) {
    if (sl.tex == VMK_INVALID) return sl.v[0];
    return eval_slot3(S, sl, uv, cnt).x;
}

// =====================================================================================================
// a2. spectrum — render_core/spectrum/{srgb,hero}.cpp, base/color/{spd,spectrum}.cpp
// A SampledSpectrum is a Spec (dmath.h): a V3 for dimension 3 in both modes — the (R, G, B) channels for srgb, the values at the
// path's wavelengths (hero, hero + 1/3, hero + 2/3 of the sampling domain) for hero — or four values for hero dimension 4.
// =====================================================================================================
#if VMK_HERO
struct Swl { Spec lambda, pdf; }; // SampledWavelengths (spectrum.h:18-56): pdf == 0 marks an invalidated secondary wavelength
VD float sample_visible_wavelength(float u) { return 538.f - 138.888889f * atanh_(0.85691062f - 1.82750197f * u); } // hero.cpp:15-18
VD float visible_wavelength_PDF(float lambda) { return 0.0039398042f / sqr(cosh_(0.0072f * (lambda - 538.f))); }     // hero.cpp:21-24
// HeroWavelengthSpectrum::sample_wavelength (hero.cpp:286-299), 1 draw
VD Swl sample_wavelengths(Sampler &sampler) {
    Swl swl;
    float u = sampler.next_1d();
    float l[kSpecDim], p[kSpecDim];
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) {
        float offset = (float) i * (1.f / (float) kSpecDim);
        float up = fract_(u + offset);
        l[i] = sample_visible_wavelength(up);
        p[i] = visible_wavelength_PDF(l[i]);
    }
#if VMK_SPEC_DIM == 4
    swl.lambda = {l[0], l[1], l[2], l[3]}; swl.pdf = {p[0], p[1], p[2], p[3]};
#else
    swl.lambda = {l[0], l[1], l[2]}; swl.pdf = {p[0], p[1], p[2]};
#endif
    return swl;
}
// SPD::eval (spd.cpp:79-86)
VD float spd_eval(const float *f, float interval, float lambda) {
    float t = (clamp_(lambda, 360.f, 830.f) - 360.f) / interval;
    uint32_t sample_count = (uint32_t) ((830.f - 360.f) / interval) + 1u;
    uint32_t i = (uint32_t) fmin_(t, (float) (sample_count - 2u));
    float l = f[i], r = f[i + 1u];
    return lerp_(fract_(t), l, r);
}
VD Spec spd_eval3(const float *f, float interval, const Swl &swl) { return smap(swl.lambda, [&](float lambda) { return spd_eval(f, interval, lambda); }); }
// RGBSigmoidPolynomial (hero.cpp:27-49)
VD float sigmoid_polynomial(V3 c, float lambda) {
    float x = fma_(fma_(c.x, lambda, c.y), lambda, c.z);
    float s = 0.5f * fma_(x, rsqrt_(fma_(x, x, 1.f)), 1.f);
    return isinf_(x) ? (x > 0.0f ? 1.f : 0.f) : s;
}
VD float inverse_smooth_step(float x) { return 0.5f - sin_(asin_(1.0f - 2.0f * x) * (1.0f / 3.0f)); } // hero.cpp:66-68
// RGBToSpectrumTable::decode_albedo, device variant (hero.cpp:142-171): the three 64^3 float4 textures, trilinear,
// clamp addressing, texel centres at (i + 0.5) / res.  Kept out of line: ~10 call sites in the material code.
__device__ __noinline__ V3 rgb2spec_fetch_ool(const float *table, uint32_t maxc, float cx, float cy, float cz) {
    const int N = (int) VMK_RGB2SPEC_RES;
    const float4 *t = reinterpret_cast<const float4 *>(table) + (size_t) maxc * N * N * N;
    float x = cx * (float) N - 0.5f, y = cy * (float) N - 0.5f, z = cz * (float) N - 0.5f;
    float fx0 = floor_(x), fy0 = floor_(y), fz0 = floor_(z);
    float tx = x - fx0, ty = y - fy0, tz = z - fz0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
    int z0 = clampi((int) fz0, 0, N - 1), z1 = clampi((int) fz0 + 1, 0, N - 1);
    auto at = [&](int xi, int yi, int zi) { float4 v = ldg(t + ((size_t) zi * N + yi) * N + xi); return V3{v.x, v.y, v.z}; };
    V3 a = lerp3(ty, lerp3(tx, at(x0, y0, z0), at(x1, y0, z0)), lerp3(tx, at(x0, y1, z0), at(x1, y1, z0)));
    V3 b = lerp3(ty, lerp3(tx, at(x0, y0, z1), at(x1, y0, z1)), lerp3(tx, at(x0, y1, z1), at(x1, y1, z1)));
    return lerp3(tz, a, b);
}
VD V3 rgb2spec_albedo_coeffs(const DScene &S, V3 rgb_in) {
    V3 rgb = {clamp_(rgb_in.x, 0.f, 1.f), clamp_(rgb_in.y, 0.f, 1.f), clamp_(rgb_in.z, 0.f, 1.f)};
    V3 c = {0.0f, 0.0f, (rgb.x - 0.5f) * rsqrt_(rgb.x * (1.0f - rgb.x))};
    if (!(rgb.x == rgb.y && rgb.y == rgb.z)) {
        uint32_t maxc = rgb.x > rgb.y ? (rgb.x > rgb.z ? 0u : 2u) : (rgb.y > rgb.z ? 1u : 2u);
        float v[3] = {rgb.x, rgb.y, rgb.z};
        float z = v[maxc];
        float x = v[(maxc + 1u) % 3u] / z;
        float y = v[(maxc + 2u) % 3u] / z;
        float zz = inverse_smooth_step(inverse_smooth_step(z));
        const float res = (float) VMK_RGB2SPEC_RES;
        const float sc = (res - 1.0f) / res, of = 0.5f / res;
        c = rgb2spec_fetch_ool(S.hero.rgb2spec, maxc, fma_(x, sc, of), fma_(y, sc, of), fma_(zz, sc, of));
    }
    return c;
}
// decode_unbound (hero.cpp:173-179): (c0, c1, c2, scale)
VD V3 rgb2spec_unbound_coeffs(const DScene &S, V3 rgb_in, float *scale_out) {
    V3 rgb = {fmax_(rgb_in.x, 0.f), fmax_(rgb_in.y, 0.f), fmax_(rgb_in.z, 0.f)};
    float m = max_comp(rgb);
    float scale = 2.f * m;
    *scale_out = scale;
    return rgb2spec_albedo_coeffs(S, scale == 0.f ? mk3(0.f) : rgb / scale);
}
VD Spec sigmoid3(V3 c, const Swl &swl) { return smap(swl.lambda, [&](float lambda) { return sigmoid_polynomial(c, lambda); }); }
#endif
// decode_to_albedo / decode_to_unbound_spectrum / decode_to_illumination (srgb.cpp:49-57, hero.cpp:331-342)
VD Spec spec_albedo(const DScene &S, V3 rgb SWL_P) {
#if VMK_HERO
    return sigmoid3(rgb2spec_albedo_coeffs(S, rgb), swl);
#else
    return rgb;
#endif
}
VD Spec spec_unbound(const DScene &S, V3 rgb SWL_P) {
#if VMK_HERO
    float scale; V3 c = rgb2spec_unbound_coeffs(S, rgb, &scale);
    return sigmoid3(c, swl) * scale; // RGBUnboundSpectrum::eval hero.cpp:201-203
#else
    return rgb;
#endif
}
VD Spec spec_illumination(const DScene &S, V3 rgb SWL_P) {
#if VMK_HERO
    float scale; V3 c = rgb2spec_unbound_coeffs(S, rgb, &scale);
    return (sigmoid3(c, swl) * scale) * spd_eval3(S.hero.spd_data + S.hero.spd_cie[3], S.hero.spd_cie_interval, swl); // RGBIlluminationSpectrum::eval hero.cpp:219-221
#else
    return rgb;
#endif
}
// Spectrum::linear_srgb (srgb.cpp:46-48; hero.cpp:265-267,281-291 + cie::xyz_to_linear_srgb cie.h:413-420)
VD V3 spec_linear_srgb(const DScene &S, Spec sp SWL_P) {
#if VMK_HERO
    const float *X = S.hero.spd_data + S.hero.spd_cie[0], *Y = S.hero.spd_data + S.hero.spd_cie[1], *Z = S.hero.spd_data + S.hero.spd_cie[2];
    float l[kSpecDim], p[kSpecDim], v[kSpecDim];
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) { l[i] = scomp(swl.lambda, i); p[i] = scomp(swl.pdf, i); v[i] = scomp(sp, i); }
    V3 sum = mk3(0.f);
    uint32_t valid = 0;
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) {
        float x = spd_eval(X, S.hero.spd_cie_interval, l[i]) * v[i], y = spd_eval(Y, S.hero.spd_cie_interval, l[i]) * v[i], z = spd_eval(Z, S.hero.spd_cie_interval, l[i]) * v[i];
        sum += V3{p[i] == 0.f ? 0.f : x / p[i], p[i] == 0.f ? 0.f : y / p[i], p[i] == 0.f ? 0.f : z / p[i]};
        valid += p[i] > 0.f ? 1u : 0u;
    }
    float factor = 1.f / ((float) valid * S.hero.cie_y_integral);
    V3 xyz = sum * factor;
    return {3.240479f * xyz.x + -1.537150f * xyz.y + -0.498535f * xyz.z,
            -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
            0.055648f * xyz.x + -0.204043f * xyz.y + 1.057311f * xyz.z};
#else
    return sp;
#endif
}
// a colour slot evaluated to a spectrum: ShaderNodeSlot::eval_albedo_spectrum / eval_illumination_spectrum (shader_node.cpp:317-333)
VD Spec eval_slot_albedo(const DScene &S, const vmk_slot &sl, V2 uv, DCounters &cnt SWL_P) { return spec_albedo(S, eval_slot3(S, sl, uv, cnt) SWL_A); }
VD Spec eval_slot_illumination(const DScene &S, const vmk_slot &sl, V2 uv, DCounters &cnt SWL_P) { return spec_illumination(S, eval_slot3(S, sl, uv, cnt) SWL_A); }
// a number slot that may be an "spd" node in hero mode (metal eta / k, dispersive glass ior): one value per wavelength
VD Spec eval_slot_spd(const DScene &S, const vmk_slot &sl, V2 uv, DCounters &cnt SWL_P) {
#if VMK_HERO
    if (sl.tex == VMK_SLOT_SPD) return spd_eval3(S.hero.spd_data + f2u(sl.v[0]), sl.v[2], swl); // SPDNode::evaluate spd.cpp:36-39
#endif
#if VMK_SPEC_DIM == 4
    return mks(eval_slot1(S, sl, uv, cnt)); // (a complete spectrum feeds these slots from "spd" nodes, metal.cpp:113-117, glass.cpp:229-233; a plain number is one value)
#else
    return eval_slot3(S, sl, uv, cnt);
#endif
}

// =====================================================================================================
// a5. Interaction — Geometry::compute_surface_interaction (base/mgr/geometry.cpp:79-166)
// =====================================================================================================
struct Interaction {
    V3 pos, wo, ng;
    V2 uv;
    Frame shading;
    float prim_area;
    uint32_t prim_id, mat_id, light_id;
};
VD V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
VD V2 ld2(const float *p) { return {p[0], p[1]}; }
VD V3 triangle_lerp(V2 b, V3 a0, V3 a1, V3 a2) { return a0 * (1.f - b.x - b.y) + a1 * b.x + a2 * b.y; }
VD V2 triangle_lerp2(V2 b, V2 a0, V2 a1, V2 a2) {
    float w = 1.f - b.x - b.y;
    return {a0.x * w + a1.x * b.x + a2.x * b.y, a0.y * w + a1.y * b.x + a2.y * b.y};
}
// `tri` indexes the Morton-ordered triangle arrays. COMPLETE = is_complete of the reference.
template<bool COMPLETE>
VD void compute_surface_interaction(const DScene &S, uint32_t tri, uint32_t inst_id, uint32_t prim_id, V2 bary, Interaction &it) {
    const vmk_tri_pos *tp = S.tri_pos + tri;
    const vmk_tri_attr *ta = S.tri_attr + tri;
    const vmk_instance *inst = S.instances + inst_id;
    it.prim_id = prim_id; it.light_id = inst->light_id; it.mat_id = inst->mat_id;
    V3 p0 = ld3(tp->p0), p1 = ld3(tp->p1), p2 = ld3(tp->p2);
    it.pos = triangle_lerp(bary, p0, p1, p2);
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    V3 ng_un = cross(dp02, dp12);
    it.prim_area = 0.5f * length(ng_un);
    V2 t0 = ld2(ta->uv0), t1 = ld2(ta->uv1), t2 = ld2(ta->uv2);
    it.uv = triangle_lerp2(bary, t0, t1, t2);
    V3 ngn = normalize(ng_un);
    it.ng = ngn;
    if constexpr (COMPLETE) {
        V2 duv02 = t0 - t2, duv12 = t1 - t2;
        float det = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = abs_(det) < 1e-8f;
        V3 dp_du, dp_dv;
        if (!degenerate_uv) {
            float inv_det = 1.f / det;
            dp_du = normalize((dp02 * duv12.y - dp12 * duv02.y) * inv_det);
            dp_dv = normalize((dp02 * (-duv12.x) + dp12 * duv02.x) * inv_det);
        } else {
            dp_du = normalize(p1 - p0);
            dp_dv = normalize(p2 - p0);
        }
        it.shading = {dp_du, dp_dv, ngn};
        V3 normal = triangle_lerp(bary, ld3(ta->n0), ld3(ta->n1), ld3(ta->n2));
        if (!is_zero(normal)) { // PartialDerivative::update (interaction.h:101-105)
            V3 ns = normalize(mul3x3(inst->n2w, normal));
            it.shading.z = ns;
            it.shading.x = normalize(cross(ns, it.shading.y)) * length(it.shading.x);
            it.shading.y = normalize(cross(ns, it.shading.x)) * length(it.shading.y);
        }
    } else {
        it.shading = {dp02, dp12, ngn};
    }
}

struct Ray { V3 o, d; float t_max; };
// a6. spawn rays — interaction.h:279-309
VD Ray spawn_ray(V3 pos, V3 normal, V3 dir) {
    normal = normal * (dot(normal, dir) > 0.f ? 1.f : -1.f);
    return {offset_ray_origin(pos, normal), dir, RayTMax};
}
VD Ray spawn_ray_to(V3 p_start, V3 n_start, V3 p_target) {
    V3 dir = p_target - p_start;
    n_start = n_start * (dot(n_start, dir) > 0.f ? 1.f : -1.f);
    return {offset_ray_origin(p_start, n_start), dir, 1.f - ShadowEpsilon};
}
VD V3 robust_pos(V3 pos, V3 ng, V3 dir, float factor) { // interaction.h:321-324
    float f = dot(ng, dir) > 0.f ? 1.f : -1.f;
    return offset_ray_origin(pos, (ng * f) * factor);
}

// =====================================================================================================
// a13. GGX microfacet — base/scattering/microfacet.{h,cpp}
// =====================================================================================================
VD V2 calculate_alpha(float alpha, float anisotropic) { // microfacet.h:43-58
    float ax = anisotropic < 0.f ? alpha / (1.f + anisotropic) : alpha * (1.f - anisotropic);
    float ay = anisotropic < 0.f ? alpha * (1.f + anisotropic) : alpha / (1.f - anisotropic);
    if (abs_(anisotropic) <= 1e-4f) return {alpha, alpha};
    return {ax, ay};
}
VD float bsdf_D(V3 wh, float ax, float ay) { // microfacet.cpp:18-22
    V3 H = {wh.x / ax, wh.y / ay, wh.z / 1.f};
    float alpha2 = ax * ay;
    return InvPi / (alpha2 * sqr(length_squared(H)));
}
VD float bsdf_lambda(V3 w, float ax, float ay) { // microfacet.cpp:43-47
    float sqr_alpha_tan_n = (sqr(ax * w.x) + sqr(ay * w.y)) / sqr(w.z);
    float ret = 0.5f * (sqrt_(1.0f + sqr_alpha_tan_n) - 1.0f);
    return w.z == 0.f ? 0.f : ret;
}
VD float bsdf_G1(V3 w, float ax, float ay) { return 1.f / (1.f + bsdf_lambda(w, ax, ay)); }
VD float bsdf_G(V3 wo, V3 wi, float ax, float ay) { return 1.f / (1.f + bsdf_lambda(wo, ax, ay) + bsdf_lambda(wi, ax, ay)); }
VD V3 sample_GGX_VNDF(V3 Ve, V2 u, float ax, float ay) { // microfacet.cpp:73-95
    V3 Vh = normalize(mk3(ax * Ve.x, ay * Ve.y, Ve.z));
    float lenSq = Vh.x * Vh.x + Vh.y * Vh.y;
    V3 T1 = lenSq > 1e-7f ? mk3(-Vh.y, Vh.x, 0.0f) / sqrt_(lenSq) : mk3(1, 0, 0);
    V3 T2 = lenSq > 1e-7f ? cross(Vh, T1) : mk3(0.0f, 1.0f, 0.0f);
    V2 t = square_to_disk(u);
    t.y = lerp_(0.5f * (1.0f + Vh.z), safe_sqrt(1.0f - sqr(t.x)), t.y);
    V3 Nh = T1 * t.x + T2 * t.y + Vh * safe_sqrt(1.0f - (t.x * t.x + t.y * t.y));
    return normalize(mk3(ax * Nh.x, ay * Nh.y, fmax_(0.0f, Nh.z)));
}
VD V3 sample_wh(V3 wo, V2 u, float ax, float ay) { // microfacet.cpp:102-112
    bool flip = wo.z < 0.f;
    V3 wh = sample_GGX_VNDF(flip ? -wo : wo, u, ax, ay);
    return flip ? -wh : wh;
}
VD float PDF_wh(V3 wo, V3 wh, float ax, float ay) { // microfacet.cpp:152-159 (sample_visible)
    return bsdf_D(wh, ax, ay) * bsdf_G1(wo, ax, ay) * abs_dot(wo, wh) / abs_cos_theta(wo);
}
VD float PDF_wi_reflection(V3 wo, V3 wh, float ax, float ay) { return PDF_wh(wo, wh, ax, ay) / (4.f * abs_dot(wo, wh)); }
VD float PDF_wi_transmission(V3 wo, V3 wh, V3 wi, float eta, float ax, float ay) { // microfacet.h:159-165
    float denom = sqr(dot(wi, wh) * eta + dot(wo, wh));
    float dwh_dwi = abs_dot(wi, wh) / denom;
    return PDF_wh(wo, wh, ax, ay) * dwh_dwi;
}
VD float BRDF_div_fr(V3 wo, V3 wh, V3 wi, float ax, float ay) { // microfacet.h:168-175
    return bsdf_D(wh, ax, ay) * bsdf_G(wo, wi, ax, ay) / abs_(4.f * cos_theta(wo) * cos_theta(wi));
}
VD float BTDF_div_ft(V3 wo, V3 wh, V3 wi, float eta, float ax, float ay, bool radiance) { // microfacet.cpp:166-179
    float cos_i = cos_theta(wi), cos_o = cos_theta(wo);
    float numerator = bsdf_D(wh, ax, ay) * bsdf_G(wo, wi, ax, ay) * abs_(dot(wi, wh) * dot(wo, wh));
    float denom = sqr(dot(wi, wh) * eta + dot(wo, wh)) * abs_(cos_i * cos_o);
    float ft = numerator / denom;
    float factor = radiance ? rcp(sqr(eta)) : 1.f;
    ft = denom == 0.f ? 0.f : ft;
    return ft * factor;
}

// =====================================================================================================
// a14. Fresnel — math/optics.h, math/complex.h, base/scattering/fresnel.h, metal.cpp:14-26
// =====================================================================================================
VD bool refract(V3 wi, V3 n, float eta, V3 *wt) { // optics.h:28-39
    float cos_i = dot(n, wi);
    float sin_i_2 = fmax_(0.f, 1.f - sqr(cos_i));
    float sin_t_2 = sin_i_2 / sqr(eta);
    bool valid = sin_t_2 < 1.f;
    float cos_t = safe_sqrt(1.f - sin_t_2);
    *wt = -wi / eta + n * (cos_i / eta - cos_t);
    return valid;
}
VD float schlick_weight(float cos_t) { return pow5(clamp_(1.f - cos_t, 0.f, 1.f)); }
VD float schlick_F0_from_ior(float ior) { return sqr((ior - 1.0f) / (ior + 1.0f)); }
VD float schlick_ior_from_F0(float f0) { float s = sqrt_(clamp_(f0, 0.0f, 0.99f)); return (1.0f + s) / (1.0f - s); }
VD float fresnel_dielectric(float abs_cos_i, float eta) { // optics.h:71-78
    float sin_i_2 = 1.f - sqr(abs_cos_i);
    float sin_t_2 = sin_i_2 / sqr(eta);
    float cos_t = safe_sqrt(1.f - sin_t_2);
    float r_parl = (eta * abs_cos_i - cos_t) / (eta * abs_cos_i + cos_t);
    float r_perp = (abs_cos_i - eta * cos_t) / (abs_cos_i + eta * cos_t);
    return sin_t_2 >= 1.f ? 1.f : (sqr(r_parl) + sqr(r_perp)) * 0.5f;
}
struct Cpx { float re, im; };
VD Cpx cadd(Cpx a, Cpx b) { return {a.re + b.re, a.im + b.im}; }
VD Cpx csub(Cpx a, Cpx b) { return {a.re - b.re, a.im - b.im}; }
VD Cpx cmul(Cpx a, Cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
VD Cpx cdiv(Cpx a, Cpx z) { float sc = 1.f / (z.re * z.re + z.im * z.im); return {sc * (a.re * z.re + a.im * z.im), sc * (a.im * z.re - a.re * z.im)}; }
VD float cnorm_sqr(Cpx z) { return z.re * z.re + z.im * z.im; }
VD Cpx csqrt(Cpx z) { // complex.h:61-69
    float n = sqrt_(cnorm_sqr(z));
    float t1 = sqrt_(0.5f * (n + abs_(z.re)));
    float t2 = 0.5f * z.im / t1;
    Cpx r;
    r.re = n == 0.f ? 0.f : (z.re >= 0.f ? t1 : abs_(t2));
    r.im = n == 0.f ? 0.f : (z.re >= 0.f ? t2 : u2f((f2u(t1) & 0x7fffffffu) | (f2u(z.im) & 0x80000000u)));
    return r;
}
VD float fresnel_complex(float cos_i, float eta_re, float k) { // optics.h:93-102
    Cpx eta = {eta_re, k};
    float sin_i_2 = 1.f - sqr(cos_i);
    Cpx sin_t_2 = cdiv(Cpx{sin_i_2, 0.f}, cmul(eta, eta));
    Cpx cos_t = csqrt(csub(Cpx{1.f, 0.f}, sin_t_2));
    Cpx ci = {cos_i, 0.f};
    Cpx r_parl = cdiv(csub(cmul(eta, ci), cos_t), cadd(cmul(eta, ci), cos_t));
    Cpx r_perp = cdiv(csub(ci, cmul(eta, cos_t)), cadd(ci, cmul(eta, cos_t)));
    return (cnorm_sqr(r_parl) + cnorm_sqr(r_perp)) * .5f;
}
enum : int { FR_CONSTANT = 0, FR_CONDUCTOR, FR_DIELECTRIC, FR_SCHLICK, FR_F82 };
struct Fresnel {
    int kind;
    Spec a, b; // conductor eta,k | schlick F0 | F82 F0,B
    float eta;
#if VMK_HERO
    bool eta_sp; // FresnelDielectric over an "spd" ior (dispersive glass): a = the per-wavelength eta, eta = a.x (fresnel.h:83-91)
#endif
    VD Spec evaluate(float cos_t) const {
        if (kind == FR_CONDUCTOR) return smap2(a, b, [&](float e, float k) { return fresnel_complex(cos_t, e, k); });
#if VMK_HERO
        if (kind == FR_DIELECTRIC && eta_sp) return smap(a, [&](float e) { return fresnel_dielectric(cos_t, e); });
#endif
        if (kind == FR_DIELECTRIC) { float f = fresnel_dielectric(cos_t, eta); return mks(f); }
        if (kind == FR_SCHLICK) { // fresnel.h:60-67
            float F_real = fresnel_dielectric(cos_t, eta);
            float F0_real = schlick_F0_from_ior(eta);
            float t = clamp_(inverse_lerp(F_real, F0_real, 1.f), 0.f, 1.f);
            return lerp3(t, a, mks(1.f));
        }
        if (kind == FR_F82) { // fresnel.h:123-129
            float mu = saturate_(1.f - cos_t);
            float mu5 = pow5(mu);
            Spec f_schlick = lerp3(mu5, a, mks(1.f));
            return saturate3(f_schlick - b * cos_t * mu5 * mu);
        }
        return mks(1.f);
    }
};

// =====================================================================================================
// a11-a18. lobes + materials — base/scattering/{bxdf,lobe,material}.cpp, render_core/material/*.cpp
// The reference builds a polymorphic Lobe tree per hit; here every material expands to a short, fixed list of
// flat lobe records sharing the interaction's shading frame, evaluated by a switch on the lobe kind.
// =====================================================================================================
namespace flag {
constexpr uint32_t Unset = 1, Reflection = 2, Transmission = 4, Diffuse = 8, Glossy = 16;
constexpr uint32_t DiffRefl = Diffuse | Reflection, GlossyRefl = Glossy | Reflection, GlossyTrans = Glossy | Transmission;
}
struct ScatterEval { Spec f; float pdf; uint32_t flags; };
struct BSDFSample { ScatterEval eval; V3 wi; float eta; };
enum : int { LB_LAMBERT = 0, LB_OREN_NAYAR, LB_MICROFACET, LB_FRESNEL_BLEND, LB_DIELECTRIC, LB_SHEEN, LB_PLASTIC };
struct Lobe {
    int kind;
    Spec kr, rs;
    float A, B, ax, ay;
    Fresnel fr;
    bool compensate;
    float weight, sample_weight;
};

VD float dielectric_refl_prob(const Lobe &l, Spec F) { // lobe.cpp:315-319
    Spec T = 1.f - F;
    Spec total = T * l.kr + F;
    return average(F) / average(total);
}
// The albedo tables the lobe code reads.  The out-of-line lobe routine gets these three pointers instead of the scene view,
// so that no pointer to the view escapes the kernel: the view then lives in scalar registers, not in a per-lane scratch copy.
struct LobeLuts { const float *pure_reflection, *dielectric, *dielectric_inv; };
VD float dielectric_lut_x(const LobeLuts &S, const Lobe &l, V3 wo, float eta) { // lobe.cpp:263-285
    const float *lut = eta > 1.f ? S.dielectric : S.dielectric_inv;
    float x = sqrt_(sqrt_(l.ax * l.ay));
    float y = abs_cos_theta(wo);
    float z = eta > 1.f ? inverse_lerp(eta, 1.003f, 5.f) : inverse_lerp(rcp(eta), 1.003f, 5.f);
    float out[2]; sample_lut3d<2>(lut, mk3(x, y, z), out);
    return out[0];
}
VD Spec blend_f_specular(const Lobe &l, V3 wo, V3 wi, V3 wh) { // FresnelBlend::f_specular substrate.cpp:31-37
    Spec specular = lerp3(schlick_weight(dot(wi, wh)), l.rs, mks(1.f)) *
                  (bsdf_D(wh, l.ax, l.ay) / (4.f * abs_dot(wi, wh) * fmax_(abs_cos_theta(wi), abs_cos_theta(wo))));
    return specular * (is_zero(wh) ? 0.f : 1.f);
}

// Lobe::evaluate_local_impl of every lobe class (local frame, before the |cos_i| factor)
VD ScatterEval eval_local(const LobeLuts &S, const Lobe &l, V3 wo, V3 wi, float *eta_out) {
    ScatterEval se; se.f = mks(0.f); se.pdf = 0.f; se.flags = flag::Unset;
    switch (l.kind) {
        case LB_LAMBERT: case LB_OREN_NAYAR: { // bxdf.cpp:34-46, bxdf.h:92-95, bxdf.cpp:103-121
            bool sh = same_hemisphere(wo, wi);
            Spec f;
            if (l.kind == LB_LAMBERT) f = l.kr * InvPi;
            else {
                float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
                float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
                float max_cos = fmax_(0.f, d_cos);
                bool cond = abs_cos_theta(wi) > abs_cos_theta(wo);
                float sin_alpha = cond ? sin_o : sin_i;
                float tan_beta = cond ? sin_i / abs_cos_theta(wi) : sin_o / abs_cos_theta(wo);
                f = l.kr * InvPi * (l.A + l.B * max_cos * sin_alpha * tan_beta);
            }
            se.f = sh ? f : mks(0.f);
            se.pdf = sh ? cosine_hemisphere_PDF(abs_cos_theta(wi)) : 0.f;
            se.flags = flag::DiffRefl;
            break;
        }
        case LB_MICROFACET: { // lobe.cpp:213-218, bxdf.cpp:65-78, lobe.cpp:716-729
            bool sh = same_hemisphere(wo, wi);
            V3 wh = normalize(wo + wi);
            V3 whf = face_forward(wh, mk3(0, 0, 1));
            Spec F = l.fr.evaluate(abs_dot(wo, whf));
            Spec f = (F * BRDF_div_fr(wo, whf, wi, l.ax, l.ay)) * l.kr;
            float pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay);
            se.f = sh ? f : mks(0.f);
            se.pdf = sh ? pdf : 0.f;
            se.flags = flag::GlossyRefl;
            if (l.compensate) {
                float alpha = sqrt_(l.ax * l.ay);
                float v; sample_lut2d<1>(S.pure_reflection, alpha, cos_theta(wo), &v);
                se.f *= 1.f / v;
            }
            break;
        }
        case LB_FRESNEL_BLEND: { // substrate.cpp:12-70
            bool sh = same_hemisphere(wo, wi);
            V3 wh = normalize(wi + wo);
            Spec specular = blend_f_specular(l, wo, wi, wh);
            Spec diffuse = (28.f / (23.f * Pi)) * l.kr * (mks(1.f) - l.rs) *
                         (1.f - pow5(1.f - .5f * abs_cos_theta(wi))) * (1.f - pow5(1.f - .5f * abs_cos_theta(wo)));
            Spec f = specular + diffuse;
            float fr = l.fr.evaluate(abs_cos_theta(wo)).x;
            float pdf = lerp_(fr, cosine_hemisphere_PDF(abs_cos_theta(wi)), PDF_wi_reflection(wo, wh, l.ax, l.ay));
            se.f = sh ? f : mks(0.f);
            se.pdf = sh ? pdf : 0.f;
            se.flags = flag::Reflection;
            break;
        }
        case LB_PLASTIC: { // PlasticLobe::evaluate_local_impl plastic.cpp:31-43 (no hemisphere test of its own)
            V3 wh = normalize(wo + wi);
            Spec F = l.fr.evaluate(abs_dot(wh, wo));
            se.f = (l.kr * InvPi) * (1.f - F);
            se.f += BRDF_div_fr(wo, wh, wi, l.ax, l.ay) * F;
            se.pdf = lerp_(average(F), cosine_hemisphere_PDF(abs_cos_theta(wi)), PDF_wi_reflection(wo, wh, l.ax, l.ay));
            se.flags = flag::GlossyRefl;
            break;
        }
        case LB_DIELECTRIC: { // lobe.cpp:321-412
            bool refl = same_hemisphere(wo, wi);
            float eta = l.fr.eta;
            float eta_p = refl ? 1.f : eta;
            if (eta_out) *eta_out = eta_p;
            V3 wh = normalize(wo + wi * eta_p);
            wh = face_forward(wh, wo);
            Spec F = l.fr.evaluate(abs_dot(wh, wo));
            float lutx = dielectric_lut_x(S, l, wo, eta);
            if (refl) {
                se.f = F * BRDF_div_fr(wo, wh, wi, l.ax, l.ay);
                se.pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay) * dielectric_refl_prob(l, F);
                se.flags = flag::GlossyRefl;
                se.f *= rcp(lutx);
            } else {
                V3 new_wh = face_forward(wh, wo);
                V3 wh2 = normalize(wo + wi * eta);
                Spec tr = (1.f - F) * BTDF_div_ft(wo, wh2, wi, eta, l.ax, l.ay, true);
                se.f = tr * l.kr;
                se.pdf = PDF_wi_transmission(wo, new_wh, wi, eta, l.ax, l.ay) * (1.f - dielectric_refl_prob(l, F));
                se.flags = flag::GlossyTrans;
                se.f *= rcp(lutx);
            }
            break;
        }
        default: { // LB_SHEEN principled_bsdf.cpp:58-72,110-117
            float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
            V3 w = mk3(l.A * wi.x + l.B * wi.z, l.A * wi.y, wi.z);
            float len = length(w);
            w = w / len;
            float jacobian = sqr(l.A) / (len * len * len);
            float ltc = cosine_hemisphere_PDF(cos_theta(w)) * jacobian;
            se.f = l.kr * ltc / cos_i;
            se.pdf = ltc;
            if (cos_i < 0.f || cos_o < 0.f) se.f = mks(0.f);
            break;
        }
    }
    return se;
}

// sample_wi_local_impl of every lobe class
VD V3 sample_wi_local(const Lobe &l, V3 wo, Sampler &sampler, bool *valid) {
    V3 wi;
    *valid = true;
    switch (l.kind) {
        case LB_LAMBERT: case LB_OREN_NAYAR: { // bxdf.cpp:48-52
            wi = square_to_cosine_hemisphere(sampler.next_2d());
            wi.z = wo.z < 0.f ? -wi.z : wi.z;
            break;
        }
        case LB_MICROFACET: { // bxdf.cpp:80-84
            V3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            wi = reflect(wo, wh);
            *valid = same_hemisphere(wo, wi);
            break;
        }
        case LB_FRESNEL_BLEND: { // substrate.cpp:52-69
            V2 u = sampler.next_2d();
            float fr = l.fr.evaluate(abs_cos_theta(wo)).x;
            if (u.x < fr) {
                u.x = remapping(u.x, 0.f, fr);
                V3 wh = sample_wh(wo, u, l.ax, l.ay);
                wi = reflect(wo, wh);
            } else {
                u.x = remapping(u.x, fr, 1.f);
                wi = square_to_cosine_hemisphere(u);
                wi.z = wo.z < 0.f ? -wi.z : wi.z;
            }
            break;
        }
        case LB_PLASTIC: { // PlasticLobe::sample_wi_local_impl plastic.cpp:45-59: 2 + 1 draws, 2 more on the diffuse branch
            V3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            Spec F = l.fr.evaluate(abs_cos_theta(wo));
            float uc = sampler.next_1d();
            if (uc < average(F)) { wi = reflect(wo, wh); *valid = same_hemisphere(wo, wi); }
            else wi = square_to_cosine_hemisphere(sampler.next_2d());
            break;
        }
        case LB_DIELECTRIC: { // lobe.cpp:431-449
            V3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            float d = dot(wo, wh);
            Spec F = l.fr.evaluate(abs_(d));
            float uc = sampler.next_1d();
            if (uc < dielectric_refl_prob(l, F)) {
                wi = reflect(wo, wh);
                *valid = same_hemisphere(wo, wi);
            } else {
                bool v = refract(wo, wh, l.fr.eta, &wi);
                *valid = v && !same_hemisphere(wo, wi);
            }
            break;
        }
        default: { // LB_SHEEN principled_bsdf.cpp:100-108
            V3 w = square_to_cosine_hemisphere(sampler.next_2d());
            w = mk3(w.x / l.A - w.z * l.B / l.A, w.y / l.A, w.z);
            wi = normalize(w);
            break;
        }
    }
    return wi;
}

// Out-of-line instances of the two big lobe routines.  Inlined into the 2-pass evaluate/sample loop they cost ~170 live
// VGPRs (LICM hoists every wo-only term of every lobe kind out of the loop); as real calls the lobe code is compiled
// once at <= 71 VGPRs and the megakernel fits 128 VGPRs (4 waves/SIMD) with far fewer spills.
// The lobe itself crosses the call through LDS: the wave's traversal scratch (dbvh.h WaveScratch) is idle while a vertex is
// shaded, so the caller parks the 20 dwords of the lobe there ([field][lane], one ds_write each) and the callee reads them
// with ds_read — instead of 20 scratch stores in the caller and 20 flat loads (64-bit address arithmetic, HBM-backed,
// counted on vmcnt and lgkmcnt) at the top of the callee, three times per vertex.  `lds` is the lane's byte offset in LDS.
constexpr uint32_t kLobeLdsDwords = 20 + 4 * (kSpecDim - 3u); // x 64 lanes x 4 B = 5 KiB (6 KiB with four wavelengths) <= sizeof(WaveScratch)
#define VMK_AS3 __attribute__((address_space(3)))
VD void stage_lobe(uint32_t lds, const Lobe &l) {
    VMK_AS3 uint32_t *p = reinterpret_cast<VMK_AS3 uint32_t *>(lds);
    uint32_t bits = l.compensate ? 1u : 0u;
#if VMK_HERO
    bits |= l.fr.eta_sp ? 2u : 0u;
#endif
    const uint32_t w[kLobeLdsDwords] = {(uint32_t) l.kind, f2u(l.kr.x), f2u(l.kr.y), f2u(l.kr.z), f2u(l.rs.x), f2u(l.rs.y), f2u(l.rs.z), f2u(l.A), f2u(l.B), f2u(l.ax), f2u(l.ay),
                                        (uint32_t) l.fr.kind, f2u(l.fr.a.x), f2u(l.fr.a.y), f2u(l.fr.a.z), f2u(l.fr.b.x), f2u(l.fr.b.y), f2u(l.fr.b.z), f2u(l.fr.eta), bits
#if VMK_SPEC_DIM == 4
                                        , f2u(l.kr.w), f2u(l.rs.w), f2u(l.fr.a.w), f2u(l.fr.b.w)
#endif
    };
#pragma unroll
    for (uint32_t k = 0; k < kLobeLdsDwords; ++k) p[k * 64u] = w[k];
}
VD Lobe staged_lobe(uint32_t lds) {
    const VMK_AS3 uint32_t *p = reinterpret_cast<const VMK_AS3 uint32_t *>(lds);
    uint32_t w[kLobeLdsDwords];
#pragma unroll
    for (uint32_t k = 0; k < kLobeLdsDwords; ++k) w[k] = p[k * 64u];
    Lobe l;
#if VMK_SPEC_DIM == 4
    l.kind = (int) w[0]; l.kr = {u2f(w[1]), u2f(w[2]), u2f(w[3]), u2f(w[20])}; l.rs = {u2f(w[4]), u2f(w[5]), u2f(w[6]), u2f(w[21])};
    l.A = u2f(w[7]); l.B = u2f(w[8]); l.ax = u2f(w[9]); l.ay = u2f(w[10]);
    l.fr.kind = (int) w[11]; l.fr.a = {u2f(w[12]), u2f(w[13]), u2f(w[14]), u2f(w[22])}; l.fr.b = {u2f(w[15]), u2f(w[16]), u2f(w[17]), u2f(w[23])}; l.fr.eta = u2f(w[18]);
#else
    l.kind = (int) w[0]; l.kr = mk3(u2f(w[1]), u2f(w[2]), u2f(w[3])); l.rs = mk3(u2f(w[4]), u2f(w[5]), u2f(w[6]));
    l.A = u2f(w[7]); l.B = u2f(w[8]); l.ax = u2f(w[9]); l.ay = u2f(w[10]);
    l.fr.kind = (int) w[11]; l.fr.a = mk3(u2f(w[12]), u2f(w[13]), u2f(w[14])); l.fr.b = mk3(u2f(w[15]), u2f(w[16]), u2f(w[17])); l.fr.eta = u2f(w[18]);
#endif
    l.compensate = (w[19] & 1u) != 0u; l.weight = 1.f; l.sample_weight = 1.f; // (the weights are the caller's business)
#if VMK_HERO
    l.fr.eta_sp = (w[19] & 2u) != 0u;
#endif
    return l;
}
// Results come back BY VALUE (6 — 7 with four wavelengths — and 5 dwords: VGPR returns under the AMDGPU calling convention), the sampler state goes
// in and out by value too: no pointer into the caller's private frame crosses the call, so nothing of the caller's is
// forced into scratch and no flat access to the stack aperture exists in these routines.
struct EvalRet { ScatterEval se; float eta; };
struct SampleRet { V3 wi; uint32_t sampler_state; uint32_t valid; };
__device__ __noinline__ EvalRet eval_local_ool(const float *lut_pure_reflection, const float *lut_dielectric, const float *lut_dielectric_inv, uint32_t lobe_lds,
                                               float wox, float woy, float woz, float wix, float wiy, float wiz) {
#ifdef VMK_OOL_BARRIER
    asm volatile("" ::: "memory");
#endif
    EvalRet r;
    r.eta = 0.f; // 0 = "not a dielectric lobe: leave the caller's eta alone" (a dielectric writes eta' > 0, lobe.cpp:333)
    r.se = eval_local(LobeLuts{lut_pure_reflection, lut_dielectric, lut_dielectric_inv}, staged_lobe(lobe_lds), mk3(wox, woy, woz), mk3(wix, wiy, wiz), &r.eta);
    return r;
}
__device__ __noinline__ SampleRet sample_wi_local_ool(uint32_t lobe_lds, float wox, float woy, float woz, uint32_t sampler_state) {
#ifdef VMK_OOL_BARRIER
    asm volatile("" ::: "memory");
#endif
    Sampler s; s.state = sampler_state;
    bool valid;
    SampleRet r;
    r.wi = sample_wi_local(staged_lobe(lobe_lds), mk3(wox, woy, woz), s, &valid);
    r.sampler_state = s.state; r.valid = valid ? 1u : 0u;
    return r;
}
// (the lobe must have been parked with stage_lobe(lds, .) first)
VD ScatterEval eval_local_call(const DScene &S, uint32_t lds, V3 wo, V3 wi, float *eta) {
    EvalRet r = eval_local_ool(S.lut_pure_reflection, S.lut_dielectric, S.lut_dielectric_inv, lds, wo.x, wo.y, wo.z, wi.x, wi.y, wi.z);
    if (eta && r.eta != 0.f) *eta = r.eta;
    return r.se;
}
VD V3 sample_wi_local_call(uint32_t lds, V3 wo, Sampler &sampler, bool *valid) {
    SampleRet r = sample_wi_local_ool(lds, wo.x, wo.y, wo.z, sampler.state);
    sampler.state = r.sampler_state; *valid = r.valid != 0u;
    return r.wi;
}

// Material::compute_shading_frame (material.cpp:331-353): the frame every lobe of the material evaluates in.  Without a "normal"
// slot it is the interaction's; with one, the rotation that takes (0,0,1) to the slot's value (used as it comes) is applied to
// the world shading normal, the result is clamped against the geometric normal (detail::clamp_ns :305-310) and the tangents are
// re-derived (PartialDerivative::update(n, s), interaction.h:106-112).  Quaternion::from_axis_angle / to_float3x3 are ocarina's:
// restated as Rodrigues' rotation about normalize(axis), like the oracle (parity unpinned).
// Out of line and by value: inlined into path_bounce its acos / sincos / normalisations cost the megakernel 38 % on scenes that
// have no normal map at all (register pressure around the shading code), as a call they cost those scenes one flag test.
struct FrameRet { V3 x, y, z; };
__device__ __noinline__ FrameRet normal_mapped_frame_ool(const vmk_texture *textures, const uint8_t *tex_data, const float *srgb_lut,
                                                         float s0, float s1, float s2, uint32_t stex, float uvx, float uvy,
                                                         float xx, float xy, float xz, float zx, float zy, float zz,
                                                         float gx, float gy, float gz, float wx, float wy, float wz) {
    V3 normal = {s0, s1, s2};
    if (stex != VMK_INVALID) { // ShaderNodeSlot::evaluate (eval_slot3) against the three texture tables
        float4 t = sample_image_ool(textures, tex_data, srgb_lut, stex & 0xffffu, uvx, uvy);
        float c[4] = {t.x * s0, t.y * s0, t.z * s0, t.w * s0};
        uint32_t sw = stex >> 16;
        auto pick = [&](uint32_t k) { return k == 0 ? c[0] : (k == 1 ? c[1] : (k == 2 ? c[2] : c[3])); };
        normal = {pick(sw & 3u), pick((sw >> 2) & 3u), pick((sw >> 4) & 3u)};
    }
    const V3 sx = {xx, xy, xz}, sz = {zx, zy, zz}, ng = {gx, gy, gz}, wo = {wx, wy, wz};
    V3 n = mk3(0.f, 0.f, 1.f);
    V3 axis = cross(n, normal);
    float theta = acos_(clamp_(dot(n, normal), -1.f, 1.f));
    V3 world_normal = sz;
    float len = length(axis);
    if (len > 0.f) {
        V3 k = axis / len;
        float st, ct; sincos_(theta, &st, &ct);
        world_normal = sz * ct + cross(k, sz) * st + k * (dot(k, sz) * (1.f - ct));
    }
    world_normal = normalize(world_normal);
    { // clamp_ns(ns, ng, w = wo)
        V3 w_refl = reflect(wo, world_normal);
        V3 w_refl_clip = same_hemisphere(wo, w_refl, ng) ? w_refl : normalize(w_refl - ng * dot(w_refl, ng));
        world_normal = normalize(w_refl_clip + wo);
    }
    world_normal = normalize(face_forward(world_normal, sz));
    V3 ss = normalize(sx - world_normal * dot(world_normal, sx));
    V3 tt = normalize(cross(world_normal, ss));
    return FrameRet{ss, tt, world_normal};
}
VD Frame compute_shading_frame(const DScene &S, const vmk_material *m, const Interaction &it, DCounters &cnt) {
    if (!(m->flags & VMK_MATF_HAS_NORMAL)) return it.shading;
    if (m->normal.tex != VMK_INVALID) cnt.tex++;
    FrameRet r = normal_mapped_frame_ool(S.textures, S.tex_data, S.srgb_lut, m->normal.v[0], m->normal.v[1], m->normal.v[2], m->normal.tex, it.uv.x, it.uv.y,
                                         it.shading.x.x, it.shading.x.y, it.shading.x.z, it.shading.z.x, it.shading.z.y, it.shading.z.z,
                                         it.ng.x, it.ng.y, it.ng.z, it.wo.x, it.wo.y, it.wo.z);
    return Frame{r.x, r.y, r.z};
}

VD void microfacet_alpha(const DScene &S, const vmk_material *m, int slot_r, int slot_a, V2 uv, float rmin, float *ax, float *ay, DCounters &cnt) {
    float roughness = clamp_(eval_slot1(S, m->slot[slot_r], uv, cnt), rmin, 1.f);
    float anisotropic = clamp_(eval_slot1(S, m->slot[slot_a], uv, cnt), -0.9f, 0.9f);
    roughness = (m->flags & VMK_MATF_REMAP_ROUGHNESS) ? sqr(roughness) : roughness;
    V2 a = calculate_alpha(roughness, anisotropic);
    *ax = a.x; *ay = a.y;
}
VD void lobe_defaults(Lobe &l) {
    l.kind = LB_LAMBERT; l.kr = mks(1.f); l.rs = mks(0.f); l.A = 0.f; l.B = 0.f; l.ax = 0.f; l.ay = 0.f;
    l.fr.kind = FR_CONSTANT; l.fr.a = mks(1.f); l.fr.b = mks(0.f); l.fr.eta = 1.f;
#if VMK_HERO
    l.fr.eta_sp = false;
#endif
    l.compensate = false; l.weight = 1.f; l.sample_weight = 1.f;
}
// create_lobe_set of the single-lobe material plugins
VD void build_simple_lobe(const DScene &S, const vmk_material *m, const Interaction &it, Lobe &l, DCounters &cnt SWL_P) {
    lobe_defaults(l);
    switch (m->type) {
        case VMK_MAT_DIFFUSE: { // diffuse.cpp:21-30, bxdf.cpp:94-101
            l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A);
            if (m->flags & VMK_MATF_HAS_SIGMA) {
                float sigma = eval_slot1(S, m->slot[1], it.uv, cnt);
                sigma = sigma * PiOver2;
                float sigma2 = sqr(sigma * sigma);
                l.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                l.B = 0.45f * sigma2 / (sigma2 + 0.09f);
                l.kind = LB_OREN_NAYAR;
            }
            break;
        }
        case VMK_MAT_MIRROR: { // mirror.cpp:60-74
            l.kind = LB_MICROFACET; l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A);
            microfacet_alpha(S, m, 1, 2, it.uv, 0.0001f, &l.ax, &l.ay, cnt);
            l.compensate = true;
            break;
        }
        case VMK_MAT_METAL: { // metal.cpp:137-156
            l.kind = LB_MICROFACET;
            microfacet_alpha(S, m, 2, 3, it.uv, 0.0001f, &l.ax, &l.ay, cnt);
            l.fr.kind = FR_CONDUCTOR; l.fr.a = eval_slot_spd(S, m->slot[0], it.uv, cnt SWL_A); l.fr.b = eval_slot_spd(S, m->slot[1], it.uv, cnt SWL_A);
            l.compensate = true;
            break;
        }
        case VMK_MAT_PLASTIC: { // plastic.cpp:103-122 (same double roughness_to_alpha as substrate)
            l.kind = LB_PLASTIC;
            l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A);
            Spec Rs = eval_slot_albedo(S, m->slot[1], it.uv, cnt SWL_A);
            float ior = eval_slot1(S, m->slot[2], it.uv, cnt);
            float ax, ay; microfacet_alpha(S, m, 3, 4, it.uv, 0.0001f, &ax, &ay, cnt);
            if (m->flags & VMK_MATF_REMAP_ROUGHNESS) { ax = sqr(ax); ay = sqr(ay); }
            l.ax = clamp_(ax, 0.0001f, 1.f); l.ay = clamp_(ay, 0.0001f, 1.f);
            l.fr.kind = FR_SCHLICK; l.fr.a = schlick_F0_from_ior(ior) * Rs; l.fr.eta = ior;
            break;
        }
        case VMK_MAT_METALLIC: { // metallic.cpp:42-60: MetallicLobe = PureReflectionLobe with compensation, F82-tint Fresnel
            l.kind = LB_MICROFACET; l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A);
            microfacet_alpha(S, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay, cnt);
            Spec edge_tint = eval_slot_albedo(S, m->slot[1], it.uv, cnt SWL_A);
            const float f = 6.f / 7.f;
            const float f5 = pow5(f);
            Spec f_schlick = lerp3(f5, l.kr, mks(1.f)); // FresnelF82Tint::init_from_F82 fresnel.h:115-121
            l.fr.kind = FR_F82; l.fr.a = l.kr; l.fr.b = f_schlick * (7.f / (f5 * f)) * (mks(1.f) - edge_tint);
            l.compensate = true;
            break;
        }
        case VMK_MAT_GLASS: { // glass.cpp:240-257, interaction.cpp:80-83
            l.kind = LB_DIELECTRIC; l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A);
            float cos_t = dot(it.wo, it.ng);
#if VMK_HERO
            if (m->slot[1].tex == VMK_SLOT_SPD) { // dispersive: one ior per wavelength, directions follow the hero wavelength (eta[0])
                Spec iors = eval_slot_spd(S, m->slot[1], it.uv, cnt SWL_A);
                iors = cos_t > 0.f ? iors : smap(iors, [](float v) { return rcp(v); });
                microfacet_alpha(S, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay, cnt);
                l.fr.kind = FR_DIELECTRIC; l.fr.eta = iors.x; l.fr.a = iors; l.fr.eta_sp = true;
                break;
            }
#endif
            float ior = eval_slot1(S, m->slot[1], it.uv, cnt);
            ior = cos_t > 0.f ? ior : rcp(ior);
            microfacet_alpha(S, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay, cnt);
            l.fr.kind = FR_DIELECTRIC; l.fr.eta = ior;
            break;
        }
        default: { // VMK_MAT_SUBSTRATE substrate.cpp:126-149
            l.kind = LB_FRESNEL_BLEND;
            l.kr = eval_slot_albedo(S, m->slot[0], it.uv, cnt SWL_A); l.rs = eval_slot_albedo(S, m->slot[1], it.uv, cnt SWL_A);
            float ax, ay; microfacet_alpha(S, m, 2, 3, it.uv, 0.0001f, &ax, &ay, cnt);
            if (m->flags & VMK_MATF_REMAP_ROUGHNESS) { ax = sqr(ax); ay = sqr(ay); }
            l.ax = clamp_(ax, 0.0001f, 1.f); l.ay = clamp_(ay, 0.0001f, 1.f);
            l.fr.kind = FR_DIELECTRIC; l.fr.eta = 1.5f;
            break;
        }
    }
}

VD Spec layering_weight(Spec layer_albedo, Spec weight) { // principled_bsdf.cpp:209-214
    Spec tmp = smap2(layer_albedo, weight, [](float a, float w) { return w == 0.f ? 0.f : a / w; });
    return weight * saturate_(1.f - max_comp(tmp));
}
// Per-hit material context: what create_lobe_set computes once.  Principled keeps only the per-lobe colours
// and sampling weights (PrincipledBSDF::create_lobe_set principled_bsdf.cpp:352-461); lobes are re-expanded
// on demand by mat_lobe().
struct MatCtx {
    const vmk_material *m;
    int n;        // lobes
    bool is_set;  // LobeSet semantics (weights, valid_world_factor, 3 burnt draws)
    Lobe single;  // simple materials: the lobe itself
    uint32_t lobe_lds; // this lane's slot in the wave's idle traversal scratch (byte offset in LDS), set by the caller of mat_prepare
    float mixw[2], mixsw[2]; // mix / add: lobe weights and sampling weights of the two children
    const vmk_material *pm;  // the principled_bsdf material the principled fields below describe: m itself, or the ONE principled child
                             // of a mix / add whose lobes LobeSet::flatten (lobe.cpp:534-562) merges into the parent's list
    int pchild;              // mix / add: which child (0 / 1) is the principled one, -1: both are single-lobe
    // principled
    int first;    // 0 with sheen, 1 without
    Spec color, spec_tint, kr_sheen, kr_coat, kr_metal, kr_spec, kr_diff;
    float sheen_a, sheen_b, ax, ay, cc_alpha, cc_ior, ior, eta, w_trans;
    Spec f82_b, f0_spec, f0_trans;
    float sw[6];
};
// FULL = the scene contains mix / principled_bsdf materials.  Like the reference, which JIT-compiles only the material
// types a scene uses, the megakernel exists in two ahead-of-time variants; the single-lobe variant carries no lobe-set
// state at all (lower VGPR pressure, no lobe loop).
template<bool FULL>
VD void mat_prepare(const DScene &S, const vmk_material *m, const Interaction &it, MatCtx &mc, DCounters &cnt SWL_P) {
    mc.m = m;
    if constexpr (!FULL) {
        mc.n = 1; mc.is_set = false;
        build_simple_lobe(S, m, it, mc.single, cnt SWL_A);
        return;
    }
    mc.pm = m; mc.pchild = -1;
    if (m->type == VMK_MAT_MIX || m->type == VMK_MAT_ADD) {
        if (m->type == VMK_MAT_MIX) { // mix.cpp:66-71 + LobeSet::create_mix (lobe.cpp:495-508)
            float frac = eval_slot1(S, m->slot[0], it.uv, cnt);
            mc.mixw[0] = 1.f - frac; mc.mixw[1] = frac;
            mc.mixsw[0] = 1.f - frac; mc.mixsw[1] = frac;
        } else { // add.cpp:57-60 + LobeSet::create_add (lobe.cpp:510-522): weights {1, 1}, sampling weights normalised
            mc.mixw[0] = 1.f; mc.mixw[1] = 1.f;
            mc.mixsw[0] = 1.f / (1.f + 1.f); mc.mixsw[1] = 1.f / (1.f + 1.f);
        }
        mc.n = 2; mc.is_set = true;
        const vmk_material *c0 = S.materials + m->child0, *c1 = S.materials + m->child1;
        if (c0->type != VMK_MAT_PRINCIPLED && c1->type != VMK_MAT_PRINCIPLED) return;
        mc.pchild = c0->type == VMK_MAT_PRINCIPLED ? 0 : 1; // (the host admits one principled child at most)
        mc.pm = mc.pchild == 0 ? c0 : c1;
        m = mc.pm; // fall through: prepare the principled child; its lobes are flattened into this set by mat_lobe
    } else if (m->type != VMK_MAT_PRINCIPLED) {
        mc.n = 1; mc.is_set = false;
        build_simple_lobe(S, m, it, mc.single, cnt SWL_A);
        return;
    }
    mc.is_set = true;
    V2 uv = it.uv;
    mc.color = eval_slot_albedo(S, m->slot[VMK_P_COLOR], uv, cnt SWL_A);
    mc.ior = eval_slot1(S, m->slot[VMK_P_IOR], uv, cnt);
    float roughness = clamp_(eval_slot1(S, m->slot[VMK_P_ROUGHNESS], uv, cnt), 0.0001f, 1.f);
    float anisotropic = eval_slot1(S, m->slot[VMK_P_ANISOTROPIC], uv, cnt);
    mc.spec_tint = eval_slot_albedo(S, m->slot[VMK_P_SPEC_TINT], uv, cnt SWL_A);
    float aspect = sqrt_(1.f - anisotropic * 0.9f);
    mc.ax = fmax_(0.001f, sqr(roughness) / aspect); mc.ay = fmax_(0.001f, sqr(roughness) * aspect);
    Spec weight = mks(1.f);
    float cos_t = dot(it.wo, it.ng);
    float front_factor = cos_t > 0.f ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) mc.sw[i] = 0.f;
    mc.first = 1;
    if (S.lut_sheen_approx) { // sheen (SheenLTC ctor principled_bsdf.cpp:37-45)
        mc.first = 0;
        Spec sheen_tint = eval_slot_albedo(S, m->slot[VMK_P_SHEEN_TINT], uv, cnt SWL_A);
        float sheen_weight = eval_slot1(S, m->slot[VMK_P_SHEEN_WEIGHT], uv, cnt) * front_factor;
        float sheen_roughness = eval_slot1(S, m->slot[VMK_P_SHEEN_ROUGHNESS], uv, cnt);
        float c[4]; sample_lut2d<4>(S.lut_sheen_approx, cos_t, sheen_roughness, c);
        mc.sheen_a = c[0]; mc.sheen_b = c[1];
        mc.kr_sheen = (sheen_tint * sheen_weight * weight) * c[2];
        mc.sw[0] = average(mc.kr_sheen);
        weight = layering_weight(mc.kr_sheen, weight);
    }
    { // coat
        float cc_weight = eval_slot1(S, m->slot[VMK_P_COAT_WEIGHT], uv, cnt) * front_factor;
        float cc_roughness = clamp_(eval_slot1(S, m->slot[VMK_P_COAT_ROUGHNESS], uv, cnt), 0.0001f, 1.f);
        cc_roughness = sqr(cc_roughness);
        mc.cc_alpha = cc_roughness;
        mc.cc_ior = eval_slot1(S, m->slot[VMK_P_COAT_IOR], uv, cnt);
        Spec cc_tint = eval_slot_albedo(S, m->slot[VMK_P_COAT_TINT], uv, cnt SWL_A);
        mc.kr_coat = (weight * cc_weight) * cc_tint;
        float x = sqrt_(sqrt_(cc_roughness * cc_roughness));
        float z = inverse_lerp(mc.cc_ior, 1.003f, 4.f);
        float sv; sample_lut3d<1>(S.lut_coat, mk3(x, cos_t, z), &sv);
        Spec albedo = mc.kr_coat * sv;
        mc.sw[1] = average(albedo);
        weight = layering_weight(albedo, weight);
    }
    { // metallic (FresnelF82Tint::init_from_F82 fresnel.h:115-121)
        float metallic = eval_slot1(S, m->slot[VMK_P_METALLIC], uv, cnt) * front_factor;
        const float f = 6.f / 7.f;
        const float f5 = pow5(f);
        Spec f_schlick = lerp3(f5, mc.color, mks(1.f));
        mc.f82_b = f_schlick * (7.f / (f5 * f)) * (mks(1.f) - mc.spec_tint);
        mc.kr_metal = weight * metallic;
        mc.sw[2] = metallic * average(weight);
        weight *= (1.0f - metallic);
    }
    { // transmission
        float trans_weight = eval_slot1(S, m->slot[VMK_P_TRANS_WEIGHT], uv, cnt);
        mc.eta = cos_t > 0.f ? mc.ior : rcp(mc.ior);
        Spec t_weight = weight * trans_weight;
        mc.f0_trans = mc.spec_tint * schlick_F0_from_ior(mc.eta);
        mc.w_trans = average(t_weight);
        mc.sw[3] = mc.w_trans;
        weight *= (1.0f - trans_weight);
    }
    { // specular
        float f0 = schlick_F0_from_ior(mc.ior);
        mc.f0_spec = mc.spec_tint * f0;
        mc.kr_spec = weight;
        float x = sqrt_(sqrt_(mc.ax * mc.ay));
        float z = sqrt_(abs_((mc.ior - 1.0f) / (mc.ior + 1.0f)));
        float sv; sample_lut3d<1>(S.lut_specular, mk3(x, cos_t, z), &sv);
        Spec albedo = lerp3(sv, mc.f0_spec, mks(1.f)) * mc.kr_spec;
        mc.sw[4] = average(albedo);
        weight = layering_weight(albedo, weight);
    }
    { // diffuse
        mc.kr_diff = mc.color * weight * front_factor;
        mc.sw[5] = average(mc.kr_diff);
    }
    float weight_sum = 0.f; // LobeSet::normalize_sampled_weight lobe.cpp:524-532
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i >= mc.first) weight_sum += mc.sw[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i >= mc.first) mc.sw[i] = mc.sw[i] / weight_sum;
    mc.n = 6 - mc.first;
    if (mc.pchild >= 0) mc.n += 1; // + the single-lobe sibling
}
// expand lobe `i` (0-based within the material's lobe list)
template<bool FULL>
VD void mat_lobe(const DScene &S, const MatCtx &mc, const Interaction &it, int i, Lobe &l, DCounters &cnt SWL_P) {
    if constexpr (!FULL) { l = mc.single; return; }
    const vmk_material *m = mc.m;
    float parent = 1.f; // LobeSet::flatten: a principled child's sub-lobes take the parent's SAMPLING weight on both of their weights
    if (m->type == VMK_MAT_MIX || m->type == VMK_MAT_ADD) {
        const int np = mc.pchild >= 0 ? mc.n - 1 : 0; // lobes of the principled child in the flat list
        const int child = mc.pchild < 0 ? i : (mc.pchild == 0 ? (i < np ? 0 : 1) : (i == 0 ? 0 : 1));
        if (child != mc.pchild) {
            build_simple_lobe(S, S.materials + (child == 0 ? m->child0 : m->child1), it, l, cnt SWL_A);
            l.weight = child == 0 ? mc.mixw[0] : mc.mixw[1]; l.sample_weight = child == 0 ? mc.mixsw[0] : mc.mixsw[1];
            return;
        }
        parent = child == 0 ? mc.mixsw[0] : mc.mixsw[1];
        i = mc.pchild == 0 ? i : i - 1;
    } else if (m->type != VMK_MAT_PRINCIPLED) { l = mc.single; return; }
    lobe_defaults(l);
    int k = i + mc.first;
    l.sample_weight = k == 0 ? mc.sw[0] : k == 1 ? mc.sw[1] : k == 2 ? mc.sw[2] : k == 3 ? mc.sw[3] : k == 4 ? mc.sw[4] : mc.sw[5];
    switch (k) {
        case 0: l.kind = LB_SHEEN; l.A = mc.sheen_a; l.B = mc.sheen_b; l.kr = mc.kr_sheen; break;
        case 1: l.kind = LB_MICROFACET; l.ax = l.ay = mc.cc_alpha; l.fr.kind = FR_DIELECTRIC; l.fr.eta = mc.cc_ior; l.kr = mc.kr_coat; break;
        case 2: l.kind = LB_MICROFACET; l.ax = mc.ax; l.ay = mc.ay; l.fr.kind = FR_F82; l.fr.a = mc.color; l.fr.b = mc.f82_b; l.kr = mc.kr_metal; l.compensate = true; break;
        case 3: l.kind = LB_DIELECTRIC; l.ax = mc.ax; l.ay = mc.ay; l.fr.kind = FR_SCHLICK; l.fr.a = mc.f0_trans; l.fr.eta = mc.eta; l.kr = mc.color; l.weight = mc.w_trans; break;
        case 4: l.kind = LB_MICROFACET; l.ax = mc.ax; l.ay = mc.ay; l.fr.kind = FR_SCHLICK; l.fr.a = mc.f0_spec; l.fr.eta = mc.ior; l.kr = mc.kr_spec; break;
        default: l.kind = LB_LAMBERT; l.kr = mc.kr_diff; break;
    }
    if (mc.pchild >= 0) { l.weight *= parent; l.sample_weight *= parent; }
}
// MaterialEvaluator::albedo (material.cpp:91-98) = LobeSet::albedo (lobe.cpp:564-570) over the per-class Lobe::albedo
// (bxdf.h:91,153; substrate.cpp:22; lobe.cpp:208-210,308-313; CoatLobe / SpecularLobe principled_bsdf.cpp:154-160,198-205);
// used by the AOV pass only (frame_buffer.cpp:192-196: `linear_srgb(bsdf.albedo(wo), swl)`), always instantiated with FULL = true.
VD Spec lobe_albedo(const DScene &S, const MatCtx &mc, int k, bool principled_lobe, const Lobe &l, float cos_theta) {
    switch (l.kind) {
        case LB_MICROFACET: {
            if (mc.is_set && principled_lobe && (k == 1 || k == 4)) {
                float x = sqrt_(sqrt_(l.ax * l.ay));
                if (k == 1) { float sv; sample_lut3d<1>(S.lut_coat, mk3(x, cos_theta, inverse_lerp(mc.cc_ior, 1.003f, 4.f)), &sv); return sv * l.kr; }
                float z = sqrt_(abs_((mc.ior - 1.0f) / (mc.ior + 1.0f)));
                float sv; sample_lut3d<1>(S.lut_specular, mk3(x, cos_theta, z), &sv);
                return lerp3(sv, l.fr.a, mks(1.f)) * l.kr;
            }
            return l.kr * l.fr.evaluate(cos_theta);
        }
        case LB_DIELECTRIC: { Spec F = l.fr.evaluate(abs_(cos_theta)); return l.kr * (1.f - F) + F; }
        case LB_PLASTIC: return l.fr.evaluate(cos_theta); // MicrofacetLobe::albedo with the specular bxdf's kr = 1 (plastic.cpp:119)
        default: return l.kr; // Lambert / Oren-Nayar / FresnelBlend Rd / sheen (albedo folded into kr at build time)
    }
}
VD Spec mat_albedo(const DScene &S, const MatCtx &mc, const Interaction &it, DCounters &cnt SWL_P) {
    float cos_theta = dot(it.shading.z, it.wo);
    Lobe l;
    if (!mc.is_set) { mat_lobe<true>(S, mc, it, 0, l, cnt SWL_A); return lobe_albedo(S, mc, 0, false, l, cos_theta); }
    Spec sum = mks(0.f);
    for (int i = 0; i < mc.n; ++i) {
        mat_lobe<true>(S, mc, it, i, l, cnt SWL_A);
        bool pl = mc.m->type == VMK_MAT_PRINCIPLED; int ip = i;
        if (mc.pchild >= 0) { const int np = mc.n - 1; pl = mc.pchild == 0 ? i < np : i > 0; ip = mc.pchild == 0 ? i : i - 1; }
        sum += lobe_albedo(S, mc, pl ? ip + mc.first : 0, pl, l, cos_theta) * l.weight;
    }
    return sum;
}
// Lobe::evaluate / LobeSet::evaluate_impl (lobe.cpp:53-75,673-688) in world space; all lobes share it.shading.
// A single-lobe material is the n = 1 case of the same loop (its weights are 1, so the products are exact).
template<bool FULL>
VD ScatterEval mat_evaluate_world(const DScene &S, const MatCtx &mc, const Interaction &it, V3 world_wo, V3 world_wi, float *eta, DCounters &cnt SWL_P) {
    V3 wo = it.shading.to_local(world_wo), wi = it.shading.to_local(world_wi);
    if constexpr (!FULL) {
        ScatterEval se = eval_local_call(S, mc.lobe_lds, wo, wi, eta); // (mat_evaluate_and_sample parked mc.single)
        se.f *= abs_cos_theta(wi);
        return se;
    }
    ScatterEval ret; ret.f = mks(0.f); ret.pdf = 0.f; ret.flags = flag::Unset;
    bool sh_world = same_hemisphere(world_wo, world_wi, it.shading.z);
#pragma unroll 1
    for (int i = 0; i < mc.n; ++i) {
        Lobe l; mat_lobe<FULL>(S, mc, it, i, l, cnt SWL_A);
        stage_lobe(mc.lobe_lds, l);
        ScatterEval se = eval_local_call(S, mc.lobe_lds, wo, wi, eta);
        se.f *= abs_cos_theta(wi);
        if (!mc.is_set) { ret = se; break; }
        float factor = l.kind == LB_DIELECTRIC ? 1.f : (sh_world ? 1.f : 0.f); // valid_world_factor lobe.cpp:35-38,373-375
        se.f *= l.weight * factor;
        se.pdf *= l.sample_weight * factor;
        ret.f += se.f;
        ret.pdf += se.pdf;
        ret.flags |= se.flags;
    }
    return ret;
}
// LobeSet::sample_wi_impl (lobe.cpp:629-658) / Lobe::sample_wi_impl: pick the lobe (3 burnt draws for a set), sample locally
template<bool FULL>
VD V3 mat_sample_wi(const DScene &S, const MatCtx &mc, const Interaction &it, Sampler &sampler, bool *valid, DCounters &cnt SWL_P) {
    V3 wo = it.shading.to_local(it.wo);
    if constexpr (!FULL) return it.shading.to_world(sample_wi_local_call(mc.lobe_lds, wo, sampler, valid));
    int strategy = 0;
    if (mc.is_set) {
        float uc = sampler.next_1d();
        (void) sampler.next_2d();
        float sum_weights = 0.f;
        for (int i = 0; i < mc.n; ++i) {
            float sw = 0.f; // (always assigned below: a lobe set is a mix / add or a principled_bsdf)
            int ip = i; float parent = 1.f; bool principled = mc.m->type == VMK_MAT_PRINCIPLED;
            if (mc.m->type == VMK_MAT_MIX || mc.m->type == VMK_MAT_ADD) {
                const int np = mc.pchild >= 0 ? mc.n - 1 : 0;
                const int child = mc.pchild < 0 ? i : (mc.pchild == 0 ? (i < np ? 0 : 1) : (i == 0 ? 0 : 1));
                sw = child == 0 ? mc.mixsw[0] : mc.mixsw[1];
                if (child == mc.pchild) { principled = true; parent = sw; ip = mc.pchild == 0 ? i : i - 1; }
            }
            if (principled) { int k = ip + mc.first; sw = (k == 0 ? mc.sw[0] : k == 1 ? mc.sw[1] : k == 2 ? mc.sw[2] : k == 3 ? mc.sw[3] : k == 4 ? mc.sw[4] : mc.sw[5]) * parent; }
            strategy = uc > sum_weights ? i : strategy;
            sum_weights += sw;
        }
        if (mc.n == 1) strategy = 0;
    }
    Lobe l; mat_lobe<FULL>(S, mc, it, strategy, l, cnt SWL_A);
    stage_lobe(mc.lobe_lds, l);
    V3 wi_local = sample_wi_local_call(mc.lobe_lds, wo, sampler, valid);
    return it.shading.to_world(wi_local);
}
// MaterialEvaluator::evaluate (towards `wi_light`) followed by MaterialEvaluator::sample (material.cpp:132-184,
// direct_lighting integrator.cpp:20-37): the two evaluations share ONE instance of the lobe code (2-trip loop).
template<bool FULL>
VD void mat_evaluate_and_sample(const DScene &S, const MatCtx &mc, const Interaction &it, V3 wi_light, Sampler &sampler,
                                ScatterEval &se_light, BSDFSample &bs, DCounters &cnt SWL_P) {
    bs.eta = 1.f; bs.wi = mk3(0.f);
    if constexpr (!FULL) stage_lobe(mc.lobe_lds, mc.single); // one lobe for all three calls
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        V3 wi = wi_light;
        bool valid = true;
#ifdef VMK_STAGE_EACH_CALL
        if constexpr (!FULL) stage_lobe(mc.lobe_lds, mc.single);
#endif
        if (pass == 1) { wi = mat_sample_wi<FULL>(S, mc, it, sampler, &valid, cnt SWL_A); bs.wi = wi; }
        ScatterEval e = mat_evaluate_world<FULL>(S, mc, it, it.wo, wi, pass == 1 ? &bs.eta : nullptr, cnt SWL_A);
        e.pdf *= valid ? 1.f : 0.f;
        bool discard = same_hemisphere(it.wo, wi, it.ng) == ((e.flags & flag::Transmission) != 0);
        if (discard) e.pdf = 0.f;
        if (pass == 0) se_light = e; else bs.eval = e;
    }
}

// =====================================================================================================
// a7-a10. lights — base/illumination/lightsampler.cpp, render_core/light/{area,environments/spherical}.cpp,
//                  render_core/warper/{alias.h,alias2d.cpp}
// =====================================================================================================
struct LightEval { Spec L; float pdf; };
struct LightSample { LightEval eval; V3 p_light; };

VD void alias_offset_u_remapped(const DScene &S, uint32_t base, uint32_t size, float u, uint32_t *idx_out, float *u_remapped) { // alias.h:148-158
    u = u * (float) size;
    uint32_t idx = min((uint32_t) u, size - 1u);
    u = fmin_(u - (float) idx, OneMinusEpsilon);
    float prob = S.alias_prob[base + idx];
    uint32_t alias = S.alias_idx[base + idx];
    *u_remapped = u < prob ? fmin_(u / prob, OneMinusEpsilon) : fmin_((1.f - u) / (1.f - prob), OneMinusEpsilon);
    *idx_out = u < prob ? idx : alias;
}
VD float alias_PMF(const DScene &S, const vmk_light *l, uint32_t i) { // alias.h:50-52
    return l->alias_integral > 0.f ? S.alias_func[l->alias_offset + i] / (l->alias_integral * (float) l->alias_count) : 0.f;
}
VD float alias_PDF(const DScene &S, const vmk_light *l, uint32_t i) { // alias.h:44-46
    return l->alias_integral > 0.f ? S.alias_func[l->alias_offset + i] / l->alias_integral : 0.f;
}
// LightSampler::PMF / select_light (lightsampler.cpp:159-197) over the sampler's own PMF_ / select_light_:
// uniform (uniform.cpp:13-34, punctual lights only + correct_index when the environment is sampled separately) or
// power (power.cpp:13-28: alias table over luminance(power()), all lights, the environment weighing 0 when separate)
VD float light_pmf_inner(const DScene &S, const vmk_render_params *P, uint32_t index) {
    uint32_t n = S.n_lights;
    if (P->light_sampler == 1)
        return S.light_alias_integral > 0.f ? S.alias_func[S.light_alias_offset + index] / (S.light_alias_integral * (float) n) : 0.f;
    bool sep = P->env_separate && S.env_light != VMK_INVALID;
    return 1.f / (float) (sep ? n - 1u : n);
}
VD uint32_t light_select_inner(const DScene &S, const vmk_render_params *P, float u) {
    uint32_t n = S.n_lights;
    if (P->light_sampler == 1) { uint32_t idx; float ur; alias_offset_u_remapped(S, S.light_alias_offset, n, u, &idx, &ur); return idx; }
    bool sep = P->env_separate && S.env_light != VMK_INVALID;
    if (sep) {
        uint32_t punctual = n - 1u;
        uint32_t idx = (uint32_t) fmin_(u * (float) punctual, (float) punctual - 1.f);
        return idx < S.env_light ? idx : idx + 1u; // correct_index lightsampler.cpp:33-38
    }
    return (uint32_t) fmin_(u * (float) n, (float) n - 1.f);
}
VD float light_select_PMF(const DScene &S, const vmk_render_params *P, uint32_t index) {
    if (P->env_separate && S.env_light != VMK_INVALID) {
        float env_prob = P->env_prob;
        if (index == S.env_light) return env_prob;
        return (1.f - env_prob) * light_pmf_inner(S, P, index);
    }
    return light_pmf_inner(S, P, index);
}
VD void light_select(const DScene &S, const vmk_render_params *P, float u, uint32_t *index, float *pmf) {
    if (P->env_separate && S.env_light != VMK_INVALID) {
        float env_prob = P->env_prob;
        if (u < env_prob) { *index = S.env_light; *pmf = env_prob; return; }
        u = remapping(u, env_prob, 1.f);
        *index = light_select_inner(S, P, u);
        *pmf = light_pmf_inner(S, P, *index) * (1.f - env_prob);
        return;
    }
    *index = light_select_inner(S, P, u);
    *pmf = light_pmf_inner(S, P, *index);
}
VD Spec area_L(const DScene &S, const vmk_light *l, V2 uv, V3 ng, V3 w, DCounters &cnt SWL_P) { // area.cpp:91-95
    Spec radiance = eval_slot_illumination(S, l->color, uv, cnt SWL_A) * l->scale;
    return radiance * ((dot(w, ng) > 0.f || l->two_sided) ? 1.f : 0.f);
}
VD float area_PDF_wi(float pdf_pos, V3 ng, V3 w) { // area.cpp:114-118
    float ret = PDF_wi(pdf_pos, ng, w);
    return (isinf_(ret) || isnan_(ret)) ? 0.f : ret;
}
VD LightSample area_sample_wi(const DScene &S, const vmk_render_params *P, const vmk_light *l, V3 p_ref, V2 u, DCounters &cnt SWL_P) { // area.cpp:120-149
    uint32_t prim; float ur;
    alias_offset_u_remapped(S, l->alias_offset, l->alias_count, u.x, &prim, &ur);
    float pmf = alias_PMF(S, l, prim);
    u.x = ur;
    V2 bary = square_to_triangle(u);
    uint32_t tri = S.tri_lookup[S.instances[l->inst_id].tri_offset + prim];
    Interaction it;
    compute_surface_interaction<false>(S, tri, l->inst_id, prim, bary, it);
    float pdf_pos = (1.f / it.prim_area) * pmf;
    LightSample ret;
    V3 w = p_ref - it.pos;
    ret.eval.L = area_L(S, l, it.uv, it.ng, w, cnt SWL_A);
    ret.eval.pdf = area_PDF_wi(pdf_pos, it.ng, w);
    ret.p_light = robust_pos(it.pos, it.ng, w, P->ray_offset_factor);
    return ret;
}
VD Spec env_L(const DScene &S, const vmk_light *l, V3 local_dir, DCounters &cnt SWL_P) { // spherical.cpp:60-68
    V2 uv = {spherical_phi(local_dir) * Inv2Pi, spherical_theta(local_dir) * InvPi};
    return eval_slot_illumination(S, l->color, uv, cnt SWL_A) * l->scale;
}
VD float env_map_PDF(const DScene &S, const vmk_light *l, V2 p) { // alias2d.cpp:102-106
    uint32_t iu = min((uint32_t) (p.x * (float) l->res_x), l->res_x - 1u);
    uint32_t iv = min((uint32_t) (p.y * (float) l->res_y), l->res_y - 1u);
    return l->alias_integral > 0.f ? S.alias_func[l->cond_offset + iv * l->res_x + iu] / l->alias_integral : 0.f;
}
VD LightEval env_evaluate_wi(const DScene &S, const vmk_light *l, V3 p_ref_pos, V3 p_light_pos, DCounters &cnt SWL_P) { // spherical.cpp:86-103
    LightEval ret;
    V3 world_dir = normalize(p_light_pos - p_ref_pos);
    V3 local_dir = mul3x3(l->w2o, world_dir);
    float theta = spherical_theta(local_dir), phi = spherical_phi(local_dir);
    float sin_t = sin_(theta);
    V2 uv = {phi * Inv2Pi, theta * InvPi};
    ret.L = env_L(S, l, local_dir, cnt SWL_A);
    float pdf = env_map_PDF(S, l, uv) / (_2Pi * Pi * sin_t);
    ret.pdf = sin_t == 0.f ? 0.f : pdf;
    return ret;
}
VD LightSample env_sample_wi(const DScene &S, const vmk_light *l, V3 p_ref, V2 u, DCounters &cnt SWL_P) { // spherical.cpp:105-125,162-168; alias2d.cpp:110-129
    uint32_t iv; float urv;
    alias_offset_u_remapped(S, l->alias_offset, l->alias_count, u.y, &iv, &urv);
    float fv = ((float) iv + urv) / (float) l->alias_count;
    float pdf_v = alias_PDF(S, l, iv);
    uint32_t buffer_offset = l->res_x * iv;
    uint32_t iu; float uru;
    alias_offset_u_remapped(S, l->cond_offset + buffer_offset, l->res_x, u.x, &iu, &uru);
    float fu = ((float) iu + uru) / (float) l->res_x;
    float integral_u = S.alias_func[l->alias_offset + iv];
    float func_u = S.alias_func[l->cond_offset + buffer_offset + iu];
    float pdf_u = integral_u > 0.f ? func_u / integral_u : 0.f;
    float pdf_map = pdf_u * pdf_v;
    LightSample ret;
    float theta = fv * Pi, phi = fu * _2Pi;
    float sin_t, cos_t; sincos_(theta, &sin_t, &cos_t);
    V3 local_dir = spherical_direction(sin_t, cos_t, phi);
    V3 world_dir = normalize(mul3x3(l->o2w, local_dir));
    float pdf_dir = pdf_map / (_2Pi * Pi * sin_t);
    ret.eval.pdf = isinf_(pdf_dir) ? 0.f : pdf_dir;
    ret.eval.L = env_L(S, l, local_dir, cnt SWL_A);
    ret.p_light = p_ref + world_dir * l->world_diameter;
    return ret;
}
// IPointLight::sample_wi (light.cpp:49-58) with PointLight::Le (point.cpp:43-48) / SpotLight::Le + falloff (spot.cpp:56-79);
// PDF_wi = -1 marks a delta light (light.h:227-231)
VD LightSample point_sample_wi(const DScene &S, const vmk_light *l, V3 p_ref, DCounters &cnt SWL_P) {
    LightSample ls;
    V3 pos = ld3(l->position);
    if (l->type == VMK_LIGHT_PROJECTOR) { // Projector::Le (projector.cpp:98-110): the image seen through the light's frustum, over d^2
        V3 p = transform_point4(l->w2o4, p_ref);
        const float d2 = length_squared(p);
        bool valid = p.z > 0.f;
        p = p / p.z;
        const V2 tan_xy = {l->tan_xy[0], l->tan_xy[1]};
        const V2 uv = {(p.x + tan_xy.x) / (2.f * tan_xy.x), (p.y + tan_xy.y) / (2.f * tan_xy.y)};
        valid = valid && uv.x >= 0.f && uv.x <= 1.f && uv.y >= 0.f && uv.y <= 1.f;
        // (outside the frustum the reference multiplies the fetched colour by 0; the fetch itself is skipped here: uv may be anything)
        ls.eval.L = valid ? ((1.f * eval_slot_illumination(S, l->color, uv, cnt SWL_A)) / d2) * l->scale : mks(0.f);
        ls.eval.pdf = -1.f;
        ls.p_light = pos;
        return ls;
    }
    V3 w_un = p_ref - pos;
    Spec value = eval_slot_illumination(S, l->color, V2{0.f, 0.f}, cnt SWL_A) * l->scale;
    if (l->type == VMK_LIGHT_SPOT) {
        V3 w = normalize(w_un);
        float cos_theta = clamp_(dot(ld3(l->direction), w), l->cos_angle, l->cos_falloff_start);
        float factor = (cos_theta - l->cos_angle) / (l->cos_falloff_start - l->cos_angle);
        ls.eval.L = value / length_squared(w_un) * pow4(factor);
    } else ls.eval.L = value / length_squared(w_un);
    ls.eval.pdf = -1.f;
    ls.p_light = pos;
    return ls;
}
VD LightSample light_sample_wi(const DScene &S, const vmk_render_params *P, V3 p_ref, Sampler &sampler, DCounters &cnt SWL_P) { // lightsampler.cpp:199-216
    float u_light = sampler.next_1d();
    V2 u_surface = sampler.next_2d();
    uint32_t index; float pmf;
    light_select(S, P, u_light, &index, &pmf);
    const vmk_light *l = S.lights + index;
    // (point / spot ignore u_surface, light.cpp:49-58)
    LightSample ls = l->type == VMK_LIGHT_AREA ? area_sample_wi(S, P, l, p_ref, u_surface, cnt SWL_A)
                   : (l->type == VMK_LIGHT_SPHERICAL ? env_sample_wi(S, l, p_ref, u_surface, cnt SWL_A) : point_sample_wi(S, l, p_ref, cnt SWL_A));
    ls.eval.pdf *= pmf;
    return ls;
}
VD LightEval light_evaluate_hit_wi(const DScene &S, const vmk_render_params *P, V3 p_ref, const Interaction &it, DCounters &cnt SWL_P) { // lightsampler.cpp:252-267
    LightEval ret; ret.L = mks(0.f); ret.pdf = 0.f;
    const vmk_light *l = S.lights + it.light_id;
    if (l->type != VMK_LIGHT_AREA) return ret;
    float pdf_pos = (1.f / it.prim_area) * alias_PMF(S, l, it.prim_id);
    V3 w = p_ref - it.pos;
    ret.L = area_L(S, l, it.uv, it.ng, w, cnt SWL_A);
    ret.pdf = area_PDF_wi(pdf_pos, it.ng, w);
    ret.pdf *= light_select_PMF(S, P, it.light_id);
    return ret;
}
VD LightEval light_evaluate_miss_wi(const DScene &S, const vmk_render_params *P, V3 p_ref, V3 wi, DCounters &cnt SWL_P) { // lightsampler.cpp:290-300
    const vmk_light *l = S.lights + S.env_light;
    LightEval ret = env_evaluate_wi(S, l, p_ref, p_ref + wi, cnt SWL_A);
    ret.pdf *= light_select_PMF(S, P, S.env_light);
    return ret;
}

// =====================================================================================================
// a3. ray generation — sampler.h:65-73, box.cpp:16-20, triangle.cpp:16-18, fitted_curve.h:76-113,
//     sensor.cpp:44-56, thin_lens.cpp:34-42
// =====================================================================================================
VD void table_offset(const float *prob, const uint32_t *alias, uint32_t size, float u, uint32_t *idx_out, float *u_remapped) {
    u = u * (float) size;
    uint32_t idx = min((uint32_t) u, size - 1u);
    u = fmin_(u - (float) idx, OneMinusEpsilon);
    float p = prob[idx];
    *u_remapped = u < p ? fmin_(u / p, OneMinusEpsilon) : fmin_((1.f - u) / (1.f - p), OneMinusEpsilon);
    *idx_out = u < p ? idx : alias[idx];
}
VD V2 filter_sample(const vmk_render_params *P, V2 u) {
    if (P->filter_type == VMK_FILTER_BOX) return {lerp_(u.x, -P->filter_radius[0], P->filter_radius[0]), lerp_(u.y, -P->filter_radius[1], P->filter_radius[1])};
    if (P->filter_type == VMK_FILTER_TRIANGLE) return {sample_tent(u.x, P->filter_radius[0]), sample_tent(u.y, P->filter_radius[1])};
    const uint32_t N = VMK_FILTER_TABLE_SIZE;
    V2 v = {u.x * 2.f - 1.f, u.y * 2.f - 1.f};
    V2 a = {abs_(v.x), abs_(v.y)};
    uint32_t iv; float urv; table_offset(P->filter_marginal_prob, P->filter_marginal_alias, N, a.y, &iv, &urv);
    float fv = ((float) iv + urv) / (float) N;
    uint32_t iu; float uru; table_offset(P->filter_cond_prob + iv * N, P->filter_cond_alias + iv * N, N, a.x, &iu, &uru);
    float fu = ((float) iu + uru) / (float) N;
    float sx = v.x > 0.f ? 1.f : (v.x < 0.f ? -1.f : 0.f), sy = v.y > 0.f ? 1.f : (v.y < 0.f ? -1.f : 0.f);
    return {fu * sx * P->filter_radius[0], fv * sy * P->filter_radius[1]};
}
VD Ray generate_ray(const vmk_render_params *P, uint32_t px, uint32_t py, Sampler &sampler, V2 *p_film_out = nullptr) {
    V2 fs = filter_sample(P, sampler.next_2d());
    V2 p_film = {(float) px + 0.5f + fs.x, (float) py + 0.5f + fs.y};
    if (p_film_out) *p_film_out = p_film;
    V2 p_lens_u = sampler.next_2d();
    (void) sampler.next_1d();
    V3 p_sensor = transform_point4(P->raster_to_sensor, mk3(p_film.x, p_film.y, 0.f));
    V3 dir = normalize(p_sensor);
    V2 pl = square_to_disk(p_lens_u) * P->lens_radius;
    float ft = P->focal_distance / dir.z;
    V3 p_focus = dir * ft;
    V3 org = mk3(pl.x, pl.y, 0.f);
    dir = normalize(p_focus - org);
    Ray r;
    r.o = transform_point4(P->c2w, org);
    r.d = transform_vector4(P->c2w, dir);
    r.t_max = RayTMax;
    return r;
}

}// namespace vmkd
