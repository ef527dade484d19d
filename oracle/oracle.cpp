// oracle/oracle.cpp — TEST INFRASTRUCTURE: CPU restatement of Vision's megakernel path-tracing hot path.
//
// NOT part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
// link or call this file.  It restates, function by function, the arithmetic of the reference
// (Royalvice/Vision, paths relative to its `src/`), consuming the same flat scene tables (include/vmk.h) the
// HIP backend consumes, in plain scalar C++ (float32, -ffp-contract=off).
//
// Pinning: the ocarina submodule (DSL/RHI/OptiX layer) is absent from the reference checkout, so everything
// that crosses into it (omath.h) is "parity unpinned".  The lobe / microfacet / Fresnel code is pinned by
// re-integrating the reference's own precomputed albedo tables (tests/golden/lut_subgrid.json, extracted from
// base/scattering/precomputed_table.h by tools/make_golden_luts.py) — see tests/test_oracle_golden.py.
#include "omath.h"
#include "../include/vmk.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace orc {

// =====================================================================================================
// a1. Sampler — render_core/sampler/independent.cpp:24-42, math/util.h:13-33, base/sampler.h:60-73
// =====================================================================================================
inline uint32_t tea(uint32_t v0, uint32_t v1) { // util.h:13-24, N = 4
    uint32_t s0 = 0;
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
struct Sampler {
    uint32_t state{0};
    void start(uint32_t px, uint32_t py, uint32_t sample_index, uint32_t dim) { // independent.cpp:24-28
        state = tea(tea(px, py), tea(sample_index, dim));
    }
    float next_1d() { // util.h:27-33
        state = 1664525u * state + 1013904223u;
        return ((float) (state & 0x00ffffffu) * 1.f) * (1.f / 16777216.f);
    }
    float2 next_2d() { float x = next_1d(); float y = next_1d(); return {x, y}; } // sampler.h:60-64
};

// =====================================================================================================
// warps / MIS — math/warp.h
// =====================================================================================================
inline float2 square_to_disk(float2 u) { // warp.h:26-31
    float r = sqrtf(u.x);
    float theta = _2Pi * u.y;
    float s, c; sincos_(theta, &s, &c);
    return {r * c, r * s};
}
inline float3 square_to_cosine_hemisphere(float2 u) { // warp.h:38-43
    float2 d = square_to_disk(u);
    float z = sqrtf(fmax_(0.f, 1.f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
inline float cosine_hemisphere_PDF(float cos_theta) { return cos_theta * InvPi; } // warp.h:46-49
inline float2 square_to_triangle(float2 u) { float su0 = sqrtf(u.x); return {1.f - su0, u.y * su0}; } // warp.h:65-69
inline float sample_linear(float u, float a, float b) { // warp.h:122-128
    float x = u * (a + b) / (a + sqrtf(lerp_(u, sqr(a), sqr(b))));
    float ret = fmin_(x, OneMinusEpsilon);
    return (u == 0.f && a == 0.f) ? 0.f : ret;
}
inline float sample_tent(float u, float r) { // warp.h:131-146
    return u < 0.5f ? -r * sample_linear((0.5f - u) * 2.f, 1.f, 0.f) : r * sample_linear((u - 0.5f) * 2.f, 1.f, 0.f);
}
inline float MIS_weight(float f_pdf, float g_pdf) { // warp.h:149-155,186-199: balance heuristic, nf = ng = 1
    return (1.f * f_pdf) / (1.f * f_pdf + 1.f * g_pdf);
}
inline float PDF_wi(float pdf_point, float3 normal, float3 wo_un) { // warp.h:89-94
    float cos_t = abs_(dot(normal, normalize(wo_un)));
    return pdf_point * length_squared(wo_un) / cos_t;
}
inline float remapping(float a, float low, float high) { return (a - low) / (high - low); } // warp.h:18-23

// =====================================================================================================
// textures & LUTs — ocarina `tex.sample(n, uv)` (App. B: normalised coords, bilinear, texel centres at
// (i+0.5)/N, repeat wrap for images, clamp for LUTs); image decode image_pool.cpp:13-35
// =====================================================================================================
struct SceneView;
static float g_srgb_lut[256];
static void init_srgb_lut() {
    static bool done = false;
    if (done) return;
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        g_srgb_lut[i] = (float) (c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    done = true;
}
inline float4 fetch_texel(const vmk_scene *s, const vmk_texture &t, int x, int y) {
    const uint8_t *base = s->tex_data + t.offset;
    size_t i = (size_t) y * t.width + (size_t) x;
    if (t.format == VMK_TEX_RGBA32F) {
        const float *p = (const float *) base + i * 4;
        return {p[0], p[1], p[2], p[3]};
    }
    const uint8_t *p = base + i * 4;
    if (t.format == VMK_TEX_RGBA8_SRGB) return {g_srgb_lut[p[0]], g_srgb_lut[p[1]], g_srgb_lut[p[2]], (float) p[3] * (1.f / 255.f)};
    return {(float) p[0] * (1.f / 255.f), (float) p[1] * (1.f / 255.f), (float) p[2] * (1.f / 255.f), (float) p[3] * (1.f / 255.f)};
}
inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
inline float4 lerp4(float t, float4 a, float4 b) {
    return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)};
}
inline float4 sample_image(const vmk_scene *s, uint32_t tex_id, float2 uv) {
    const vmk_texture &t = s->textures[tex_id];
    float x = uv.x * (float) t.width - 0.5f, y = uv.y * (float) t.height - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float tx = x - fx0, ty = y - fy0;
    int x0 = wrap_repeat((int) fx0, (int) t.width), y0 = wrap_repeat((int) fy0, (int) t.height);
    int x1 = wrap_repeat((int) fx0 + 1, (int) t.width), y1 = wrap_repeat((int) fy0 + 1, (int) t.height);
    float4 c00 = fetch_texel(s, t, x0, y0), c10 = fetch_texel(s, t, x1, y0);
    float4 c01 = fetch_texel(s, t, x0, y1), c11 = fetch_texel(s, t, x1, y1);
    return lerp4(ty, lerp4(tx, c00, c10), lerp4(tx, c01, c11));
}
inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
// 2-D LUT, `nc` interleaved channels, clamp addressing
inline void sample_lut2d(const float *lut, int nc, float u, float v, float *out) {
    const int N = VMK_LUT_RES;
    float x = u * (float) N - 0.5f, y = v * (float) N - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float tx = x - fx0, ty = y - fy0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
    for (int c = 0; c < nc; ++c) {
        float c00 = lut[(y0 * N + x0) * nc + c], c10 = lut[(y0 * N + x1) * nc + c];
        float c01 = lut[(y1 * N + x0) * nc + c], c11 = lut[(y1 * N + x1) * nc + c];
        out[c] = lerp_(ty, lerp_(tx, c00, c10), lerp_(tx, c01, c11));
    }
}
inline void sample_lut3d(const float *lut, int nc, float3 uvw, float *out) {
    const int N = VMK_LUT_RES;
    float x = uvw.x * (float) N - 0.5f, y = uvw.y * (float) N - 0.5f, z = uvw.z * (float) N - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    float tx = x - fx0, ty = y - fy0, tz = z - fz0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
    int z0 = clampi((int) fz0, 0, N - 1), z1 = clampi((int) fz0 + 1, 0, N - 1);
    for (int c = 0; c < nc; ++c) {
        auto at = [&](int xi, int yi, int zi) { return lut[((zi * N + yi) * N + xi) * nc + c]; };
        float a = lerp_(ty, lerp_(tx, at(x0, y0, z0), at(x1, y0, z0)), lerp_(tx, at(x0, y1, z0), at(x1, y1, z0)));
        float b = lerp_(ty, lerp_(tx, at(x0, y0, z1), at(x1, y0, z1)), lerp_(tx, at(x0, y1, z1), at(x1, y1, z1)));
        out[c] = lerp_(tz, a, b);
    }
}

// slot evaluation — shader_node.cpp:242-273 (swizzle), number.cpp:106-109, image.cpp:87-97
inline float3 eval_slot3(const vmk_scene *s, const vmk_slot &sl, float2 uv) {
    if (sl.tex == VMK_INVALID) return {sl.v[0], sl.v[1], sl.v[2]};
    float4 t = sample_image(s, sl.tex & 0xffffu, uv);
    const bool tinted = (sl.tex & VMK_SLOT_TINTED) != 0u; // "multiply" node: image x constant (math.cpp:78-90, BinaryOpNode always multiplies)
    const float scale = tinted ? 1.f : sl.v[0];
    float c[4] = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
    uint32_t sw = sl.tex >> 16;
    float3 r = {c[sw & 3u], c[(sw >> 2) & 3u], c[(sw >> 4) & 3u]};
    if (tinted) r = r * make_float3(sl.v[0], sl.v[1], sl.v[2]);
    return r;
}
inline float eval_slot1(const vmk_scene *s, const vmk_slot &sl, float2 uv) {
    if (sl.tex == VMK_INVALID) return sl.v[0];
    return eval_slot3(s, sl, uv).x;
}

// =====================================================================================================
// a2. spectrum — render_core/spectrum/{srgb,hero}.cpp, base/color/{spd,spectrum}.cpp.
// A SampledSpectrum is a Spec (omath.h): a float3 for dimension 3 in both modes, four samples in the ORC_SPEC_DIM = 4 build
// (spectrum/hero, "dimension": 4).  In hero mode the path's SampledWavelengths live in a
// thread-local (set by Li for the path being traced); every decode below reads them.
// PARITY UNPINNED for the hero branch's data: Vision's sRGB->spectrum table (srgb2spec.h) is absent from the reference
// checkout, so the table these functions read is regenerated by the host (vision_amd/csrc/host/rgb2spec_opt.h, the
// published Jakob-Hanika fit) and pinned only by properties (round trip through the CIE observer, convergence of a hero
// render to the sRGB render; tests/test_host.py, tests/test_oracle_render.py).  The arithmetic itself follows hero.cpp /
// spd.cpp line by line, including SPD's 471/94 nm sample spacing.
// =====================================================================================================
struct Swl { float lambda[kSpecDim]; float pdf[kSpecDim]; };
static thread_local Swl *tl_swl = nullptr; // non-null only while a hero-spectrum path is traced
inline bool is_hero(const vmk_scene *s) { return s->spectrum == VMK_SPECTRUM_HERO; }
inline float sample_visible_wavelength(float u) { return 538.f - 138.888889f * atanh_(0.85691062f - 1.82750197f * u); } // hero.cpp:15-18
inline float visible_wavelength_PDF(float lambda) { return 0.0039398042f / sqr(cosh_(0.0072f * (lambda - 538.f))); }     // hero.cpp:21-24
inline Swl sample_wavelengths(Sampler &sampler) { // hero.cpp:286-299, 1 draw
    Swl swl;
    float u = sampler.next_1d();
    for (uint32_t i = 0; i < kSpecDim; ++i) {
        float offset = (float) i * (1.f / (float) kSpecDim);
        float up = fract_(u + offset);
        swl.lambda[i] = sample_visible_wavelength(up);
        swl.pdf[i] = visible_wavelength_PDF(swl.lambda[i]);
    }
    return swl;
}
inline float spd_eval(const float *f, float interval, float lambda) { // SPD::eval spd.cpp:79-86
    float t = (clamp_(lambda, 360.f, 830.f) - 360.f) / interval;
    uint32_t sample_count = (uint32_t) ((830.f - 360.f) / interval) + 1u;
    uint32_t i = (uint32_t) fmin_(t, (float) (sample_count - 2u));
    float l = f[i], r = f[i + 1u];
    return lerp_(fract_(t), l, r);
}
inline Spec spd_eval3(const float *f, float interval, const Swl &swl) { return smap(spec_from_array(swl.lambda), [&](float lambda) { return spd_eval(f, interval, lambda); }); }
inline float sigmoid_polynomial(float3 c, float lambda) { // RGBSigmoidPolynomial hero.cpp:27-49
    float x = fma_(fma_(c.x, lambda, c.y), lambda, c.z);
    float v = 0.5f * fma_(x, rsqrt_(fma_(x, x, 1.f)), 1.f);
    return isinf_(x) ? (x > 0.0f ? 1.f : 0.f) : v;
}
inline float inverse_smooth_step(float x) { return 0.5f - sin_(asin_(1.0f - 2.0f * x) * (1.0f / 3.0f)); } // hero.cpp:66-68
inline float3 rgb2spec_fetch(const float *table, uint32_t maxc, float cx, float cy, float cz) { // tex3D, trilinear, clamp (App. B)
    const int N = (int) VMK_RGB2SPEC_RES;
    const float *t = table + (size_t) maxc * N * N * N * 4;
    float x = cx * (float) N - 0.5f, y = cy * (float) N - 0.5f, z = cz * (float) N - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    float tx = x - fx0, ty = y - fy0, tz = z - fz0;
    int x0 = clampi((int) fx0, 0, N - 1), x1 = clampi((int) fx0 + 1, 0, N - 1);
    int y0 = clampi((int) fy0, 0, N - 1), y1 = clampi((int) fy0 + 1, 0, N - 1);
    int z0 = clampi((int) fz0, 0, N - 1), z1 = clampi((int) fz0 + 1, 0, N - 1);
    auto at = [&](int xi, int yi, int zi) { const float *v = t + (((size_t) zi * N + yi) * N + xi) * 4; return make_float3(v[0], v[1], v[2]); };
    float3 a = lerp3(ty, lerp3(tx, at(x0, y0, z0), at(x1, y0, z0)), lerp3(tx, at(x0, y1, z0), at(x1, y1, z0)));
    float3 b = lerp3(ty, lerp3(tx, at(x0, y0, z1), at(x1, y0, z1)), lerp3(tx, at(x0, y1, z1), at(x1, y1, z1)));
    return lerp3(tz, a, b);
}
inline float3 rgb2spec_albedo_coeffs(const vmk_scene *s, float3 rgb_in) { // RGBToSpectrumTable::decode_albedo, device variant hero.cpp:142-171
    float3 rgb = {clamp_(rgb_in.x, 0.f, 1.f), clamp_(rgb_in.y, 0.f, 1.f), clamp_(rgb_in.z, 0.f, 1.f)};
    float3 c = {0.0f, 0.0f, (rgb.x - 0.5f) * rsqrt_(rgb.x * (1.0f - rgb.x))};
    if (!(rgb.x == rgb.y && rgb.y == rgb.z)) {
        uint32_t maxc = rgb.x > rgb.y ? (rgb.x > rgb.z ? 0u : 2u) : (rgb.y > rgb.z ? 1u : 2u);
        float v[3] = {rgb.x, rgb.y, rgb.z};
        float z = v[maxc];
        float x = v[(maxc + 1u) % 3u] / z;
        float y = v[(maxc + 2u) % 3u] / z;
        float zz = inverse_smooth_step(inverse_smooth_step(z));
        const float res = (float) VMK_RGB2SPEC_RES;
        const float sc = (res - 1.0f) / res, of = 0.5f / res;
        c = rgb2spec_fetch(s->rgb2spec, maxc, fma_(x, sc, of), fma_(y, sc, of), fma_(zz, sc, of));
    }
    return c;
}
inline float3 rgb2spec_unbound_coeffs(const vmk_scene *s, float3 rgb_in, float *scale_out) { // decode_unbound hero.cpp:173-179
    float3 rgb = {fmax_(rgb_in.x, 0.f), fmax_(rgb_in.y, 0.f), fmax_(rgb_in.z, 0.f)};
    float m = fmax_(fmax_(rgb.x, rgb.y), rgb.z);
    float scale = 2.f * m;
    *scale_out = scale;
    return rgb2spec_albedo_coeffs(s, scale == 0.f ? make_float3(0.f) : rgb / scale);
}
inline Spec sigmoid3(float3 c, const Swl &swl) { return smap(spec_from_array(swl.lambda), [&](float lambda) { return sigmoid_polynomial(c, lambda); }); }
// decode_to_albedo / decode_to_unbound_spectrum / decode_to_illumination (srgb.cpp:49-57, hero.cpp:331-342)
#if ORC_SPEC_DIM == 3
inline Spec rgb_as_spec(float3 rgb) { return rgb; } // spectrum/srgb: the channels are the samples (srgb.cpp:49-57)
#else
inline Spec rgb_as_spec(float3) { fprintf(stderr, "oracle (ORC_SPEC_DIM = 4): not a spectrum/hero scene\n"); abort(); }
#endif
inline Spec spec_albedo(const vmk_scene *s, float3 rgb) {
    if (!is_hero(s)) return rgb_as_spec(rgb);
    return sigmoid3(rgb2spec_albedo_coeffs(s, rgb), *tl_swl);
}
inline Spec spec_unbound(const vmk_scene *s, float3 rgb) {
    if (!is_hero(s)) return rgb_as_spec(rgb);
    float scale; float3 c = rgb2spec_unbound_coeffs(s, rgb, &scale);
    return sigmoid3(c, *tl_swl) * scale; // RGBUnboundSpectrum::eval hero.cpp:201-203
}
inline Spec spec_illumination(const vmk_scene *s, float3 rgb) {
    if (!is_hero(s)) return rgb_as_spec(rgb);
    float scale; float3 c = rgb2spec_unbound_coeffs(s, rgb, &scale);
    return (sigmoid3(c, *tl_swl) * scale) * spd_eval3(s->spd_data + s->spd_cie[3], s->spd_cie_interval, *tl_swl); // hero.cpp:219-221
}
// Spectrum::linear_srgb (srgb.cpp:46-48; hero.cpp:265-267,281-291 + cie::xyz_to_linear_srgb cie.h:413-420)
#if ORC_SPEC_DIM == 3
inline float3 spec_as_rgb(Spec sp) { return sp; } // Spectrum::linear_srgb of srgb.cpp:46-48
#else
inline float3 spec_as_rgb(Spec) { fprintf(stderr, "oracle (ORC_SPEC_DIM = 4): not a spectrum/hero scene\n"); abort(); }
#endif
inline float3 spec_linear_srgb(const vmk_scene *s, Spec sp) {
    if (!is_hero(s)) return spec_as_rgb(sp);
    const Swl &swl = *tl_swl;
    const float *X = s->spd_data + s->spd_cie[0], *Y = s->spd_data + s->spd_cie[1], *Z = s->spd_data + s->spd_cie[2];
    float v[kSpecDim];
    for (uint32_t i = 0; i < kSpecDim; ++i) v[i] = scomp(sp, i);
    float3 sum = make_float3(0.f);
    uint32_t valid = 0;
    for (uint32_t i = 0; i < kSpecDim; ++i) {
        float p = swl.pdf[i];
        float x = spd_eval(X, s->spd_cie_interval, swl.lambda[i]) * v[i], y = spd_eval(Y, s->spd_cie_interval, swl.lambda[i]) * v[i], z = spd_eval(Z, s->spd_cie_interval, swl.lambda[i]) * v[i];
        sum = sum + make_float3(p == 0.f ? 0.f : x / p, p == 0.f ? 0.f : y / p, p == 0.f ? 0.f : z / p);
        valid += p > 0.f ? 1u : 0u;
    }
    float factor = 1.f / ((float) valid * s->cie_y_integral);
    float3 xyz = sum * factor;
    return {3.240479f * xyz.x + -1.537150f * xyz.y + -0.498535f * xyz.z,
            -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
            0.055648f * xyz.x + -0.204043f * xyz.y + 1.057311f * xyz.z};
}
inline Spec eval_slot_albedo(const vmk_scene *s, const vmk_slot &sl, float2 uv) { return spec_albedo(s, eval_slot3(s, sl, uv)); }             // shader_node.cpp:317-321
inline Spec eval_slot_illumination(const vmk_scene *s, const vmk_slot &sl, float2 uv) { return spec_illumination(s, eval_slot3(s, sl, uv)); } // shader_node.cpp:329-333
inline Spec eval_slot_spd(const vmk_scene *s, const vmk_slot &sl, float2 uv) { // number slot that may be an "spd" node (spd.cpp:36-39)
    if (sl.tex == VMK_SLOT_SPD) return spd_eval3(s->spd_data + f2u(sl.v[0]), sl.v[2], *tl_swl);
#if ORC_SPEC_DIM == 4
    return make_spec(eval_slot1(s, sl, uv)); // (a complete spectrum feeds these slots from "spd" nodes, metal.cpp:113-117; a plain number is one value)
#else
    return eval_slot3(s, sl, uv);
#endif
}

// =====================================================================================================
// a4. ray / triangle / BVH — Geometry::trace_closest / trace_occlusion (geometry.cpp:168-185)
// The reference delegates to OptiX, whose triangle test is WATERTIGHT: a ray aimed at an edge or a vertex shared by two triangles hits
// (at least) one of them.  The restatement is the published watertight test of Woop, Benthin and Wald (JCGT 2013, "Watertight
// Ray/Triangle Intersection") in float32 — vertices translated to the ray origin, permuted so that the ray's dominant axis is z, sheared
// so that the ray runs along +z, then three 2-D edge functions.  An edge function is computed from the two sheared vertices of ITS edge
// only, with commutative products and one subtraction, so the two triangles that share an edge see values that are equal or exactly
// negated: the ray cannot be outside both.  Zeros count as inside for either sign (no double-precision fallback: an edge-on ray may
// report both neighbours, and the selection rule below picks one).  The hit distance is taken from the plane equation (below).  Hit-selection rule: valid hits 0 < t < t_max, smallest t wins,
// ties resolved towards the smaller (inst, prim).  Barycentrics: u weighs p1, v weighs p2.
// =====================================================================================================
struct Ray { float3 o; float3 d; float t_max; };
struct Hit { uint32_t inst{VMK_INVALID}, prim{VMK_INVALID}; float2 bary{0, 0}; uint32_t tri{VMK_INVALID};
             bool is_miss() const { return inst == VMK_INVALID; } };

// The permutation puts the ray's dominant axis kz (first maximum of |d|) in z; the other two axes go to x and y in whichever order
// costs one select each (every quantity below is invariant under swapping them): kz = 0 -> (y, z, x), 1 -> (x, z, y), 2 -> (x, y, z).
inline float3 tri_permute(float3 v, bool k0, bool k2) {
    float3 r;
    r.x = k0 ? v.y : v.x;
    r.y = k2 ? v.y : v.z;
    r.z = k2 ? v.z : (k0 ? v.x : v.y);
    return r;
}
inline bool intersect_tri(const vmk_tri_pos &tp, float3 o, float3 d, float *t_out, float *u_out, float *v_out) {
    float3 p0 = {tp.p0[0], tp.p0[1], tp.p0[2]}, p1 = {tp.p1[0], tp.p1[1], tp.p1[2]}, p2 = {tp.p2[0], tp.p2[1], tp.p2[2]};
    // per-ray constants (the device keeps them with the ray): dominant axis, shear
    float ax = abs_(d.x), ay = abs_(d.y), az = abs_(d.z);
    bool k1 = ay > ax;
    bool k2 = az > (k1 ? ay : ax);
    bool k0 = !k1 && !k2;
    float3 dp = tri_permute(d, k0, k2);
    float Sz = 1.f / dp.z, Sx = dp.x * Sz, Sy = dp.y * Sz;
    // vertices relative to the ray origin, permuted
    float3 A = tri_permute(p0 - o, k0, k2), B = tri_permute(p1 - o, k0, k2), C = tri_permute(p2 - o, k0, k2);
    // sheared x, y: the ray runs along +z through the origin
    float Ax = A.x - Sx * A.z, Ay = A.y - Sy * A.z;
    float Bx = B.x - Sx * B.z, By = B.y - Sy * B.z;
    float Cx = C.x - Sx * C.z, Cy = C.y - Sy * C.z;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    // outside iff the edge functions disagree in sign (zeros are inside for either sign; minNum / maxNum ignore a NaN operand)
    float lo = fmin_(fmin_(U, V), W), hi = fmax_(fmax_(U, V), W);
    if (lo < 0.f && hi > 0.f) return false;
    float det = U + V + W;
    if (det == 0.f) return false;
    // The distance comes from the triangle's PLANE in the unsheared frame, not from interpolating the sheared depths (the paper's
    // T = U Az + V Bz + W Cz carries an absolute error of eps * |vertex - o|: on a large floor or wall quad that exceeds the offset a
    // spawned ray starts with, and shadow rays would hit the surface they leave):  t = (A . N) / (d' . N),  N = (C - A) x (B - A),
    // d' . N = d_z (Sx Nx + Sy Ny + Nz) = d_z * det  (the z component of a normal is what a shear along z leaves unchanged), so the one
    // reciprocal of det serves t, u and v.  For an axis-aligned quad the in-plane components of N are exact zeros and t is exact.
    float3 e1 = B - A, e2 = C - A;
    float3 N = cross(e2, e1); // this orientation has N . (Sx, Sy, 1) = U + V + W
    float inv = 1.f / det;
    float t = dot(A, N) * Sz * inv;
    // A hit lies within the triangle's own extent along the ray's dominant axis: t * d_z in [min z, max z] of the three vertices
    // (padded by 2^-14).  For a well-conditioned test this is implied; for a triangle seen edge-on (U, V, W and det all rounding noise,
    // t arbitrary) it is what keeps a hit from being reported far outside the triangle's bounding box — such a hit would be found or
    // not depending on whether its leaf was culled against an earlier hit, i.e. on the traversal order.
    float zlo = fmin_(fmin_(A.z, B.z), C.z) * Sz, zhi = fmax_(fmax_(A.z, B.z), C.z) * Sz;
    float tlo = fmin_(zlo, zhi), thi = fmax_(zlo, zhi);
    const float pad = 1.f / 16384.f;
    if (!(t * (1.f + pad) >= tlo && t * (1.f - pad) <= thi)) return false;
    *t_out = t; *u_out = V * inv; *v_out = W * inv;
    return true;
}

struct BVHNode { float bmin[3], bmax[3]; int left, right; int first, count; };
struct Counters { std::atomic<uint64_t> closest{0}, shadow{0}, nodes{0}, tris{0}, paths{0}, hits{0}, tex{0}; };
// per-thread tallies (flushed into the scene's atomics when a worker finishes): no shared cache line on the hot path
struct LocalCounters { uint64_t closest{0}, shadow{0}, nodes{0}, tris{0}, paths{0}, hits{0}, tex{0}; };
static thread_local LocalCounters tl_cnt;
// optional ray capture (orc_dump_rays): every ray the integrator traces is appended as 10 words
// [o.xyz, d.xyz, t_max, kind (0 closest / 1 occlusion), path id, sequence number within the path]
struct RayDump { std::vector<uint32_t> words; uint32_t path{0}, seq{0}; };
static thread_local RayDump *tl_dump = nullptr;

struct SceneView {
    const vmk_scene *s{};
    std::vector<BVHNode> nodes;
    std::vector<uint32_t> order; // BVH leaf order -> global triangle index
    Counters cnt;

    void flush_thread_counters() {
        cnt.closest += tl_cnt.closest; cnt.shadow += tl_cnt.shadow; cnt.nodes += tl_cnt.nodes; cnt.tris += tl_cnt.tris;
        cnt.paths += tl_cnt.paths; cnt.hits += tl_cnt.hits; cnt.tex += tl_cnt.tex;
        tl_cnt = LocalCounters{};
    }
    void build() {
        uint32_t n = s->n_tris;
        order.resize(n);
        for (uint32_t i = 0; i < n; ++i) order[i] = i;
        std::vector<float> cen(3 * (size_t) n);
        for (uint32_t i = 0; i < n; ++i) {
            const vmk_tri_pos &t = s->tri_pos[i];
            for (int a = 0; a < 3; ++a) cen[3 * (size_t) i + a] = (t.p0[a] + t.p1[a] + t.p2[a]) * (1.f / 3.f);
        }
        nodes.clear();
        nodes.reserve(2 * (size_t) n + 1);
        if (n) build_rec(0, n, cen);
    }
    int build_rec(uint32_t lo, uint32_t hi, const std::vector<float> &cen) {
        int idx = (int) nodes.size();
        nodes.push_back({});
        float bmin[3] = {1e30f, 1e30f, 1e30f}, bmax[3] = {-1e30f, -1e30f, -1e30f};
        float cmin[3] = {1e30f, 1e30f, 1e30f}, cmax[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t i = lo; i < hi; ++i) {
            const vmk_tri_pos &t = s->tri_pos[order[i]];
            for (int a = 0; a < 3; ++a) {
                bmin[a] = std::min(bmin[a], std::min(t.p0[a], std::min(t.p1[a], t.p2[a])));
                bmax[a] = std::max(bmax[a], std::max(t.p0[a], std::max(t.p1[a], t.p2[a])));
                float c = cen[3 * (size_t) order[i] + a];
                cmin[a] = std::min(cmin[a], c); cmax[a] = std::max(cmax[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) { nodes[idx].bmin[a] = bmin[a]; nodes[idx].bmax[a] = bmax[a]; }
        if (hi - lo <= 4) { nodes[idx].left = nodes[idx].right = -1; nodes[idx].first = (int) lo; nodes[idx].count = (int) (hi - lo); return idx; }
        int axis = 0;
        if (cmax[1] - cmin[1] > cmax[axis] - cmin[axis]) axis = 1;
        if (cmax[2] - cmin[2] > cmax[axis] - cmin[axis]) axis = 2;
        uint32_t mid = (lo + hi) / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](uint32_t a, uint32_t b) {
            return cen[3 * (size_t) a + axis] < cen[3 * (size_t) b + axis];
        });
        int l = build_rec(lo, mid, cen);
        int r = build_rec(mid, hi, cen);
        nodes[idx].left = l; nodes[idx].right = r; nodes[idx].count = 0;
        return idx;
    }
    // conservative slab test (double precision, padded): never rejects a box containing a valid float32 hit
    static bool hit_box(const BVHNode &nd, const Ray &r, float t_far) {
        double t0 = 0.0, t1 = (double) t_far;
        const float o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
        for (int a = 0; a < 3; ++a) {
            double pad = 1e-4 * (1.0 + std::fabs((double) nd.bmin[a]) + std::fabs((double) nd.bmax[a]));
            double lo = (double) nd.bmin[a] - pad, hi = (double) nd.bmax[a] + pad;
            if (d[a] == 0.f) { if ((double) o[a] < lo || (double) o[a] > hi) return false; continue; }
            double inv = 1.0 / (double) d[a];
            double ta = (lo - (double) o[a]) * inv, tb = (hi - (double) o[a]) * inv;
            if (ta > tb) std::swap(ta, tb);
            t0 = std::max(t0, ta); t1 = std::min(t1, tb);
            if (t0 > t1 * 1.0000001 + 1e-9) return false;
        }
        return true;
    }
    static void dump_ray(const Ray &r, uint32_t kind) {
        if (!tl_dump) return;
        float f[7] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.t_max};
        for (float x : f) tl_dump->words.push_back(f2u(x));
        tl_dump->words.push_back(kind); tl_dump->words.push_back(tl_dump->path); tl_dump->words.push_back(tl_dump->seq++);
    }
    Hit trace_closest(const Ray &r) {
        tl_cnt.closest++;
        dump_ray(r, 0);
        Hit best; float best_t = r.t_max;
        if (nodes.empty()) return best;
        int stack[128]; int sp = 0; stack[sp++] = 0;
        uint64_t nn = 0, nt = 0;
        while (sp) {
            const BVHNode &nd = nodes[stack[--sp]];
            ++nn;
            if (!hit_box(nd, r, best_t)) continue;
            if (nd.left < 0) {
                for (int i = 0; i < nd.count; ++i) {
                    uint32_t gi = order[nd.first + i];
                    const vmk_tri_pos &tp = s->tri_pos[gi];
                    float t, u, v; ++nt;
                    if (!intersect_tri(tp, r.o, r.d, &t, &u, &v)) continue;
                    if (!(t > 0.f && t < r.t_max)) continue;
                    bool better = t < best_t || (t == best_t && !best.is_miss() && (tp.inst < best.inst || (tp.inst == best.inst && tp.prim < best.prim)));
                    if (best.is_miss() && t <= best_t) better = true;
                    if (better) { best_t = t; best.inst = tp.inst; best.prim = tp.prim; best.bary = {u, v}; best.tri = gi; }
                }
            } else { stack[sp++] = nd.left; stack[sp++] = nd.right; }
        }
        tl_cnt.nodes += nn; tl_cnt.tris += nt;
        return best;
    }
    bool trace_occlusion(const Ray &r) {
        tl_cnt.shadow++;
        dump_ray(r, 1);
        if (nodes.empty()) return false;
        int stack[128]; int sp = 0; stack[sp++] = 0;
        uint64_t nn = 0, nt = 0; bool occ = false;
        while (sp && !occ) {
            const BVHNode &nd = nodes[stack[--sp]];
            ++nn;
            if (!hit_box(nd, r, r.t_max)) continue;
            if (nd.left < 0) {
                for (int i = 0; i < nd.count; ++i) {
                    float t, u, v; ++nt;
                    if (!intersect_tri(s->tri_pos[order[nd.first + i]], r.o, r.d, &t, &u, &v)) continue;
                    if (t > 0.f && t < r.t_max) { occ = true; break; }
                }
            } else { stack[sp++] = nd.left; stack[sp++] = nd.right; }
        }
        tl_cnt.nodes += nn; tl_cnt.tris += nt;
        return occ;
    }
};

// =====================================================================================================
// a5. Interaction — Geometry::compute_surface_interaction (geometry.cpp:79-166), interaction.h:87-117
// =====================================================================================================
struct Interaction {
    float3 pos, wo, ng;
    float2 uv;
    Frame shading;
    float prim_area{0.f};
    uint32_t prim_id{VMK_INVALID}, mat_id{VMK_INVALID}, light_id{VMK_INVALID};
    uint32_t med_inside{VMK_INVALID}, med_outside{VMK_INVALID}; // MediumInterface, interaction.h:121-133
    bool phase{false}; float g{0.f};                            // HenyeyGreenstein valid() / g_, interaction.h:160-175
    bool has_material() const { return mat_id != VMK_INVALID; }
    bool has_emission() const { return light_id != VMK_INVALID; }
    bool has_phase() const { return phase; }
};
inline float3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
inline float2 ld2(const float *p) { return {p[0], p[1]}; }
inline float3 triangle_lerp(float2 b, float3 a0, float3 a1, float3 a2) { // App. B: (1-u-v) p0 + u p1 + v p2
    return a0 * (1.f - b.x - b.y) + a1 * b.x + a2 * b.y;
}
inline float2 triangle_lerp2(float2 b, float2 a0, float2 a1, float2 a2) {
    float w = 1.f - b.x - b.y;
    return {a0.x * w + a1.x * b.x + a2.x * b.y, a0.y * w + a1.y * b.x + a2.y * b.y};
}
inline Interaction compute_surface_interaction(const vmk_scene *s, uint32_t tri, uint32_t inst_id, uint32_t prim_id,
                                               float2 bary, bool is_complete) {
    Interaction it;
    const vmk_instance &inst = s->instances[inst_id];
    const vmk_tri_pos &tp = s->tri_pos[tri];
    const vmk_tri_attr &ta = s->tri_attr[tri];
    it.prim_id = prim_id; it.light_id = inst.light_id; it.mat_id = inst.mat_id;
    it.med_inside = inst.inside_medium; it.med_outside = inst.outside_medium; // geometry.cpp:90
    float3 p0 = ld3(tp.p0), p1 = ld3(tp.p1), p2 = ld3(tp.p2);
    it.pos = triangle_lerp(bary, p0, p1, p2);
    float3 dp02 = p0 - p2, dp12 = p1 - p2;
    float3 ng_un = cross(dp02, dp12);
    it.prim_area = 0.5f * length(ng_un);
    float2 t0 = ld2(ta.uv0), t1 = ld2(ta.uv1), t2 = ld2(ta.uv2);
    float2 duv02 = t0 - t2, duv12 = t1 - t2;
    float det = duv02.x * duv12.y - duv02.y * duv12.x;
    bool degenerate_uv = abs_(det) < 1e-8f;
    Frame frame;
    if (is_complete) {
        float3 dp_du, dp_dv;
        if (!degenerate_uv) {
            float inv_det = 1.f / det;
            dp_du = normalize((dp02 * duv12.y - dp12 * duv02.y) * inv_det);
            dp_dv = normalize((dp02 * (-duv12.x) + dp12 * duv02.x) * inv_det);
        } else {
            dp_du = normalize(p1 - p0);
            dp_dv = normalize(p2 - p0);
        }
        frame = {dp_du, dp_dv, normalize(ng_un)};
        float3 normal = triangle_lerp(bary, ld3(ta.n0), ld3(ta.n1), ld3(ta.n2));
        it.shading = frame;
        if (!is_zero(normal)) { // geometry.cpp:130-133 + PartialDerivative::update (interaction.h:101-105)
            float3 ns = normalize(mul3x3(inst.n2w, normal));
            it.shading.z = ns;
            it.shading.x = normalize(cross(ns, it.shading.y)) * length(it.shading.x);
            it.shading.y = normalize(cross(ns, it.shading.x)) * length(it.shading.y);
        }
    } else {
        frame = {dp02, dp12, normalize(ng_un)};
        it.shading = frame;
    }
    it.uv = triangle_lerp2(bary, t0, t1, t2);
    it.ng = frame.z;
    return it;
}
// geometry.h:64-69
inline Interaction compute_surface_interaction(const vmk_scene *s, const Hit &hit, Ray &ray) {
    Interaction it = compute_surface_interaction(s, hit.tri, hit.inst, hit.prim, hit.bary, true);
    it.wo = normalize(-ray.d);
    ray.t_max = length(it.pos - ray.o) / length(ray.d);
    return it;
}

// a6. spawn rays — interaction.h:279-309, interaction.cpp:114-134
inline Ray spawn_ray(float3 pos, float3 normal, float3 dir) {
    normal = normal * (dot(normal, dir) > 0.f ? 1.f : -1.f);
    return {offset_ray_origin(pos, normal), dir, RayTMax};
}
inline Ray spawn_ray_to(float3 p_start, float3 n_start, float3 p_target) {
    float3 dir = p_target - p_start;
    n_start = n_start * (dot(n_start, dir) > 0.f ? 1.f : -1.f);
    return {offset_ray_origin(p_start, n_start), dir, 1.f - ShadowEpsilon};
}
inline float3 robust_pos(float3 pos, float3 ng, float3 dir, float factor) { // interaction.h:321-324
    float f = dot(ng, dir) > 0.f ? 1.f : -1.f;
    return offset_ray_origin(pos, (ng * f) * factor);
}

// =====================================================================================================
// a13. microfacet — base/scattering/microfacet.{h,cpp} (GGX, sample_visible = true)
// =====================================================================================================
inline float2 calculate_alpha(float alpha, float anisotropic) { // microfacet.h:43-58
    float ax = anisotropic < 0.f ? alpha / (1.f + anisotropic) : alpha * (1.f - anisotropic);
    float ay = anisotropic < 0.f ? alpha * (1.f + anisotropic) : alpha / (1.f - anisotropic);
    if (abs_(anisotropic) <= 1e-4f) return {alpha, alpha};
    return {ax, ay};
}
inline float bsdf_D(float3 wh, float ax, float ay) { // microfacet.cpp:12-23
    float3 H = {wh.x / ax, wh.y / ay, wh.z / 1.f};
    float alpha2 = ax * ay;
    return InvPi / (alpha2 * sqr(length_squared(H)));
}
inline float bsdf_lambda(float3 w, float ax, float ay) { // microfacet.cpp:41-47
    float sqr_alpha_tan_n = (sqr(ax * w.x) + sqr(ay * w.y)) / sqr(w.z);
    float ret = 0.5f * (sqrtf(1.0f + sqr_alpha_tan_n) - 1.0f);
    return w.z == 0.f ? 0.f : ret;
}
inline float bsdf_G1(float3 w, float ax, float ay) { return 1.f / (1.f + bsdf_lambda(w, ax, ay)); } // microfacet.h:97-101
inline float bsdf_G(float3 wo, float3 wi, float ax, float ay) { // microfacet.h:108-114
    return 1.f / (1.f + bsdf_lambda(wo, ax, ay) + bsdf_lambda(wi, ax, ay));
}
inline float3 sample_GGX_VNDF(float3 Ve, float2 u, float ax, float ay) { // microfacet.cpp:73-95
    float3 Vh = normalize(make_float3(ax * Ve.x, ay * Ve.y, Ve.z));
    float lenSq = Vh.x * Vh.x + Vh.y * Vh.y;
    float3 T1 = lenSq > 1e-7f ? make_float3(-Vh.y, Vh.x, 0.0f) / sqrtf(lenSq) : make_float3(1, 0, 0);
    float3 T2 = lenSq > 1e-7f ? cross(Vh, T1) : make_float3(0.0f, 1.0f, 0.0f);
    float2 t = square_to_disk(u);
    t.y = lerp_(0.5f * (1.0f + Vh.z), safe_sqrt(1.0f - sqr(t.x)), t.y);
    float3 Nh = T1 * t.x + T2 * t.y + Vh * safe_sqrt(1.0f - (t.x * t.x + t.y * t.y));
    return normalize(make_float3(ax * Nh.x, ay * Nh.y, fmax_(0.0f, Nh.z)));
}
inline float3 sample_wh(float3 wo, float2 u, float ax, float ay) { // microfacet.cpp:102-112
    bool flip = wo.z < 0.f;
    float3 wh = sample_GGX_VNDF(flip ? -wo : wo, u, ax, ay);
    return flip ? -wh : wh;
}
inline float PDF_wh(float3 wo, float3 wh, float ax, float ay) { // microfacet.cpp:152-159
    return bsdf_D(wh, ax, ay) * bsdf_G1(wo, ax, ay) * abs_dot(wo, wh) / abs_cos_theta(wo);
}
inline float PDF_wi_reflection(float3 wo, float3 wh, float ax, float ay) { // microfacet.h:134-146
    return PDF_wh(wo, wh, ax, ay) / (4.f * abs_dot(wo, wh));
}
inline float PDF_wi_transmission(float3 wo, float3 wh, float3 wi, float eta, float ax, float ay) { // microfacet.h:159-165
    float denom = sqr(dot(wi, wh) * eta + dot(wo, wh));
    float dwh_dwi = abs_dot(wi, wh) / denom;
    return PDF_wh(wo, wh, ax, ay) * dwh_dwi;
}
inline float BRDF_div_fr(float3 wo, float3 wh, float3 wi, float ax, float ay) { // microfacet.h:168-175
    return bsdf_D(wh, ax, ay) * bsdf_G(wo, wi, ax, ay) / abs_(4.f * cos_theta(wo) * cos_theta(wi));
}
inline float BTDF_div_ft(float3 wo, float3 wh, float3 wi, float eta, float ax, float ay, bool radiance) { // microfacet.cpp:166-179
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
inline bool refract(float3 wi, float3 n, float eta, float3 *wt) { // optics.h:28-39
    float cos_i = dot(n, wi);
    float sin_i_2 = fmax_(0.f, 1.f - sqr(cos_i));
    float sin_t_2 = sin_i_2 / sqr(eta);
    bool valid = sin_t_2 < 1.f;
    float cos_t = safe_sqrt(1.f - sin_t_2);
    *wt = -wi / eta + n * (cos_i / eta - cos_t);
    return valid;
}
inline float schlick_weight(float cos_t) { return pow5(clamp_(1.f - cos_t, 0.f, 1.f)); } // optics.h:42-45
inline float schlick_F0_from_ior(float ior) { return sqr((ior - 1.0f) / (ior + 1.0f)); } // optics.h:60-62
inline float schlick_ior_from_F0(float f0) { float s = sqrtf(clamp_(f0, 0.0f, 0.99f)); return (1.0f + s) / (1.0f - s); } // optics.h:65-68
inline float fresnel_dielectric(float abs_cos_i, float eta) { // optics.h:71-78
    float sin_i_2 = 1.f - sqr(abs_cos_i);
    float sin_t_2 = sin_i_2 / sqr(eta);
    float cos_t = safe_sqrt(1.f - sin_t_2);
    float r_parl = (eta * abs_cos_i - cos_t) / (eta * abs_cos_i + cos_t);
    float r_perp = (abs_cos_i - eta * cos_t) / (abs_cos_i + eta * cos_t);
    return sin_t_2 >= 1.f ? 1.f : (sqr(r_parl) + sqr(r_perp)) * 0.5f;
}
struct Cpx { float re, im; };
inline Cpx cadd(Cpx a, Cpx b) { return {a.re + b.re, a.im + b.im}; }
inline Cpx csub(Cpx a, Cpx b) { return {a.re - b.re, a.im - b.im}; }
inline Cpx cmul(Cpx a, Cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cpx cdiv(Cpx a, Cpx z) { float sc = 1.f / (z.re * z.re + z.im * z.im); return {sc * (a.re * z.re + a.im * z.im), sc * (a.im * z.re - a.re * z.im)}; }
inline float cnorm_sqr(Cpx z) { return z.re * z.re + z.im * z.im; }
inline Cpx csqrt(Cpx z) { // complex.h:61-69
    float n = sqrtf(cnorm_sqr(z));
    float t1 = sqrtf(0.5f * (n + abs_(z.re)));
    float t2 = 0.5f * z.im / t1;
    Cpx r;
    r.re = n == 0.f ? 0.f : (z.re >= 0.f ? t1 : abs_(t2));
    r.im = n == 0.f ? 0.f : (z.re >= 0.f ? t2 : u2f((f2u(t1) & 0x7fffffffu) | (f2u(z.im) & 0x80000000u)));
    return r;
}
inline float fresnel_complex(float cos_i, float eta_re, float k) { // optics.h:93-102
    Cpx eta = {eta_re, k};
    float sin_i_2 = 1.f - sqr(cos_i);
    Cpx sin_t_2 = cdiv(Cpx{sin_i_2, 0.f}, cmul(eta, eta));
    Cpx cos_t = csqrt(csub(Cpx{1.f, 0.f}, sin_t_2));
    Cpx ci = {cos_i, 0.f};
    Cpx r_parl = cdiv(csub(cmul(eta, ci), cos_t), cadd(cmul(eta, ci), cos_t));
    Cpx r_perp = cdiv(csub(ci, cmul(eta, cos_t)), cadd(ci, cmul(eta, cos_t)));
    return (cnorm_sqr(r_parl) + cnorm_sqr(r_perp)) * .5f;
}
enum FresnelKind { FR_CONSTANT, FR_CONDUCTOR, FR_DIELECTRIC, FR_SCHLICK, FR_F82 };
struct Fresnel {
    int kind{FR_CONSTANT};
    Spec a = make_spec(1.f); // conductor eta | schlick F0 | F82 F0
    Spec b = make_spec(0.f); // conductor k   | F82 B
    float eta{1.f};    // dielectric / schlick eta[0]
    bool eta_sp{false}; // hero: FresnelDielectric over an "spd" ior — a = the per-wavelength eta, eta = a.x (fresnel.h:83-91)
    Spec evaluate(float cos_t) const {
        switch (kind) {
            case FR_CONDUCTOR: return smap2(a, b, [&](float e, float k) { return fresnel_complex(cos_t, e, k); });
            case FR_DIELECTRIC: {
                if (eta_sp) return smap(a, [&](float e) { return fresnel_dielectric(cos_t, e); });
                float f = fresnel_dielectric(cos_t, eta); return make_spec(f);
            }
            case FR_SCHLICK: { // fresnel.h:60-67
                float F_real = fresnel_dielectric(cos_t, eta);
                float F0_real = schlick_F0_from_ior(eta);
                float t = clamp_(inverse_lerp(F_real, F0_real, 1.f), 0.f, 1.f);
                return lerp3(t, a, make_spec(1.f));
            }
            case FR_F82: { // fresnel.h:123-129
                float mu = saturate_(1.f - cos_t);
                float mu5 = pow5(mu);
                Spec f_schlick = lerp3(mu5, a, make_spec(1.f));
                return saturate3(f_schlick - b * cos_t * mu5 * mu);
            }
            default: return make_spec(1.f);
        }
    }
};
inline void f82_init(Fresnel &fr, Spec F82) { // fresnel.h:115-121
    const float f = 6.f / 7.f;
    const float f5 = pow5(f);
    Spec one = make_spec(1.f);
    Spec f_schlick = lerp3(f5, fr.a, one);
    fr.b = f_schlick * (7.f / (f5 * f)) * (one - F82);
}

// =====================================================================================================
// a11-a18. lobes — base/scattering/{bxdf,lobe}.{h,cpp}, render_core/material/*.cpp
// =====================================================================================================
namespace flag {
constexpr uint32_t Unset = 1, Reflection = 2, Transmission = 4, Diffuse = 8, Glossy = 16, Specular = 32, NearSpec = 64;
constexpr uint32_t DiffRefl = Diffuse | Reflection, GlossyRefl = Glossy | Reflection, GlossyTrans = Glossy | Transmission;
}
struct ScatterEval { Spec f = make_spec(0.f); float pdf{0.f}; uint32_t flags{flag::Unset}; bool valid() const { return pdf > 0.f; } };
struct BSDFSample { ScatterEval eval; float3 wi{0, 0, 0}; float eta{1.f}; bool valid() const { return eval.valid(); } };
struct SampledDirection { float3 wi{0, 0, 0}; bool valid{true}; };

enum LobeKind { LB_LAMBERT, LB_OREN_NAYAR, LB_MICROFACET, LB_FRESNEL_BLEND, LB_DIELECTRIC, LB_SHEEN, LB_PLASTIC };
struct Lobe {
    int kind{LB_LAMBERT};
    Frame frame;
    Spec kr = make_spec(1.f); // Lambert Kr / OrenNayar R / MicrofacetReflection kr / dielectric kt / sheen tint / blend Rd
    Spec rs = make_spec(0.f); // FresnelBlend Rs
    float A{0}, B{0};       // Oren-Nayar | sheen a,b
    float ax{0}, ay{0};
    Fresnel fr;
    bool compensate{false}; // PureReflectionLobe::compensate (mirror / conductor / metallic)
    uint32_t bxdf_flags{flag::DiffRefl};
    float weight{1.f}, sample_weight{1.f};
    int albedo_lut{0};      // 0: class default, 1: CoatLobe::albedo (principled_bsdf.cpp:154-160), 2: SpecularLobe::albedo (:198-205)
    float lut_x{0}, lut_z{0};
};
struct LobeSet { int n{0}; bool is_set{false}; Lobe lobes[12]; };
struct MatCtx { const vmk_scene *s; };

inline float pure_reflection_compensate(const vmk_scene *s, const Lobe &l, float3 wo) { // lobe.cpp:716-720
    float alpha = sqrtf(l.ax * l.ay); // MicrofacetBxDF::alpha_average (bxdf.h:116-118)
    float v; sample_lut2d(s->luts.pure_reflection, 1, alpha, cos_theta(wo), &v);
    return 1.f / v;
}
inline float dielectric_to_ratio_x(const Lobe &l) { return sqrtf(sqrtf(l.ax * l.ay)); } // lobe.h:250-255
inline float2 dielectric_sample_lut(const vmk_scene *s, const Lobe &l, float3 wo, float eta) { // lobe.cpp:263-285
    const float *lut = eta > 1.f ? s->luts.dielectric : s->luts.dielectric_inv;
    float x = dielectric_to_ratio_x(l);
    float y = abs_cos_theta(wo);
    float z = eta > 1.f ? inverse_lerp(eta, 1.003f, 5.f) : inverse_lerp(rcp(eta), 1.003f, 5.f);
    float out[2]; sample_lut3d(lut, 2, make_float3(x, y, z), out);
    return {out[0], out[1]};
}
inline float dielectric_refl_prob(const Lobe &l, Spec F) { // lobe.cpp:315-319
    Spec T = 1.f - F;
    Spec total = T * l.kr + F;
    return average(F) / average(total);
}

// FresnelBlend::f_specular (substrate.cpp:31-37): D(wh) / (4 |wi.wh| max(|cos_i|,|cos_o|)) * fresnel_schlick(Rs, wi.wh)
inline Spec blend_f_specular(const Lobe &l, float3 wo, float3 wi, float3 wh) {
    Spec specular = lerp3(schlick_weight(dot(wi, wh)), l.rs, make_spec(1.f)) *
                      (bsdf_D(wh, l.ax, l.ay) / (4.f * abs_dot(wi, wh) * fmax_(abs_cos_theta(wi), abs_cos_theta(wo))));
    return specular * (is_zero(wh) ? 0.f : 1.f);
}

// ---- local evaluate (Lobe::evaluate_local_impl of each lobe class) ----
inline ScatterEval eval_local(const vmk_scene *s, const Lobe &l, float3 wo, float3 wi, bool radiance, float *eta_out) {
    ScatterEval se;
    switch (l.kind) {
        case LB_LAMBERT: case LB_OREN_NAYAR: { // DiffuseLobe -> BxDF::safe_evaluate (bxdf.cpp:34-46)
            bool sh = same_hemisphere(wo, wi);
            Spec f;
            if (l.kind == LB_LAMBERT) f = l.kr * InvPi; // bxdf.h:92-95
            else { // OrenNayar::f bxdf.cpp:103-121
                float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
                float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
                float max_cos = fmax_(0.f, d_cos);
                bool cond = abs_cos_theta(wi) > abs_cos_theta(wo);
                float sin_alpha = cond ? sin_o : sin_i;
                float tan_beta = cond ? sin_i / abs_cos_theta(wi) : sin_o / abs_cos_theta(wo);
                f = l.kr * InvPi * (l.A + l.B * max_cos * sin_alpha * tan_beta);
            }
            se.f = sh ? f : make_spec(0.f);
            se.pdf = sh ? cosine_hemisphere_PDF(abs_cos_theta(wi)) : 0.f;
            se.flags = flag::DiffRefl;
            return se;
        }
        case LB_MICROFACET: { // MicrofacetLobe::evaluate_local_impl (lobe.cpp:213-218) + MicrofacetReflection (bxdf.cpp:65-78)
            bool sh = same_hemisphere(wo, wi);
            float3 wh = normalize(wo + wi);
            float3 whf = face_forward(wh, make_float3(0, 0, 1));
            Spec F = l.fr.evaluate(abs_dot(wo, whf));
            Spec f = (F * BRDF_div_fr(wo, whf, wi, l.ax, l.ay)) * l.kr;
            float pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay);
            se.f = sh ? f : make_spec(0.f);
            se.pdf = sh ? pdf : 0.f;
            se.flags = flag::GlossyRefl;
            if (l.compensate) se.f *= pure_reflection_compensate(s, l, wo); // lobe.cpp:722-729
            return se;
        }
        case LB_FRESNEL_BLEND: { // substrate.cpp:12-70 through BxDF::safe_evaluate
            bool sh = same_hemisphere(wo, wi);
            float3 wh = normalize(wi + wo);
            Spec specular = blend_f_specular(l, wo, wi, wh);
            Spec diffuse = (28.f / (23.f * Pi)) * l.kr * (make_spec(1.f) - l.rs) *
                             (1.f - pow5(1.f - .5f * abs_cos_theta(wi))) * (1.f - pow5(1.f - .5f * abs_cos_theta(wo)));
            Spec f = specular + diffuse;
            float fr = l.fr.evaluate(abs_cos_theta(wo)).x;
            float pdf = lerp_(fr, cosine_hemisphere_PDF(abs_cos_theta(wi)), PDF_wi_reflection(wo, wh, l.ax, l.ay));
            se.f = sh ? f : make_spec(0.f);
            se.pdf = sh ? pdf : 0.f;
            se.flags = flag::Reflection;
            return se;
        }
        case LB_PLASTIC: { // PlasticLobe::evaluate_local_impl plastic.cpp:31-43 (no hemisphere test of its own)
            float3 wh = normalize(wo + wi);
            Spec F = l.fr.evaluate(abs_dot(wh, wo));
            se.f = (l.kr * InvPi) * (1.f - F);
            se.f += BRDF_div_fr(wo, wh, wi, l.ax, l.ay) * F;
            se.pdf = lerp_(average(F), cosine_hemisphere_PDF(abs_cos_theta(wi)), PDF_wi_reflection(wo, wh, l.ax, l.ay));
            se.flags = flag::GlossyRefl;
            return se;
        }
        case LB_DIELECTRIC: { // lobe.cpp:321-412
            bool refl = same_hemisphere(wo, wi);
            float eta = l.fr.eta;
            float eta_p = refl ? 1.f : eta;
            if (eta_out) *eta_out = eta_p;
            float3 wh = normalize(wo + wi * eta_p);
            wh = face_forward(wh, wo);
            Spec F = l.fr.evaluate(abs_dot(wh, wo));
            float2 lut = dielectric_sample_lut(s, l, wo, eta);
            if (refl) { // evaluate_reflection lobe.cpp:340-353
                se.f = F * BRDF_div_fr(wo, wh, wi, l.ax, l.ay);
                se.pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay) * dielectric_refl_prob(l, F);
                se.flags = flag::GlossyRefl;
                se.f *= rcp(lut.x);
            } else { // evaluate_transmission lobe.cpp:355-371
                float3 new_wh = face_forward(wh, wo);
                float3 wh2 = normalize(wo + wi * eta); // GGXMicrofacet::BTDF(wo, wi, Ft, eta, tm) recomputes wh
                Spec tr = (1.f - F) * BTDF_div_ft(wo, wh2, wi, eta, l.ax, l.ay, radiance);
                se.f = tr * l.kr;
                se.pdf = PDF_wi_transmission(wo, new_wh, wi, eta, l.ax, l.ay) * (1.f - dielectric_refl_prob(l, F));
                se.flags = flag::GlossyTrans;
                se.f *= rcp(lut.x);
            }
            return se;
        }
        case LB_SHEEN: { // principled_bsdf.cpp:58-72,110-117
            float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
            float3 w = make_float3(l.A * wi.x + l.B * wi.z, l.A * wi.y, wi.z); // inv_M
            float len = length(w);
            w = w / len;
            float jacobian = sqr(l.A) / (len * len * len);
            float ltc = cosine_hemisphere_PDF(cos_theta(w)) * jacobian;
            se.f = l.kr * ltc / cos_i;
            se.pdf = ltc;
            if (cos_i < 0.f || cos_o < 0.f) se.f = make_spec(0.f);
            se.flags = flag::Unset; // ScatterEval default: SheenLTC never assigns flags
            return se;
        }
    }
    return se;
}

// ---- local direction sampling (sample_wi_local_impl of each lobe class) ----
inline SampledDirection sample_wi_local(const Lobe &l, float3 wo, Sampler &sampler) {
    SampledDirection sd;
    switch (l.kind) {
        case LB_LAMBERT: case LB_OREN_NAYAR: { // BxDF::sample_wi bxdf.cpp:48-52
            float3 wi = square_to_cosine_hemisphere(sampler.next_2d());
            wi.z = wo.z < 0.f ? -wi.z : wi.z;
            sd.wi = wi; sd.valid = true;
            return sd;
        }
        case LB_MICROFACET: { // MicrofacetReflection::sample_wi bxdf.cpp:80-84
            float3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            sd.wi = reflect(wo, wh);
            sd.valid = same_hemisphere(wo, sd.wi);
            return sd;
        }
        case LB_FRESNEL_BLEND: { // FresnelBlend::sample_wi substrate.cpp:52-69
            float2 u = sampler.next_2d();
            float fr = l.fr.evaluate(abs_cos_theta(wo)).x;
            if (u.x < fr) {
                u.x = remapping(u.x, 0.f, fr);
                float3 wh = sample_wh(wo, u, l.ax, l.ay);
                sd.wi = reflect(wo, wh);
            } else {
                u.x = remapping(u.x, fr, 1.f);
                sd.wi = square_to_cosine_hemisphere(u);
                sd.wi.z = wo.z < 0.f ? -sd.wi.z : sd.wi.z;
            }
            sd.valid = true;
            return sd;
        }
        case LB_PLASTIC: { // PlasticLobe::sample_wi_local_impl plastic.cpp:45-59: 2 + 1 draws, 2 more on the diffuse branch
            float3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            Spec F = l.fr.evaluate(abs_cos_theta(wo));
            float uc = sampler.next_1d();
            if (uc < average(F)) { sd.wi = reflect(wo, wh); sd.valid = same_hemisphere(wo, sd.wi); }
            else sd.wi = square_to_cosine_hemisphere(sampler.next_2d());
            return sd;
        }
        case LB_DIELECTRIC: { // lobe.cpp:431-449
            float3 wh = sample_wh(wo, sampler.next_2d(), l.ax, l.ay);
            float d = dot(wo, wh);
            Spec F = l.fr.evaluate(abs_(d));
            float uc = sampler.next_1d();
            if (uc < dielectric_refl_prob(l, F)) {
                sd.wi = reflect(wo, wh);
                sd.valid = same_hemisphere(wo, sd.wi);
            } else {
                bool valid = refract(wo, wh, l.fr.eta, &sd.wi);
                sd.valid = valid && !same_hemisphere(wo, sd.wi);
            }
            return sd;
        }
        case LB_SHEEN: { // principled_bsdf.cpp:100-108
            float3 wi = square_to_cosine_hemisphere(sampler.next_2d());
            wi = make_float3(wi.x / l.A - wi.z * l.B / l.A, wi.y / l.A, wi.z); // M
            sd.wi = normalize(wi);
            return sd;
        }
    }
    return sd;
}

// Lobe::evaluate_impl (lobe.cpp:53-63,65-75): world -> local with the lobe's shading frame, f *= |cos_i|
inline ScatterEval lobe_evaluate(const vmk_scene *s, const Lobe &l, float3 world_wo, float3 world_wi, bool radiance, float *eta) {
    float3 wo = l.frame.to_local(world_wo), wi = l.frame.to_local(world_wi);
    ScatterEval se = eval_local(s, l, wo, wi, radiance, eta);
    se.f *= abs_cos_theta(wi);
    return se;
}
inline float valid_world_factor(const Lobe &l, float3 wo, float3 wi) { // lobe.cpp:35-38, 373-375
    if (l.kind == LB_DIELECTRIC) return 1.f;
    return same_hemisphere(wo, wi, l.frame.z) ? 1.f : 0.f;
}
// LobeSet::evaluate_impl (lobe.cpp:673-688) or a single lobe
inline ScatterEval set_evaluate(const vmk_scene *s, const LobeSet &ls, float3 world_wo, float3 world_wi, bool radiance, float *eta) {
    if (!ls.is_set) return lobe_evaluate(s, ls.lobes[0], world_wo, world_wi, radiance, eta);
    ScatterEval ret;
    for (int i = 0; i < ls.n; ++i) {
        const Lobe &l = ls.lobes[i];
        ScatterEval se = lobe_evaluate(s, l, world_wo, world_wi, radiance, eta);
        float factor = valid_world_factor(l, world_wo, world_wi);
        se.f *= l.weight * factor;
        se.pdf *= l.sample_weight * factor;
        ret.f += se.f;
        ret.pdf += se.pdf;
        ret.flags |= se.flags;
    }
    return ret;
}
// Lobe::sample (lobe.cpp:111-119) with LobeSet::sample_wi_impl (lobe.cpp:629-658)
inline BSDFSample set_sample(const vmk_scene *s, const LobeSet &ls, float3 world_wo, Sampler &sampler, bool radiance) {
    BSDFSample ret;
    SampledDirection sd;
    if (ls.is_set) {
        float uc = sampler.next_1d();
        (void) sampler.next_2d();
        int strategy = 0;
        float sum_weights = 0.f;
        for (int i = 0; i < ls.n; ++i) {
            strategy = uc > sum_weights ? i : strategy;
            sum_weights += ls.lobes[i].sample_weight;
        }
        const Lobe &l = ls.lobes[ls.n == 1 ? 0 : strategy];
        sd = sample_wi_local(l, l.frame.to_local(world_wo), sampler);
        sd.wi = l.frame.to_world(sd.wi);
    } else {
        const Lobe &l = ls.lobes[0];
        sd = sample_wi_local(l, l.frame.to_local(world_wo), sampler);
        sd.wi = l.frame.to_world(sd.wi);
    }
    ret.wi = sd.wi;
    ret.eval = set_evaluate(s, ls, world_wo, sd.wi, radiance, &ret.eta);
    ret.eval.pdf *= sd.valid ? 1.f : 0.f;
    return ret;
}

// MaterialEvaluator::evaluate / sample with individual_ns (material.cpp:132-184)
inline ScatterEval evaluator_evaluate(const vmk_scene *s, const LobeSet &ls, float3 ng, float3 wo, float3 wi) {
    ScatterEval ret = set_evaluate(s, ls, wo, wi, true, nullptr);
    bool discard = same_hemisphere(wo, wi, ng) == ((ret.flags & flag::Transmission) != 0);
    if (discard) ret.pdf = 0.f;
    return ret;
}
inline BSDFSample evaluator_sample(const vmk_scene *s, const LobeSet &ls, float3 ng, float3 wo, Sampler &sampler) {
    BSDFSample ret = set_sample(s, ls, wo, sampler, true);
    bool discard = same_hemisphere(wo, ret.wi, ng) == ((ret.eval.flags & flag::Transmission) != 0);
    if (discard) ret.eval.pdf = 0.f;
    return ret;
}

// ---- material -> lobes (create_lobe_set of each material plugin) ----
inline float layering_weight_max(Spec layer_albedo, Spec weight) { // principled_bsdf.cpp:209-214
    Spec tmp = smap2(layer_albedo, weight, [](float a, float w) { return w == 0.f ? 0.f : a / w; });
    return max_comp(tmp);
}
inline Spec layering_weight(Spec layer_albedo, Spec weight) {
    return weight * saturate_(1.f - layering_weight_max(layer_albedo, weight));
}
// Material::compute_shading_frame (material.cpp:331-353) with detail::clamp_ns (:305-310) and PartialDerivative::update(n, s)
// (interaction.h:106-112).  The normal slot's value is used as it comes (no 2x-1 remap in the reference), the rotation
// that takes (0,0,1) to it is applied to the WORLD shading normal (as the reference writes it).  Quaternion::from_axis_angle /
// to_float3x3 live in the absent ocarina layer: restated as Rodrigues' rotation about normalize(axis) — parity unpinned (App. B).
inline float3 clamp_ns(float3 ns, float3 ng, float3 w) {
    float3 w_refl = reflect(w, ns);
    float3 w_refl_clip = same_hemisphere(w, w_refl, ng) ? w_refl : normalize(w_refl - ng * dot(w_refl, ng));
    return normalize(w_refl_clip + w);
}
inline Frame compute_shading_frame(const vmk_scene *s, const vmk_material &m, const Interaction &it) {
    Frame ret = it.shading;
    if (!(m.flags & VMK_MATF_HAS_NORMAL)) return ret;
    float3 normal = eval_slot3(s, m.normal, it.uv);
    float3 n = make_float3(0.f, 0.f, 1.f);
    float3 axis = cross(n, normal);
    float theta = acos_(clamp_(dot(n, normal), -1.f, 1.f));
    float3 world_normal = ret.z;
    float len = length(axis);
    if (len > 0.f) { // rotate ret.z about k = axis / |axis| by theta
        float3 k = axis / len;
        float st, ct; sincos_(theta, &st, &ct);
        world_normal = ret.z * ct + cross(k, ret.z) * st + k * (dot(k, ret.z) * (1.f - ct));
    }
    world_normal = normalize(world_normal);
    world_normal = clamp_ns(world_normal, it.ng, it.wo);
    world_normal = normalize(face_forward(world_normal, it.shading.z));
    float3 ss = normalize(ret.x - world_normal * dot(world_normal, ret.x));
    float3 tt = normalize(cross(world_normal, ss));
    ret.z = world_normal; ret.x = ss; ret.y = tt;
    return ret;
}
inline void microfacet_alpha(const vmk_scene *s, const vmk_material &m, int slot_r, int slot_a, float2 uv, float rmin,
                             float *ax, float *ay) { // metal.cpp:140-144, mirror.cpp:63-67, glass.cpp:245-249
    float roughness = clamp_(eval_slot1(s, m.slot[slot_r], uv), rmin, 1.f);
    float anisotropic = clamp_(eval_slot1(s, m.slot[slot_a], uv), -0.9f, 0.9f);
    roughness = (m.flags & VMK_MATF_REMAP_ROUGHNESS) ? sqr(roughness) : roughness;
    float2 a = calculate_alpha(roughness, anisotropic);
    *ax = a.x; *ay = a.y;
}
inline void build_simple_lobe(const vmk_scene *s, const vmk_material &m, const Interaction &it, Lobe &l) {
    l = Lobe{};
    l.frame = compute_shading_frame(s, m, it); // Material::compute_shading_frame (material.cpp:331-353)
    switch (m.type) {
        case VMK_MAT_DIFFUSE: { // diffuse.cpp:21-30
            l.kr = eval_slot_albedo(s, m.slot[0], it.uv);
            if (m.flags & VMK_MATF_HAS_SIGMA) { // OrenNayar ctor bxdf.cpp:94-101
                float sigma = eval_slot1(s, m.slot[1], it.uv);
                sigma = sigma * PiOver2;
                float sigma2 = sqr(sigma * sigma);
                l.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                l.B = 0.45f * sigma2 / (sigma2 + 0.09f);
                l.kind = LB_OREN_NAYAR;
            } else l.kind = LB_LAMBERT;
            l.bxdf_flags = flag::DiffRefl;
            break;
        }
        case VMK_MAT_MIRROR: { // mirror.cpp:60-74
            l.kind = LB_MICROFACET; l.kr = eval_slot_albedo(s, m.slot[0], it.uv);
            microfacet_alpha(s, m, 1, 2, it.uv, 0.0001f, &l.ax, &l.ay);
            l.fr.kind = FR_CONSTANT; l.compensate = true; l.bxdf_flags = flag::GlossyRefl;
            break;
        }
        case VMK_MAT_METAL: { // metal.cpp:137-156
            l.kind = LB_MICROFACET; l.kr = make_spec(1.f);
            microfacet_alpha(s, m, 2, 3, it.uv, 0.0001f, &l.ax, &l.ay);
            l.fr.kind = FR_CONDUCTOR; l.fr.a = eval_slot_spd(s, m.slot[0], it.uv); l.fr.b = eval_slot_spd(s, m.slot[1], it.uv);
            l.compensate = true; l.bxdf_flags = flag::GlossyRefl;
            break;
        }
        case VMK_MAT_PLASTIC: { // plastic.cpp:103-122 (same double roughness_to_alpha as substrate)
            l.kind = LB_PLASTIC;
            l.kr = eval_slot_albedo(s, m.slot[0], it.uv);
            Spec Rs = eval_slot_albedo(s, m.slot[1], it.uv);
            float ior = eval_slot1(s, m.slot[2], it.uv);
            float ax, ay; microfacet_alpha(s, m, 3, 4, it.uv, 0.0001f, &ax, &ay);
            if (m.flags & VMK_MATF_REMAP_ROUGHNESS) { ax = sqr(ax); ay = sqr(ay); }
            l.ax = clamp_(ax, 0.0001f, 1.f); l.ay = clamp_(ay, 0.0001f, 1.f);
            l.fr.kind = FR_SCHLICK; l.fr.a = schlick_F0_from_ior(ior) * Rs; l.fr.eta = ior;
            l.bxdf_flags = flag::GlossyRefl;
            break;
        }
        case VMK_MAT_METALLIC: { // metallic.cpp:42-60: MetallicLobe = PureReflectionLobe with compensation, F82-tint Fresnel
            l.kind = LB_MICROFACET; l.kr = eval_slot_albedo(s, m.slot[0], it.uv);
            microfacet_alpha(s, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay);
            l.fr.kind = FR_F82; l.fr.a = l.kr; f82_init(l.fr, eval_slot_albedo(s, m.slot[1], it.uv));
            l.compensate = true; l.bxdf_flags = flag::GlossyRefl;
            break;
        }
        case VMK_MAT_GLASS: { // glass.cpp:240-257
            l.kind = LB_DIELECTRIC; l.kr = eval_slot_albedo(s, m.slot[0], it.uv);
            float cos_t = dot(it.wo, it.ng); // Interaction::correct_eta interaction.cpp:80-83
            if (m.slot[1].tex == VMK_SLOT_SPD) { // hero, dispersive: one ior per wavelength, directions follow eta[0] (lobe.cpp:353,383,407)
                Spec iors = eval_slot_spd(s, m.slot[1], it.uv);
                iors = cos_t > 0.f ? iors : smap(iors, [](float v) { return rcp(v); });
                microfacet_alpha(s, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay);
                l.fr.kind = FR_DIELECTRIC; l.fr.eta = iors.x; l.fr.a = iors; l.fr.eta_sp = true;
                break;
            }
            float ior = eval_slot1(s, m.slot[1], it.uv);
            ior = cos_t > 0.f ? ior : rcp(ior);
            microfacet_alpha(s, m, 2, 3, it.uv, 0.01f, &l.ax, &l.ay);
            l.fr.kind = FR_DIELECTRIC; l.fr.eta = ior;
            break;
        }
        case VMK_MAT_SUBSTRATE: { // substrate.cpp:126-149
            l.kind = LB_FRESNEL_BLEND;
            l.kr = eval_slot_albedo(s, m.slot[0], it.uv); l.rs = eval_slot_albedo(s, m.slot[1], it.uv);
            float ax, ay; microfacet_alpha(s, m, 2, 3, it.uv, 0.0001f, &ax, &ay);
            if (m.flags & VMK_MATF_REMAP_ROUGHNESS) { ax = sqr(ax); ay = sqr(ay); }
            l.ax = clamp_(ax, 0.0001f, 1.f); l.ay = clamp_(ay, 0.0001f, 1.f);
            l.fr.kind = FR_DIELECTRIC; l.fr.eta = 1.5f; l.bxdf_flags = flag::Reflection;
            break;
        }
        default: break;
    }
}
inline void build_principled(const vmk_scene *s, const vmk_material &m, const Interaction &it_in, LobeSet &out) { // principled_bsdf.cpp:352-461
    Interaction it = it_in; it.shading = compute_shading_frame(s, m, it_in); // :353 — every lobe below takes it.shading as its frame
    out.is_set = true; out.n = 0;
    float2 uv = it.uv;
    Spec color = eval_slot_albedo(s, m.slot[VMK_P_COLOR], uv);
    float ior = eval_slot1(s, m.slot[VMK_P_IOR], uv);
    float roughness = clamp_(eval_slot1(s, m.slot[VMK_P_ROUGHNESS], uv), 0.0001f, 1.f);
    float anisotropic = eval_slot1(s, m.slot[VMK_P_ANISOTROPIC], uv);
    Spec specular_tint = eval_slot_albedo(s, m.slot[VMK_P_SPEC_TINT], uv);
    float aspect = sqrtf(1.f - anisotropic * 0.9f);
    float ax = fmax_(0.001f, sqr(roughness) / aspect), ay = fmax_(0.001f, sqr(roughness) * aspect);
    Spec weight = make_spec(1.f);
    float cos_t = dot(it.wo, it.ng);
    float front_factor = cos_t > 0.f ? 1.f : 0.f;
    auto push = [&](const Lobe &l) { out.lobes[out.n++] = l; };
    if (s->luts.sheen_approx) { // sheen (Approximate mode default, principled_bsdf.cpp:260)
        Spec sheen_tint = eval_slot_albedo(s, m.slot[VMK_P_SHEEN_TINT], uv);
        float sheen_weight = eval_slot1(s, m.slot[VMK_P_SHEEN_WEIGHT], uv) * front_factor;
        float sheen_roughness = eval_slot1(s, m.slot[VMK_P_SHEEN_ROUGHNESS], uv);
        Lobe l; l.kind = LB_SHEEN; l.frame = it.shading;
        float c[4]; sample_lut2d(s->luts.sheen_approx, 4, cos_t, sheen_roughness, c);
        l.A = c[0]; l.B = c[1];
        l.kr = (sheen_tint * sheen_weight * weight) * c[2];
        l.bxdf_flags = flag::GlossyRefl;
        l.sample_weight = average(l.kr); l.weight = 1.f;
        Spec albedo = l.kr;
        push(l);
        weight = layering_weight(albedo, weight);
    }
    { // coat
        float cc_weight = eval_slot1(s, m.slot[VMK_P_COAT_WEIGHT], uv) * front_factor;
        float cc_roughness = clamp_(eval_slot1(s, m.slot[VMK_P_COAT_ROUGHNESS], uv), 0.0001f, 1.f);
        cc_roughness = sqr(cc_roughness);
        float cc_ior = eval_slot1(s, m.slot[VMK_P_COAT_IOR], uv);
        Spec cc_tint = eval_slot_albedo(s, m.slot[VMK_P_COAT_TINT], uv);
        Lobe l; l.kind = LB_MICROFACET; l.frame = it.shading; l.ax = l.ay = cc_roughness;
        l.fr.kind = FR_DIELECTRIC; l.fr.eta = cc_ior;
        l.kr = (weight * cc_weight) * cc_tint;
        l.bxdf_flags = flag::GlossyRefl;
        float x = sqrtf(sqrtf(l.ax * l.ay)); // MicrofacetLobe::to_ratio_x lobe.cpp:187-192
        float z = inverse_lerp(cc_ior, 1.003f, 4.f);
        float sv; sample_lut3d(s->luts.coat, 1, make_float3(x, cos_t, z), &sv);
        Spec albedo = l.kr * sv;
        l.sample_weight = average(albedo); l.weight = 1.f;
        l.albedo_lut = 1; l.lut_x = x; l.lut_z = z;
        weight = layering_weight(albedo, weight);
        push(l);
    }
    { // metallic
        float metallic = eval_slot1(s, m.slot[VMK_P_METALLIC], uv) * front_factor;
        Lobe l; l.kind = LB_MICROFACET; l.frame = it.shading; l.ax = ax; l.ay = ay;
        l.fr.kind = FR_F82; l.fr.a = color; f82_init(l.fr, specular_tint);
        l.kr = weight * metallic; l.compensate = true; l.bxdf_flags = flag::GlossyRefl;
        l.sample_weight = metallic * average(weight); l.weight = 1.f;
        push(l);
        weight *= (1.0f - metallic);
    }
    { // transmission
        float trans_weight = eval_slot1(s, m.slot[VMK_P_TRANS_WEIGHT], uv);
        float eta = cos_t > 0.f ? ior : rcp(ior);
        Spec t_weight = weight * trans_weight;
        Lobe l; l.kind = LB_DIELECTRIC; l.frame = it.shading; l.ax = ax; l.ay = ay;
        l.fr.kind = FR_SCHLICK; l.fr.a = specular_tint * schlick_F0_from_ior(eta); l.fr.eta = eta;
        l.kr = color;
        l.sample_weight = average(t_weight); l.weight = average(t_weight);
        push(l);
        weight *= (1.0f - trans_weight);
    }
    { // specular
        float f0 = schlick_F0_from_ior(ior);
        Lobe l; l.kind = LB_MICROFACET; l.frame = it.shading; l.ax = ax; l.ay = ay;
        l.fr.kind = FR_SCHLICK; l.fr.a = specular_tint * f0; l.fr.eta = ior;
        l.kr = weight; l.bxdf_flags = flag::GlossyRefl;
        float x = sqrtf(sqrtf(ax * ay));
        float z = sqrtf(abs_((ior - 1.0f) / (ior + 1.0f))); // ior_to_ratio_z lobe.h:174-176
        float sv; sample_lut3d(s->luts.specular, 1, make_float3(x, cos_t, z), &sv);
        Spec albedo = lerp3(sv, l.fr.a, make_spec(1.f)) * l.kr;
        l.sample_weight = average(albedo); l.weight = 1.f;
        l.albedo_lut = 2; l.lut_x = x; l.lut_z = z;
        push(l);
        weight = layering_weight(albedo, weight);
    }
    { // diffuse
        Spec diff_weight = color * weight * front_factor;
        Lobe l; l.kind = LB_LAMBERT; l.frame = it.shading; l.kr = diff_weight; l.bxdf_flags = flag::DiffRefl;
        l.sample_weight = average(diff_weight); l.weight = 1.f;
        push(l);
    }
    // LobeSet::initialize -> normalize_sampled_weight (lobe.cpp:524-532)
    float weight_sum = 0.f;
    for (int i = 0; i < out.n; ++i) weight_sum += out.lobes[i].sample_weight;
    for (int i = 0; i < out.n; ++i) out.lobes[i].sample_weight = out.lobes[i].sample_weight / weight_sum;
}
inline void build_lobe_set(const vmk_scene *s, const vmk_material &m, const Interaction &it, LobeSet &out) {
    if (m.type == VMK_MAT_PRINCIPLED) { build_principled(s, m, it, out); return; }
    if (m.type == VMK_MAT_MIX || m.type == VMK_MAT_ADD) { // mix.cpp:66-71 / add.cpp:57-60; LobeSet::create_mix / create_add + flatten (lobe.cpp:495-522,534-562)
        float frac = m.type == VMK_MAT_MIX ? eval_slot1(s, m.slot[0], it.uv) : 0.f;
        float w[2] = {1.f - frac, frac};          // lobe weights
        float sw[2] = {1.f - frac, frac};         // sampling weights
        if (m.type == VMK_MAT_ADD) { w[0] = w[1] = 1.f; sw[0] = sw[1] = 1.f / (1.f + 1.f); } // {1, 1} then normalize_sampled_weight
        out.is_set = true; out.n = 0;
        for (int c = 0; c < 2; ++c) {
            const vmk_material &cm = s->materials[c == 0 ? m.child0 : m.child1];
            if (cm.type == VMK_MAT_PRINCIPLED) {
                LobeSet sub; build_principled(s, cm, it, sub);
                // flatten (lobe.cpp:546-553): parent_weight = the parent's SAMPLING weight, applied to both sub-lobe weights
                for (int i = 0; i < sub.n; ++i) { Lobe l = sub.lobes[i]; l.sample_weight *= sw[c]; l.weight *= sw[c]; out.lobes[out.n++] = l; }
            } else {
                Lobe l; build_simple_lobe(s, cm, it, l);
                l.sample_weight = sw[c]; l.weight = w[c];
                out.lobes[out.n++] = l;
            }
        }
        return;
    }
    out.is_set = false; out.n = 1;
    build_simple_lobe(s, m, it, out.lobes[0]);
}

// =====================================================================================================
// a7-a10. lights — base/illumination/lightsampler.cpp, render_core/light/area.cpp, environments/spherical.cpp,
//                 render_core/warper/alias.h, alias2d.cpp
// =====================================================================================================
struct LightEval { Spec L = make_spec(0.f); float pdf{0.f}; };
struct LightSample { LightEval eval; float3 p_light{0, 0, 0}; bool valid() const { return eval.pdf > 0.f; } };
struct LightSampleContext { float3 pos, ng; };

inline void alias_offset_u_remapped(const vmk_scene *s, uint32_t base, uint32_t size, float u, uint32_t *idx_out, float *u_remapped) { // alias.h:148-158
    u = u * (float) size;
    uint32_t idx = std::min((uint32_t) u, size - 1u);
    u = fmin_(u - (float) idx, OneMinusEpsilon);
    float prob = s->alias_prob[base + idx];
    uint32_t alias = s->alias_idx[base + idx];
    *u_remapped = u < prob ? fmin_(u / prob, OneMinusEpsilon) : fmin_((1.f - u) / (1.f - prob), OneMinusEpsilon);
    *idx_out = u < prob ? idx : alias;
}
inline float alias_PMF(const vmk_scene *s, const vmk_light &l, uint32_t i) { // alias.h:50-52
    return l.alias_integral > 0.f ? s->alias_func[l.alias_offset + i] / (l.alias_integral * (float) l.alias_count) : 0.f;
}
inline float alias_PDF(const vmk_scene *s, const vmk_light &l, uint32_t i) { // alias.h:44-46
    return l.alias_integral > 0.f ? s->alias_func[l.alias_offset + i] / l.alias_integral : 0.f;
}
struct LightCtx { const vmk_scene *s; const vmk_render_params *p; };

// LightSampler::PMF / select_light (lightsampler.cpp:159-197) over the sampler's own PMF_ / select_light_:
// uniform (uniform.cpp:13-34, punctual lights only + correct_index when the environment is sampled separately) or
// power (power.cpp:13-28: alias table over luminance(power()), all lights, the environment weighing 0 when separate)
inline float light_pmf_inner(const LightCtx &c, uint32_t index) {
    uint32_t n = c.s->n_lights;
    if (c.p->light_sampler == 1)
        return c.s->light_alias_integral > 0.f ? c.s->alias_func[c.s->light_alias_offset + index] / (c.s->light_alias_integral * (float) n) : 0.f;
    bool sep = c.p->env_separate && c.s->env_light != VMK_INVALID;
    return 1.f / (float) (sep ? n - 1u : n);
}
inline uint32_t light_select_inner(const LightCtx &c, float u) {
    uint32_t n = c.s->n_lights;
    if (c.p->light_sampler == 1) { uint32_t idx; float ur; alias_offset_u_remapped(c.s, c.s->light_alias_offset, n, u, &idx, &ur); return idx; }
    bool sep = c.p->env_separate && c.s->env_light != VMK_INVALID;
    if (sep) {
        uint32_t punctual = n - 1u;
        uint32_t idx = (uint32_t) fmin_(u * (float) punctual, (float) punctual - 1.f);
        return idx < c.s->env_light ? idx : idx + 1u; // correct_index lightsampler.cpp:33-38
    }
    return (uint32_t) fmin_(u * (float) n, (float) n - 1.f);
}
inline float light_select_PMF(const LightCtx &c, uint32_t index) {
    if (c.p->env_separate && c.s->env_light != VMK_INVALID) {
        float env_prob = c.p->env_prob;
        if (index == c.s->env_light) return env_prob;
        return (1.f - env_prob) * light_pmf_inner(c, index);
    }
    return light_pmf_inner(c, index);
}
inline void light_select(const LightCtx &c, float u, uint32_t *index, float *pmf) {
    if (c.p->env_separate && c.s->env_light != VMK_INVALID) {
        float env_prob = c.p->env_prob;
        if (u < env_prob) { *index = c.s->env_light; *pmf = env_prob; return; }
        u = remapping(u, env_prob, 1.f);
        *index = light_select_inner(c, u);
        *pmf = light_pmf_inner(c, *index) * (1.f - env_prob);
        return;
    }
    *index = light_select_inner(c, u);
    *pmf = light_pmf_inner(c, *index);
}
inline Spec area_L(const vmk_scene *s, const vmk_light &l, float2 uv, float3 ng, float3 w) { // area.cpp:91-95
    Spec radiance = eval_slot_illumination(s, l.color, uv) * l.scale;
    return radiance * ((dot(w, ng) > 0.f || l.two_sided) ? 1.f : 0.f);
}
inline float area_PDF_wi(float pdf_pos, float3 ng, float3 w) { // area.cpp:114-118
    float ret = PDF_wi(pdf_pos, ng, w);
    return (isinf_(ret) || isnan_(ret)) ? 0.f : ret;
}
inline LightSample area_sample_wi(const LightCtx &c, const vmk_light &l, const LightSampleContext &p_ref, float2 u) { // area.cpp:120-149
    const vmk_scene *s = c.s;
    uint32_t prim; float ur;
    alias_offset_u_remapped(s, l.alias_offset, l.alias_count, u.x, &prim, &ur);
    float pmf = alias_PMF(s, l, prim);
    u.x = ur;
    float2 bary = square_to_triangle(u);
    uint32_t tri = s->instances[l.inst_id].tri_offset + prim;
    Interaction it = compute_surface_interaction(s, tri, l.inst_id, prim, bary, false);
    float pdf_pos = (1.f / it.prim_area) * pmf; // LightEvalContext(it) interaction.h:372-373
    LightSample ret;
    float3 w = p_ref.pos - it.pos;
    ret.eval.L = area_L(s, l, it.uv, it.ng, w);
    ret.eval.pdf = area_PDF_wi(pdf_pos, it.ng, w);
    ret.p_light = robust_pos(it.pos, it.ng, w, c.p->ray_offset_factor);
    return ret;
}
inline Spec env_L(const vmk_scene *s, const vmk_light &l, float3 local_dir) { // spherical.cpp:60-68
    float2 uv = {spherical_phi(local_dir) * Inv2Pi, spherical_theta(local_dir) * InvPi};
    return eval_slot_illumination(s, l.color, uv) * l.scale;
}
inline float env_func_at(const vmk_scene *s, const vmk_light &l, uint32_t iu, uint32_t iv) { return s->alias_func[l.cond_offset + iv * l.res_x + iu]; }
inline float env_map_PDF(const vmk_scene *s, const vmk_light &l, float2 p) { // alias2d.cpp:102-106
    uint32_t iu = std::min((uint32_t) (p.x * (float) l.res_x), l.res_x - 1u);
    uint32_t iv = std::min((uint32_t) (p.y * (float) l.res_y), l.res_y - 1u);
    return l.alias_integral > 0.f ? env_func_at(s, l, iu, iv) / l.alias_integral : 0.f;
}
inline LightEval env_evaluate_wi(const vmk_scene *s, const vmk_light &l, float3 p_ref_pos, float3 p_light_pos) { // spherical.cpp:86-103
    LightEval ret;
    float3 world_dir = normalize(p_light_pos - p_ref_pos);
    float3 local_dir = mul3x3(l.w2o, world_dir);
    float theta = spherical_theta(local_dir), phi = spherical_phi(local_dir);
    float sin_t = sin_(theta);
    float2 uv = {phi * Inv2Pi, theta * InvPi};
    ret.L = env_L(s, l, local_dir);
    float pdf = env_map_PDF(s, l, uv) / (_2Pi * Pi * sin_t);
    ret.pdf = sin_t == 0.f ? 0.f : pdf;
    return ret;
}
inline LightSample env_sample_wi(const vmk_scene *s, const vmk_light &l, const LightSampleContext &p_ref, float2 u) { // spherical.cpp:105-125,162-168; alias2d.cpp:110-129
    uint32_t iv; float urv;
    alias_offset_u_remapped(s, l.alias_offset, l.alias_count, u.y, &iv, &urv);
    float fv = ((float) iv + urv) / (float) l.alias_count;
    float pdf_v = alias_PDF(s, l, iv);
    uint32_t buffer_offset = l.res_x * iv;
    uint32_t iu; float uru;
    alias_offset_u_remapped(s, l.cond_offset + buffer_offset, l.res_x, u.x, &iu, &uru);
    float fu = ((float) iu + uru) / (float) l.res_x;
    float integral_u = s->alias_func[l.alias_offset + iv];
    float func_u = s->alias_func[l.cond_offset + buffer_offset + iu];
    float pdf_u = integral_u > 0.f ? func_u / integral_u : 0.f;
    float pdf_map = pdf_u * pdf_v;
    float2 uv = {fu, fv};
    LightSample ret;
    float theta = uv.y * Pi, phi = uv.x * _2Pi;
    float sin_t, cos_t; sincos_(theta, &sin_t, &cos_t);
    float3 local_dir = spherical_direction(sin_t, cos_t, phi);
    float3 world_dir = normalize(mul3x3(l.o2w, local_dir));
    float pdf_dir = pdf_map / (_2Pi * Pi * sin_t);
    ret.eval.pdf = isinf_(pdf_dir) ? 0.f : pdf_dir;
    ret.eval.L = env_L(s, l, local_dir);
    ret.p_light = p_ref.pos + world_dir * l.world_diameter;
    return ret;
}
// IPointLight::sample_wi (light.cpp:49-58) with PointLight::Le (point.cpp:43-48) / SpotLight::Le + falloff (spot.cpp:56-79);
// PDF_wi = -1 marks a delta light (light.h:227-231)
inline LightSample point_sample_wi(const vmk_scene *s, const vmk_light &l, const LightSampleContext &p_ref) {
    LightSample ls;
    float3 pos = ld3(l.position);
    if (l.type == VMK_LIGHT_PROJECTOR) { // Projector::Le projector.cpp:98-110
        float3 p = transform_point4(l.w2o4, p_ref.pos);
        float d2 = length_squared(p);
        bool valid = p.z > 0.f;
        p = p / p.z;
        float2 tan_xy = make_float2(l.tan_xy[0], l.tan_xy[1]);
        float2 uv = make_float2((p.x + tan_xy.x) / (2.f * tan_xy.x), (p.y + tan_xy.y) / (2.f * tan_xy.y));
        valid = valid && uv.x >= 0.f && uv.x <= 1.f && uv.y >= 0.f && uv.y <= 1.f;
        ls.eval.L = valid ? ((1.f * eval_slot_illumination(s, l.color, uv)) / d2) * l.scale : make_spec(0.f); // select(valid, 1, 0) * colour / d2 * scale
        ls.eval.pdf = -1.f;
        ls.p_light = pos;
        return ls;
    }
    float3 w_un = p_ref.pos - pos;
    Spec value = eval_slot_illumination(s, l.color, make_float2(0.f, 0.f)) * l.scale;
    if (l.type == VMK_LIGHT_SPOT) {
        float3 w = normalize(w_un);
        float cos_theta = clamp_(dot(ld3(l.direction), w), l.cos_angle, l.cos_falloff_start);
        float factor = (cos_theta - l.cos_angle) / (l.cos_falloff_start - l.cos_angle);
        ls.eval.L = value / length_squared(w_un) * pow4(factor);
    } else ls.eval.L = value / length_squared(w_un);
    ls.eval.pdf = -1.f;
    ls.p_light = pos;
    return ls;
}
inline LightSample light_sample_wi(const LightCtx &c, const LightSampleContext &lsc, Sampler &sampler) { // lightsampler.cpp:199-216
    float u_light = sampler.next_1d();
    float2 u_surface = sampler.next_2d();
    uint32_t index; float pmf;
    light_select(c, u_light, &index, &pmf);
    const vmk_light &l = c.s->lights[index];
    LightSample ls;
    switch (l.type) {
        case VMK_LIGHT_AREA: ls = area_sample_wi(c, l, lsc, u_surface); break;
        case VMK_LIGHT_SPHERICAL: ls = env_sample_wi(c.s, l, lsc, u_surface); break;
        default: ls = point_sample_wi(c.s, l, lsc); break; // point / spot (u_surface unused, light.cpp:49-58)
    }
    ls.eval.pdf *= pmf;
    return ls;
}
inline LightEval light_evaluate_hit_wi(const LightCtx &c, const LightSampleContext &p_ref, const Interaction &it) { // lightsampler.cpp:252-267
    LightEval ret;
    const vmk_light &l = c.s->lights[it.light_id];
    if (l.type != VMK_LIGHT_AREA) return ret;
    float pdf_pos = (1.f / it.prim_area) * alias_PMF(c.s, l, it.prim_id);
    float3 w = p_ref.pos - it.pos;
    ret.L = area_L(c.s, l, it.uv, it.ng, w);
    ret.pdf = area_PDF_wi(pdf_pos, it.ng, w);
    ret.pdf *= light_select_PMF(c, it.light_id);
    return ret;
}
inline LightEval light_evaluate_miss_wi(const LightCtx &c, const LightSampleContext &p_ref, float3 wi) { // lightsampler.cpp:290-300
    const vmk_light &l = c.s->lights[c.s->env_light];
    LightEval ret = env_evaluate_wi(c.s, l, p_ref.pos, p_ref.pos + wi);
    ret.pdf *= light_select_PMF(c, c.s->env_light);
    return ret;
}

// =====================================================================================================
// a3. ray generation — sampler.h:65-73, box.cpp:16-20, triangle.cpp:16-18, fitted_curve.h:76-113,
//     sensor.cpp:44-56, thin_lens.cpp:34-42
// =====================================================================================================
inline void table_offset(const float *prob, const uint32_t *alias, uint32_t size, float u, uint32_t *idx_out, float *u_remapped) {
    u = u * (float) size;
    uint32_t idx = std::min((uint32_t) u, size - 1u);
    u = fmin_(u - (float) idx, OneMinusEpsilon);
    float p = prob[idx];
    *u_remapped = u < p ? fmin_(u / p, OneMinusEpsilon) : fmin_((1.f - u) / (1.f - p), OneMinusEpsilon);
    *idx_out = u < p ? idx : alias[idx];
}
inline float2 filter_sample(const vmk_render_params &p, float2 u) {
    if (p.filter_type == VMK_FILTER_BOX) return {lerp_(u.x, -p.filter_radius[0], p.filter_radius[0]), lerp_(u.y, -p.filter_radius[1], p.filter_radius[1])};
    if (p.filter_type == VMK_FILTER_TRIANGLE) return {sample_tent(u.x, p.filter_radius[0]), sample_tent(u.y, p.filter_radius[1])};
    // FilterSampler::sample (fitted_curve.h:76-83): alias-2D over |f| on the positive quadrant, mirrored by sign(u)
    const uint32_t N = VMK_FILTER_TABLE_SIZE;
    float2 v = {u.x * 2.f - 1.f, u.y * 2.f - 1.f};
    float2 a = {abs_(v.x), abs_(v.y)};
    uint32_t iv; float urv; table_offset(p.filter_marginal_prob, p.filter_marginal_alias, N, a.y, &iv, &urv);
    float fv = ((float) iv + urv) / (float) N;
    uint32_t iu; float uru; table_offset(p.filter_cond_prob + iv * N, p.filter_cond_alias + iv * N, N, a.x, &iu, &uru);
    float fu = ((float) iu + uru) / (float) N;
    float sx = v.x > 0.f ? 1.f : (v.x < 0.f ? -1.f : 0.f), sy = v.y > 0.f ? 1.f : (v.y < 0.f ? -1.f : 0.f);
    return {fu * sx * p.filter_radius[0], fv * sy * p.filter_radius[1]};
}
inline Ray generate_ray(const vmk_render_params &p, uint32_t px, uint32_t py, Sampler &sampler, float2 *p_film_out = nullptr) {
    float2 fs = filter_sample(p, sampler.next_2d());
    float2 p_film = {(float) px + 0.5f + fs.x, (float) py + 0.5f + fs.y};
    if (p_film_out) *p_film_out = p_film;
    float2 p_lens_u = sampler.next_2d();
    (void) sampler.next_1d(); // time
    float3 p_sensor = transform_point4(p.raster_to_sensor, make_float3(p_film.x, p_film.y, 0.f));
    float3 dir = normalize(p_sensor);
    float2 pl = square_to_disk(p_lens_u) * p.lens_radius;
    float ft = p.focal_distance / dir.z;
    float3 p_focus = dir * ft; // ray->at(ft) with origin 0
    float3 org = make_float3(pl.x, pl.y, 0.f);
    dir = normalize(p_focus - org);
    Ray r;
    r.o = transform_point4(p.c2w, org);
    r.d = transform_vector4(p.c2w, dir);
    r.t_max = RayTMax;
    return r;
}

// =====================================================================================================
// §8f-3. AOVs of the primary hit — FrameBuffer::compile_compute_geom (frame_buffer.cpp:156-219): shading normal,
//        linear depth (sensor.cpp:192-195), MaterialEvaluator::albedo (material.cpp:91-98) = LobeSet::albedo
//        (lobe.cpp:564-570) over the per-class Lobe::albedo, emission via evaluate_hit_wi
// =====================================================================================================
inline Spec lobe_albedo(const vmk_scene *s, const Lobe &l, float cos_theta) {
    switch (l.kind) {
        case LB_LAMBERT: case LB_OREN_NAYAR: case LB_FRESNEL_BLEND: return l.kr; // bxdf.h:91,153; substrate.cpp:22
        case LB_MICROFACET: {
            if (l.albedo_lut == 1) { float sv; sample_lut3d(s->luts.coat, 1, make_float3(l.lut_x, cos_theta, l.lut_z), &sv); return sv * l.kr; }
            if (l.albedo_lut == 2) { float sv; sample_lut3d(s->luts.specular, 1, make_float3(l.lut_x, cos_theta, l.lut_z), &sv); return lerp3(sv, l.fr.a, make_spec(1.f)) * l.kr; }
            return l.kr * l.fr.evaluate(cos_theta); // MicrofacetLobe::albedo lobe.cpp:208-210
        }
        case LB_PLASTIC: return l.fr.evaluate(cos_theta); // MicrofacetLobe::albedo with the specular bxdf's kr = 1 (plastic.cpp:119)
        case LB_DIELECTRIC: { Spec F = l.fr.evaluate(abs_(cos_theta)); return l.kr * (1.f - F) + F; } // lobe.cpp:308-313
        default: return l.kr; // sheen: its directional albedo is folded into kr at build time (principled_bsdf.cpp:54-57)
    }
}
struct PixelAov { float3 normal, albedo, emission; float depth; float2 motion; bool hit; };
inline Swl *path_wavelengths(const vmk_scene *s, uint32_t x, uint32_t y, uint32_t frame, Swl &storage);
inline PixelAov primary_aov(SceneView &sv, const vmk_render_params &p, const float *w2c, const float *s2r, uint32_t px, uint32_t py, uint32_t frame) {
    PixelAov a{make_float3(0.f), make_float3(0.f), make_float3(0.f), 0.f, make_float2(0.f, 0.f), false};
    const vmk_scene *s = sv.s;
    Swl swl_store;
    tl_swl = path_wavelengths(s, px, py, frame, swl_store); // RenderEnv::initial (frame_buffer.cpp:169-170); null for srgb
    Sampler sampler; sampler.start(px, py, frame, 0);
    float2 p_film;
    Ray ray = generate_ray(p, px, py, sampler, &p_film);
    Hit hit = sv.trace_closest(ray);
    if (hit.is_miss()) return a;
    a.hit = true;
    Interaction it = compute_surface_interaction(s, hit, ray);
    a.normal = it.shading.z;
    a.depth = w2c[2] * it.pos.x + w2c[6] * it.pos.y + w2c[10] * it.pos.z + w2c[14]; // transform_point(w2c, pos).z, column-major
    { // compute_motion_vec (frame_buffer.cpp:483-491) against Sensor::prev_raster_coord (sensor.cpp:95-100); prev == current camera
        float3 ps = transform_point4(w2c, it.pos);
        ps = ps / ps.z;
        float3 rc = transform_point4(s2r, ps);
        a.motion = make_float2(p_film.x - rc.x, p_film.y - rc.y);
    }
    if (it.has_material()) {
        LobeSet lobes; build_lobe_set(s, s->materials[it.mat_id], it, lobes);
        float cos_theta = dot(it.shading.z, it.wo);
        // frame_buffer.cpp:192-196: linear_srgb(bsdf.albedo(wo), swl) — the lobe set's weighted sum as a spectrum, converted once
        Spec sum = make_spec(0.f);
        if (!lobes.is_set) sum = lobe_albedo(s, lobes.lobes[0], cos_theta);
        else for (int i = 0; i < lobes.n; ++i) sum += lobe_albedo(s, lobes.lobes[i], cos_theta) * lobes.lobes[i].weight;
        a.albedo = spec_linear_srgb(s, sum);
    }
    if (it.has_emission()) {
        LightCtx lc{s, &p};
        LightSampleContext p_ref{ray.o, ray.d};
        a.emission = spec_linear_srgb(s, light_evaluate_hit_wi(lc, p_ref, it).L); // :197-203
    }
    return a;
}

// =====================================================================================================
// §8f-1. Homogeneous medium + Henyey-Greenstein phase function — render_core/medium/homogeneous.cpp:30-70,
//        base/scattering/interaction.h:136-139, interaction.cpp:12-32,114-134, geometry.cpp:187-199
// =====================================================================================================
// sigma_t / sigma_s as spectra: decode_to_unbound_spectrum of the RGB coefficients (homogeneous.cpp:30-31,34,54-55)
inline Spec medium_sigma_t(const vmk_scene *s, const vmk_medium &m) { return spec_unbound(s, (ld3(m.sigma_a) + ld3(m.sigma_s)) * m.scale); }
inline Spec medium_sigma_s(const vmk_scene *s, const vmk_medium &m) { return spec_unbound(s, ld3(m.sigma_s) * m.scale); }
inline Spec exp3(Spec v) { return smap(v, [](float x) { return exp_(x); }); }
inline Spec medium_Tr(const vmk_scene *s, const vmk_medium &m, float t) { return exp3((-1.f * medium_sigma_t(s, m)) * fmin_(RayTMax, t)); } // :33-36
inline Spec medium_Tr_ray(const vmk_scene *s, const vmk_medium &m, const Ray &r) { return medium_Tr(s, m, length(r.d) * r.t_max); }        // :45-48
// Geometry::Tr geometry.cpp:187-199
inline Spec geometry_Tr(const vmk_scene *s, const vmk_render_params &p, const Ray &r, uint32_t medium) {
    if (p.process_mediums && medium != VMK_INVALID) return medium_Tr_ray(s, s->mediums[medium], r);
    return make_spec(1.f);
}
// the medium a ray spawned at `it` towards dir travels in — Interaction::spawn_ray_state interaction.cpp:114-123
inline uint32_t spawn_medium(const vmk_render_params &p, const Interaction &it, float3 dir) {
    if (!p.process_mediums) return VMK_INVALID;
    return dot(it.ng, dir) > 0.f ? it.med_outside : it.med_inside;
}
inline float phase_HG(float cos_theta, float g) { // interaction.h:136-139
    float denom = 1.f + sqr(g) + 2.f * g * cos_theta;
    return Inv4Pi * (1.f - sqr(g)) / (denom * sqrtf(denom));
}
// HomogeneousMedium::sample homogeneous.cpp:50-70 (2 draws); may replace `it` by a medium interaction
inline Spec medium_sample(const vmk_scene *s, const vmk_medium &m, uint32_t medium_id, const Ray &ray, Interaction &it, Sampler &sampler) {
    Spec sigma_t = medium_sigma_t(s, m), sigma_s = medium_sigma_s(s, m);
    uint32_t channel = (uint32_t) (sampler.next_1d() * (float) kSpecDim); if (channel > kSpecDim - 1u) channel = kSpecDim - 1u; // min(uint(u * dimension), dimension - 1)
    float st_c = scomp(sigma_t, channel);
    float dist = -log_(1.f - sampler.next_1d()) / st_c;
    float t = fmin_(dist / length(ray.d), ray.t_max);
    bool sampled_medium = t < ray.t_max;
    if (sampled_medium) { // Interaction(ray->at(t), -ray->direction(), true); init_phase; set_medium
        Interaction mi;
        mi.pos = ray.o + ray.d * t; mi.wo = -1.f * ray.d; mi.ng = make_float3(0.f);
        mi.uv = make_float2(0.f, 0.f);
        mi.phase = true; mi.g = m.g;
        mi.med_inside = medium_id; mi.med_outside = medium_id;
        it = mi;
    }
    Spec tr = medium_Tr(s, m, t);
    Spec density = sampled_medium ? sigma_t * tr : tr;
    float pdf = average(density);
    return sampled_medium ? tr * sigma_s / pdf : tr / pdf;
}
// HenyeyGreenstein::sample interaction.cpp:16-32 (2 draws)
inline float3 hg_sample(float3 wo, float g, Sampler &sampler, float *f_out) {
    float2 u = sampler.next_2d();
    float sqr_term = (1.f - sqr(g)) / (1.f + g - 2.f * g * u.x);
    float cos_theta = -(1.f + sqr(g) - sqr(sqr_term)) / (2.f * g);
    cos_theta = abs_(g) < 1e-3f ? 1.f - 2.f * u.x : cos_theta;
    float sin_theta = safe_sqrt(1.f - sqr(cos_theta));
    float phi = 2.f * Pi * u.y;
    float3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    float sp, cp; sincos_(phi, &sp, &cp);
    float3 wi = sin_theta * cp * v1 + sin_theta * sp * v2 + cos_theta * wo; // spherical_direction(sin, cos, phi, x, y, z)
    *f_out = phase_HG(cos_theta, g);
    return wi;
}

// =====================================================================================================
// a20. IlluminationIntegrator::Li — base/integral/integrator.cpp:160-311, direct_lighting :20-37,
//      evaluate_miss :137-158
// =====================================================================================================
struct PathStats { uint32_t closest{0}, shadow{0}, hits{0}; };

// `swl`: the path's SampledWavelengths (RenderEnv::initial integrator.cpp:48-57), hero spectrum only.  L is linear sRGB
// (every contribution goes through Spectrum::linear_srgb when it is added), T a spectrum.
inline float3 Li(SceneView &sv, const vmk_render_params &p, Ray ray, Sampler &sampler, float *dbg = nullptr, Swl *swl = nullptr) {
    int vtx = 0;
    const vmk_scene *s = sv.s;
    tl_swl = swl;
    LightCtx lc{s, &p};
    float3 L = make_float3(0.f);
    Spec T = make_spec(1.f);
    float scatter_pdf = 1e16f;
    float eta_scale = 1.f;
    float3 prev_surface_ng = ray.d;
    uint32_t ray_medium = p.process_mediums ? p.camera_medium : VMK_INVALID; // RayState::medium, sensor.cpp:48
    auto correct_bsdf_weight = [&](float weight, uint32_t bounce) { // integrator.h:146-159
        if (p.mis_mode == 2) return 1.f;
        if (p.mis_mode == 1) return bounce == 0 ? weight : 0.f;
        return weight;
    };
    const float3 primary_dir = ray.d;
    // mis_bsdf (integrator.cpp:175-232): trace, environment on a miss, medium sampling, pass-through, emitter on a hit.
    // Returns 0 = the caller goes on with `it` (NEE + scattering), 1 = `$super_break`, 2 = `$super_continue`.
    auto mis_bsdf = [&](uint32_t &bounces, bool inner, Interaction &it) -> int {
        if (!inner && ray.d.x == primary_dir.x && ray.d.y == primary_dir.y && ray.d.z == primary_dir.z) return 1; // primary_miss :178-183 (before the trace is used)
        Hit hit = sv.trace_closest(ray);
        float *rec = (dbg && vtx < 8) ? dbg + 8 * vtx : nullptr;
        ++vtx;
        if (rec) { rec[0] = u2f(hit.inst); rec[1] = u2f(hit.prim); rec[2] = hit.bary.x; rec[3] = hit.bary.y; }
        if (hit.is_miss()) { // evaluate_miss :137-158
            if (s->env_light != VMK_INVALID) {
                LightSampleContext p_ref{ray.o, prev_surface_ng};
                Spec tr = make_spec(1.f);
                if (p.process_mediums) { // :146-151: rs.ray.dir_max.w = world_diameter; tr = geometry.Tr(scene, swl, rs)
                    ray.t_max = s->lights[s->env_light].world_diameter;
                    tr = geometry_Tr(s, p, ray, ray_medium);
                }
                LightEval eval = light_evaluate_miss_wi(lc, p_ref, ray.d);
                float weight = correct_bsdf_weight(MIS_weight(scatter_pdf, eval.pdf), bounces);
                L += spec_linear_srgb(s, (eval.L * tr * weight) * T);
            }
            return 1;
        }
        it = compute_surface_interaction(s, hit, ray);
        if (p.process_mediums && ray_medium != VMK_INVALID) // integrator.cpp:199-206
            T *= medium_sample(s, s->mediums[ray_medium], ray_medium, ray, it, sampler);
        if (!it.has_material() && !it.has_phase()) { // integrator.cpp:208-214
            ray_medium = spawn_medium(p, it, ray.d);
            ray = spawn_ray(it.pos, it.ng, ray.d);
            bounces -= 1;
            return 2;
        }
        if (!it.has_phase() && inner) tl_cnt.hits++; // (counter of shaded surface vertices: one per shadow ray)
        if (it.has_emission()) { // integrator.cpp:221-231
            LightSampleContext p_ref{ray.o, prev_surface_ng};
            LightEval eval = light_evaluate_hit_wi(lc, p_ref, it);
            Spec tr = geometry_Tr(s, p, ray, ray_medium);
            float weight = correct_bsdf_weight(MIS_weight(scatter_pdf, eval.pdf), bounces);
            L += spec_linear_srgb(s, eval.L * T * weight * tr);
        }
        prev_surface_ng = it.ng;
        return 0;
    };
    for (uint32_t bounces = 0; bounces < p.max_depth; ++bounces) {
        Interaction it;
        int st = mis_bsdf(bounces, true, it);
        if (st == 1) break;
        if (st == 2) continue;
        float *rec = (dbg && vtx - 1 < 8) ? dbg + 8 * (vtx - 1) : nullptr;
        // NEE
        LightSampleContext lsc{it.pos, it.ng};
        LightSample ls = light_sample_wi(lc, lsc, sampler);
        Ray shadow_ray = spawn_ray_to(it.pos, it.ng, ls.p_light);
        bool occluded = sv.trace_occlusion(shadow_ray);
        Spec tr = geometry_Tr(s, p, shadow_ray, spawn_medium(p, it, shadow_ray.d)); // geometry.cpp:176-185, integrator.cpp:244
        // direct_lighting (integrator.cpp:20-37) via direct_light_mis (integrator.h:164-176)
        float3 wi = normalize(ls.p_light - it.pos);
        ScatterEval scatter_eval;
        BSDFSample bs;
        if (it.has_phase()) { // integrator.cpp:271-279: the phase function stands in for the BSDF
            float f = phase_HG(dot(it.wo, wi), it.g);
            scatter_eval.f = make_spec(f); scatter_eval.pdf = f; scatter_eval.flags = 0;
            float fs;
            bs.wi = hg_sample(it.wo, it.g, sampler, &fs);
            bs.eval.f = make_spec(fs); bs.eval.pdf = fs; bs.eval.flags = 0;
        } else {
            LobeSet lobes;
            build_lobe_set(s, s->materials[it.mat_id], it, lobes);
            // SampledWavelengths::check_dispersive (spectrum.cpp:32-39, integrator.cpp:264): a dispersive lobe keeps the hero wavelength only
            if (swl && s->materials[it.mat_id].type == VMK_MAT_GLASS && (s->materials[it.mat_id].flags & VMK_MATF_DISPERSIVE)) { for (uint32_t i = 1; i < kSpecDim; ++i) swl->pdf[i] = 0.f; } // invalidation_secondary spectrum.cpp:21-30
            scatter_eval = evaluator_evaluate(s, lobes, it.ng, it.wo, wi);
            bs = evaluator_sample(s, lobes, it.ng, it.wo, sampler);
        }
        if (rec) { rec[4] = ls.eval.pdf; rec[5] = scatter_eval.pdf; rec[6] = bs.eval.pdf; rec[7] = occluded ? 1.f : 0.f; }
        bool is_delta_light = ls.eval.pdf < 0.f;
        bool mis = p.mis_mode != 1;
        float weight = mis ? (is_delta_light ? 1.f : MIS_weight(ls.eval.pdf, scatter_eval.pdf)) : 1.f;
        ls.eval.pdf = is_delta_light ? -ls.eval.pdf : ls.eval.pdf;
        Spec Ld = make_spec(0.f);
        if (!occluded && scatter_eval.valid() && ls.valid()) Ld = ls.eval.L * scatter_eval.f * weight / ls.eval.pdf;
        if (p.mis_mode == 2) Ld = Ld * 0.f;
        L += spec_linear_srgb(s, T * Ld * tr);
        eta_scale *= sqr(bs.eta);
        float lum = max_comp(T);
        if (!bs.valid() || lum == 0.f) break;
        T *= bs.eval.f / bs.eval.pdf;
        if (eta_scale * lum < p.rr_threshold && bounces >= p.min_depth) { // integrator.cpp:292-299
            float q = fmin_(0.95f, lum);
            float rr = sampler.next_1d();
            if (q < rr) break;
            T = T / q;
        }
        scatter_pdf = bs.eval.pdf;
        ray_medium = spawn_medium(p, it, bs.wi);
        ray = spawn_ray(it.pos, it.ng, bs.wi);
    }
    if (p.max_depth < 2 && p.mis_mode == 0) { // integrator.cpp:302-307: `only_direct && mis_mode_ == EBoth` — the BSDF-sampling half of the direct-light MIS
        for (uint32_t bounce = 0; bounce < 1u; ++bounce) {
            Interaction it;
            if (mis_bsdf(bounce, false, it) == 1) break;
        }
    }
    return L;
}

// RenderEnv::initial (integrator.cpp:48-57): sampler->temporary { start(pixel, frame, -1); sample_wavelength } — its own
// sampler state, 1 draw; null for the srgb spectrum
inline Swl *path_wavelengths(const vmk_scene *s, uint32_t x, uint32_t y, uint32_t frame, Swl &storage) {
    if (!is_hero(s)) return nullptr;
    Sampler tmp; tmp.start(x, y, frame, 0xFFFFFFFFu);
    storage = sample_wavelengths(tmp);
    return &storage;
}

// tile ownership: include/vmk.h vmk_tiles — owner(tx, ty) = (tx + skew * ty) mod world, skew = the smallest odd number
// >= 0.38 * world coprime to world (restated here, not linked from the product)
inline uint32_t tile_skew(uint32_t world) {
    if (world <= 2) return 1;
    uint32_t s = (uint32_t) (0.38 * (double) world + 0.999999);
    if (!(s & 1u)) ++s;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
    while (gcd(s, world) != 1) s += 2;
    return s;
}
inline bool tile_owned(const vmk_tiles *t, uint32_t px, uint32_t py, uint32_t width, uint32_t height) {
    if (!t || t->tile_size == 0 || t->world <= 1) return true;
    uint32_t tx = px / t->tile_size, ty = py / t->tile_size;
    return (tx + (uint64_t) tile_skew(t->world) * ty) % t->world == t->rank;
}

}// namespace orc

// =========================================================================================================
// C entry points (ctypes)
// =========================================================================================================
using namespace orc;

struct orc_scene_handle { SceneView sv; };

extern "C" {

uint32_t orc_spec_dim(void) { return kSpecDim; }
void *orc_scene_create(const vmk_scene *scene) {
    // each build of this file serves one SampledSpectrum dimension (omath.h): refuse the other one's scenes
    const uint32_t dim = scene->spectrum == VMK_SPECTRUM_HERO && scene->spectrum_dimension == 4 ? 4u : 3u;
    if (dim != kSpecDim) { fprintf(stderr, "oracle: scene needs the ORC_SPEC_DIM = %u build, this is %u\n", dim, kSpecDim); return nullptr; }
    init_srgb_lut();
    auto *h = new orc_scene_handle();
    h->sv.s = scene;
    h->sv.build();
    return h;
}
void orc_scene_destroy(void *h) { delete (orc_scene_handle *) h; }

// accum (width*height*4 floats) is updated in place: acc = lerp(1/(f+1), acc, L_f) (frame_buffer.cpp:117-126)
int orc_render(void *h, const vmk_render_params *p, uint32_t frame_begin, uint32_t frame_count, const vmk_tiles *tiles,
               float *accum, uint32_t n_threads, vmk_counters *counters) {
    SceneView &sv = ((orc_scene_handle *) h)->sv;
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    // work queue: one 32 x 32 pixel block per claim (the tile of vmk_tiles when tiles are given), so that a sharded or sub-sampled job —
    // bench.py's cpu_baseline renders every 4th tile — still hands every thread many items (whole rows left ~4 items per thread on a
    // 256-thread host and the measured rate FELL with the thread count)
    const uint32_t bs = (tiles && tiles->tile_size) ? tiles->tile_size : 32u;
    const uint32_t bx = (p->width + bs - 1) / bs, by = (p->height + bs - 1) / bs;
    std::atomic<uint32_t> next_block{0};
    auto worker = [&]() {
        for (;;) {
            uint32_t b = next_block.fetch_add(1);
            if (b >= bx * by) break;
            const uint32_t x0 = (b % bx) * bs, y0 = (b / bx) * bs;
            if (!tile_owned(tiles, x0, y0, p->width, p->height)) continue;
            for (uint32_t y = y0; y < std::min(y0 + bs, p->height); ++y)
            for (uint32_t x = x0; x < std::min(x0 + bs, p->width); ++x) {
                float *px = accum + ((size_t) y * p->width + x) * 4;
                float4 acc = {px[0], px[1], px[2], px[3]};
                for (uint32_t f = frame_begin; f < frame_begin + frame_count; ++f) {
                    Sampler sampler;
                    sampler.start(x, y, f, 0); // rt_geom ray generation, frame_buffer.cpp:172-177
                    Ray ray = generate_ray(*p, x, y, sampler);
                    sampler.start(x, y, f, 1); // path_tracing kernel, integrator.cpp:93
                    tl_cnt.paths++;
                    Swl swl_store; Swl *swl = path_wavelengths(sv.s, x, y, f, swl_store);
                    float3 L = Li(sv, *p, ray, sampler, nullptr, swl);
                    float a = 1.f / (float) (f + 1u);
                    float4 val = {L.x, L.y, L.z, 1.f};
                    acc = lerp4(a, acc, val);
                }
                px[0] = acc.x; px[1] = acc.y; px[2] = acc.z; px[3] = acc.w;
            }
        }
        sv.flush_thread_counters();
    };
    std::vector<std::thread> pool;
    for (uint32_t i = 1; i < n_threads; ++i) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (counters) {
        counters->closest_rays = sv.cnt.closest; counters->shadow_rays = sv.cnt.shadow; counters->nodes_visited = sv.cnt.nodes;
        counters->tris_tested = sv.cnt.tris; counters->paths = sv.cnt.paths; counters->surface_hits = sv.cnt.hits; counters->tex_fetches = sv.cnt.tex;
    }
    return 0;
}

// Ray capture for the traversal-only replay (SURVEY §8d "replay kernel reads ray buffers dumped from C3"): runs frame
// `frame` of every `stride`-th pixel in x and y single-threaded and returns every ray Li() traced, 10 words each (see
// RayDump).  out may be null to query the count; returns the number of rays (clamped to max_rays when out is given).
uint32_t orc_dump_rays(void *h, const vmk_render_params *p, uint32_t frame, uint32_t stride, uint32_t *out, uint32_t max_rays) {
    SceneView &sv = ((orc_scene_handle *) h)->sv;
    RayDump dump;
    tl_dump = &dump;
    if (stride == 0) stride = 1;
    for (uint32_t y = 0; y < p->height; y += stride)
        for (uint32_t x = 0; x < p->width; x += stride) {
            Sampler sampler;
            sampler.start(x, y, frame, 0);
            Ray ray = generate_ray(*p, x, y, sampler);
            sampler.start(x, y, frame, 1);
            dump.seq = 0;
            Swl swl_store; Swl *swl = path_wavelengths(sv.s, x, y, frame, swl_store);
            Li(sv, *p, ray, sampler, nullptr, swl);
            dump.path++;
        }
    tl_dump = nullptr;
    tl_cnt = LocalCounters{};
    uint32_t n = (uint32_t) (dump.words.size() / 10);
    if (out) { n = std::min(n, max_rays); std::memcpy(out, dump.words.data(), (size_t) n * 40); }
    return n;
}

// AOV planes of frame `frame` (frame_buffer.cpp:156-219): any output may be null.  normal/albedo/emission are RGBA (w = 1 like
// the reference's buffers), depth one float per pixel; misses leave zeros.
int orc_render_aov(void *h, const vmk_render_params *p, const float *w2c, const float *s2r, uint32_t frame, float *normal, float *albedo, float *emission, float *depth, float *motion) {
    SceneView &sv = ((orc_scene_handle *) h)->sv;
    for (uint32_t y = 0; y < p->height; ++y)
        for (uint32_t x = 0; x < p->width; ++x) {
            PixelAov a = primary_aov(sv, *p, w2c, s2r, x, y, frame);
            size_t i = (size_t) y * p->width + x;
            if (normal) { normal[4 * i] = a.normal.x; normal[4 * i + 1] = a.normal.y; normal[4 * i + 2] = a.normal.z; normal[4 * i + 3] = a.hit ? 1.f : 0.f; }
            if (albedo) { albedo[4 * i] = a.albedo.x; albedo[4 * i + 1] = a.albedo.y; albedo[4 * i + 2] = a.albedo.z; albedo[4 * i + 3] = 1.f; }
            if (emission) { emission[4 * i] = a.emission.x; emission[4 * i + 1] = a.emission.y; emission[4 * i + 2] = a.emission.z; emission[4 * i + 3] = 1.f; }
            if (depth) depth[i] = a.depth;
            if (motion) { motion[2 * i] = a.motion.x; motion[2 * i + 1] = a.motion.y; }
        }
    sv.flush_thread_counters();
    return 0;
}

void orc_reset_counters(void *h) {
    tl_cnt = LocalCounters{}; // tallies a previous orc_test_eval / orc_trace_rays left on the calling thread
    Counters &c = ((orc_scene_handle *) h)->sv.cnt;
    c.closest = 0; c.shadow = 0; c.nodes = 0; c.tris = 0; c.paths = 0; c.hits = 0; c.tex = 0;
}

int orc_trace_rays(void *h, uint32_t n, const float *org, const float *dir, const float *tmax, int any_hit, uint32_t *hit_out) {
    SceneView &sv = ((orc_scene_handle *) h)->sv;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r{{org[3 * i], org[3 * i + 1], org[3 * i + 2]}, {dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]}, tmax[i]};
        if (any_hit) { hit_out[4 * i] = sv.trace_occlusion(r) ? 1u : 0u; hit_out[4 * i + 1] = hit_out[4 * i + 2] = hit_out[4 * i + 3] = 0; }
        else {
            Hit hh = sv.trace_closest(r);
            hit_out[4 * i] = hh.inst; hit_out[4 * i + 1] = hh.prim; hit_out[4 * i + 2] = f2u(hh.bary.x); hit_out[4 * i + 3] = f2u(hh.bary.y);
        }
    }
    sv.flush_thread_counters();
    return 0;
}

// exposure / tone map / gamma — frame_buffer.cpp:72-74,135-154, tonemapper/impl.cpp:16-45, pipeline.cpp:337-354
static inline float tone1(uint32_t tm, float x) {
    if (tm == 1) { float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f; return saturate_((x * (a * x + b)) / (x * (c * x + d) + e)); }
    if (tm == 2) return x / (x + 1.f);
    return x;
}
static inline float linear_to_srgb1(float x) { return x <= 0.0031308f ? 12.92f * x : 1.055f * (float) pow((double) x, 1.0 / 2.4) - 0.055f; }
int orc_tonemap(const vmk_render_params *p, const float *accum, int final_picture, float *out) {
    size_t n = (size_t) p->width * p->height;
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) {
            float v = accum[4 * i + c];
            v = 1.f - exp_(-v * p->exposure);
            v = tone1(p->tone_mapper, v);
            if (final_picture) { v = tone1(p->tone_mapper, v); v = linear_to_srgb1(v); }
            out[4 * i + c] = v;
        }
        out[4 * i + 3] = 1.f;
    }
    return 0;
}

// ---- unit entry points mirrored by vmk_test_eval (tests/test_device_units.py documents the layouts) ----
// kind 0: rng        in: px,py,frame,dim (as float-encoded uint bits)  out: 8 draws
// kind 1: elementary in: x,y                                         out: sin x, cos x, acos clamp(x), atan2(y,x), exp(-|x|), sqrt|x|
// kind 2: warps      in: u0,u1                                       out: disk(2) coshemi(3) tri(2) tent(1)
// kind 3: microfacet in: wo(3) u(2) ax ay eta                        out: wh(3) D G1 pdf_wh Fd(eta) Fc(eta,k=3.5)
// kind 4: bsdf       in: mat_id, px,py,frame, wo(3), wi(3), uvx, uvy  out: f(3) pdf flags | wi(3) f(3) pdf eta
// kind 5: camera     in: px,py,frame                                 out: o(3) d(3)
int orc_test_eval(void *h, const vmk_render_params *p, uint32_t kind, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride) {
    SceneView *sv = h ? &((orc_scene_handle *) h)->sv : nullptr;
    for (uint32_t i = 0; i < n; ++i) {
        const float *a = in + (size_t) i * in_stride;
        float *o = out + (size_t) i * out_stride;
        switch (kind) {
            case 0: { Sampler s; s.start(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3])); for (int k = 0; k < 8; ++k) o[k] = s.next_1d(); break; }
            case 1: {
                float s, c; sincos_(a[0], &s, &c);
                o[0] = s; o[1] = c; o[2] = acos_(clamp_(a[0], -1.f, 1.f)); o[3] = atan2_(a[1], a[0]); o[4] = exp_(-abs_(a[0])); o[5] = sqrtf(abs_(a[0]));
                if (out_stride >= 7) o[6] = log_(abs_(a[0]) * 0.125f + 5.9604645e-8f);
                break;
            }
            case 2: {
                float2 u = {a[0], a[1]};
                float2 d = square_to_disk(u); float3 c = square_to_cosine_hemisphere(u); float2 t = square_to_triangle(u);
                o[0] = d.x; o[1] = d.y; o[2] = c.x; o[3] = c.y; o[4] = c.z; o[5] = t.x; o[6] = t.y; o[7] = sample_tent(a[0], 0.5f);
                break;
            }
            case 3: {
                float3 wo = normalize(make_float3(a[0], a[1], a[2]));
                float3 wh = sample_wh(wo, {a[3], a[4]}, a[5], a[6]);
                o[0] = wh.x; o[1] = wh.y; o[2] = wh.z; o[3] = bsdf_D(wh, a[5], a[6]); o[4] = bsdf_G1(wo, a[5], a[6]);
                o[5] = PDF_wh(wo, wh, a[5], a[6]); o[6] = fresnel_dielectric(abs_dot(wo, wh), a[7]); o[7] = fresnel_complex(abs_dot(wo, wh), a[7], 3.5f);
                break;
            }
            case 4: {
                const vmk_scene *s = sv->s;
                uint32_t mat_id = f2u(a[0]);
                Interaction it;
                it.pos = make_float3(0, 0, 0); it.ng = make_float3(0, 0, 1);
                it.shading = {make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, 1)};
                it.wo = normalize(make_float3(a[4], a[5], a[6]));
                it.uv = {a[10], a[11]};
                it.mat_id = mat_id;
                float3 wi = normalize(make_float3(a[7], a[8], a[9]));
                LobeSet ls; build_lobe_set(s, s->materials[mat_id], it, ls);
                ScatterEval se = evaluator_evaluate(s, ls, it.ng, it.wo, wi);
                Sampler smp; smp.start(f2u(a[1]), f2u(a[2]), f2u(a[3]), 1);
                BSDFSample bs = evaluator_sample(s, ls, it.ng, it.wo, smp);
                o[0] = se.f.x; o[1] = se.f.y; o[2] = se.f.z; o[3] = se.pdf; o[4] = u2f(se.flags);
                o[5] = bs.wi.x; o[6] = bs.wi.y; o[7] = bs.wi.z; o[8] = bs.eval.f.x; o[9] = bs.eval.f.y; o[10] = bs.eval.f.z; o[11] = bs.eval.pdf; o[12] = bs.eta;
                break;
            }
            case 5: {
                Sampler s; s.start(f2u(a[0]), f2u(a[1]), f2u(a[2]), 0);
                Ray r = generate_ray(*p, f2u(a[0]), f2u(a[1]), s);
                o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
                break;
            }
            case 6: { // whole path of one (pixel, frame): 8 floats per vertex (first 8 vertices), then L
                uint32_t px = f2u(a[0]), py = f2u(a[1]), frame = f2u(a[2]);
                Sampler s; s.start(px, py, frame, 0);
                Ray r = generate_ray(*p, px, py, s);
                s.start(px, py, frame, 1);
                for (int k = 0; k < 64; ++k) o[k] = 0.f;
                Swl swl_store; Swl *swl = path_wavelengths(sv->s, px, py, frame, swl_store);
                float3 L = Li(*sv, *p, r, s, o, swl);
                o[64] = L.x; o[65] = L.y; o[66] = L.z;
                break;
            }
            case 60: { // oracle-only, hero scenes: in (r, g, b, u) -> lambda[N], pdf[N], albedo / unbound / illumination spectra [N each], linear_srgb of each [3 each]; N = kSpecDim
                if (!sv || !is_hero(sv->s)) return -1;
                Sampler smp; smp.state = 0;
                Swl swl;
                for (uint32_t k = 0; k < kSpecDim; ++k) { // sample_wavelength with the draw replaced by the given u (hero.cpp:286-299)
                    float up = fract_(a[3] + (float) k * (1.f / (float) kSpecDim));
                    swl.lambda[k] = sample_visible_wavelength(up); swl.pdf[k] = visible_wavelength_PDF(swl.lambda[k]);
                }
                tl_swl = &swl;
                float3 rgb = make_float3(a[0], a[1], a[2]);
                Spec al = spec_albedo(sv->s, rgb), un = spec_unbound(sv->s, rgb), il = spec_illumination(sv->s, rgb);
                float3 ral = spec_linear_srgb(sv->s, al), run = spec_linear_srgb(sv->s, un), ril = spec_linear_srgb(sv->s, il);
                float *q = o;
                for (uint32_t k = 0; k < kSpecDim; ++k) *q++ = swl.lambda[k];
                for (uint32_t k = 0; k < kSpecDim; ++k) *q++ = swl.pdf[k];
                for (uint32_t k = 0; k < kSpecDim; ++k) *q++ = scomp(al, k);
                for (uint32_t k = 0; k < kSpecDim; ++k) *q++ = scomp(un, k);
                for (uint32_t k = 0; k < kSpecDim; ++k) *q++ = scomp(il, k);
                const float v[9] = {ral.x, ral.y, ral.z, run.x, run.y, run.z, ril.x, ril.y, ril.z};
                for (int k = 0; k < 9; ++k) *q++ = v[k];
                tl_swl = nullptr;
                break;
            }
            case 50: { // oracle-only: refract (optics.h:28-39; src/tests/test_bxdf.cpp:28-40 vector), Fresnel helpers
                float3 wt; bool valid = refract(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5]), a[6], &wt);
                float c = abs_(dot(make_float3(a[0], a[1], a[2]), make_float3(a[3], a[4], a[5])));
                o[0] = valid ? 1.f : 0.f; o[1] = wt.x; o[2] = wt.y; o[3] = wt.z; o[4] = fresnel_dielectric(c, a[6]); o[5] = schlick_weight(c); o[6] = schlick_F0_from_ior(a[6]);
                break;
            }
            case 51: { o[0] = u2f(tea(f2u(a[0]), f2u(a[1]))); break; }
            default: return -1;
        }
    }
    return 0;
}

// ---- albedo-table re-integration (pins the lobe code against the reference's precomputed_table.h) ----
// Material::precompute_lobe (material.h:121-163): texel (x,y,z) of a res^3 (or res^2) grid, ratio = idx/(res-1),
// sampler.start((x,y), 0, 0), Lobe::precompute_with_radio + integral_albedo (lobe.cpp:13-33) in Importance mode.
// which: 0 PureReflection (mirror.cpp:53-57, lobe.h:344-356), 1 Dielectric, 2 DielectricInv (glass.cpp:14-76),
//        3 Specular (principled_bsdf.cpp:177-190), 4 Coat (principled_bsdf.cpp:135-146)
int orc_integrate_albedo(uint32_t which, uint32_t res, uint32_t x, uint32_t y, uint32_t z, uint32_t sample_num, float *out2) {
    Sampler sampler; sampler.start(x, y, 0, 0);
    float rx = (float) x / (float) (res - 1), ry = (float) y / (float) (res - 1), rz = which == 0 ? 0.f : (float) z / (float) (res - 1);
    Lobe l;
    l.frame = {make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, 1)};
    float a = clamp_(sqr(rx), 0.001f, 1.f); // from_ratio_x lobe.cpp:183-185 / glass.cpp:36-40
    l.ax = l.ay = a;
    float cos_t = clamp_(ry, 1e-4f, 1.0f); // from_ratio_y lobe.cpp:158-164
    float3 wo = make_float3(sqrtf(1.f - sqr(cos_t)), 0.f, cos_t);
    switch (which) {
        case 0: l.kind = LB_MICROFACET; l.fr.kind = FR_CONSTANT; l.kr = make_spec(1.f); break;
        case 1: l.kind = LB_DIELECTRIC; l.fr.kind = FR_DIELECTRIC; l.fr.eta = lerp_(rz, 1.003f, 5.f); l.kr = make_spec(1.f); break;
        case 2: l.kind = LB_DIELECTRIC; l.fr.kind = FR_DIELECTRIC; l.fr.eta = rcp(lerp_(rz, 1.003f, 5.f)); l.kr = make_spec(1.f); break;
        case 3: l.kind = LB_MICROFACET; l.fr.kind = FR_SCHLICK; l.fr.a = make_spec(0.04f); l.fr.eta = schlick_ior_from_F0(pow4(rz)); l.kr = make_spec(1.f); break;
        case 4: l.kind = LB_MICROFACET; l.fr.kind = FR_DIELECTRIC; l.fr.eta = lerp_(rz, 1.003f, 4.f); l.kr = make_spec(1.f); break;
        default: return -1;
    }
    double acc0 = 0.0, acc1 = 0.0;
    static const float unit_lut[2] = {1.f, 1.f};
    for (uint32_t i = 0; i < sample_num; ++i) {
        SampledDirection sd = sample_wi_local(l, wo, sampler);
        float eta_dummy;
        ScatterEval se;
        if (l.kind == LB_DIELECTRIC) { // DielectricPrecompute::compensate() == false (glass.cpp:17)
            // evaluate without LUT compensation
            bool refl = same_hemisphere(wo, sd.wi);
            float eta = l.fr.eta, eta_p = refl ? 1.f : eta;
            float3 wh = face_forward(normalize(wo + sd.wi * eta_p), wo);
            Spec F = l.fr.evaluate(abs_dot(wh, wo));
            if (refl) { se.f = F * BRDF_div_fr(wo, wh, sd.wi, l.ax, l.ay); se.pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay) * dielectric_refl_prob(l, F); }
            else {
                float3 wh2 = normalize(wo + sd.wi * eta);
                se.f = ((1.f - F) * BTDF_div_ft(wo, wh2, sd.wi, eta, l.ax, l.ay, false)) * l.kr;
                se.pdf = PDF_wi_transmission(wo, face_forward(wh, wo), sd.wi, eta, l.ax, l.ay) * (1.f - dielectric_refl_prob(l, F));
            }
        } else {
            se = eval_local(nullptr, l, wo, sd.wi, false, &eta_dummy);
        }
        se.pdf *= sd.valid ? 1.f : 0.f;
        if (se.pdf > 0.f) {
            float r = (se.f.x / se.pdf) * abs_cos_theta(sd.wi);
            acc0 += (double) r;
            if (same_hemisphere(sd.wi, wo)) acc1 += (double) r;
        }
    }
    (void) unit_lut;
    out2[0] = (float) (acc0 / sample_num);
    out2[1] = (float) (acc1 / sample_num);
    return 0;
}


// whole table, threaded; nc = 2 floats per texel for the dielectric tables (total, reflected), else 1
int orc_integrate_albedo_table(uint32_t which, uint32_t res, uint32_t sample_num, float *out, uint32_t n_threads) {
    uint32_t depth = which == 0 ? 1u : res;
    uint32_t nc = (which == 1 || which == 2) ? 2u : 1u;
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<uint32_t> next{0};
    uint32_t total = res * res * depth;
    auto worker = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= total) break;
            uint32_t x = i % res, y = (i / res) % res, z = i / (res * res);
            float v[2];
            orc_integrate_albedo(which, res, x, y, z, sample_num, v);
            for (uint32_t c = 0; c < nc; ++c) out[(size_t) i * nc + c] = v[c];
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}

}// extern "C"
