// dmath.h — device-side math for the gfx950 path-tracing megakernel.
//
// IEEE-754 binary32 throughout; the library is compiled with -ffp-contract=off and HIP's default correctly-rounded
// division/sqrt, and every transcendental is an explicit polynomial kernel (no ocml calls), so results do not depend
// on libm/ocml versions and are reproducible bit for bit.  Restates the ocarina math semantics Vision's kernels
// rely on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VD __device__ __forceinline__

namespace vmkd {

constexpr float Pi = 3.14159265358979323846f;
constexpr float InvPi = 0.31830988618379067154f;
constexpr float Inv2Pi = 0.15915494309189533577f;
constexpr float PiOver2 = 1.57079632679489661923f;
constexpr float PiOver4 = 0.78539816339744830961f;
constexpr float _2Pi = 6.28318530717958647692f;
constexpr float OneMinusEpsilon = 0x1.fffffep-1f;
constexpr float ShadowEpsilon = 1e-4f;
constexpr float RayTMax = 3.402823466e+38f;

VD uint32_t f2u(float f) { return __float_as_uint(f); }
VD float u2f(uint32_t u) { return __uint_as_float(u); }

VD float fmin_(float a, float b) { return a < b ? a : b; }
VD float fmax_(float a, float b) { return a > b ? a : b; }
VD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
VD float saturate_(float x) { return clamp_(x, 0.f, 1.f); }
VD float sqr(float x) { return x * x; }
VD float sqrt_(float x) { return __builtin_sqrtf(x); } // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is NOT
VD float div_(float a, float b) { return a / b; }
VD float safe_sqrt(float x) { return sqrt_(fmax_(x, 0.f)); }
VD float lerp_(float t, float a, float b) { return a + t * (b - a); }
VD float inverse_lerp(float x, float a, float b) { return (x - a) / (b - a); }
VD float rcp(float x) { return 1.f / x; }
VD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
VD float pow4(float x) { float x2 = x * x; return x2 * x2; }
VD float floor_(float x) { return __builtin_floorf(x); }
VD bool isinf_(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
VD bool isnan_(float x) { return (f2u(x) & 0x7fffffffu) > 0x7f800000u; }
VD float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// ---- elementary functions: Cephes single-precision kernels, Cody-Waite reduction (valid for |x| < ~8e3) ----
VD void sincos_(float x, float *s, float *c) {
    float q = floor_(x * 0.636619772367581343f + 0.5f);
    float r = x - q * 1.5703125f;
    r = r - q * 4.837512969970703125e-4f;
    r = r - q * 7.54978995489188216e-8f;
    int k = (int) q;
    float r2 = r * r;
    float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * (-1.9515295891e-4f)));
    float cp = 1.f - 0.5f * r2 +
               r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    bool swap = (k & 1) != 0;
    float ss = swap ? cp : sp, cc = swap ? sp : cp;
    *s = (k & 2) ? -ss : ss;
    *c = (((k + 1) & 2) != 0) ? -cc : cc;
}
VD float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
VD float asin_poly(float s, float z) {
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * s + s;
}
VD float asin_(float x) {
    float a = abs_(x);
    float r;
    if (a > 0.5f) {
        float z = 0.5f * (1.f - a);
        float s = sqrt_(z);
        r = PiOver2 - 2.f * asin_poly(s, z);
    } else {
        r = asin_poly(a, a * a);
    }
    return x < 0.f ? -r : r;
}
VD float acos_(float x) {
    if (x < -0.5f) return Pi - 2.f * asin_(sqrt_(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * asin_(sqrt_(0.5f * (1.f - x)));
    return PiOver2 - asin_(x);
}
VD float atan_(float xx) {
    float x = abs_(xx);
    float y;
    if (x > 2.414213562373095f) { y = PiOver2; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = PiOver4; x = (x - 1.f) / (x + 1.f); }
    else { y = 0.f; }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return xx < 0.f ? -y : y;
}
VD float atan2_(float y, float x) {
    if (x > 0.f) return atan_(y / x);
    if (x < 0.f) return y >= 0.f ? atan_(y / x) + Pi : atan_(y / x) - Pi;
    if (y > 0.f) return PiOver2;
    if (y < 0.f) return -PiOver2;
    return 0.f;
}
VD float exp_(float x) {
    if (x > 88.f) x = 88.f;
    if (x < -87.f) return 0.f;
    float z = floor_(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int) z;
    float x2 = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
                1.6666665459e-1f) * x + 5.0000001201e-1f) * x2 + x + 1.f;
    return p * u2f((uint32_t) (n + 127) << 23);
}
VD float log_(float x) { // Cephes logf for normal x > 0 (callers pass 1 - u with u in [0,1))
    if (!(x > 0.f)) return x == 0.f ? -__builtin_inff() : __builtin_nanf("");
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 126; // x = m * 2^e, m in [0.5, 1)
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.f; } else m = m - 1.f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                   1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
               3.3333331174e-1f) * m * z;
    float fe = (float) e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    float r = m + y;
    r += 0.693359375f * fe;
    return r;
}
// hero spectrum helpers (render_core/spectrum/hero.cpp): explicit fused multiply-add where the reference writes fma(),
// rsqrt / atanh / cosh through the kernels above (ocarina's device intrinsics are unpinned)
VD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VD float fract_(float x) { return x - floor_(x); }
VD float rsqrt_(float x) { return 1.f / sqrt_(x); }
VD float atanh_(float x) { return 0.5f * log_((1.f + x) / (1.f - x)); }
VD float cosh_(float x) { return 0.5f * (exp_(x) + exp_(-x)); }
constexpr float Inv4Pi = 0.07957747154594766788f;

// ---- vectors ----
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
VD V2 mk2(float x, float y) { return {x, y}; }
VD V3 mk3(float x, float y, float z) { return {x, y, z}; }
VD V3 mk3(float v) { return {v, v, v}; }
VD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
VD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
VD V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
VD V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
VD V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
VD V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
VD V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
VD V3 operator-(float s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
VD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
VD V3 &operator+=(V3 &a, V3 b) { a = a + b; return a; }
VD V3 &operator*=(V3 &a, V3 b) { a = a * b; return a; }
VD V3 &operator*=(V3 &a, float s) { a = a * s; return a; }
VD V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
VD V2 operator*(V2 a, float s) { return {a.x * s, a.y * s}; }
VD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
VD float length_squared(V3 a) { return dot(a, a); }
VD float length(V3 a) { return sqrt_(dot(a, a)); }
VD V3 normalize(V3 a) { float inv = 1.f / sqrt_(dot(a, a)); return a * inv; }
VD float abs_dot(V3 a, V3 b) { return abs_(dot(a, b)); }
VD bool is_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
VD float max_comp(V3 a) { return fmax_(fmax_(a.x, a.y), a.z); }
VD float average(V3 a) { return (a.x + a.y + a.z) / 3.f; }
VD V3 lerp3(float t, V3 a, V3 b) { return a + (b - a) * t; }
VD V3 saturate3(V3 a) { return {saturate_(a.x), saturate_(a.y), saturate_(a.z)}; }
VD V4 lerp4(float t, V4 a, V4 b) { return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)}; }

// ---- sampled spectrum ----
// A SampledSpectrum (base/color/spectrum.h:60-170) of the scene's dimension.  Three values — the (R, G, B) channels of spectrum/srgb,
// or hero dimension 3 — are a V3: the type and the arithmetic the path code had before the dimension became a build parameter.
// VMK_SPEC_DIM = 4 (vmk_hero4.hip: spectrum/hero with "dimension": 4, cbox-prism.json:692-697) makes it four samples.  Directions,
// positions and RGB colours stay V3; only values carried per wavelength are Spec.
#ifndef VMK_SPEC_DIM
#define VMK_SPEC_DIM 3
#endif
struct S4 { float x, y, z, w; };
VD S4 operator+(S4 a, S4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
VD S4 operator-(S4 a, S4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
VD S4 operator*(S4 a, S4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
VD S4 operator/(S4 a, S4 b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
VD S4 operator*(S4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
VD S4 operator*(float s, S4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
VD S4 operator/(S4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
VD S4 operator-(float s, S4 a) { return {s - a.x, s - a.y, s - a.z, s - a.w}; }
VD S4 operator-(S4 a) { return {-a.x, -a.y, -a.z, -a.w}; }
VD S4 &operator+=(S4 &a, S4 b) { a = a + b; return a; }
VD S4 &operator*=(S4 &a, S4 b) { a = a * b; return a; }
VD S4 &operator*=(S4 &a, float s) { a = a * s; return a; }
VD bool is_zero(S4 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f && a.w == 0.f; }
VD float max_comp(S4 a) { return fmax_(fmax_(fmax_(a.x, a.y), a.z), a.w); }   // SampledSpectrum::max: a left fold (spectrum.h)
VD float average(S4 a) { return (a.x + a.y + a.z + a.w) / 4.f; }               // SampledSpectrum::average: sum() / dimension
VD S4 lerp3(float t, S4 a, S4 b) { return a + (b - a) * t; }
VD S4 saturate3(S4 a) { return {saturate_(a.x), saturate_(a.y), saturate_(a.z), saturate_(a.w)}; }
#if VMK_SPEC_DIM == 4
typedef S4 Spec;
VD Spec mks(float v) { return {v, v, v, v}; }
template<class F> VD Spec smap(Spec a, F f) { return {f(a.x), f(a.y), f(a.z), f(a.w)}; }
template<class F> VD Spec smap2(Spec a, Spec b, F f) { return {f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w)}; }
VD float scomp(Spec a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)); }
VD void sput(Spec &a, uint32_t i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else if (i == 2) a.z = v; else a.w = v; }
VD float ssum(Spec a) { return a.x + a.y + a.z + a.w; }
#elif VMK_SPEC_DIM == 3
typedef V3 Spec;
VD Spec mks(float v) { return {v, v, v}; }
template<class F> VD Spec smap(Spec a, F f) { return {f(a.x), f(a.y), f(a.z)}; }
template<class F> VD Spec smap2(Spec a, Spec b, F f) { return {f(a.x, b.x), f(a.y, b.y), f(a.z, b.z)}; }
VD float scomp(Spec a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
VD void sput(Spec &a, uint32_t i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
VD float ssum(Spec a) { return a.x + a.y + a.z; }
#else
#error "VMK_SPEC_DIM must be 3 or 4"
#endif
constexpr uint32_t kSpecDim = VMK_SPEC_DIM;

// ---- local shading geometry (z-up) ----
VD float cos_theta(V3 w) { return w.z; }
VD float abs_cos_theta(V3 w) { return abs_(w.z); }
VD float sin_theta_2(V3 w) { return fmax_(0.f, 1.f - w.z * w.z); }
VD float sin_theta(V3 w) { return sqrt_(sin_theta_2(w)); }
VD float cos_phi(V3 w) { float s = sin_theta(w); return s == 0.f ? 1.f : clamp_(w.x / s, -1.f, 1.f); }
VD float sin_phi(V3 w) { float s = sin_theta(w); return s == 0.f ? 0.f : clamp_(w.y / s, -1.f, 1.f); }
VD bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.f; }
VD bool same_hemisphere(V3 a, V3 b, V3 n) { return dot(a, n) * dot(b, n) > 0.f; }
VD V3 face_forward(V3 v, V3 n) { return dot(v, n) < 0.f ? -v : v; }
VD V3 reflect(V3 wo, V3 n) { return -wo + n * (2.f * dot(wo, n)); }
VD void coordinate_system(V3 v1, V3 *v2, V3 *v3) {
    if (abs_(v1.x) > abs_(v1.y)) { float inv = 1.f / sqrt_(v1.x * v1.x + v1.z * v1.z); *v2 = {-v1.z * inv, 0.f, v1.x * inv}; }
    else { float inv = 1.f / sqrt_(v1.y * v1.y + v1.z * v1.z); *v2 = {0.f, v1.z * inv, -v1.y * inv}; }
    *v3 = cross(v1, *v2);
}
VD V3 spherical_direction(float sin_t, float cos_t, float phi) {
    float s, c; sincos_(phi, &s, &c);
    return {sin_t * c, sin_t * s, cos_t};
}
VD float spherical_theta(V3 v) { return acos_(clamp_(v.z, -1.f, 1.f)); }
VD float spherical_phi(V3 v) { float p = atan2_(v.y, v.x); return p < 0.f ? p + _2Pi : p; }

struct Frame {
    V3 x, y, z;
    VD V3 to_local(V3 v) const { return mk3(dot(v, x), dot(v, y), dot(v, z)); }
    VD V3 to_world(V3 v) const { return x * v.x + y * v.y + z * v.z; }
};

VD V3 mul3x3(const float *m, V3 v) {
    return mk3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
VD V3 transform_vector4(const float *m, V3 v) {
    return mk3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
VD V3 transform_point4(const float *m, V3 p) {
    float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    if (w == 1.f) return mk3(x, y, z);
    return mk3(x / w, y / w, z / w);
}

// Waechter & Binder self-intersection offset (interaction.h:177-199 -> ocarina offset_ray_origin)
VD V3 offset_ray_origin(V3 p, V3 n) {
    constexpr float origin = 1.f / 32.f, float_scale = 1.f / 65536.f, int_scale = 256.f;
    int ix = (int) (int_scale * n.x), iy = (int) (int_scale * n.y), iz = (int) (int_scale * n.z);
    float px = u2f((uint32_t) ((int32_t) f2u(p.x) + (p.x < 0.f ? -ix : ix)));
    float py = u2f((uint32_t) ((int32_t) f2u(p.y) + (p.y < 0.f ? -iy : iy)));
    float pz = u2f((uint32_t) ((int32_t) f2u(p.z) + (p.z < 0.f ? -iz : iz)));
    return mk3(abs_(p.x) < origin ? p.x + float_scale * n.x : px, abs_(p.y) < origin ? p.y + float_scale * n.y : py,
               abs_(p.z) < origin ? p.z + float_scale * n.z : pz);
}

}// namespace vmkd
