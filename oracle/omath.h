// oracle/omath.h — TEST INFRASTRUCTURE (CPU oracle). Not part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, link or call anything under oracle/.
//
// Restatement of the ocarina math semantics Vision's hot path relies on (the ocarina submodule is absent from
// the reference checkout, SURVEY.md F1/App. B — every item here is "parity unpinned" against ocarina itself)
// plus deterministic float32 elementary functions.  All arithmetic is plain IEEE-754 binary32 with no
// contraction (build with -ffp-contract=off), so the HIP path can reproduce it bit for bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

constexpr float Pi = 3.14159265358979323846f;
constexpr float InvPi = 0.31830988618379067154f;
constexpr float Inv2Pi = 0.15915494309189533577f;
constexpr float Inv4Pi = 0.07957747154594766788f;
constexpr float PiOver2 = 1.57079632679489661923f;
constexpr float PiOver4 = 0.78539816339744830961f;
constexpr float _2Pi = 6.28318530717958647692f;
constexpr float OneMinusEpsilon = 0x1.fffffep-1f;
constexpr float ShadowEpsilon = 1e-4f;
constexpr float RayTMax = 3.402823466e+38f;

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ---- scalar helpers (explicit comparisons so NaN behaviour is identical on both sides) ----
inline float fmin_(float a, float b) { return a < b ? a : b; }
inline float fmax_(float a, float b) { return a > b ? a : b; }
inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline float saturate_(float x) { return clamp_(x, 0.f, 1.f); }
inline float sqr(float x) { return x * x; }
inline float safe_sqrt(float x) { return sqrtf(fmax_(x, 0.f)); }
inline float lerp_(float t, float a, float b) { return a + t * (b - a); }
inline float inverse_lerp(float x, float a, float b) { return (x - a) / (b - a); }
inline float rcp(float x) { return 1.f / x; }
inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
inline float pow4(float x) { float x2 = x * x; return x2 * x2; }
inline float fract_(float x) { return x - floorf(x); }
inline bool isinf_(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }
inline bool isnan_(float x) { return (f2u(x) & 0x7fffffffu) > 0x7f800000u; }
inline float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }

// ---- deterministic elementary functions (Cephes single-precision kernels; +,-,*,/,sqrt,floor only) ----
inline void sincos_(float x, float *s, float *c) {
    float q = floorf(x * 0.636619772367581343f + 0.5f);
    float r = x - q * 1.5703125f;
    r = r - q * 4.837512969970703125e-4f;
    r = r - q * 7.54978995489188216e-8f;
    int k = (int) q;
    float r2 = r * r;
    float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * (-1.9515295891e-4f)));
    float cp = 1.f - 0.5f * r2 +
               r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    switch (k & 3) {
        case 0: *s = sp; *c = cp; break;
        case 1: *s = cp; *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}
inline float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
inline float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

inline float asin_poly(float s, float z) {
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * s + s;
}
inline float asin_(float x) {
    float a = abs_(x);
    float r;
    if (a > 0.5f) {
        float z = 0.5f * (1.f - a);
        float s = sqrtf(z);
        r = PiOver2 - 2.f * asin_poly(s, z);
    } else {
        r = asin_poly(a, a * a);
    }
    return x < 0.f ? -r : r;
}
inline float acos_(float x) {
    if (x < -0.5f) return Pi - 2.f * asin_(sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * asin_(sqrtf(0.5f * (1.f - x)));
    return PiOver2 - asin_(x);
}
inline float atan_(float xx) {
    float x = abs_(xx);
    float y;
    if (x > 2.414213562373095f) { y = PiOver2; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = PiOver4; x = (x - 1.f) / (x + 1.f); }
    else { y = 0.f; }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return xx < 0.f ? -y : y;
}
inline float atan2_(float y, float x) {
    if (x > 0.f) return atan_(y / x);
    if (x < 0.f) return y >= 0.f ? atan_(y / x) + Pi : atan_(y / x) - Pi;
    if (y > 0.f) return PiOver2;
    if (y < 0.f) return -PiOver2;
    return 0.f;
}
inline float exp_(float x) {
    if (x > 88.f) x = 88.f;
    if (x < -87.f) return 0.f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int) z;
    float x2 = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
                1.6666665459e-1f) * x + 5.0000001201e-1f) * x2 + x + 1.f;
    return p * u2f((uint32_t) (n + 127) << 23);
}
inline float log_(float x) { // Cephes logf for normal x > 0 (callers pass 1 - u with u in [0,1))
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
inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
inline float rsqrt_(float x) { return 1.f / sqrtf(x); }
inline float atanh_(float x) { return 0.5f * log_((1.f + x) / (1.f - x)); }
inline float cosh_(float x) { return 0.5f * (exp_(x) + exp_(-x)); }

// ---- vectors ----
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float v) { return {v, v, v}; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 operator+(float3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline float3 operator-(float s, float3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 &operator+=(float3 &a, float3 b) { a = a + b; return a; }
inline float3 &operator*=(float3 &a, float3 b) { a = a * b; return a; }
inline float3 &operator*=(float3 &a, float s) { a = a * s; return a; }
inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length_squared(float3 a) { return dot(a, a); }
inline float length(float3 a) { return sqrtf(dot(a, a)); }
inline float3 normalize(float3 a) { float inv = 1.f / sqrtf(dot(a, a)); return a * inv; }
inline float abs_dot(float3 a, float3 b) { return abs_(dot(a, b)); }
inline bool is_zero(float3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
inline float max_comp(float3 a) { return fmax_(fmax_(a.x, a.y), a.z); }
inline float average(float3 a) { return (a.x + a.y + a.z) / 3.f; }
inline float3 lerp3(float t, float3 a, float3 b) { return a + (b - a) * t; }
inline float3 saturate3(float3 a) { return {saturate_(a.x), saturate_(a.y), saturate_(a.z)}; }
inline float3 select3(bool c, float3 a, float3 b) { return c ? a : b; }
inline float luminance(float3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }

// ---- sampled spectrum (base/color/spectrum.h:60-170) ----
// ORC_SPEC_DIM = 3 (default): a float3 — the (R, G, B) channels of spectrum/srgb or the three wavelengths of spectrum/hero, the
// arithmetic this oracle always had.  ORC_SPEC_DIM = 4 (liboracle4.so, the second build of the same source): four samples, for
// spectrum/hero scenes with "dimension": 4.  Directions, positions and RGB colours stay float3.
#ifndef ORC_SPEC_DIM
#define ORC_SPEC_DIM 3
#endif
struct spec4 { float x, y, z, w; };
inline spec4 operator+(spec4 a, spec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline spec4 operator-(spec4 a, spec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline spec4 operator*(spec4 a, spec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline spec4 operator/(spec4 a, spec4 b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
inline spec4 operator*(spec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline spec4 operator*(float s, spec4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline spec4 operator/(spec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline spec4 operator-(float s, spec4 a) { return {s - a.x, s - a.y, s - a.z, s - a.w}; }
inline spec4 operator-(spec4 a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline spec4 &operator+=(spec4 &a, spec4 b) { a = a + b; return a; }
inline spec4 &operator*=(spec4 &a, spec4 b) { a = a * b; return a; }
inline spec4 &operator*=(spec4 &a, float s) { a = a * s; return a; }
inline bool is_zero(spec4 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f && a.w == 0.f; }
inline float max_comp(spec4 a) { return fmax_(fmax_(fmax_(a.x, a.y), a.z), a.w); }
inline float average(spec4 a) { return (a.x + a.y + a.z + a.w) / 4.f; } // sum() * (1.0 / dimension), spectrum.h:147-149
inline spec4 lerp3(float t, spec4 a, spec4 b) { return a + (b - a) * t; }
inline spec4 saturate3(spec4 a) { return {saturate_(a.x), saturate_(a.y), saturate_(a.z), saturate_(a.w)}; }
#if ORC_SPEC_DIM == 4
typedef spec4 Spec;
inline Spec make_spec(float v) { return {v, v, v, v}; }
template<class F> inline Spec smap(Spec a, F f) { return {f(a.x), f(a.y), f(a.z), f(a.w)}; }
template<class F> inline Spec smap2(Spec a, Spec b, F f) { return {f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w)}; }
inline float scomp(Spec a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)); }
inline Spec spec_from_array(const float *v) { return {v[0], v[1], v[2], v[3]}; }
#elif ORC_SPEC_DIM == 3
typedef float3 Spec;
inline Spec make_spec(float v) { return {v, v, v}; }
template<class F> inline Spec smap(Spec a, F f) { return {f(a.x), f(a.y), f(a.z)}; }
template<class F> inline Spec smap2(Spec a, Spec b, F f) { return {f(a.x, b.x), f(a.y, b.y), f(a.z, b.z)}; }
inline float scomp(Spec a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline Spec spec_from_array(const float *v) { return {v[0], v[1], v[2]}; }
#else
#error "ORC_SPEC_DIM must be 3 or 4"
#endif
constexpr uint32_t kSpecDim = ORC_SPEC_DIM;

// ---- pbrt-style local shading geometry (math/geometry.h of ocarina; z-up local frame) ----
inline float cos_theta(float3 w) { return w.z; }
inline float abs_cos_theta(float3 w) { return abs_(w.z); }
inline float cos_theta_2(float3 w) { return w.z * w.z; }
inline float sin_theta_2(float3 w) { return fmax_(0.f, 1.f - w.z * w.z); }
inline float sin_theta(float3 w) { return sqrtf(sin_theta_2(w)); }
inline float cos_phi(float3 w) { float s = sin_theta(w); return s == 0.f ? 1.f : clamp_(w.x / s, -1.f, 1.f); }
inline float sin_phi(float3 w) { float s = sin_theta(w); return s == 0.f ? 0.f : clamp_(w.y / s, -1.f, 1.f); }
inline bool same_hemisphere(float3 a, float3 b) { return a.z * b.z > 0.f; }
inline bool same_hemisphere(float3 a, float3 b, float3 n) { return dot(a, n) * dot(b, n) > 0.f; }
inline float3 face_forward(float3 v, float3 n) { return dot(v, n) < 0.f ? -v : v; }
inline float3 reflect(float3 wo, float3 n) { return -wo + n * (2.f * dot(wo, n)); }
inline float3 spherical_direction(float sin_t, float cos_t, float phi) {
    float s, c; sincos_(phi, &s, &c);
    return {sin_t * c, sin_t * s, cos_t};
}
inline float spherical_theta(float3 v) { return acos_(clamp_(v.z, -1.f, 1.f)); }
inline float spherical_phi(float3 v) { float p = atan2_(v.y, v.x); return p < 0.f ? p + _2Pi : p; }
inline void coordinate_system(float3 v1, float3 *v2, float3 *v3) {
    if (abs_(v1.x) > abs_(v1.y)) { float inv = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z); *v2 = make_float3(-v1.z * inv, 0.f, v1.x * inv); }
    else { float inv = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z); *v2 = make_float3(0.f, v1.z * inv, -v1.y * inv); }
    *v3 = cross(v1, *v2);
}

struct Frame { // Frame<T,false>: axes used as given (interaction.h:87-117)
    float3 x, y, z;
    float3 to_local(float3 v) const { return make_float3(dot(v, x), dot(v, y), dot(v, z)); }
    float3 to_world(float3 v) const { return x * v.x + y * v.y + z * v.z; }
};

// column-major 3x3 / 4x4 helpers (m[col*N+row])
inline float3 mul3x3(const float *m, float3 v) {
    return make_float3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
                       m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
inline float3 transform_vector4(const float *m, float3 v) {
    return make_float3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
                       m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
inline float3 transform_point4(const float *m, float3 p) {
    float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    if (w == 1.f) return make_float3(x, y, z);
    return make_float3(x / w, y / w, z / w);
}

// offset_ray_origin (interaction.h:177-199 -> ocarina): Waechter & Binder, "A Fast and Robust Method for
// Avoiding Self-Intersection" (Ray Tracing Gems ch. 6), n already on the ray's side.
inline float3 offset_ray_origin(float3 p, float3 n) {
    constexpr float origin = 1.f / 32.f, float_scale = 1.f / 65536.f, int_scale = 256.f;
    int ix = (int) (int_scale * n.x), iy = (int) (int_scale * n.y), iz = (int) (int_scale * n.z);
    float px = u2f((uint32_t) ((int32_t) f2u(p.x) + (p.x < 0.f ? -ix : ix)));
    float py = u2f((uint32_t) ((int32_t) f2u(p.y) + (p.y < 0.f ? -iy : iy)));
    float pz = u2f((uint32_t) ((int32_t) f2u(p.z) + (p.z < 0.f ? -iz : iz)));
    return make_float3(abs_(p.x) < origin ? p.x + float_scale * n.x : px,
                       abs_(p.y) < origin ? p.y + float_scale * n.y : py,
                       abs_(p.z) < origin ? p.z + float_scale * n.z : pz);
}

}// namespace orc
