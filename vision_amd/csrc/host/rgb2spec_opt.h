// sRGB -> sigmoid-polynomial spectrum coefficient table ("srgb2spec"), regenerated on the host.
//
// Vision's hero spectrum uplifts every RGB colour through `sRGBToSpectrumTable_Data` (render_core/spectrum/hero.cpp:52-76,
// include "srgb2spec.h"), a 3 x 64^3 table of the three coefficients of  s(c0*l^2 + c1*l + c2),  s(x) = 1/2 + x / (2 sqrt(1 + x^2)).
// The header that holds the table is NOT part of the reference checkout (generated / third-party data), so it is regenerated
// here with the published algorithm it comes from — W. Jakob and J. Hanika, "A Low-Dimensional Function Space for Efficient
// Spectral Upsampling", Eurographics 2019, tool `rgb2spec_opt` (res 64, sRGB gamut, D65): per table cell a Gauss-Newton fit
// (<= 15 iterations, central-difference Jacobian with eps 1e-4, coefficients clamped to |c| <= 200) of the CIELAB residual
// between the target colour and the colour of the sigmoid spectrum under D65, integrated with Simpson's 3/8 rule over
// 283 samples of the 5 nm CIE tables; cells are solved outwards from z = res/5 using the neighbour's solution as the
// start value.  PARITY UNPINNED against Vision's own table (absent); pinned by the round-trip property instead
// (tests/test_host.py: the uplifted spectrum of an sRGB colour integrates back to that colour).
#pragma once
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace rgb2spec {

constexpr int kRes = 64;
constexpr double kLambdaMin = 360.0, kLambdaMax = 830.0;
constexpr int kCoarse = 95;                       // 5 nm tables
constexpr int kFine = (kCoarse - 1) * 3 + 1;      // 283 quadrature nodes
constexpr double kEps = 1e-4;

struct Tables {
    double lambda[kFine];
    double rgb[3][kFine];
    double white[3];
};

inline double interp5(const double *tab, double lambda) { // piecewise-linear lookup in a 95-entry 5 nm table
    double x = (lambda - kLambdaMin) * ((kCoarse - 1) / (kLambdaMax - kLambdaMin));
    int off = (int) x;
    if (off < 0) off = 0;
    if (off > kCoarse - 2) off = kCoarse - 2;
    double w = x - off;
    return (1.0 - w) * tab[off] + w * tab[off + 1];
}

static const double kXyzToSrgb[3][3] = {{3.240479, -1.537150, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
static const double kSrgbToXyz[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};

// cie: X, Y, Z, D65 at 1 nm over 360..830 (471 samples each); the optimiser works on the 5 nm subsets
inline void init_tables(const float *cie, Tables &T) {
    double x5[kCoarse], y5[kCoarse], z5[kCoarse], d5[kCoarse];
    for (int i = 0; i < kCoarse; ++i) { x5[i] = cie[0 * 471 + 5 * i]; y5[i] = cie[1 * 471 + 5 * i]; z5[i] = cie[2 * 471 + 5 * i]; d5[i] = cie[3 * 471 + 5 * i]; }
    const double h = (kLambdaMax - kLambdaMin) / (kFine - 1);
    double wsum_y = 0.0;
    std::vector<double> weight(kFine);
    for (int i = 0; i < kFine; ++i) {
        double w = 3.0 / 8.0 * h;
        if (i == 0 || i == kFine - 1) {} else if ((i - 1) % 3 == 2) w *= 2.0; else w *= 3.0;
        weight[i] = w;
        T.lambda[i] = kLambdaMin + i * h;
        wsum_y += interp5(y5, T.lambda[i]) * interp5(d5, T.lambda[i]) * w;
    }
    for (int k = 0; k < 3; ++k) { T.white[k] = 0.0; for (int i = 0; i < kFine; ++i) T.rgb[k][i] = 0.0; }
    for (int i = 0; i < kFine; ++i) {
        double l = T.lambda[i];
        double xyz[3] = {interp5(x5, l), interp5(y5, l), interp5(z5, l)};
        double I = interp5(d5, l) / wsum_y; // illuminant normalised so that the white point has Y = 1
        for (int k = 0; k < 3; ++k) {
            for (int j = 0; j < 3; ++j) T.rgb[k][i] += kXyzToSrgb[k][j] * xyz[j] * I * weight[i];
            T.white[k] += xyz[k] * I * weight[i];
        }
    }
}

inline void cie_lab(const Tables &T, double *p) {
    double X = 0, Y = 0, Z = 0;
    for (int j = 0; j < 3; ++j) { X += p[j] * kSrgbToXyz[0][j]; Y += p[j] * kSrgbToXyz[1][j]; Z += p[j] * kSrgbToXyz[2][j]; }
    auto f = [](double t) { const double d = 6.0 / 29.0; return t > d * d * d ? std::cbrt(t) : t / (3.0 * d * d) + 4.0 / 29.0; };
    double fx = f(X / T.white[0]), fy = f(Y / T.white[1]), fz = f(Z / T.white[2]);
    p[0] = 116.0 * fy - 16.0; p[1] = 500.0 * (fx - fy); p[2] = 200.0 * (fy - fz);
}

inline void eval_residual(const Tables &T, const double *c, const double *rgb, double *res) {
    double out[3] = {0, 0, 0};
    for (int i = 0; i < kFine; ++i) {
        double l = (T.lambda[i] - kLambdaMin) / (kLambdaMax - kLambdaMin);
        double x = 0.0;
        for (int k = 0; k < 3; ++k) x = x * l + c[k];
        double s = 0.5 * x / std::sqrt(1.0 + x * x) + 0.5;
        for (int j = 0; j < 3; ++j) out[j] += T.rgb[j][i] * s;
    }
    cie_lab(T, out);
    std::memcpy(res, rgb, 3 * sizeof(double));
    cie_lab(T, res);
    for (int j = 0; j < 3; ++j) res[j] -= out[j];
}

inline void eval_jacobian(const Tables &T, const double *c, const double *rgb, double jac[3][3]) {
    double r0[3], r1[3], tmp[3];
    for (int i = 0; i < 3; ++i) {
        std::memcpy(tmp, c, sizeof(tmp)); tmp[i] -= kEps; eval_residual(T, tmp, rgb, r0);
        std::memcpy(tmp, c, sizeof(tmp)); tmp[i] += kEps; eval_residual(T, tmp, rgb, r1);
        for (int j = 0; j < 3; ++j) jac[j][i] = (r1[j] - r0[j]) * (1.0 / (2.0 * kEps));
    }
}

// 3x3 solve by LU with partial pivoting; false when singular
inline bool solve3(double A[3][3], const double *b, double *x) {
    int P[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) {
        double maxA = 0.0; int imax = i;
        for (int k = i; k < 3; ++k) if (std::fabs(A[k][i]) > maxA) { maxA = std::fabs(A[k][i]); imax = k; }
        if (maxA < 1e-15) return false;
        if (imax != i) { std::swap(P[i], P[imax]); for (int j = 0; j < 3; ++j) std::swap(A[i][j], A[imax][j]); }
        for (int j = i + 1; j < 3; ++j) {
            A[j][i] /= A[i][i];
            for (int k = i + 1; k < 3; ++k) A[j][k] -= A[j][i] * A[i][k];
        }
    }
    for (int i = 0; i < 3; ++i) { x[i] = b[P[i]]; for (int k = 0; k < i; ++k) x[i] -= A[i][k] * x[k]; }
    for (int i = 2; i >= 0; --i) { for (int k = i + 1; k < 3; ++k) x[i] -= A[i][k] * x[k]; x[i] /= A[i][i]; }
    return true;
}

inline void gauss_newton(const Tables &T, const double *rgb, double *c) {
    for (int it = 0; it < 15; ++it) {
        double r[3], J[3][3], x[3];
        eval_residual(T, c, rgb, r);
        eval_jacobian(T, c, rgb, J);
        if (!solve3(J, r, x)) break;
        double err = 0.0;
        for (int j = 0; j < 3; ++j) { c[j] -= x[j]; err += r[j] * r[j]; }
        double mx = std::fmax(std::fmax(std::fabs(c[0]), std::fabs(c[1])), std::fabs(c[2]));
        if (mx > 200.0) for (int j = 0; j < 3; ++j) c[j] *= 200.0 / mx;
        if (err < 1e-6) break;
    }
}

inline double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

// out: float[3][res][res][res][4] (c0, c1, c2 for wavelengths in nm, w = 0) — the float4 layout hero.cpp:54 uploads
inline void optimise(const float *cie, float *out, unsigned threads) {
    Tables T;
    init_tables(cie, T);
    double scale[kRes];
    for (int k = 0; k < kRes; ++k) scale[k] = smoothstep(smoothstep((double) k / (kRes - 1)));
    std::atomic<int> next{0};
    auto worker = [&]() {
        for (;;) {
            int job = next.fetch_add(1);
            if (job >= 3 * kRes) return;
            int l = job / kRes, j = job % kRes;
            const double y = (double) j / (kRes - 1);
            for (int i = 0; i < kRes; ++i) {
                const double x = (double) i / (kRes - 1);
                const int start = kRes / 5;
                auto solve_cell = [&](int k, double *c) {
                    double b = scale[k], rgb[3];
                    rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                    gauss_newton(T, rgb, c);
                    const double c0 = kLambdaMin, c1 = 1.0 / (kLambdaMax - kLambdaMin);
                    const double A = c[0], B = c[1], C = c[2];
                    size_t idx = (((size_t) l * kRes + k) * kRes + j) * kRes + i;
                    out[4 * idx + 0] = (float) (A * c1 * c1);
                    out[4 * idx + 1] = (float) (B * c1 - 2.0 * A * c0 * c1 * c1);
                    out[4 * idx + 2] = (float) (C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
                    out[4 * idx + 3] = 0.f;
                };
                double c[3] = {0, 0, 0};
                for (int k = start; k < kRes; ++k) solve_cell(k, c);
                c[0] = c[1] = c[2] = 0.0;
                for (int k = start; k >= 0; --k) solve_cell(k, c);
            }
        }
    };
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
}

}// namespace rgb2spec
