// rgb2spec.cpp — RGB -> sigmoid-polynomial spectrum for the host side (setup only).
// Own implementation of the Jakob & Hanika (2019) model the reference reaches through
// srgb_model_fetch (src/librender/srgb.cpp:11-28 -> ext/rgb2spec): instead of interpolating a
// precomputed 64^3 table it fits the three coefficients of the requested colour directly
// (Gauss-Newton on the CIELAB residual, homotopy from mid-grey).  Same algorithm as
// misaki-render_amd/rgb2spec.py; evaluation on the device follows render/srgb.h:8-19.
#include <misaki/render.h>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>

namespace misaki {
namespace {
const double XYZ_TO_SRGB[9] = {3.240479, -1.537150, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
const double SRGB_TO_XYZ[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};

struct Quadrature {
    std::vector<double> t;          // normalised wavelength
    std::vector<double> rgb[3];     // spectrum -> linear sRGB weights
    double white[3];
    Quadrature() {
        const float *cie = cie1931_xyz_table(), *d65 = d65_table();
        const int n = 471 * 2 - 1;                       // 0.5 nm
        auto interp = [](const float *tab, double lam) {
            double x = (lam - 360.0) / 5.0; int i = std::min(93, std::max(0, (int) x)); double w = x - i;
            return (1 - w) * tab[i] + w * tab[i + 1];
        };
        std::vector<double> cmf[3], ill(n), w(n);
        double ysum = 0;
        for (int k = 0; k < n; ++k) {
            double lam = 360.0 + 470.0 * k / (n - 1);
            t.push_back((lam - 360.0) / 470.0);
            w[k] = 470.0 / (n - 1) * ((k == 0 || k == n - 1) ? 0.5 : 1.0);
            for (int c = 0; c < 3; ++c) cmf[c].push_back(interp(cie + 95 * c, lam));
            ill[k] = interp(d65, lam);
            ysum += cmf[1][k] * ill[k] * w[k];
        }
        for (int c = 0; c < 3; ++c) { rgb[c].assign(n, 0.0); white[c] = 0; }
        for (int k = 0; k < n; ++k) {
            double s = ill[k] / ysum * w[k];
            for (int c = 0; c < 3; ++c) {
                rgb[c][k] = (XYZ_TO_SRGB[c * 3] * cmf[0][k] + XYZ_TO_SRGB[c * 3 + 1] * cmf[1][k] + XYZ_TO_SRGB[c * 3 + 2] * cmf[2][k]) * s;
                white[c] += cmf[c][k] * s;
            }
        }
    }
};
const Quadrature &quad() { static Quadrature q; return q; }

void model_rgb(const double c[3], double out[3]) {
    const Quadrature &q = quad();
    out[0] = out[1] = out[2] = 0;
    for (size_t k = 0; k < q.t.size(); ++k) {
        double x = (c[0] * q.t[k] + c[1]) * q.t[k] + c[2];
        double s = 0.5 + 0.5 * x / std::sqrt(1.0 + x * x);
        for (int j = 0; j < 3; ++j) out[j] += q.rgb[j][k] * s;
    }
}
void lab(const double rgb[3], double out[3]) {
    const Quadrature &q = quad();
    double xyz[3];
    for (int i = 0; i < 3; ++i) xyz[i] = SRGB_TO_XYZ[i * 3] * rgb[0] + SRGB_TO_XYZ[i * 3 + 1] * rgb[1] + SRGB_TO_XYZ[i * 3 + 2] * rgb[2];
    auto f = [](double v) { const double d = 6.0 / 29.0; return v > d * d * d ? std::cbrt(std::max(v, 0.0)) : v / (3 * d * d) + 4.0 / 29.0; };
    double fx = f(xyz[0] / q.white[0]), fy = f(xyz[1] / q.white[1]), fz = f(xyz[2] / q.white[2]);
    out[0] = 116 * fy - 16; out[1] = 500 * (fx - fy); out[2] = 200 * (fy - fz);
}
bool solve3(double J[3][3], const double r[3], double x[3]) {
    double a[3][4];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a[i][j] = J[i][j]; a[i][3] = r[i]; }
    for (int c = 0; c < 3; ++c) {
        int p = c;
        for (int i = c + 1; i < 3; ++i) if (std::fabs(a[i][c]) > std::fabs(a[p][c])) p = i;
        if (std::fabs(a[p][c]) < 1e-300) return false;
        for (int j = 0; j < 4; ++j) std::swap(a[p][j], a[c][j]);
        for (int i = 0; i < 3; ++i) if (i != c) { double f = a[i][c] / a[c][c]; for (int j = c; j < 4; ++j) a[i][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 3; ++i) x[i] = a[i][3] / a[i][i];
    return true;
}
void gauss_newton(double c[3], const double target_lab[3]) {
    for (int it = 0; it < 20; ++it) {
        double rgb[3], l[3], r[3];
        model_rgb(c, rgb); lab(rgb, l);
        for (int i = 0; i < 3; ++i) r[i] = l[i] - target_lab[i];
        if (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] < 1e-12) break;
        double J[3][3];
        const double eps = 1e-4;
        for (int i = 0; i < 3; ++i) {
            double cp[3] = {c[0], c[1], c[2]}, cm[3] = {c[0], c[1], c[2]}, lp[3], lm[3];
            cp[i] += eps; cm[i] -= eps;
            model_rgb(cp, rgb); lab(rgb, lp);
            model_rgb(cm, rgb); lab(rgb, lm);
            for (int j = 0; j < 3; ++j) J[j][i] = (lp[j] - lm[j]) / (2 * eps);
        }
        double step[3];
        if (!solve3(J, r, step)) break;
        for (int i = 0; i < 3; ++i) c[i] -= step[i];
        double m = std::max(std::fabs(c[0]), std::max(std::fabs(c[1]), std::fabs(c[2])));
        if (m > 200) for (int i = 0; i < 3; ++i) c[i] *= 200 / m;
    }
}
}  // namespace

Color3 srgb_model_fetch(const Color3 &rgb_) {
    static std::mutex mtx;
    static std::map<std::tuple<float, float, float>, Color3> cache;
    std::lock_guard<std::mutex> lock(mtx);
    auto key = std::make_tuple(rgb_.r, rgb_.g, rgb_.b);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    double target[3] = {std::min(1.0, std::max(0.0, (double) rgb_.r)), std::min(1.0, std::max(0.0, (double) rgb_.g)),
                        std::min(1.0, std::max(0.0, (double) rgb_.b))};
    Color3 out;
    if (target[0] == target[1] && target[1] == target[2]) {
        double v = target[0];
        out = Color3{0.f, 0.f, v <= 0 ? -INFINITY : v >= 1 ? INFINITY : (float) ((v - 0.5) / std::sqrt(v * (1 - v)))};
    } else {
        double c[3] = {0, 0, 0};
        for (int k = 1; k <= 16; ++k) {
            double s = k / 16.0, mix[3], l[3];
            for (int i = 0; i < 3; ++i) mix[i] = (1 - s) * 0.5 + s * target[i];
            lab(mix, l);
            gauss_newton(c, l);
        }
        const double c0 = 360.0, c1 = 1.0 / 470.0, A = c[0], B = c[1], C = c[2];
        out = Color3{(float) (A * c1 * c1), (float) (B * c1 - 2 * A * c0 * c1 * c1), (float) (C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1))};
    }
    cache[key] = out;
    return out;
}

}  // namespace misaki
