// rgb2spec_table.cpp — the RGB -> sigmoid-polynomial coefficient table of Jakob & Hanika (2019) for the sRGB gamut:
// its optimiser and the trilinear fetch.  Replaces ext/rgb2spec of the reference (build-time tool
// ext/rgb2spec/rgb2spec_opt.cpp -> data/srgb.coeff, run-time ext/rgb2spec/rgb2spec.c:77-119) with code of this
// library.  What is reproduced, so that a fetch here returns the coefficients a fetch there returns:
//   * the quadrature of rgb2spec_opt.cpp:94-168: 5-nm CIE 1931 / D65 tables refined to 283 nodes (h = 470/282) with
//     Simpson-3/8 weights, D65 normalised to unit luminance, spectrum -> linear sRGB through the 1931 matrices;
//   * the residual in CIELAB (:56-83,:170-199), a central-difference Jacobian with step 1e-4 (:201-216), at most 15
//     Gauss-Newton steps that stop once |residual|^2 < 1e-6 and rescale the coefficients when the largest exceeds 200
//     (:218-255) — the table stores the iterate the loop stops at, not the converged optimum, so the stopping rule and
//     the warm start are part of the table's definition;
//   * the grid (:297-366): three sheets (largest component l), brightness scale[k] = smoothstep(smoothstep(k/(res-1))),
//     the other two components x*b, y*b; along k every (l, y, x) column is continued from k = res/5 upwards and,
//     restarted from zero, downwards; the polynomial is stored for wavelengths in nanometres;
//   * the file layout (:369-379): "SPEC", uint32 res, float scale[res], float data[3][res][res][res][3].
// All arithmetic is IEEE double evaluated left to right (the library is built with -ffp-contract=off), the stored
// values are rounded to float once.
#include <misaki/render.h>

#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace misaki {
const double *cie1931_xyz_table_f64();      // build/spectral_data.cpp: 3 x 95 doubles
const double *d65_table_f64();              // 95 doubles, relative SPD as tabulated (100 at 560 nm)

namespace {
constexpr int kCoarse = 95, kFine = (kCoarse - 1) * 3 + 1;
constexpr double kLambdaMin = 360.0, kLambdaMax = 830.0;
constexpr double kD65Luminance = 10566.864005283874576;     // Y of the tabulated D65 under this quadrature: unit luminance after division
const double kXyzToSrgb[3][3] = {{3.240479, -1.537150, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
const double kSrgbToXyz[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};

// piecewise-linear read of a 95-entry table (cie1931.h:255-265)
double table_at(const double *tab, double lambda) {
    double x = lambda - kLambdaMin;
    x *= (kCoarse - 1) / (kLambdaMax - kLambdaMin);
    int i = (int) x;
    i = i < 0 ? 0 : (i > kCoarse - 2 ? kCoarse - 2 : i);
    const double w = x - i;
    return (1.0 - w) * tab[i] + w * tab[i + 1];
}

struct Quadrature {
    double t[kFine];            // wavelength mapped to [0, 1]
    double to_rgb[3][kFine];    // weight of S(lambda_i) in each linear-sRGB component
    double white[3];            // XYZ of the illuminant
    Quadrature() {
        const double *cmf = cie1931_xyz_table_f64(), *spd = d65_table_f64();
        double d65[kCoarse];
        for (int i = 0; i < kCoarse; ++i) d65[i] = spd[i] / kD65Luminance;
        const double h = (kLambdaMax - kLambdaMin) / (kFine - 1);
        std::memset(to_rgb, 0, sizeof(to_rgb));
        white[0] = white[1] = white[2] = 0.0;
        for (int i = 0; i < kFine; ++i) {
            const double lambda = kLambdaMin + i * h;
            const double xyz[3] = {table_at(cmf, lambda), table_at(cmf + kCoarse, lambda), table_at(cmf + 2 * kCoarse, lambda)};
            const double illum = table_at(d65, lambda);
            double w = 3.0 / 8.0 * h;                               // Simpson 3/8: 1 3 3 2 3 3 2 ... 3 3 1
            if (i != 0 && i != kFine - 1) w *= ((i - 1) % 3 == 2) ? 2.0 : 3.0;
            t[i] = (lambda - kLambdaMin) / (kLambdaMax - kLambdaMin);
            for (int c = 0; c < 3; ++c)
                for (int j = 0; j < 3; ++j) to_rgb[c][i] += kXyzToSrgb[c][j] * xyz[j] * illum * w;
            for (int j = 0; j < 3; ++j) white[j] += xyz[j] * illum * w;
        }
    }
};
const Quadrature &quadrature() { static const Quadrature q; return q; }

struct Lab { double v[3]; };

double lab_f(double t) {
    const double delta = 6.0 / 29.0;
    return t > delta * delta * delta ? std::cbrt(t) : t / (delta * delta * 3.0) + (4.0 / 29.0);
}
Lab rgb_to_lab(const Quadrature &q, const double rgb[3]) {
    double xyz[3] = {0.0, 0.0, 0.0};
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r) xyz[r] += rgb[j] * kSrgbToXyz[r][j];
    const double fx = lab_f(xyz[0] / q.white[0]), fy = lab_f(xyz[1] / q.white[1]), fz = lab_f(xyz[2] / q.white[2]);
    return Lab{{116.0 * fy - 16.0, 500.0 * (fx - fy), 200.0 * (fy - fz)}};
}
// CIELAB of the colour the spectrum with polynomial c (over t in [0,1]) has under D65
Lab spectrum_lab(const Quadrature &q, const double c[3]) {
    double rgb[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < kFine; ++i) {
        const double t = q.t[i];
        double x = 0.0;
        for (int k = 0; k < 3; ++k) x = x * t + c[k];
        const double s = 0.5 * x / std::sqrt(1.0 + x * x) + 0.5;
        for (int j = 0; j < 3; ++j) rgb[j] += q.to_rgb[j][i] * s;
    }
    return rgb_to_lab(q, rgb);
}
void residual(const Quadrature &q, const double c[3], const Lab &target, double r[3]) {
    const Lab got = spectrum_lab(q, c);
    for (int j = 0; j < 3; ++j) r[j] = target.v[j] - got.v[j];
}

// 3x3 solve, LU with row pivoting in place (Doolittle; the first of equal pivots wins); false if singular
bool solve3(double a[3][3], const double b[3], double x[3]) {
    int perm[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) {
        int p = c;
        double best = 0.0;
        for (int r = c; r < 3; ++r) { const double v = std::fabs(a[r][c]); if (v > best) { best = v; p = r; } }
        if (best < 1e-15) return false;
        if (p != c) { std::swap(perm[c], perm[p]); for (int k = 0; k < 3; ++k) std::swap(a[c][k], a[p][k]); }
        for (int r = c + 1; r < 3; ++r) {
            a[r][c] /= a[c][c];
            for (int k = c + 1; k < 3; ++k) a[r][k] -= a[r][c] * a[c][k];
        }
    }
    for (int r = 0; r < 3; ++r) { x[r] = b[perm[r]]; for (int k = 0; k < r; ++k) x[r] -= a[r][k] * x[k]; }
    for (int r = 2; r >= 0; --r) { for (int k = r + 1; k < 3; ++k) x[r] -= a[r][k] * x[k]; x[r] = x[r] / a[r][r]; }
    return true;
}

// the table's definition of "the coefficients of rgb, continued from c" (rgb2spec_opt.cpp:218-255)
bool fit_step(const Quadrature &q, const double rgb[3], double c[3]) {
    const double eps = 1e-4;
    const Lab target = rgb_to_lab(q, rgb);
    for (int it = 0; it < 15; ++it) {
        double r[3], jac[3][3];
        residual(q, c, target, r);
        for (int k = 0; k < 3; ++k) {
            double lo[3] = {c[0], c[1], c[2]}, hi[3] = {c[0], c[1], c[2]}, rl[3], rh[3];
            lo[k] -= eps; hi[k] += eps;
            residual(q, lo, target, rl);
            residual(q, hi, target, rh);
            for (int j = 0; j < 3; ++j) jac[j][k] = (rh[j] - rl[j]) * 1.0 / (2 * eps);
        }
        double step[3];
        if (!solve3(jac, r, step)) return false;
        double rr = 0.0;
        for (int j = 0; j < 3; ++j) { c[j] -= step[j]; rr += r[j] * r[j]; }
        const double top = std::max(std::max(c[0], c[1]), c[2]);
        if (top > 200) for (int j = 0; j < 3; ++j) c[j] *= 200 / top;
        if (rr < 1e-6) break;
    }
    return true;
}

double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }
}  // namespace

// ---- optimiser: fills scale[res] and data[3*res^3*3]; `threads` <= 0 = hardware concurrency
void rgb2spec_build_table(int res, std::vector<float> &scale, std::vector<float> &data, int threads) {
    if (res < 2) Throw("rgb2spec_build_table: resolution {} < 2", res);
    const Quadrature &q = quadrature();
    scale.resize(res);
    for (int k = 0; k < res; ++k) scale[k] = (float) smoothstep(smoothstep(k / double(res - 1)));
    data.assign((size_t) 9 * res * res * res, 0.f);
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    auto column = [&](int l, int j, int i) {
        const double x = i / double(res - 1), y = j / double(res - 1);
        const int start = res / 5;
        auto run = [&](int k0, int k1, int dk) {
            double c[3] = {0.0, 0.0, 0.0};
            for (int k = k0; k != k1; k += dk) {
                const double b = (double) scale[k];
                double rgb[3];
                rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                if (!fit_step(q, rgb, c)) { failed = true; return; }
                // polynomial over t = (lambda - 360) / 470  ->  polynomial over lambda in nm
                const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0), A = c[0], B = c[1], C = c[2];
                float *o = &data[(size_t) 3 * ((((size_t) l * res + k) * res + j) * res + i)];
                o[0] = float(A * (c1 * c1));
                o[1] = float(B * c1 - 2 * A * c0 * (c1 * c1));
                o[2] = float(C - B * c0 * c1 + A * ((c0 * c1) * (c0 * c1)));
            }
        };
        run(start, res, +1);
        run(start, -1, -1);
    };
    auto worker = [&]() {
        for (int job; (job = next.fetch_add(1)) < 3 * res && !failed;)
            for (int i = 0; i < res; ++i) column(job / res, job % res, i);
    };
    int n = threads > 0 ? threads : (int) std::thread::hardware_concurrency();
    n = std::max(1, std::min(n, 64));
    std::vector<std::thread> pool;
    for (int t = 1; t < n; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (failed) Throw("rgb2spec_build_table: singular Jacobian");
}

// Written to a temporary file this PROCESS created exclusively (mkstemp: several ranks that find no table each compute it and
// write at the same time), then renamed over `path`: a reader sees either no file or a complete one.
void rgb2spec_write_table(const std::string &path, const std::vector<float> &scale, const std::vector<float> &data) {
    std::string tmp = path + ".tmp.XXXXXX";
    const int fd = mkstemp(&tmp[0]);
    if (fd < 0) Throw("Could not create \"{}\"", tmp);
    (void) fchmod(fd, 0644);          // mkstemp creates 0600: the table beside a shared install must be readable by its other users
    FILE *f = fdopen(fd, "wb");
    if (!f) { close(fd); std::remove(tmp.c_str()); Throw("Could not create \"{}\"", tmp); }
    const uint32_t res = (uint32_t) scale.size();
    bool ok = std::fwrite("SPEC", 4, 1, f) == 1 && std::fwrite(&res, 4, 1, f) == 1 &&
              std::fwrite(scale.data(), sizeof(float), scale.size(), f) == scale.size() &&
              std::fwrite(data.data(), sizeof(float), data.size(), f) == data.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); Throw("Could not write \"{}\"", path); }
}

bool rgb2spec_read_table(const std::string &path, std::vector<float> &scale, std::vector<float> &data) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[4];
    uint32_t res = 0;
    bool ok = std::fread(magic, 4, 1, f) == 1 && std::memcmp(magic, "SPEC", 4) == 0 && std::fread(&res, 4, 1, f) == 1 && res >= 2 && res <= 1024;
    if (ok) {
        scale.resize(res);
        data.resize((size_t) 9 * res * res * res);
        ok = std::fread(scale.data(), sizeof(float), scale.size(), f) == scale.size() &&
             std::fread(data.data(), sizeof(float), data.size(), f) == data.size();
    }
    std::fclose(f);
    // a table is finite numbers throughout (a torn or zero-filled file is not taken for one: all-zero coefficients, or scale
    // values that do not rise from 0 to 1, mean "no table here")
    if (ok) {
        ok = scale.front() == 0.f && scale.back() == 1.f;
        for (size_t i = 1; ok && i < scale.size(); ++i) ok = scale[i] > scale[i - 1];
        bool any = false;
        for (size_t i = 0; ok && i < data.size(); ++i) { ok = std::isfinite(data[i]); any = any || data[i] != 0.f; }
        ok = ok && any;
    }
    return ok;
}

// ---- fetch (rgb2spec.c:56-119): fp32, the products and sums in the order written there
void rgb2spec_fetch_table(int res, const float *scale, const float *data, const float rgb_in[3], float out[3]) {
    float rgb[3];
    for (int j = 0; j < 3; ++j) rgb[j] = std::max(std::min(rgb_in[j], 1.f), 0.f);
    int l = 0;
    for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[l]) l = j;                      // the last of equal maxima
    const float z = rgb[l], s = (res - 1) / z, x = rgb[(l + 1) % 3] * s, y = rgb[(l + 2) % 3] * s;
    const uint32_t xi = std::min((uint32_t) x, (uint32_t) (res - 2)), yi = std::min((uint32_t) y, (uint32_t) (res - 2));
    // largest zi <= res - 2 with scale[zi] <= z, by bisection over scale[1 .. res-2] (0 if there is none)
    int zi = 0;
    for (int n = res - 2; n > 0;) {
        const int half = n >> 1, mid = zi + half + 1;
        if (scale[mid] <= z) { zi = mid; n -= half + 1; } else n = half;
    }
    zi = std::min(zi, res - 2);
    const size_t dx = 3, dy = (size_t) 3 * res, dz = (size_t) 3 * res * res;
    size_t o = ((((size_t) l * res + zi) * res + yi) * res + xi) * 3;
    const float x1 = x - xi, x0 = 1.f - x1, y1 = y - yi, y0 = 1.f - y1,
                z1 = (z - scale[zi]) / (scale[zi + 1] - scale[zi]), z0 = 1.f - z1;
    for (int j = 0; j < 3; ++j, ++o)
        out[j] = ((data[o] * x0 + data[o + dx] * x1) * y0 + (data[o + dy] * x0 + data[o + dy + dx] * x1) * y1) * z0 +
                 ((data[o + dz] * x0 + data[o + dz + dx] * x1) * y0 + (data[o + dz + dy] * x0 + data[o + dz + dy + dx] * x1) * y1) * z1;
}

}  // namespace misaki
