/*
 * oracle_math.h — TEST INFRASTRUCTURE ONLY (see oracle.cpp header).
 *
 * Scalar fp32 restatement of the reference's math core, written without Eigen.
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  Rules that make the restatement well defined:
 *
 *   R1  every fp32 operation is one IEEE-754 binary32 op, no fused multiply-add
 *       (built with -ffp-contract=off), sqrt and divide correctly rounded;
 *   R2  Eigen expression order (Eigen is an un-vendored, un-pinned dependency,
 *       vcpkg.json:4-14): 3-vector reductions follow Eigen 3.3/3.4's
 *       redux_novec_unroller, a0 + (a1 + a2); 4-wide Array reductions follow
 *       the SSE2 predux, (a0 + a2) + (a1 + a3); everything else is evaluated
 *       left to right, component-wise;
 *   R3  transcendental functions the hot path calls per sample (atanh, cosh in
 *       core/spectrum.h:158-166; sin, cos in core/warp.h:31) are evaluated by
 *       the det_* family below: an fp64 polynomial of explicit fused multiply-adds, rounded once to fp32.  libm
 *       and the GPU's ocml differ from each other in the last ulp, the det_*
 *       functions are bit-identical everywhere.  ORACLE_LIBM=1 switches back
 *       to libm so the size of this deviation can be measured.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace orc {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct S4 { float v[4]; };   // Spectrum / Wavelength = SpectrumArray<float,4>  (core/fwd.h:40-41)

static inline V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }

// R2: Eigen redux_novec_unroller<.,.,0,3>  ==  a0 + (a1 + a2).
// (Embree's dot(Vec3) has the same association: madd(a.x,b.x,madd(a.y,b.y,a.z*b.z)).)
static inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline float squared_norm(V3 a) { return dot(a, a); }
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen MatrixBase::normalized(): z = squaredNorm; z > 0 ? n / sqrt(z) : n
static inline V3 normalized(V3 a) {
    float z = squared_norm(a);
    return z > 0.f ? a / std::sqrt(z) : a;
}
static inline float norm(V3 a) { return std::sqrt(squared_norm(a)); }
static inline float max_abs(V3 a) {   // p.cwiseAbs().maxCoeff()
    return std::max(std::fabs(a.x), std::max(std::fabs(a.y), std::fabs(a.z)));
}

static inline S4 s4(float c) { return S4{{c, c, c, c}}; }
#define ORC_S4_OP(op)                                                          \
    static inline S4 operator op(S4 a, S4 b) {                                 \
        S4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b.v[i]; return r; }\
    static inline S4 operator op(S4 a, float b) {                              \
        S4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b; return r; }
ORC_S4_OP(+) ORC_S4_OP(-) ORC_S4_OP(*) ORC_S4_OP(/)
#undef ORC_S4_OP
// R2: SSE2 predux of a Packet4f
static inline float sum4(S4 a) { return (a.v[0] + a.v[2]) + (a.v[1] + a.v[3]); }
static inline float mean4(S4 a) { return sum4(a) / 4.f; }
static inline float max4(S4 a) { return std::max(std::max(a.v[0], a.v[1]), std::max(a.v[2], a.v[3])); }

// ---------------------------------------------------------------------------
// constants — core/mathutils.h:10-20
// ---------------------------------------------------------------------------
static const float kPi      = float(3.14159265358979323846);
static const float kInvPi   = float(0.31830988618379067154);
static const float kEpsilon = 5.9604644775390625e-08f;        // numeric_limits<float>::epsilon()/2
static const float kRayEpsilon    = kEpsilon * 1500;
static const float kShadowEpsilon = kRayEpsilon * 10;
static const float kInf = INFINITY;

// ---------------------------------------------------------------------------
// R3: deterministic transcendental functions (fp64 polynomial, one rounding)
// ---------------------------------------------------------------------------
extern int g_use_libm;   // oracle.cpp; 1 = call libm like the reference does

static inline uint64_t orc_bits(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
static inline double orc_from_bits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }
// Every polynomial step is ONE fused multiply-add (IEEE fma: one rounding, the same bits from std::fma on the host and
// v_fma_f64 on the device, whatever -ffp-contract says); the series are cut where the next term is below 2e-16 of the result on
// the reduced range, i.e. the fp64 value is accurate to a few ulps of a double before its single rounding to fp32.
static inline double det_sin_poly(double y) {   // |y| <= pi/4 (+ulps): y - y z (1/3! - z/5! + ... ), z = y^2; next term (pi/4)^19/19! = 8e-20
    const double z = y * y;
    double p = -1.0 / 355687428096000.0;                    // -1/17!
    p = __builtin_fma(p, z, 1.0 / 1307674368000.0);                 //  1/15!
    p = __builtin_fma(p, z, -1.0 / 6227020800.0);                   // -1/13!
    p = __builtin_fma(p, z, 1.0 / 39916800.0);                      //  1/11!
    p = __builtin_fma(p, z, -1.0 / 362880.0);                       // -1/9!
    p = __builtin_fma(p, z, 1.0 / 5040.0);                          //  1/7!
    p = __builtin_fma(p, z, -1.0 / 120.0);                          // -1/5!
    p = __builtin_fma(p, z, 1.0 / 6.0);                             //  1/3!   (sign folded below)
    return __builtin_fma(-(y * z), p, y);
}
static inline double det_cos_poly(double y) {   // next term (pi/4)^18/18! = 2e-18
    const double z = y * y;
    double p = 1.0 / 20922789888000.0;                      //  1/16!
    p = __builtin_fma(p, z, -1.0 / 87178291200.0);                  // -1/14!
    p = __builtin_fma(p, z, 1.0 / 479001600.0);                     //  1/12!
    p = __builtin_fma(p, z, -1.0 / 3628800.0);                      // -1/10!
    p = __builtin_fma(p, z, 1.0 / 40320.0);                         //  1/8!
    p = __builtin_fma(p, z, -1.0 / 720.0);                          // -1/6!
    p = __builtin_fma(p, z, 1.0 / 24.0);                            //  1/4!
    p = __builtin_fma(p, z, -0.5);                                  // -1/2!
    return __builtin_fma(z, p, 1.0);
}
// quadrant reduction of an fp32 angle (|phi| < 2^20 pi/2): phi = k pi/2 + y, |y| <= pi/4
static inline double det_reduce_pio2(float phi, long long *k_out) {
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632679489655800e+00;   // pi/2 rounded to double
    const double pio2_lo = 6.12323399573676603587e-17;   // pi/2 - pio2_hi
    const double x = (double) phi;
    const double k = std::rint(x * two_over_pi);
    *k_out = (long long) k;
    return __builtin_fma(-k, pio2_lo, __builtin_fma(-k, pio2_hi, x));
}
// sin and cos of an fp32 angle; valid for |phi| < 2^20 * pi/2
static inline void det_sincos(float phi, float *s, float *c) {
    if (g_use_libm) { *s = std::sin(phi); *c = std::cos(phi); return; }
    long long k;
    const double y = det_reduce_pio2(phi, &k);
    const int q = (int) (k & 3);
    double sy = det_sin_poly(y), cy = det_cos_poly(y);
    double sv, cv;
    switch (q) {
        case 0:  sv = sy;  cv = cy;  break;
        case 1:  sv = cy;  cv = -sy; break;
        case 2:  sv = -sy; cv = -cy; break;
        default: sv = -cy; cv = sy;  break;
    }
    *s = (float) sv; *c = (float) cv;
}
// atanh of an fp32 x in (-1, 1): 1/2 ln((1 + x) / (1 - x)) with ONE division — N = 1 + x and D = 1 - x are exact in fp64; with
// N = 2^a n, D = 2^b d (n, d in [1, 2), one of them halved when n / d leaves [1/sqrt 2, sqrt 2]) it is
// (a - b) ln2 / 2 + s (1 + z/3 + z^2/5 + ... + z^9/19), s = (n - d) / (n + d) (numerator and denominator exact), z = s^2 <= 0.0295:
// the next term, z^10 / 21, is below 3e-17.
static inline double det_atanh_d(double xd) {
    const double N = 1.0 + xd, D = 1.0 - xd;
    uint64_t bn = orc_bits(N), bd = orc_bits(D);
    int e = (int) ((bn >> 52) & 0x7ff) - (int) ((bd >> 52) & 0x7ff);
    double n = orc_from_bits((bn & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    double d = orc_from_bits((bd & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    const double r2 = 1.41421356237309514547;
    if (n > r2 * d) { n = n * 0.5; e += 1; }
    else if (d > r2 * n) { d = d * 0.5; e -= 1; }
    const double s = (n - d) / (n + d), z = s * s;
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    p = __builtin_fma(p, z, 1.0);
    return __builtin_fma((double) e, 0.5 * 0.69314718055994528623, s * p);
}
// cosh of an fp32 x (|x| < 700) without a division: x = k ln2 + r, |r| <= ln2 / 2; e^(+-r) = E(r^2) +- r O(r^2) with the even and
// the odd half of the exponential series (through r^12 / 12! and r^13 / 13!: the next terms are below 5e-18), and
// cosh x = (2^k (E + r O) + 2^-k (E - r O)) / 2, the powers of two applied exactly.
static inline double det_cosh_d(double xd) {
    const double inv_ln2 = 1.44269504088896338700;
    const double ln2_hi  = 6.93147180369123816490e-01;
    const double ln2_lo  = 1.90821492927058770002e-10;
    const double k = std::rint(xd * inv_ln2);
    const double r = __builtin_fma(-k, ln2_lo, __builtin_fma(-k, ln2_hi, xd)), w = r * r;
    double E = 1.0 / 479001600.0;             // 1/12!
    E = __builtin_fma(E, w, 1.0 / 3628800.0);
    E = __builtin_fma(E, w, 1.0 / 40320.0);
    E = __builtin_fma(E, w, 1.0 / 720.0);
    E = __builtin_fma(E, w, 1.0 / 24.0);
    E = __builtin_fma(E, w, 0.5);
    E = __builtin_fma(E, w, 1.0);
    double O = 1.0 / 6227020800.0;            // 1/13!
    O = __builtin_fma(O, w, 1.0 / 39916800.0);
    O = __builtin_fma(O, w, 1.0 / 362880.0);
    O = __builtin_fma(O, w, 1.0 / 5040.0);
    O = __builtin_fma(O, w, 1.0 / 120.0);
    O = __builtin_fma(O, w, 1.0 / 6.0);
    O = __builtin_fma(O, w, 1.0);
    const double ro = r * O;
    const long long ki = (long long) k;
    const double up = orc_from_bits((uint64_t) (ki + 1023) << 52), dn = orc_from_bits((uint64_t) (1023 - ki) << 52);
    return 0.5 * ((E + ro) * up + (E - ro) * dn);
}
static inline float det_atanh(float x) {
    if (g_use_libm) return std::atanh(x);
    return (float) det_atanh_d((double) x);
}
static inline float det_cosh(float x) {
    if (g_use_libm) return std::cosh(x);
    return (float) det_cosh_d((double) x);
}

// arctangent of a double (any finite z): two range reductions, then the Taylor series on |t| <= tan(pi/8)
static inline double det_atan_d(double z) {
    const double pio2 = 1.57079632679489655800, pio4 = 0.78539816339744827900;
    const double sgn = z < 0.0 ? -1.0 : 1.0;
    double a = z < 0.0 ? -z : z;
    const bool inv = a > 1.0;
    if (inv) a = 1.0 / a;
    double base = 0.0, t = a;
    if (a > 0.41421356237309503) { t = (a - 1.0) / (a + 1.0); base = pio4; }
    const double w = t * t;
    double p = 1.0 / 45.0;
    p = 1.0 / 43.0 - w * p; p = 1.0 / 41.0 - w * p; p = 1.0 / 39.0 - w * p; p = 1.0 / 37.0 - w * p;
    p = 1.0 / 35.0 - w * p; p = 1.0 / 33.0 - w * p; p = 1.0 / 31.0 - w * p; p = 1.0 / 29.0 - w * p;
    p = 1.0 / 27.0 - w * p; p = 1.0 / 25.0 - w * p; p = 1.0 / 23.0 - w * p; p = 1.0 / 21.0 - w * p;
    p = 1.0 / 19.0 - w * p; p = 1.0 / 17.0 - w * p; p = 1.0 / 15.0 - w * p; p = 1.0 / 13.0 - w * p;
    p = 1.0 / 11.0 - w * p; p = 1.0 / 9.0 - w * p; p = 1.0 / 7.0 - w * p; p = 1.0 / 5.0 - w * p;
    p = 1.0 / 3.0 - w * p; p = 1.0 - w * p;
    double r = base + t * p;
    if (inv) r = pio2 - r;
    return sgn * r;
}
static inline float det_atan(float z) {
    if (g_use_libm) return std::atan(z);
    return (float) det_atan_d((double) z);
}
// tangent of an fp32 angle: sin/cos of the det_sincos core in fp64, one rounding
static inline float det_tan(float phi) {
    if (g_use_libm) return std::tan(phi);
    long long k;
    const double y = det_reduce_pio2(phi, &k);
    const double sy = det_sin_poly(y), cy = det_cos_poly(y);
    return (float) ((k & 1) ? -cy / sy : sy / cy);
}

// ---------------------------------------------------------------------------
// PCG32 — core/mathutils.h:85-143
// ---------------------------------------------------------------------------
struct PCG32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    void seed(uint64_t initstate, uint64_t initseq) {       // mathutils.h:95-101
        state = 0U;
        inc = (initseq << 1u) | 1u;
        next_uint32();
        state += initstate;
        next_uint32();
    }
    uint32_t next_uint32() {                                  // mathutils.h:102-109
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    float next_float32() {                                    // mathutils.h:111-121
        uint32_t u = (next_uint32() >> 9) | 0x3f800000u;
        float f; std::memcpy(&f, &u, 4);
        return f - 1.0f;
    }
};

// ---------------------------------------------------------------------------
// counter RNG (the build's definition, shared bit for bit with the HIP side;
// DESIGN.md §rng).  A draw is a pure function of (seed, pixel, sample, pair).
// ---------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t z) {    // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static inline uint64_t counter_key(uint64_t seed, uint32_t pixel_index, uint32_t sample_index) {
    uint64_t id = ((uint64_t) pixel_index << 32) | (uint64_t) sample_index;
    return mix64(id + 0x9e3779b97f4a7c15ULL * (seed + 1));
}
static inline float u32_to_float01(uint32_t u) {   // same mapping as PCG32::next_float32
    uint32_t b = (u >> 9) | 0x3f800000u;
    float f; std::memcpy(&f, &b, 4);
    return f - 1.0f;
}
static inline void counter_pair(uint64_t key, uint32_t pair, float *a, float *b) {
    uint64_t r = mix64(key + 0x9e3779b97f4a7c15ULL * (uint64_t) (pair + 1));
    *a = u32_to_float01((uint32_t) (r >> 32));
    *b = u32_to_float01((uint32_t) r);
}

// ---------------------------------------------------------------------------
// core/mathutils.h:196-203  coordinate_system (Duff et al.)
// ---------------------------------------------------------------------------
static inline void coordinate_system(V3 n, V3 *s, V3 *t) {
    float sign = std::copysign(1.f, n.z);
    const float a = -1.f / (sign + n.z);
    const float b = n.x * n.y * a;
    *s = mk3(1.f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    *t = mk3(b, sign + n.y * n.y * a, -n.y);
}

// core/frame.h:11-24
struct Frame {
    V3 s, t, n;
    V3 to_local(V3 v) const { return mk3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 to_world(V3 v) const { return s * v.x + t * v.y + n * v.z; }
};

// core/warp.h:7-15
static inline float safe_sqrt(float a) { return std::sqrt(std::max(a, 0.f)); }
static inline V2 square_to_uniform_triangle(V2 sample) {
    float t = safe_sqrt(1.f - sample.x);
    return V2{1.f - t, t * sample.y};
}
// core/warp.h:7-9,46-53
static inline V3 square_to_uniform_sphere(V2 sample) {
    float z = -2.f * sample.y + 1.f, r = safe_sqrt(-z * z + 1.f);
    float t = (2.f * kPi) * sample.x;
    float s, c; det_sincos(t, &s, &c);
    return mk3(r * c, r * s, z);
}
static const float kInvFourPi = float(0.07957747154594766788);
// core/warp.h:17-32
static inline V2 square_to_uniform_disk_concentric(V2 sample) {
    float x = 2.f * sample.x - 1.f;
    float y = 2.f * sample.y - 1.f;
    float phi, r;
    if (x == 0 && y == 0) {
        r = phi = 0;
    } else if (x * x > y * y) {
        r = x;
        phi = (kPi / 4.f) * (y / x);
    } else {
        r = y;
        phi = (kPi / 2.f) - (x / y) * (kPi / 4.f);
    }
    float s, c;
    det_sincos(phi, &s, &c);
    return V2{r * c, r * s};
}
// core/warp.h:34-43
static inline V3 square_to_cosine_hemisphere(V2 sample) {
    V2 p = square_to_uniform_disk_concentric(sample);
    float z = safe_sqrt(1.f - (p.x * p.x + p.y * p.y));
    return mk3(p.x, p.y, z);
}
static inline float square_to_cosine_hemisphere_pdf(V3 v) { return kInvPi * v.z; }

// ---------------------------------------------------------------------------
// core/spectrum.h:152-181 + core/mathutils.h:166-182  wavelength sampling
// ---------------------------------------------------------------------------
static inline void sample_wavelength(float sample, S4 *wavelengths, S4 *weight) {
    for (int i = 0; i < 4; ++i) {
        float shift = (float) i / 4.f;                      // Index / Scalar(Size)
        float value = sample + shift;
        float u = (value <= 1.f) ? value : value - 1.f;     // mathutils.h:174-176
        float lam = 538.f - det_atanh(0.8569106254698279f - 1.8275019724092267f * u) *
                                138.88888888888889f;
        float tmp = det_cosh(0.0072f * (lam - 538.f));
        wavelengths->v[i] = lam;
        weight->v[i] = 253.82f * tmp * tmp;
    }
}

// render/srgb.h:8-19
static inline S4 srgb_model_eval(const float coeff[3], S4 wl) {
    S4 r;
    if (std::isinf(coeff[2])) {
        return s4(std::copysign(1.f, coeff[2]) * .5f + .5f);
    }
    for (int i = 0; i < 4; ++i) {
        float v = (coeff[0] * wl.v[i] + coeff[1]) * wl.v[i] + coeff[2];
        float rsqrt = 1.f / std::sqrt(v * v + 1.f);
        r.v[i] = std::max(.5f * v * rsqrt + .5f, 0.f);
    }
    return r;
}

}  // namespace orc
