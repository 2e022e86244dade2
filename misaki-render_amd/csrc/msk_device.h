// msk_device.h — fp32 math core of the wavefront path tracer, device side (gfx950).
//
// Numerical contract: every fp32 operation below is a single IEEE-754
// binary32 operation — the library is compiled with -ffp-contract=off, divides and square
// roots are correctly rounded (hipcc default) — and reductions use the association the
// reference's Eigen expressions have (3-vectors a0+(a1+a2); 4-wide spectra (a0+a2)+(a1+a3)).
// Transcendentals on the per-sample path are the det_* fp64 polynomials, rounded once.
// Reference file:line citations are relative to the misaki-render checkout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSK_DEV __device__ __forceinline__

namespace msk {

// 1.f / sqrtf(x), the two correctly rounded operations of render/srgb.h:16, for 1 <= x < 2^100 WITHOUT the compiler's expansions
// (v_sqrt_f32 + two next-up / next-down residual checks + denormal scaling = 16 instructions, then v_div_scale x 2, v_rcp_f32, five
// fma, v_div_fmas, v_div_fixup = 11: ~105 SIMD cycles by tools/micro/valu_ops.hip): one v_rsq_f32 seeds BOTH the square root
// (Markstein's coupled iteration, the last fma rounds) and the reciprocal of that root (two residual steps, the last fma rounds)
// — 1 transcendental + 2 multiplies + 1 add + 10 fma, ~40 cycles.  x = v * v + 1 is never below 1, so nothing here can be subnormal.
// Correct rounding is not argued but CHECKED: tools/micro/ieee_fast_check.hip compares this function with the compiler's
// 1.f / __builtin_sqrtf(x) on every one of the 2^32 bit patterns (tests/test_ieee_fast.py, -m gpu), the guard included.
#ifndef MSK_FAST_RSQRT
#define MSK_FAST_RSQRT 1
#endif
MSK_DEV float rsqrt_ieee_1_to_2p100(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r, h = .5f * r;
    const float e = __fmaf_rn(-h, s, .5f);
    s = __fmaf_rn(s, e, s); h = __fmaf_rn(h, e, h);
    s = __fmaf_rn(__fmaf_rn(-s, s, x), h, s);                  // = sqrtf(x)
    float y = h + h;
    y = __fmaf_rn(__fmaf_rn(-s, y, 1.f), y, y);
    // the last step: for a root with an all-ones mantissa (one x per binade) y is a power of two, the residual 2^-24 exactly and
    // y + e y an exact tie that rounds to even, while the true quotient y (1 + e + e^2 ...) lies just above it: that root takes
    // the next float up.  (Feeding the product y's successor instead — no compare — breaks 2 400 other inputs.)
    y = __fmaf_rn(__fmaf_rn(-s, y, 1.f), y, y);
    y = __uint_as_float(__float_as_uint(y) + (((__float_as_uint(s) & 0x7fffffu) == 0x7fffffu) ? 1u : 0u));
    return y;                                                   // = 1.f / sqrtf(x)
}
// (Round 5, measured and taken out again: sqrtf alone the same way for 2^-40 <= x < 2^40 — the first half of the above, exact on
// every bit pattern —, and a vector / float or spectrum / float as ONE reciprocal + refinement and five instructions per numerator
// for operands in [2^-40, 2^40), where v_div_scale / v_div_fmas / v_div_fixup do nothing, so that the compiler's own expansion
// is reproduced — exact on 2^32 random and 4 x 10^8 structured operand sets.  Bench step, config-3 / config-5 class renders with
// this function alone: 31.5 / 142.0 / 116.9 ms; with the square root as well: 31.5 / 141.6 / 118.1; with all three: 31.5 / 141.8 /
// 118.4 (nothing compiled by hand: 31.9 / 142.6 / 117.5): their guards — six or seven compares — eat what they save.)
// any x: the fast form where it is proven, the compiler's otherwise (a NaN fails the comparison)
MSK_DEV float rsqrt_ieee(float x) {
#if MSK_FAST_RSQRT
    if (x >= 1.f && x < 0x1p100f) return rsqrt_ieee_1_to_2p100(x);
#endif
    return 1.f / __builtin_sqrtf(x);
}


struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct spec { float v[4]; };   // Spectrum / Wavelength (core/fwd.h:40-41)

MSK_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MSK_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MSK_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MSK_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
MSK_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MSK_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
MSK_DEV float dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
MSK_DEV f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
MSK_DEV float fmin_std(float a, float b) { return (b < a) ? b : a; }   // std::min
MSK_DEV float fmax_std(float a, float b) { return (a < b) ? b : a; }   // std::max
MSK_DEV f3 normalized(f3 a) {            // Eigen MatrixBase::normalized()
    float z = dot(a, a);
    return z > 0.f ? a / __builtin_sqrtf(z) : a;
}
MSK_DEV float max_abs(f3 a) { return fmax_std(fabsf(a.x), fmax_std(fabsf(a.y), fabsf(a.z))); }

MSK_DEV spec splat(float c) { spec r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = c; return r; }
MSK_DEV spec from4(float4 a) { spec r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; return r; }
MSK_DEV float4 to4(spec a) { return make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }
#define MSK_SPEC_OP(op)                                                                         \
    MSK_DEV spec operator op(spec a, spec b) { spec r;                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b.v[i]; return r; }    \
    MSK_DEV spec operator op(spec a, float b) { spec r;                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b; return r; }
MSK_SPEC_OP(+) MSK_SPEC_OP(-) MSK_SPEC_OP(*) MSK_SPEC_OP(/)
#undef MSK_SPEC_OP
MSK_DEV float mean4(spec a) { return ((a.v[0] + a.v[2]) + (a.v[1] + a.v[3])) / 4.f; }
MSK_DEV float max4(spec a) { return fmax_std(fmax_std(a.v[0], a.v[1]), fmax_std(a.v[2], a.v[3])); }
MSK_DEV bool any_nonzero(spec a) { return a.v[0] != 0.f || a.v[1] != 0.f || a.v[2] != 0.f || a.v[3] != 0.f; }

// core/mathutils.h:10-20
#define MSK_PI_F        3.14159274101257324f      /* float(3.14159265358979323846) */
#define MSK_INV_FOUR_PI_F 0.0795774683356285095f   /* float(0.07957747154594766788) */
#define MSK_INV_PI_F    0.318309873342514038f     /* float(0.31830988618379067154) */
#define MSK_EPSILON_F   5.9604644775390625e-08f
#define MSK_RAY_EPS_F   (MSK_EPSILON_F * 1500)
#define MSK_SHADOW_EPS_F (MSK_RAY_EPS_F * 10)
#define MSK_INF_F       __builtin_inff()

// ------------------------------------------------------------------ det_* transcendentals
// (oracle/oracle_math.h holds the same functions, operation for operation: rule R3 of the numerics contract)
MSK_DEV uint64_t msk_bits(double x) { return (uint64_t) __double_as_longlong(x); }
MSK_DEV double msk_from_bits(uint64_t b) { return __longlong_as_double((long long) b); }
// The polynomial coefficients.  A v_fma_f64 takes its constant from an SGPR pair, and where the 41 of them come from decides
// the shading kernel's register budget: as plain immediates the compiler hoists their materialisation out of the chunk loop
// (~90 SGPRs live across the sweep: three waves instead of four); from constant memory (MSK_DET_CONST=0, the default: msk_det_*
// tables read by scalar loads) the LOADS are hoisted the same way — in rounds 3-4 the allocator then parked them in VGPR lanes,
// 82 v_writelane before the loop and two v_readlane, VALU instructions both, at every use inside it; since the kernel's cold
// arguments are read where they are used (msk_kernels.h: MSK_COLD_KARGS, round 5) most of them stay in SGPRs: 50 + 63 spill
// moves in k_shade_gen<true, true>, 3 690 VALU instructions.  Built and measured in round 5, not the default: MSK_DET_CONST=1
// materialises every constant on the spot with two s_mov_b32 of a literal (SALU) behind an `asm volatile` (45 + 49 spill moves,
// but 3 898 VALU instructions and the volatile statements pin the schedule: bench step 32.1-32.5 ms against 32.0), =2 the
// same as a plain asm that takes the polynomial's variable as an unused input (cannot be hoisted, may be scheduled: the same
// counts).  Same values as oracle_math.h's literals either way (the compiler's correctly rounded quotients).
#ifndef MSK_DET_CONST
#define MSK_DET_CONST 0
#endif
__constant__ double msk_det_sin[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0,
                                      -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
__constant__ double msk_det_cos[8] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0,
                                      1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0, -0.5};
__constant__ double msk_det_ath[9] = {1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};
__constant__ double msk_det_che[6] = {1.0 / 479001600.0, 1.0 / 3628800.0, 1.0 / 40320.0, 1.0 / 720.0, 1.0 / 24.0, 0.5};
__constant__ double msk_det_cho[6] = {1.0 / 6227020800.0, 1.0 / 39916800.0, 1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0};
template <uint64_t BITS> MSK_DEV double det_lit(double dep) {
    if constexpr (BITS == 0x3fe0000000000000ull || BITS == 0xbfe0000000000000ull) return __builtin_bit_cast(double, BITS);     // +-0.5: inline constants
    int lo, hi;
    if (MSK_DET_CONST == 2) {
        const int dep_lo = __double2loint(dep);
        asm("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((int) (uint32_t) BITS), "i"((int) (uint32_t) (BITS >> 32)), "v"(dep_lo));
    } else {
        asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((int) (uint32_t) BITS), "i"((int) (uint32_t) (BITS >> 32)));
    }
    return __hiloint2double(hi, lo);
}
// coefficient I of table T, whose value is the constant expression E (the same expression the table is initialised with); D: the
// polynomial's variable (see det_lit)
#define MSK_K(T, I, E, D) (MSK_DET_CONST ? det_lit<__builtin_bit_cast(uint64_t, (double) (E))>(D) : T[I])
// Every polynomial step is ONE fused multiply-add (IEEE fma: one rounding, the same bits from std::fma on the host and
// v_fma_f64 on the device, whatever -ffp-contract says); the series are cut where the next term is below 2e-16 of the result on
// the reduced range, i.e. the fp64 value is accurate to a few ulps of a double before its single rounding to fp32.
MSK_DEV double det_sin_poly(double y) {   // |y| <= pi/4 (+ulps): y - y z (1/3! - z/5! + ... ), z = y^2; next term (pi/4)^19/19! = 8e-20
    const double z = y * y;
    double p = MSK_K(msk_det_sin, 0, -1.0 / 355687428096000.0, z);                    // -1/17!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 1, 1.0 / 1307674368000.0, z));                 //  1/15!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 2, -1.0 / 6227020800.0, z));                   // -1/13!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 3, 1.0 / 39916800.0, z));                      //  1/11!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 4, -1.0 / 362880.0, z));                       // -1/9!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 5, 1.0 / 5040.0, z));                          //  1/7!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 6, -1.0 / 120.0, z));                          // -1/5!
    p = __builtin_fma(p, z, MSK_K(msk_det_sin, 7, 1.0 / 6.0, z));                             //  1/3!   (sign folded below)
    return __builtin_fma(-(y * z), p, y);
}
MSK_DEV double det_cos_poly(double y) {   // next term (pi/4)^18/18! = 2e-18
    const double z = y * y;
    double p = MSK_K(msk_det_cos, 0, 1.0 / 20922789888000.0, z);                      //  1/16!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 1, -1.0 / 87178291200.0, z));                  // -1/14!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 2, 1.0 / 479001600.0, z));                     //  1/12!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 3, -1.0 / 3628800.0, z));                      // -1/10!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 4, 1.0 / 40320.0, z));                         //  1/8!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 5, -1.0 / 720.0, z));                          // -1/6!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 6, 1.0 / 24.0, z));                            //  1/4!
    p = __builtin_fma(p, z, MSK_K(msk_det_cos, 7, -0.5, z));                                  // -1/2!
    return __builtin_fma(z, p, 1.0);
}
// quadrant reduction of an fp32 angle (|phi| < 2^20 pi/2): phi = k pi/2 + y, |y| <= pi/4
MSK_DEV double det_reduce_pio2(float phi, long long *k_out) {
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632679489655800e+00;   // pi/2 rounded to double
    const double pio2_lo = 6.12323399573676603587e-17;   // pi/2 - pio2_hi
    const double x = (double) phi;
    const double k = __builtin_rint(x * two_over_pi);
    *k_out = (long long) k;
    return __builtin_fma(-k, pio2_lo, __builtin_fma(-k, pio2_hi, x));
}
MSK_DEV void det_sincos(float phi, float *s, float *c) {
    long long k;
    const double y = det_reduce_pio2(phi, &k);
    const int q = (int) (k & 3);
    const double sy = det_sin_poly(y);
    __builtin_amdgcn_sched_barrier(0);          // one polynomial at a time: interleaved, the two fma chains cost the shading kernel 24 VGPRs (a wave per SIMD)
    const double cy = det_cos_poly(y);
    double sv = (q & 1) ? cy : sy, cv = (q & 1) ? sy : cy;
    if (q == 1 || q == 2) cv = -cv;
    if (q >= 2) sv = -sv;
    *s = (float) sv; *c = (float) cv;
}
// atanh of an fp32 x in (-1, 1): 1/2 ln((1 + x) / (1 - x)) with ONE division — N = 1 + x and D = 1 - x are exact in fp64; with
// N = 2^a n, D = 2^b d (n, d in [1, 2), one of them halved when n / d leaves [1/sqrt 2, sqrt 2]) it is
// (a - b) ln2 / 2 + s (1 + z/3 + z^2/5 + ... + z^9/19), s = (n - d) / (n + d) (numerator and denominator exact), z = s^2 <= 0.0295:
// the next term, z^10 / 21, is below 3e-17.
MSK_DEV double det_atanh_d(double xd) {
    const double N = 1.0 + xd, D = 1.0 - xd;
    uint64_t bn = msk_bits(N), bd = msk_bits(D);
    int e = (int) ((bn >> 52) & 0x7ff) - (int) ((bd >> 52) & 0x7ff);
    double n = msk_from_bits((bn & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    double d = msk_from_bits((bd & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    const double r2 = 1.41421356237309514547;
    if (n > r2 * d) { n = n * 0.5; e += 1; }
    else if (d > r2 * n) { d = d * 0.5; e -= 1; }
    const double s = (n - d) / (n + d), z = s * s;
    double p = MSK_K(msk_det_ath, 0, 1.0 / 19.0, z);
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 1, 1.0 / 17.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 2, 1.0 / 15.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 3, 1.0 / 13.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 4, 1.0 / 11.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 5, 1.0 / 9.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 6, 1.0 / 7.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 7, 1.0 / 5.0, z));
    p = __builtin_fma(p, z, MSK_K(msk_det_ath, 8, 1.0 / 3.0, z));
    p = __builtin_fma(p, z, 1.0);
    return __builtin_fma((double) e, 0.5 * 0.69314718055994528623, s * p);
}
// cosh of an fp32 x (|x| < 700) without a division: x = k ln2 + r, |r| <= ln2 / 2; e^(+-r) = E(r^2) +- r O(r^2) with the even and
// the odd half of the exponential series (through r^12 / 12! and r^13 / 13!: the next terms are below 5e-18), and
// cosh x = (2^k (E + r O) + 2^-k (E - r O)) / 2, the powers of two applied exactly.
MSK_DEV double det_cosh_d(double xd) {
    const double inv_ln2 = 1.44269504088896338700;
    const double ln2_hi  = 6.93147180369123816490e-01;
    const double ln2_lo  = 1.90821492927058770002e-10;
    const double k = __builtin_rint(xd * inv_ln2);
    const double r = __builtin_fma(-k, ln2_lo, __builtin_fma(-k, ln2_hi, xd)), w = r * r;
    double E = MSK_K(msk_det_che, 0, 1.0 / 479001600.0, w);             // 1/12!
    E = __builtin_fma(E, w, MSK_K(msk_det_che, 1, 1.0 / 3628800.0, w));
    E = __builtin_fma(E, w, MSK_K(msk_det_che, 2, 1.0 / 40320.0, w));
    E = __builtin_fma(E, w, MSK_K(msk_det_che, 3, 1.0 / 720.0, w));
    E = __builtin_fma(E, w, MSK_K(msk_det_che, 4, 1.0 / 24.0, w));
    E = __builtin_fma(E, w, MSK_K(msk_det_che, 5, 0.5, w));
    E = __builtin_fma(E, w, 1.0);
    double O = MSK_K(msk_det_cho, 0, 1.0 / 6227020800.0, w);            // 1/13!
    O = __builtin_fma(O, w, MSK_K(msk_det_cho, 1, 1.0 / 39916800.0, w));
    O = __builtin_fma(O, w, MSK_K(msk_det_cho, 2, 1.0 / 362880.0, w));
    O = __builtin_fma(O, w, MSK_K(msk_det_cho, 3, 1.0 / 5040.0, w));
    O = __builtin_fma(O, w, MSK_K(msk_det_cho, 4, 1.0 / 120.0, w));
    O = __builtin_fma(O, w, MSK_K(msk_det_cho, 5, 1.0 / 6.0, w));
    O = __builtin_fma(O, w, 1.0);
    const double ro = r * O;
    const long long ki = (long long) k;
    const double up = msk_from_bits((uint64_t) (ki + 1023) << 52), dn = msk_from_bits((uint64_t) (1023 - ki) << 52);
    return 0.5 * ((E + ro) * up + (E - ro) * dn);
}
MSK_DEV float det_atanh(float x) { return (float) det_atanh_d((double) x); }
MSK_DEV float det_cosh(float x) { return (float) det_cosh_d((double) x); }

// arctangent / tangent for the GGX azimuth (render/microfacet.h:23-26)
MSK_DEV double det_atan_d(double z) {
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
MSK_DEV float det_atan(float z) { return (float) det_atan_d((double) z); }
MSK_DEV float det_tan(float phi) {
    long long k;
    const double y = det_reduce_pio2(phi, &k);
    const double sy = det_sin_poly(y), cy = det_cos_poly(y);
    return (float) ((k & 1) ? -cy / sy : sy / cy);
}

// ------------------------------------------------------------------ counter RNG
MSK_DEV uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
MSK_DEV uint64_t counter_key(uint64_t seed, uint32_t pixel_index, uint32_t sample_index) {
    uint64_t id = ((uint64_t) pixel_index << 32) | (uint64_t) sample_index;
    return mix64(id + 0x9e3779b97f4a7c15ULL * (seed + 1));
}
MSK_DEV float u32_to_float01(uint32_t u) {     // core/mathutils.h:111-121
    return __uint_as_float((u >> 9) | 0x3f800000u) - 1.0f;
}
MSK_DEV f2 counter_pair(uint64_t key, uint32_t pair) {
    uint64_t r = mix64(key + 0x9e3779b97f4a7c15ULL * (uint64_t) (pair + 1));
    f2 o; o.x = u32_to_float01((uint32_t) (r >> 32)); o.y = u32_to_float01((uint32_t) r);
    return o;
}

// ------------------------------------------------------------------ core/mathutils.h:196-203
MSK_DEV void coordinate_system(f3 n, f3 *s, f3 *t) {
    float sign = __builtin_copysignf(1.f, n.z);
    const float a = -1.f / (sign + n.z);
    const float b = n.x * n.y * a;
    *s = mk3(1.f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    *t = mk3(b, sign + n.y * n.y * a, -n.y);
}
struct frame3 {                                  // core/frame.h:11-24
    f3 s, t, n;
    MSK_DEV f3 to_local(f3 v) const { return mk3(dot(v, s), dot(v, t), dot(v, n)); }
    MSK_DEV f3 to_world(f3 v) const { return s * v.x + t * v.y + n * v.z; }
};

MSK_DEV float safe_sqrt(float a) { return __builtin_sqrtf(fmax_std(a, 0.f)); }
MSK_DEV f2 square_to_uniform_triangle(f2 sample) {          // core/warp.h:11-15
    float t = safe_sqrt(1.f - sample.x);
    f2 r; r.x = 1.f - t; r.y = t * sample.y; return r;
}
MSK_DEV f3 square_to_uniform_sphere(f2 sample) {            // core/warp.h:7-9,46-53
    const float z = -2.f * sample.y + 1.f, r = safe_sqrt(-z * z + 1.f);
    const float t = (2.f * MSK_PI_F) * sample.x;
    float sn, cs; det_sincos(t, &sn, &cs);
    return mk3(r * cs, r * sn, z);
}
MSK_DEV f2 square_to_uniform_disk_concentric(f2 sample) {   // core/warp.h:17-32
    float x = 2.f * sample.x - 1.f;
    float y = 2.f * sample.y - 1.f;
    float phi, r;
    if (x == 0 && y == 0) {
        r = phi = 0;
    } else if (x * x > y * y) {
        r = x;
        phi = (MSK_PI_F / 4.f) * (y / x);
    } else {
        r = y;
        phi = (MSK_PI_F / 2.f) - (x / y) * (MSK_PI_F / 4.f);
    }
    float s, c;
    det_sincos(phi, &s, &c);
    f2 o; o.x = r * c; o.y = r * s; return o;
}
MSK_DEV f3 square_to_cosine_hemisphere(f2 sample) {         // core/warp.h:34-43
    f2 p = square_to_uniform_disk_concentric(sample);
    float z = safe_sqrt(1.f - (p.x * p.x + p.y * p.y));
    return mk3(p.x, p.y, z);
}

// core/spectrum.h:152-181, core/mathutils.h:166-182
MSK_DEV float wavelength_of(float sample, int i) {
    float shift = (float) i / 4.f;
    float value = sample + shift;
    float u = (value <= 1.f) ? value : value - 1.f;
    return 538.f - det_atanh(0.8569106254698279f - 1.8275019724092267f * u) * 138.88888888888889f;
}
MSK_DEV float wavelength_weight(float lam) {
    float tmp = det_cosh(0.0072f * (lam - 538.f));
    return 253.82f * tmp * tmp;
}

// render/srgb.h:8-19
MSK_DEV spec srgb_model_eval(float c0, float c1, float c2, spec wl) {
    if (__builtin_isinf(c2)) return splat(__builtin_copysignf(1.f, c2) * .5f + .5f);
    spec r;
    float x[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (c0 * wl.v[i] + c1) * wl.v[i] + c2; x[i] = v[i] * v[i] + 1.f; }
#if MSK_FAST_RSQRT
    // one guard for the four wavelengths: every x is >= 1 or a NaN (v * v >= 0), so their sum is below 2^100 exactly when none is
    // a NaN, an infinity or that large — the slow form is compiled once, behind a branch no lane of a sane scene takes
    if ((x[0] + x[1]) + (x[2] + x[3]) < 0x1p100f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[i] = fmax_std(.5f * v[i] * rsqrt_ieee_1_to_2p100(x[i]) + .5f, 0.f);
        return r;
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rsqrt = 1.f / __builtin_sqrtf(x[i]);
        r.v[i] = fmax_std(.5f * v[i] * rsqrt + .5f, 0.f);
    }
    return r;
}
// spectra/regular.cpp:73-91 on a 95-entry table over [360,830]
MSK_DEV spec regular_eval(const float *tbl, spec wl) {
    spec r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = (wl.v[i] - 360.f) * 0.2f;
        uint32_t idx = (uint32_t) x;
        idx = idx < 93u ? idx : 93u;
        float y0 = tbl[idx], y1 = tbl[idx + 1];
        float w1 = x - (float) idx, w0 = 1.f - w1;
        r.v[i] = w0 * y0 + w1 * y1;
    }
    return r;
}
// spectra/regular.cpp:73-91 on a table with its own grid (a `regular` spectrum of the scene, ABI v7): x = (l - lambda_min) *
// inv_interval, the segment index clamped to [0, last] with last = size - 2.  Below the table the conversion saturates at 0
// (the reference converts a negative float to uint32_t there — undefined in C++; the oracle takes 0 as well), above it the last
// segment is continued, as the reference does.
MSK_DEV spec regular_eval_grid(const float *tbl, float lam_min, float inv_interval, uint32_t last, spec wl) {
    spec r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = (wl.v[i] - lam_min) * inv_interval;
        uint32_t idx = (uint32_t) x;                 // v_cvt_u32_f32: saturating, a negative x gives 0
        idx = idx < last ? idx : last;
        float y0 = tbl[idx], y1 = tbl[idx + 1];
        float w1 = x - (float) idx, w0 = 1.f - w1;
        r.v[i] = w0 * y0 + w1 * y1;
    }
    return r;
}
// core/spectrum.h:82-115; cie = x[95] y[95] z[95]
MSK_DEV void spectrum_to_xyz(const float *cie, spec value, spec wl, float *X, float *Y, float *Z) {
    spec cx, cy, cz;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float t = (wl.v[s] - 360.f) * (94 / (830.f - 360.f));
        uint32_t i0 = (uint32_t) t;
        i0 = i0 < 93u ? i0 : 93u;
        float w1 = t - (float) i0, w0 = 1.f - w1;
        cx.v[s] = w0 * cie[i0] + w1 * cie[i0 + 1];
        cy.v[s] = w0 * cie[95 + i0] + w1 * cie[95 + i0 + 1];
        cz.v[s] = w0 * cie[190 + i0] + w1 * cie[190 + i0 + 1];
    }
    *X = mean4(cx * value); *Y = mean4(cy * value); *Z = mean4(cz * value);
}

// ------------------------------------------------------------------ GGX microfacet conductor
// render/microfacet.h:11-44,104-121,145-172 and render/fresnel.h:65-88, GGX branch only
MSK_DEV float eval_ggx(f3 m, float au, float av) {
    float cos_theta2 = m.z * m.z;
    float beckman_exp = ((m.x * m.x / (au * au)) + (m.y * m.y) / (av * av)) / cos_theta2;
    float root = (1.f + beckman_exp) * cos_theta2;
    return 1.f / (MSK_PI_F * au * av * root * root);
}
MSK_DEV f3 sample_ggx(f2 sample, float au, float av, float *pdf_out) {
    float phi_m = det_atan(au / av * det_tan(MSK_PI_F + 2 * MSK_PI_F * sample.y)) + MSK_PI_F * floorf(2 * sample.y + 0.5f);
    float sin_phi_m, cos_phi_m;
    det_sincos(phi_m, &sin_phi_m, &cos_phi_m);
    float cs = cos_phi_m / au, sn = sin_phi_m / av;
    float alpha_sqr = 1.f / (cs * cs + sn * sn);
    float tan_theta_m_sqr = alpha_sqr * sample.x / (1.f - sample.x);
    float cos_theta_m = rsqrt_ieee(1.f + tan_theta_m_sqr);
    float tmp = 1 + tan_theta_m_sqr / alpha_sqr;
    float pdf = MSK_INV_PI_F / (au * av * cos_theta_m * cos_theta_m * cos_theta_m * tmp * tmp);
    if (pdf < 1e-20f) pdf = 0;
    float sin_theta_m = safe_sqrt(1 - cos_theta_m * cos_theta_m);
    *pdf_out = pdf;
    return mk3(sin_theta_m * cos_phi_m, sin_theta_m * sin_phi_m, cos_theta_m);
}
MSK_DEV float distr_eval(f3 m, float au, float av) {
    if (m.z <= 0) return 0.0f;
    float result = eval_ggx(m, au, av);
    return result * m.z > 1e-20f ? result : 0.f;
}
MSK_DEV float smith_g1(f3 v, f3 m, float au, float av) {
    float xy_alpha_2 = (au * v.x) * (au * v.x) + (av * v.y) * (av * v.y), tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
    if (xy_alpha_2 == 0.f) return 1.f;
    if (dot(v, m) * v.z <= 0.f) return 0.f;
    return 2.f / (1.f + __builtin_sqrtf(1.f + tan_theta_alpha_2));
}
// render/fresnel.h:37-63
MSK_DEV void fresnel_dielectric(float cos_theta_i, float eta, float *F, float *cos_theta_t, float *eta_it, float *eta_ti) {
    if (cos_theta_i >= 0.f) { *eta_it = eta; *eta_ti = 1.f / eta; } else { *eta_it = 1.f / eta; *eta_ti = eta; }
    const float cos_theta_t_sqr = 1.f - *eta_ti * *eta_ti * (1.f - cos_theta_i * cos_theta_i);
    const float cos_theta_i_abs = fabsf(cos_theta_i);
    const float cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    const float a_s = (cos_theta_i_abs - *eta_it * cos_theta_t_abs) / (cos_theta_i_abs + *eta_it * cos_theta_t_abs);
    const float a_p = (cos_theta_t_abs - *eta_it * cos_theta_i_abs) / (cos_theta_t_abs + *eta_it * cos_theta_i_abs);
    float r;
    if (eta == 1.f || cos_theta_i_abs == 0.f) r = eta == 1.f ? 0.f : 1.f;
    else r = 0.5f * (a_s * a_s + a_p * a_p);
    *cos_theta_t = copysignf(cos_theta_t_abs, -cos_theta_i);
    *F = r;
}
MSK_DEV float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2;
    float a_2_pb_2 = __builtin_sqrtf(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r);
    float a = __builtin_sqrtf(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return .5f * (r_s + r_p);
}

MSK_DEV float mis_weight(float pdf_a, float pdf_b) {        // integrators/path.cpp:127-131
    pdf_a *= pdf_a; pdf_b *= pdf_b;
    return pdf_a > 0.f ? pdf_a / (pdf_a + pdf_b) : 0.f;
}

}  // namespace msk
