// ieee_fast_check.hip — msk_device.h's hand-expanded IEEE operation (rsqrt_ieee: the fast form AND its guard) against the
// compiler's correctly rounded division and square root:
//   * rsqrt_ieee(x) on EVERY binary32 bit pattern;
//   * against the host's arithmetic (x86 SSE: IEEE) on a sample, srgb_model_eval (its caller, with a guard of its own) against a
//     host restatement.
// (Round 5 ran the same harness over a hand-expanded sqrtf and shared-denominator divisions — exact, and worth nothing: msk_device.h.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I misaki-render_amd/csrc
//         -o gpurun_scratch/ieee_fast_check tools/micro/ieee_fast_check.hip
// (tests/test_ieee_fast.py builds and runs it on the GPU box; prints "all ok" or the first differing inputs)
#include "msk_device.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
using namespace msk;

__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__global__ void k_unary(unsigned long long *n_bad, uint32_t *first_bad, unsigned long long *n_fast) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;       // 2^24 threads x 256 patterns
    unsigned long long bad = 0, fast = 0;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t bits = tid * 256u + k;
        const float x = __uint_as_float(bits);
        float want, got;
        want = 1.f / __builtin_sqrtf(x); got = rsqrt_ieee(x); if (x >= 1.f && x < 0x1p100f) ++fast;
        if (!same_bits(want, got)) { atomicMin(first_bad, bits); ++bad; }
    }
    if (bad) atomicAdd(n_bad, bad);
    atomicAdd(n_fast, fast);
}
__global__ void k_eval(const float *c, const float *wl, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    spec w; for (int k = 0; k < 4; ++k) w.v[k] = wl[4 * i + k];
    const spec r = srgb_model_eval(c[3 * i], c[3 * i + 1], c[3 * i + 2], w);
    for (int k = 0; k < 4; ++k) out[4 * i + k] = r.v[k];
}
__global__ void k_sample(const float *x, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rsqrt_ieee(x[i]);
}

static int rc = 0;
static void report(const char *what, unsigned long long n, unsigned long long bad, unsigned long long first, unsigned long long fast) {
    printf("%s: %llu cases, %llu through the fast form, %llu differ", what, n, fast, bad);
    if (bad) { printf(" (first: 0x%llx)", first); rc = 1; }
    printf("\n");
}
int main() {
    unsigned long long *d_bad, *d_fast; uint32_t *d_first;
    if (hipMalloc(&d_bad, 8) || hipMalloc(&d_fast, 8) || hipMalloc(&d_first, 4)) return 2;
    unsigned long long bad, fast; uint32_t first;
    auto reset = [&]() { (void) hipMemset(d_bad, 0, 8); (void) hipMemset(d_fast, 0, 8); (void) hipMemset(d_first, 0xff, 4); };
    auto fetch = [&]() {
        if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); std::exit(2); }
        (void) hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); (void) hipMemcpy(&fast, d_fast, 8, hipMemcpyDeviceToHost);
        (void) hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    };
    reset(); hipLaunchKernelGGL(k_unary, dim3(65536), dim3(256), 0, nullptr, d_bad, d_first, d_fast); fetch();
    report("rsqrt_ieee(x) vs 1.f / sqrtf(x), every bit pattern", 1ull << 32, bad, first, fast);
    if (fast != (unsigned long long) (0x71800000u - 0x3f800000u)) { printf("WRONG: the guard let %llu patterns through, expected %u\n", fast, 0x71800000u - 0x3f800000u); rc = 1; }

    // against the host's IEEE arithmetic
    std::vector<float> x;
    for (uint32_t b = 0x2b800000u; b < 0x73800000u; b += 977u) { float f; std::memcpy(&f, &b, 4); x.push_back(f); }
    for (uint32_t e = 87; e < 231; ++e) for (int k = -64; k <= 64; ++k) { const uint32_t b = (e << 23) + (uint32_t) k; float f; std::memcpy(&f, &b, 4); x.push_back(f); }
    const uint32_t n = (uint32_t) x.size();
    float *dx, *dz;
    if (hipMalloc(&dx, (size_t) n * 4) || hipMalloc(&dz, (size_t) n * 4)) return 2;
    (void) hipMemcpy(dx, x.data(), (size_t) n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sample, dim3((n + 255) / 256), dim3(256), 0, nullptr, dx, n, dz);
    std::vector<float> z(n);
    (void) hipMemcpy(z.data(), dz, (size_t) n * 4, hipMemcpyDeviceToHost);
    unsigned long long host_bad = 0;
    for (uint32_t i = 0; i < n; ++i) {
        volatile float s = std::sqrt(x[i]); volatile float w = 1.f / s;
        const float want = w;
        if (std::memcmp(&want, &z[i], 4) != 0) { if (!host_bad) printf("host differs at x = %a: %a vs %a\n", x[i], want, z[i]); ++host_bad; }
    }
    printf("%u inputs against the host's sqrt and division: %llu differ\n", n, host_bad);
    if (host_bad) rc = 1;

    // srgb_model_eval on coefficient triples that make v * v + 1 ordinary, huge, infinite and NaN: compared with the host's
    // restatement of render/srgb.h:8-19 (the guard's slow side is taken by the large ones)
    std::vector<float> c, wl;
    const float cs[][3] = {{0.f, 0.f, 0.f}, {1e-4f, -0.1f, 20.f}, {-3e-5f, 0.03f, -7.f}, {1e10f, 1e20f, 1e30f}, {1e30f, 1e30f, 1e30f}, {3e38f, 3e38f, 3e38f},
                           {NAN, 0.f, 0.f}, {0.f, 0.f, 1e25f}, {1e-3f, 2.f, -1000.f}, {0.f, 0.f, 1.2e15f}};
    for (auto &t : cs) for (int j = 0; j < 64; ++j) { c.push_back(t[0]); c.push_back(t[1]); c.push_back(t[2]); for (int k = 0; k < 4; ++k) wl.push_back(360.f + 470.f * ((j * 4 + k) % 251) / 250.f); }
    const uint32_t m = (uint32_t) c.size() / 3;
    float *dc, *dw, *dout;
    if (hipMalloc(&dc, c.size() * 4) || hipMalloc(&dw, wl.size() * 4) || hipMalloc(&dout, wl.size() * 4)) return 2;
    (void) hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice); (void) hipMemcpy(dw, wl.data(), wl.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_eval, dim3((m + 63) / 64), dim3(64), 0, nullptr, dc, dw, m, dout);
    std::vector<float> got(wl.size());
    (void) hipMemcpy(got.data(), dout, wl.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long eval_bad = 0;
    for (uint32_t i = 0; i < m; ++i) for (int k = 0; k < 4; ++k) {
        const float c0 = c[3 * i], c1 = c[3 * i + 1], c2 = c[3 * i + 2], w = wl[4 * i + k];
        float want;
        if (std::isinf(c2)) want = std::copysign(1.f, c2) * .5f + .5f;
        else {
            volatile float a = c0 * w; volatile float b = a + c1; volatile float cc = b * w; volatile float v = cc + c2;
            volatile float vv = v * v; volatile float xx = vv + 1.f; volatile float s = std::sqrt(xx); volatile float r = 1.f / s;
            volatile float h = .5f * v; volatile float hr = h * r; volatile float q = hr + .5f;
            want = (q < 0.f) ? 0.f : q;           // std::max(q, 0.f): a NaN stays
        }
        const float g = got[4 * i + k];
        const bool same = std::memcmp(&want, &g, 4) == 0 || (want != want && g != g);
        if (!same) { if (!eval_bad) printf("srgb_model_eval differs: c = %g %g %g, wl %g: host %a, device %a\n", c0, c1, c2, w, want, g); ++eval_bad; }
    }
    printf("%u srgb_model_eval calls against the host: %llu values differ\n", m, eval_bad);
    if (eval_bad) rc = 1;
    printf(rc ? "WRONG\n" : "all ok\n");
    return rc;
}
