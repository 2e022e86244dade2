// hbm_stream.hip — what this box's HBM delivers to plain streaming kernels, as a yardstick for k_shade_gen's state traffic
// (which moves ~490 MB in and ~590 MB out per launch: 16-byte accesses, 1 KB contiguous per wave and array, 7 arrays in, 7 out):
//   read   : float4 loads, summed                                   (bytes = N)
//   write  : float4 stores                                          (bytes = N)
//   copy   : one array in, one array out                            (bytes = 2 N)
//   soa7   : per wave and step 1 KB from each of 7 arrays in, 1 KB to each of 7 other arrays — the shading sweep's shape, one
//            wave per 16-step "region", non-temporal like the state accesses                                  (bytes = 14 x)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_scratch/hbm_stream tools/micro/hbm_stream.hip ; prints GB/s (best of 5)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void k_read(const v4 *in, size_t n, float *out) {
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const v4 x = __builtin_nontemporal_load(in + i); acc += x; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) *out = 1.f;
}
__global__ void k_write(v4 *out, size_t n) {
    const v4 x = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) __builtin_nontemporal_store(x, out + i);
}
__global__ void k_copy(const v4 *in, v4 *out, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
// arrays of `slots` float4 each; wave w owns slots [w * steps * 64, (w + 1) * steps * 64)
struct Soa { const v4 *in[7]; v4 *out[7]; };
__global__ void __launch_bounds__(256) k_soa7(Soa a, uint32_t steps, uint32_t n_waves) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
    if (wave >= n_waves) return;
    for (uint32_t s = 0; s < steps; ++s) {
        const size_t i = ((size_t) wave * steps + s) * 64 + lane;
        v4 x[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) x[k] = __builtin_nontemporal_load(a.in[k] + i);
#pragma unroll
        for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(x[k] + x[(k + 1) % 7], a.out[k] + i);
    }
}
template <class F> static double best_ms(F launch) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    double best = 1e30;
    for (int r = 0; r < 6; ++r) {
        (void) hipEventRecord(e0); launch(); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t n = (size_t) 1 << 26;                  // 2^26 float4 = 1 GiB per array
    v4 *a, *b; float *flag;
    if (hipMalloc(&a, n * 16) || hipMalloc(&b, n * 16) || hipMalloc(&flag, 4)) return 2;
    (void) hipMemset(a, 0, n * 16); (void) hipMemset(b, 0, n * 16);
    const int grid = 256 * 16;
    const double gb = n * 16 / 1e9;
    double ms = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, nullptr, a, n, flag); });
    printf("read   %7.0f GB/s\n", gb / ms * 1e3);
    ms = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, nullptr, b, n); });
    printf("write  %7.0f GB/s\n", gb / ms * 1e3);
    ms = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, nullptr, a, b, n); });
    printf("copy   %7.0f GB/s (read + written)\n", 2 * gb / ms * 1e3);
    // soa7: 14 arrays of 6144 regions x 1024 slots (the bench pool's half: 6.3 M slots, 100 MB each, 1.4 GB in all)
    for (uint32_t waves_per_region = 1; waves_per_region <= 2; ++waves_per_region) {
        const uint32_t steps = 16 / waves_per_region, n_waves = 6144 * waves_per_region;
        const size_t slots = (size_t) n_waves * steps * 64;
        Soa s;
        v4 *mem[14];
        for (int k = 0; k < 14; ++k) { if (hipMalloc(&mem[k], slots * 16)) return 2; (void) hipMemset(mem[k], 0, slots * 16); }
        for (int k = 0; k < 7; ++k) { s.in[k] = mem[k]; s.out[k] = mem[7 + k]; }
        ms = best_ms([&] { hipLaunchKernelGGL(k_soa7, dim3((n_waves * 64 + 255) / 256), dim3(256), 0, nullptr, s, steps, n_waves); });
        printf("soa7   %7.0f GB/s (7 arrays in, 7 out, %u waves x %u steps of 1 KB per array; %.0f MB per launch, %.1f us)\n", 14 * slots * 16 / 1e9 / ms * 1e3, n_waves, steps, 14 * slots * 16 / 1e6, ms * 1e3);
        for (int k = 0; k < 14; ++k) (void) hipFree(mem[k]);
    }
    return 0;
}
