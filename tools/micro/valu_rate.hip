// valu_rate.hip — how many cycles does a wave64 VALU instruction take on this GPU?  Every wave runs a long unrolled chain of
// independent v_fma_f32 (8 accumulators), 8 waves per SIMD; prints wave-instructions per second and per SIMD-cycle.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_fma(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void __launch_bounds__(256) k_mul(float *out, int iters, float a, float b) {     // v_mul + v_add (no fma)
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
    float *out; hipMalloc(&out, (size_t) blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (which == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double insts = (double) blocks * 4 * iters * 16 * 8;       // wave-level VALU instructions
            const double per_simd_s = insts / (cus * 4) / (ms * 1e-3);
            printf("%s: %d CUs, clock %.2f GHz: %.1f G wave-instr/s total, %.3f G/s per SIMD = one instruction per %.2f cycles at %.2f GHz\n",
                   which ? "v_mul+v_add" : "v_fma", cus, p.clockRate / 1e6, insts / (ms * 1e-3) / 1e9, per_simd_s / 1e9, (p.clockRate * 1e3) / per_simd_s, p.clockRate / 1e6);
        }
    return 0;
}
