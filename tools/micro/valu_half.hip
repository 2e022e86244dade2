// valu_half.hip — does a wave64 VALU instruction cost less when one half of the wave is switched off?  Streams of v_fma_f32 /
// v_min_f32 / v_cvt with exec = all lanes, the lower 32, the lower 16, every other lane; eight waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_half tools/micro/valu_half.hip && ./valu_half
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)
#define A_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b, unsigned long long mask) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) { if (KIND == 0) { R64(A_FMA) } else { R64(A_MIN) } }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    hipDeviceProp_t p; (void) hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 4000;
    float *out; (void) hipMalloc(&out, (size_t) blocks * 256 * 4);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const struct { const char *name; unsigned long long m; } masks[] = {{"all 64 lanes", ~0ull}, {"lower 32", 0xffffffffull}, {"upper 32", 0xffffffff00000000ull},
        {"lower 16", 0xffffull}, {"every other lane", 0x5555555555555555ull}, {"lanes 0-15 + 32-47", 0x0000ffff0000ffffull}, {"one lane", 1ull}};
    for (int kind = 0; kind < 2; ++kind)
        for (auto &m : masks) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void) hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, m.m);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, m.m);
                (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
                float ms; (void) hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            const double insts = (double) blocks * 4 * iters * 64;
            printf("%-10s exec = %-20s %7.3f ms  %.2f cycles per instruction per SIMD\n", kind ? "v_min_f32" : "v_fma_f32", m.name, best,
                   (p.clockRate * 1e3) / (insts / (cus * 4) / (best * 1e-3)));
        }
    return 0;
}
