// valu_ops.hip — issue cost of the VALU instruction kinds the traversal and shading kernels are made of, on this GPU:
// every wave runs a long unrolled stream of ONE instruction kind on 8 independent registers, 8 waves per SIMD; prints
// SIMD cycles per wave64 instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ops tools/micro/valu_ops.hip && ./valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define R8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, float a, float b) {                  \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; ++i) { R64(ASM) }                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                        \
    }
#define A_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MUL(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define A_MIN(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_MAX(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_MAX3(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
#define A_CND64(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(a) : );
#define A_CMPCND(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define A_CMPCND4(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define A_CMPCNDS(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21");
#define A_CMP2CND(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define A_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_ASHR(x) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x));
#define A_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MED3(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define A_RSQ(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
#define A_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define A_MULU24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_SUBU(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_LSHR(x) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
#define A_FLOOR(x) asm volatile("v_floor_f32 %0, %0" : "+v"(x));
#define A_CVTI(x) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
#define A_CVTF(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define A_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_FMAK(x) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(x) : "v"(a));
#define A_SUB(x) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define A_FMAC(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(a));
#define A_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x));
#define A_CMP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
#define A_CMPS(x) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x), "v"(a) : "s20", "s21");
#define A_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_LSHL(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define A_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
#define A_CVTUB(x) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(x));
#define A_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define A_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_FMAMIX(x) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(a), "v"(b));
#define A_FMAMIXH(x) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(a), "v"(b));
#define A_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_CVTF16(x) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x));
#define A_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##x) : "v"(da));
KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_min, A_MIN) KERNEL(k_max, A_MAX) KERNEL(k_max3, A_MAX3)
KERNEL(k_cnd, A_CND) KERNEL(k_cmp, A_CMP) KERNEL(k_cmps, A_CMPS) KERNEL(k_and, A_AND) KERNEL(k_xor, A_XOR) KERNEL(k_addu, A_ADDU)
KERNEL(k_cnd64, A_CND64) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_sub, A_SUB) KERNEL(k_fmac, A_FMAC) KERNEL(k_min3, A_MIN3) KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_or3, A_OR3) KERNEL(k_bfe, A_BFE)
KERNEL(k_cmpcnd4, A_CMPCND4) KERNEL(k_cmpcnds, A_CMPCNDS) KERNEL(k_cmp2cnd, A_CMP2CND) KERNEL(k_minu, A_MINU) KERNEL(k_ashr, A_ASHR) KERNEL(k_andor, A_ANDOR)
KERNEL(k_perm, A_PERM) KERNEL(k_med3, A_MED3) KERNEL(k_sqrt, A_SQRT) KERNEL(k_rsq, A_RSQ) KERNEL(k_exp, A_EXP) KERNEL(k_mulu24, A_MULU24) KERNEL(k_mulhi, A_MULHI)
KERNEL(k_subu, A_SUBU) KERNEL(k_or, A_OR) KERNEL(k_lshr, A_LSHR) KERNEL(k_floor, A_FLOOR) KERNEL(k_cvti, A_CVTI) KERNEL(k_cvtf, A_CVTF) KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_fmak, A_FMAK)
KERNEL(k_fmamix, A_FMAMIX) KERNEL(k_fmamixh, A_FMAMIXH) KERNEL(k_bfi, A_BFI) KERNEL(k_cvtf16, A_CVTF16)
KERNEL(k_lshl, A_LSHL) KERNEL(k_mov, A_MOV) KERNEL(k_cvtub, A_CVTUB) KERNEL(k_rcp, A_RCP) KERNEL(k_mullo, A_MULLO) KERNEL(k_mad24, A_MAD24)
// packed fp32: two lanes' worth per instruction on 64-bit register pairs
__global__ void __launch_bounds__(256) k_pkmul(float *out, int iters, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {(float) threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f, av = {a, b};
#define A_PK(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(av));
    for (int i = 0; i < iters; ++i) { R64(A_PK) }
    f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
__global__ void __launch_bounds__(256) k_fma64(float *out, int iters, float a, float b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, da = a;
#define A_F64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(da));
    for (int i = 0; i < iters; ++i) { R64(A_F64) }
    out[blockIdx.x * 256 + threadIdx.x] = (float) (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}
typedef void (*kern_t)(float *, int, float, float);
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8;
    float *out; hipMalloc(&out, (size_t) blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char *name; kern_t k; } ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_min_f32", k_min}, {"v_max_f32", k_max},
        {"v_max3_f32", k_max3}, {"v_cndmask_b32", k_cnd}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"v_cmp+v_cndmask (PAIR)", k_cmpcnd}, {"cmp + 4 fmac + cndmask (6 instr)", k_cmpcnd4}, {"v_cmp sgpr + v_cndmask_e64 (PAIR)", k_cmpcnds}, {"cmp + 4 cndmask (5 instr)", k_cmp2cnd},
        {"v_min_u32", k_minu}, {"v_ashrrev_i32", k_ashr}, {"v_and_or_b32", k_andor}, {"v_perm_b32", k_perm}, {"v_med3_f32", k_med3}, {"v_sqrt_f32", k_sqrt}, {"v_rsq_f32", k_rsq},
        {"v_exp_f32", k_exp}, {"v_mul_u32_u24", k_mulu24}, {"v_mul_hi_u32", k_mulhi}, {"v_sub_u32", k_subu}, {"v_or_b32", k_or}, {"v_lshrrev_b32", k_lshr}, {"v_floor_f32", k_floor},
        {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_u32", k_cvtf}, {"v_mbcnt_lo", k_mbcnt}, {"v_fmaak_f32", k_fmak}, {"v_sub_f32", k_sub}, {"v_fmac_f32", k_fmac}, {"v_min3_f32", k_min3}, {"v_lshl_add_u32", k_lshladd}, {"v_or3_b32", k_or3}, {"v_bfe_u32", k_bfe}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmps}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor},
        {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl}, {"v_mov_b32", k_mov}, {"v_cvt_f32_ubyte1", k_cvtub}, {"v_rcp_f32", k_rcp}, {"v_mul_lo_u32", k_mullo},
        {"v_mad_u32_u24", k_mad24}, {"v_fma_mix_f32 (f16 lo src)", k_fmamix}, {"v_fma_mix_f32 (f16 hi src)", k_fmamixh}, {"v_bfi_b32", k_bfi}, {"v_cvt_f32_f16", k_cvtf16}, {"v_pk_mul_f32", k_pkmul}, {"v_fma_f64", k_fma64}};
    const int iters = 4000;
    for (auto &k : ks) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        const double mult = strstr(k.name, "6 instr") ? 6 : strstr(k.name, "5 instr") ? 5 : strstr(k.name, "PAIR") ? 2 : 1;
        const double insts = (double) blocks * 4 * iters * 64 * mult;
        const double per_simd_s = insts / (cus * 4) / (best * 1e-3);
        printf("%-22s %7.3f ms  %6.1f G wave-instr/s  %.2f cycles per instruction per SIMD (at %.2f GHz)\n", k.name, best, insts / (best * 1e-3) / 1e9,
               (p.clockRate * 1e3) / per_simd_s, p.clockRate / 1e6);
    }
    return 0;
}
