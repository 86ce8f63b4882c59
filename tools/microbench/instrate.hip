// Issue cost (cycles per wave64 instruction per SIMD) of the VALU instruction classes k_g2p2g is made of.
// 8 independent chains per wave, 8 waves per SIMD; reported relative to v_fma_f32 = 4 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, BODY, DECL, FIN)                                                         \
    __global__ void NAME(float* out, int iters, float a) {                                    \
        DECL                                                                                  \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)); } \
        FIN                                                                                   \
    }
#define DECLF float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#define FINF out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
#define DECLD double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#define FIND out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
#define I8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define U8(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"
#define S8(OP) OP " %0, %0, %0\n" OP " %1, %1, %1\n" OP " %2, %2, %2\n" OP " %3, %3, %3\n" OP " %4, %4, %4\n" OP " %5, %5, %5\n" OP " %6, %6, %6\n" OP " %7, %7, %7\n"
#define D8 "v_fmac_f32_dpp %0, %0, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fmac_f32_dpp %1, %1, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
           "v_fmac_f32_dpp %2, %2, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fmac_f32_dpp %3, %3, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
           "v_fmac_f32_dpp %4, %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fmac_f32_dpp %5, %5, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
           "v_fmac_f32_dpp %6, %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fmac_f32_dpp %7, %7, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
KERNEL(k_fma, I8("v_fmac_f32"), DECLF, FINF)
KERNEL(k_mul, I8("v_mul_f32"), DECLF, FINF)
KERNEL(k_dpp, D8, DECLF, FINF)
KERNEL(k_rcp, U8("v_rcp_f32"), DECLF, FINF)
KERNEL(k_rsq, U8("v_rsq_f32"), DECLF, FINF)
KERNEL(k_sqrt, U8("v_sqrt_f32"), DECLF, FINF)
KERNEL(k_exp, U8("v_exp_f32"), DECLF, FINF)
KERNEL(k_mov, U8("v_mov_b32"), DECLF, FINF)
KERNEL(k_fma64, S8("v_mul_f64"), DECLD, FIND)
KERNEL(k_add, I8("v_add_f32"), DECLF, FINF)
KERNEL(k_fma3, "v_fma_f32 %0, %0, %8, %1\nv_fma_f32 %1, %1, %8, %2\nv_fma_f32 %2, %2, %8, %3\nv_fma_f32 %3, %3, %8, %4\nv_fma_f32 %4, %4, %8, %5\nv_fma_f32 %5, %5, %8, %6\nv_fma_f32 %6, %6, %8, %7\nv_fma_f32 %7, %7, %8, %0\n", DECLF, FINF)
KERNEL(k_cnd, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n", DECLF, FINF)
// packed f32: four register PAIRS per thread
#define DECLP typedef float f2 __attribute__((ext_vector_type(2))); f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f; f2 aa = {a, a};
#define FINP out[blockIdx.x * blockDim.x + threadIdx.x] = (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7).x;
#define KERNELP(NAME, OP)                                                                     \
    __global__ void NAME(float* out, int iters, float a) {                                    \
        DECLP                                                                                 \
        for (int it = 0; it < iters; ++it) { asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n" OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa)); } \
        FINP                                                                                  \
    }
#define KERNELP2(NAME, OP)                                                                    \
    __global__ void NAME(float* out, int iters, float a) {                                    \
        DECLP                                                                                 \
        for (int it = 0; it < iters; ++it) { asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa)); } \
        FINP                                                                                  \
    }
KERNELP(k_pkfma, "v_pk_fma_f32")
KERNELP2(k_pkadd, "v_pk_add_f32")
KERNELP2(k_pkmul, "v_pk_mul_f32")

// cvt f32 -> f64 -> f32 round trip (two instructions per chain step)
__global__ void k_cvt(float* out, int iters, float a) {
    DECLF
    double d0, d1, d2, d3, d4, d5, d6, d7;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\nv_cvt_f64_f32 %12, %4\nv_cvt_f64_f32 %13, %5\nv_cvt_f64_f32 %14, %6\nv_cvt_f64_f32 %15, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <class K> double run(K k, float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4096;
    const double ref = run(k_fma, d, iters);
    struct { const char* name; double ms; } r[] = {
        {"v_fmac_f32", ref}, {"v_mul_f32", run(k_mul, d, iters)}, {"v_fmac_f32_dpp row_shl:1", run(k_dpp, d, iters)}, {"v_rcp_f32", run(k_rcp, d, iters)},
        {"v_rsq_f32", run(k_rsq, d, iters)}, {"v_sqrt_f32", run(k_sqrt, d, iters)}, {"v_exp_f32", run(k_exp, d, iters)}, {"v_mov_b32", run(k_mov, d, iters)},
        {"v_mul_f64", run(k_fma64, d, iters)}, {"v_cvt_f64_f32", run(k_cvt, d, iters)}, {"v_add_f32", run(k_add, d, iters)}, {"v_fma_f32 (3 sources)", run(k_fma3, d, iters)},
        {"v_cndmask_b32", run(k_cnd, d, iters)}, {"v_pk_fma_f32", run(k_pkfma, d, iters)}, {"v_pk_add_f32", run(k_pkadd, d, iters)}, {"v_pk_mul_f32", run(k_pkmul, d, iters)}};
    for (auto& x : r) printf("%-28s %.3f ms  = %.2f cycles/wave-instr (v_fmac_f32 := 4)\n", x.name, x.ms, 4.0 * x.ms / ref);
    return 0;
}
