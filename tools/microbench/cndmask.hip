// What does v_cndmask_b32 cost?  (instrate.hip read 31 cycles for a chain on an undefined vcc.)  Variants: vcc set by a v_cmp before the loop, an SGPR-pair mask
// (VOP3), v_cmp + v_cndmask pairs as the compiler emits them, against v_fmac_f32 = 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DECLF float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#define FINF out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
#define OUTS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)
__global__ void k_fma(float* out, int iters, float a) { DECLF for (int it = 0; it < iters; ++it) asm volatile("v_fmac_f32 %0, %0, %8\nv_fmac_f32 %1, %1, %8\nv_fmac_f32 %2, %2, %8\nv_fmac_f32 %3, %3, %8\nv_fmac_f32 %4, %4, %8\nv_fmac_f32 %5, %5, %8\nv_fmac_f32 %6, %6, %8\nv_fmac_f32 %7, %7, %8\n" OUTS); FINF }
__global__ void k_cnd_vcc(float* out, int iters, float a) { DECLF asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x0), "v"(a) : "vcc");
    for (int it = 0; it < iters; ++it) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n" OUTS : "vcc"); FINF }
__global__ void k_cnd_sgpr(float* out, int iters, float a) { DECLF unsigned long long m = __ballot(x0 > 17.f);
    for (int it = 0; it < iters; ++it) asm volatile("v_cndmask_b32 %0, %0, %8, %9\nv_cndmask_b32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, %9\nv_cndmask_b32 %4, %4, %8, %9\nv_cndmask_b32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, %9\nv_cndmask_b32 %7, %7, %8, %9\n" OUTS, "s"(m)); FINF }
// independent destinations (no read-after-write chain): is it a dependency stall?
__global__ void k_cnd_indep(float* out, int iters, float a) { DECLF float y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0; unsigned long long m = __ballot(x0 > 17.f);
    for (int it = 0; it < iters; ++it) asm volatile("v_cndmask_b32 %0, %8, %16, %17\nv_cndmask_b32 %1, %9, %16, %17\nv_cndmask_b32 %2, %10, %16, %17\nv_cndmask_b32 %3, %11, %16, %17\nv_cndmask_b32 %4, %12, %16, %17\nv_cndmask_b32 %5, %13, %16, %17\nv_cndmask_b32 %6, %14, %16, %17\nv_cndmask_b32 %7, %15, %16, %17\n"
        : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3), "=v"(y4), "=v"(y5), "=v"(y6), "=v"(y7) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(a), "s"(m));
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7; }
// v_cmp (to an SGPR pair) + v_cndmask on it, as compiled selects look
__global__ void k_cmp_cnd(float* out, int iters, float a) { DECLF
    for (int it = 0; it < iters; ++it) asm volatile("v_cmp_gt_f32 s[10:11], %0, %8\nv_cmp_gt_f32 s[12:13], %1, %8\nv_cmp_gt_f32 s[14:15], %2, %8\nv_cmp_gt_f32 s[16:17], %3, %8\ns_nop 1\n"
        "v_cndmask_b32 %4, %4, %8, s[10:11]\nv_cndmask_b32 %5, %5, %8, s[12:13]\nv_cndmask_b32 %6, %6, %8, s[14:15]\nv_cndmask_b32 %7, %7, %8, s[16:17]\n" OUTS : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17"); FINF }
__global__ void k_cmp_cnd_vcc(float* out, int iters, float a) { DECLF
    for (int it = 0; it < iters; ++it) asm volatile("v_cmp_gt_f32 vcc, %0, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_gt_f32 vcc, %1, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_gt_f32 vcc, %2, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_cmp_gt_f32 vcc, %3, %8\nv_cndmask_b32 %7, %7, %8, vcc\n" OUTS : "vcc"); FINF }
// the same with two plain VALU instructions between the compare and the select
__global__ void k_cmp_cnd_vcc_gap(float* out, int iters, float a) { DECLF
    for (int it = 0; it < iters; ++it) asm volatile("v_cmp_gt_f32 vcc, %0, %8\nv_fmac_f32 %0, %0, %8\nv_fmac_f32 %1, %1, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_gt_f32 vcc, %2, %8\nv_fmac_f32 %2, %2, %8\nv_fmac_f32 %3, %3, %8\nv_cndmask_b32 %5, %5, %8, vcc\n" OUTS : "vcc"); FINF }
__global__ void k_cmp(float* out, int iters, float a) { DECLF
    for (int it = 0; it < iters; ++it) asm volatile("v_cmp_gt_f32 s[10:11], %0, %8\nv_cmp_gt_f32 s[12:13], %1, %8\nv_cmp_gt_f32 s[14:15], %2, %8\nv_cmp_gt_f32 s[16:17], %3, %8\nv_cmp_gt_f32 s[10:11], %4, %8\nv_cmp_gt_f32 s[12:13], %5, %8\nv_cmp_gt_f32 s[14:15], %6, %8\nv_cmp_gt_f32 s[16:17], %7, %8\n" OUTS : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17"); FINF }
// bfi / and-or select without a mask register
__global__ void k_max(float* out, int iters, float a) { DECLF for (int it = 0; it < iters; ++it) asm volatile("v_max_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_max_f32 %2, %2, %8\nv_max_f32 %3, %3, %8\nv_max_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_max_f32 %6, %6, %8\nv_max_f32 %7, %7, %8\n" OUTS); FINF }
template <class K> double run(K k, float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, d, iters, 0.999f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); }
    return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4096;
    const double ref = run(k_fma, d, iters);
    struct { const char* name; double ms; } r[] = {{"v_fmac_f32", ref}, {"v_cndmask vcc (chain)", run(k_cnd_vcc, d, iters)}, {"v_cndmask sgpr mask (chain)", run(k_cnd_sgpr, d, iters)},
        {"v_cndmask sgpr mask (independent)", run(k_cnd_indep, d, iters)}, {"4 v_cmp->sgpr + 4 v_cndmask", run(k_cmp_cnd, d, iters)}, {"4 v_cmp->vcc + 4 v_cndmask vcc", run(k_cmp_cnd_vcc, d, iters)}, {"2 x (cmp vcc, 2 fmac, cndmask vcc)", run(k_cmp_cnd_vcc_gap, d, iters)}, {"v_cmp_gt_f32 -> sgpr", run(k_cmp, d, iters)}, {"v_max_f32", run(k_max, d, iters)}};
    for (auto& x : r) printf("%-36s %.3f ms  = %.2f cycles/wave-instr (v_fmac_f32 := 4)\n", x.name, x.ms, 4.0 * x.ms / ref);
    return 0;
}
