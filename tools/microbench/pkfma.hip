// Is v_pk_fma_f32 full rate on gfx950?  8 independent chains per thread, scalar vs packed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float* out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, int iters, float a, float b) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)blocks * 4 * iters * 8;   // wave instructions
        printf("scalar v_fma_f32   : %.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", ms, ms * 1e-3 * 2.4e9 / (inst / 1024));
        hipEventRecord(e0); hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 0.001f); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("packed v_pk_fma_f32: %.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", ms, ms * 1e-3 * 2.4e9 / (inst / 1024));
    }
    return 0;
}
