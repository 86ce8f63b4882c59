#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2048;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int stride_mode, int active_mod) {
    __shared__ unsigned long long t64[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) t64[i] = 0;
    __syncthreads();
    int addr = (tid * 7) & 2047;
    if (stride_mode == 1) addr = ((tid >> 3) * 5) & 2047;
    const bool active = (lane % active_mod) == 0;
    if (active) {
        if (MODE == 0) { for (int it = 0; it < ITER; ++it) atomicAdd(&t64[(addr + it * 9) & 2047], (unsigned long long)tid); }
        else if (MODE == 1) { double* d = reinterpret_cast<double*>(t64); for (int it = 0; it < ITER; ++it) atomicAdd(&d[(addr + it * 9) & 2047], (double)tid); }
        else if (MODE == 2) { unsigned* u = reinterpret_cast<unsigned*>(t64); for (int it = 0; it < ITER; ++it) atomicAdd(&u[(addr + it * 9) & 4095], (unsigned)tid); }
        else if (MODE == 3) { float* f = reinterpret_cast<float*>(t64); for (int it = 0; it < ITER; ++it) atomicAdd(&f[(addr + it * 9) & 4095], (float)tid); }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)t64[5];
}
template <int MODE> void run(const char* name, int sm, int am) {
    float* d; (void)hipMalloc(&d, 4096 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sm, am);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sm, am);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double wi = (double)blocks / 256 * 4 * ITER;
    printf("%-52s %8.3f ms -> ~%6.1f cyc per wave-instr per CU\n", name, ms, ms * 1e6 / wi * 2.1);
    (void)hipFree(d);
}
int main() {
    run<0>("ds_add_u64 64 lanes distinct", 0, 1);
    run<0>("ds_add_u64 16/64 lanes distinct", 0, 4);
    run<0>("ds_add_u64 64 lanes, 8 lanes/addr", 1, 1);
    run<1>("ds_add_f64 64 lanes distinct", 0, 1);
    run<1>("ds_add_f64 16/64 lanes", 0, 4);
    run<2>("ds_add_u32 64 lanes distinct", 0, 1);
    run<2>("ds_add_u32 16/64 lanes", 0, 4);
    run<2>("ds_add_u32 8/64 lanes", 0, 8);
    run<3>("ds_add_f32 8/64 lanes", 0, 8);
    run<3>("ds_add_f32 4/64 lanes", 0, 16);
    return 0;
}
