#include <hip/hip_runtime.h>
#include <cstdio>
// LDS op throughput per CU: 256-thread blocks, 8 blocks/CU worth of work, each thread issues ITER ops.
constexpr int ITER = 2048;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int stride_mode) {
    __shared__ float tile[4 * 512];
    __shared__ float4 tile4[512];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) tile[i] = 0.f;
    for (int i = tid; i < 512; i += 256) tile4[i] = make_float4(0, 0, 0, 0);
    __syncthreads();
    int addr = (tid * 7) & 511;           // distinct-ish addresses
    if (stride_mode == 1) addr = (tid >> 3) * 5 & 511;  // 8 lanes share an address
    const bool active = (MODE == 2) ? ((lane & 3) == 0) : true;
    float v = 1.0f + tid;
    if (MODE == 0 || MODE == 2) {
        if (active)
            for (int it = 0; it < ITER; ++it) { atomicAdd(&tile[(addr + it * 9) & 2047], v); }
    } else if (MODE == 1) {
        unsigned* t = reinterpret_cast<unsigned*>(tile);
        for (int it = 0; it < ITER; ++it) { atomicAdd(&t[(addr + it * 9) & 2047], (unsigned)tid); }
    } else if (MODE == 3) {
        for (int it = 0; it < ITER; ++it) { float4 x = tile4[(addr + it * 9) & 511]; x.x += v; x.y += v; x.z += v; x.w += v; tile4[(addr + it * 9) & 511] = x; }
    } else if (MODE == 4) {
        for (int it = 0; it < ITER; ++it) { int a = (addr + it * 9) & 2047; float x = tile[a]; tile[a] = x + v; }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = tile[5] + tile4[3].x;
}
template <int MODE> void run(const char* name, int sm, float ops_per_iter) {
    float* d; (void)hipMalloc(&d, 4096 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sm);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sm);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double wave_instr_per_cu = (double)blocks / 256 * 4 * ITER * ops_per_iter;
    printf("%-44s %8.3f ms  -> %6.1f ns per wave-instr per CU  (~%5.1f cyc @2.1GHz)\n", name, ms, ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.1);
    (void)hipFree(d);
}
int main() {
    run<0>("ds_add_f32, 64 lanes, distinct addr", 0, 1);
    run<0>("ds_add_f32, 64 lanes, 8 lanes/addr", 1, 1);
    run<2>("ds_add_f32, 16/64 lanes active, distinct", 0, 1);
    run<1>("ds_add_u32, 64 lanes, distinct addr", 0, 1);
    run<1>("ds_add_u32, 64 lanes, 8 lanes/addr", 1, 1);
    run<3>("ds_read_b128 + ds_write_b128 (RMW float4)", 0, 2);
    run<4>("ds_read_b32 + ds_write_b32 (RMW float)", 0, 2);
    return 0;
}
