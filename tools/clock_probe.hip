// Shader clock while the bench kernels run: a one-wave kernel on its own stream samples clock64() (s_memtime) against
// wall_clock64() (constant 100 MHz) every ~20 us.  Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o clock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k_probe(uint64_t* out, int n, uint64_t period_wall) {
    if (threadIdx.x != 0) return;
    uint64_t w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        uint64_t target = w0 + (uint64_t)(i + 1) * period_wall;
        while (wall_clock64() < target) __builtin_amdgcn_s_sleep(32);
        out[2 * i] = wall_clock64();
        out[2 * i + 1] = clock64();
    }
}
extern "C" int probe_launch(uint64_t* dev_out, int n, uint64_t period_wall, hipStream_t st) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, st, dev_out, n, period_wall);
    return (int)hipGetLastError();
}
