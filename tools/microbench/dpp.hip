#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__global__ void k(int* out) {
    int lane = threadIdx.x;
    out[lane] = dpp_i<0x101>(lane + 100);        // row_shl:1
    out[64 + lane] = dpp_i<0x111>(lane + 100);   // row_shr:1
    out[128 + lane] = dpp_i<0x104>(lane + 100);  // row_shl:4
}
int main() {
    int* d; (void)hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 20; ++i) printf("lane %2d shl1 %3d shr1 %3d shl4 %3d\n", i, h[i], h[64 + i], h[128 + i]);
    return 0;
}
