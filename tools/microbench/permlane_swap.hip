#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 128 * 4);
    k<<<1, 64>>>(d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %d: r0=%u r1=%u\n", i, h[i], h[64 + i]);
}
