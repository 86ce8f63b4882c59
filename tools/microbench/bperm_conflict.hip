// Does ds_bpermute_b32 pay for "bank conflicts" of its SOURCE lanes, and what does a 2-way conflict cost a ds_read_b32?
// Patterns of the run-task scatter's visits on the lattice: 48 task lanes fetch from 16 distinct sources {0, 4, ..., 60} + j (the lanes of
// one 32-lane group read sources that differ by 32), against the same 16 sources swizzled so that they differ mod 32, and against identity.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/bp tools/microbench/bperm_conflict.hip && /tmp/bp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>   // 0 bpermute, 1 ds_read_b32
__global__ __launch_bounds__(256) void k(const int* src_of_lane, float* out, int iters) {
    __shared__ float buf[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    buf[w][lane] = (float)lane;
    __syncthreads();
    int s = src_of_lane[lane];
    float x = (float)lane, acc = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float v;
            if (MODE == 0) v = __int_as_float(__builtin_amdgcn_ds_bpermute(s << 2, __float_as_int(x)));
            else v = ((volatile float*)buf[w])[s];
            acc += v;
            s = (s + 1) & 63;          // the visit index j moves every source by one
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    int h[4][64];
    for (int l = 0; l < 64; ++l) {
        const int c = l & 15;                         // 16 chunks, three z-planes (lanes 48-63 idle in the kernel: here they repeat)
        h[0][l] = l;                                  // identity
        h[1][l] = 4 * c;                              // lattice: sources 0, 4, ..., 60 (2-way mod 32)
        h[2][l] = (4 * c) ^ (((4 * c) >> 4) & 2);     // swizzled
        h[3][l] = 0;                                  // all the same source (broadcast)
    }
    int* d; float* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 1024 * 256 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[4] = {"identity", "lattice (stride 4, 2-way mod 32)", "swizzled", "broadcast"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int p = 0; p < 4; ++p) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<1024, 256>>>(d + 64 * p, o, 2000); else k<1><<<1024, 256>>>(d + 64 * p, o, 2000);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%-12s %-36s %.3f ms\n", mode == 0 ? "ds_bpermute" : "ds_read_b32", names[p], best);
        }
    return 0;
}
