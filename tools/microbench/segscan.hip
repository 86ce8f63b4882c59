#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void k(const int* keys, const float* c, float* out_s, int* out_head, int* out_m) {
    const int tid = threadIdx.x; const int lane = tid & 63;
    int cell_key = keys[tid]; float c0 = c[tid];
    const int prev_key = __shfl_up(cell_key, 1);
    const bool head = ((lane & 15) == 0) || (prev_key != cell_key);
    bool m1, m2, m4, m8;
    { const int r = lane & 15;
      m1 = (r + 1 < 16) && (__shfl_down(cell_key, 1) == cell_key);
      m2 = m1 && (r + 2 < 16) && (__shfl_down((int)m1, 1) != 0);
      m4 = m2 && (r + 4 < 16) && (__shfl_down((int)m2, 2) != 0);
      m8 = m4 && (r + 8 < 16) && (__shfl_down((int)m4, 4) != 0); }
    float t;
    t = __shfl_down(c0, 1); c0 += m1 ? t : 0.f;
    t = __shfl_down(c0, 2); c0 += m2 ? t : 0.f;
    t = __shfl_down(c0, 4); c0 += m4 ? t : 0.f;
    t = __shfl_down(c0, 8); c0 += m8 ? t : 0.f;
    out_s[tid] = c0; out_head[tid] = head; out_m[tid] = m1 | (m2<<1) | (m4<<2) | (m8<<3);
}
int main() {
    const int N = 64; std::vector<int> keys(N); std::vector<float> c(N);
    int kk = 0; for (int i = 0; i < N; ++i) { if (rand() % 3 == 0) kk++; keys[i] = kk; c[i] = 1.0f; }
    int *dk, *dh, *dm; float *dc, *ds;
    hipMalloc(&dk, N*4); hipMalloc(&dh, N*4); hipMalloc(&dm, N*4); hipMalloc(&dc, N*4); hipMalloc(&ds, N*4);
    hipMemcpy(dk, keys.data(), N*4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), N*4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dk, dc, ds, dh, dm);
    std::vector<float> s(N); std::vector<int> h(N), m(N);
    hipMemcpy(s.data(), ds, N*4, hipMemcpyDeviceToHost); hipMemcpy(h.data(), dh, N*4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, N*4, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) printf("%2d key %2d head %d m %x s %.0f\n", i, keys[i], h[i], m[i], s[i]);
    return 0;
}
