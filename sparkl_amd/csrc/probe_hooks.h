// Opt-in analysis hooks of the hot kernels (not part of the product path: every function here is empty unless an analysis build
// defines its macro; tools/ab_build.sh <name> -DSPK_EXP_DUMMY_VALU=200 builds such a variant beside the product library).
//   SPK_EXP_DUMMY_VALU=N  N extra dependency-free VALU instructions per particle batch of k_g2p2g (VALU sensitivity)
//   SPK_EXP_DUMMY_LDS=N   N extra dependent LDS reads per particle batch (LDS latency sensitivity)
//   SPK_MARKS             section markers in the ISA (device_math.h: SPK_MARK), counted by tools/isa_sections.py
// The ablations of round 2 (no stress / no DPP / no atomics / no prefetch / LDS padding) are described in
// profiles/r02_kernel_experiments.txt; their switches are gone from the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace spk {

template <class ScatterT>
__device__ __forceinline__ void probe_extra_work(float (*s_A)[64], int tid, const float* px, const float* vel, ScatterT& sc) {
#ifdef SPK_EXP_DUMMY_LDS
    {
        volatile float* vp = &s_A[0][0];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < SPK_EXP_DUMMY_LDS; ++k) acc += vp[((tid + 7 * k) & 63) + 64 * (k % 9)];
        if (acc == 123.456f) sc.val_m += 1.f;
    }
#endif
#ifdef SPK_EXP_DUMMY_VALU
    {
        float d0 = px[0], d1 = px[1], d2 = px[2], d3 = vel[0];
#pragma unroll
        for (int k = 0; k < SPK_EXP_DUMMY_VALU / 4; ++k) {
            asm volatile("v_fmac_f32 %0, %0, %1\n v_fmac_f32 %1, %1, %2\n v_fmac_f32 %2, %2, %3\n v_fmac_f32 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        }
        if (d0 + d1 + d2 + d3 == 123.456f) sc.val_m += 1.f;
    }
#endif
    (void)s_A; (void)tid; (void)px; (void)vel; (void)sc;
}

}  // namespace spk
