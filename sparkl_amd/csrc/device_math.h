// Device-side 3x3 algebra, SVD and material models for the MPM hot path (gfx950).
// All matrices are column-major like nalgebra: element (i, j) at m[i + 3 j].
//
// Reference semantics implemented here (f32):
//   QuadraticKernel::{inv_d, eval_all}          src_core/dynamics/solver/kernel.rs:10-21
//   inv_exact                                   src_core/utils/physics.rs:7-14
//   CorotatedLinearElasticity::kirchhoff_stress src_core/dynamics/models/elasticity_corotated_linear.rs:31-78
//   NeoHookeanElasticity::kirchhoff_stress      src_core/dynamics/models/elasticity_neo_hookean.rs:43-67
//   MonaghanSphEos::{pressure,kirchhoff_stress} src_core/dynamics/models/eos_monaghan_sph.rs:28-63
//   DruckerPragerPlasticity::*                  src_core/dynamics/models/plasticity_drucker_prager.rs:36-104
//   *::pos_energy, *::timestep_bound            same files
// nalgebra's `svd_unordered` (third party, absent) is replaced by a one-sided Jacobi SVD; every call
// site only uses SVD-invariant combinations (U f(S) V^T), so sign/order conventions do not matter.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../include/sparkl_hip.h"

#define SPK_DEV __device__ __forceinline__
// section markers in the ISA of an analysis build (-DSPK_MARKS): tools/isa_sections.py counts the instructions between them
#ifdef SPK_MARKS
#define SPK_MARK(name) asm volatile("; SPKMARK " name)
#else
#define SPK_MARK(name) ((void)0)
#endif
// SPK_HOT(cond): nothing in the product build.  An analysis build (-DSPK_HOT_ONLY, tools/isa_sections.py --hot) turns it into
// __builtin_assume(cond), which drops the cold side of the branch that follows: the static instruction count of such a build is the
// count of the hot path of the 10M sand column (series return map, stress taken from the return map, nothing failed, phase == 1).
#ifdef SPK_HOT_ONLY
#define SPK_HOT(cond) __builtin_assume(cond)
#else
#define SPK_HOT(cond) ((void)0)
#endif

namespace spk {

struct M3 {
    float m[9];
};

SPK_DEV M3 m3_identity() { return M3{{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; }
SPK_DEV M3 m3_zero() { return M3{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}; }
SPK_DEV M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            r.m[i + 3 * j] = a.m[i] * b.m[3 * j] + a.m[i + 3] * b.m[3 * j + 1] + a.m[i + 6] * b.m[3 * j + 2];
    return r;
}
// a * b^T
SPK_DEV M3 m3_mul_bt(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            r.m[i + 3 * j] = a.m[i] * b.m[j] + a.m[i + 3] * b.m[j + 3] + a.m[i + 6] * b.m[j + 6];
    return r;
}
SPK_DEV float m3_det(const M3& a) {
    // nalgebra's cofactor expansion (first row)
    const float m11 = a.m[0], m21 = a.m[1], m31 = a.m[2], m12 = a.m[3], m22 = a.m[4], m32 = a.m[5], m13 = a.m[6], m23 = a.m[7], m33 = a.m[8];
    const float minor_m12_m23 = m22 * m33 - m32 * m23;
    const float minor_m11_m23 = m21 * m33 - m31 * m23;
    const float minor_m11_m22 = m21 * m32 - m31 * m22;
    return m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22;
}
SPK_DEV float m3_frob(const M3& a) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += a.m[i] * a.m[i];
    return sqrtf(s);
}

SPK_DEV float inv_exact(float e) { return e == 0.f ? 0.f : 1.f / e; }

// QuadraticKernel::eval_all
SPK_DEV void eval_all(float x, float w[3]) {
    const float a = 1.5f - x, b = x - 1.0f, c = x - 0.5f;
    w[0] = 0.5f * (a * a);
    w[1] = 0.75f - (b * b);
    w[2] = 0.5f * (c * c);
}

// ---------------------------------------------------------------------------------------------- SVD
// One-sided (Hestenes) Jacobi: on return A = W V^T with W's columns mutually orthogonal;
// sigma_j = |w_j|, u_j = w_j / sigma_j.  `w` is A on entry, `v` is overwritten.
struct Svd3 {
    M3 w;        // U * diag(sigma)
    M3 v;        // V
    float s[3];  // sigma >= 0
};

// fast reciprocal / rsqrt / sqrt (1 ulp hardware ops): a Jacobi rotation only has to be *nearly* optimal,
// and c^2 + s^2 = 1 holds to ~1e-7, which keeps V orthogonal to f32 resolution.
SPK_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
SPK_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
SPK_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Orthogonalises columns p and q; returns true when their coupling |wp.wq| / (|wp||wq|) was above 1e-6 BEFORE the
// rotation, i.e. one more sweep is needed.  Why 1e-6 and not sqrt(eps): with (nearly) equal singular values -- F ~ I,
// the common case -- the rotation angle is O(1) however small the coupling, so a later rotation of the sweep carries
// couplings over linearly, and a residual coupling r shifts a near-degenerate singular value by O(r), not O(r^2).
// Measured in f32 (tools/svd_threshold_scan.py): threshold 3e-4 leaves sigma errors up to 7.5e-5 for |F - I| ~ 1e-4,
// 1e-6 keeps them <= 5e-7 at every scale for 1 (undeformed) to ~3.5 (strained) sweeps.  The rotation tangent itself is
// NOT a convergence measure (the first version used it and spun ~5 sweeps on undeformed particles).
// (a rotation with two v_rsq_f32 and no v_rcp/v_sqrt was also tried: not faster, and 6x worse parity on the sand case)
typedef float spk_f2 __attribute__((ext_vector_type(2)));
SPK_DEV bool jacobi_pair(float* wp, float* wq, float* vp, float* vq) {
    // dot products two lanes at a time: (alpha, beta) = sum_i (wp_i^2, wq_i^2) as packed FMAs
    spk_f2 ab2 = spk_f2{wp[0], wq[0]} * spk_f2{wp[0], wq[0]};
    ab2 += spk_f2{wp[1], wq[1]} * spk_f2{wp[1], wq[1]};
    ab2 += spk_f2{wp[2], wq[2]} * spk_f2{wp[2], wq[2]};
    const float alpha = ab2.x, beta = ab2.y;
    const float gamma = wp[0] * wq[0] + wp[1] * wq[1] + wp[2] * wq[2];
    const float g2 = gamma * gamma, ab = alpha * beta;
    float c = 1.f, s = 0.f;
    if (g2 > 1.0e-18f * ab) {
        const float zeta = (beta - alpha) * fast_rcp(2.f * gamma);
        const float t = copysignf(1.f, zeta) * fast_rcp(fabsf(zeta) + fast_sqrt(1.f + zeta * zeta));
        c = fast_rsq(1.f + t * t);
        s = c * t;
    }
    // the rotation acts on (w, v) pairs: v_pk_mul / v_pk_fma retire the w and the v component together
    const spk_f2 cc = {c, c}, ss = {s, s};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const spk_f2 a = {wp[i], vp[i]}, b = {wq[i], vq[i]};
        const spk_f2 np = cc * a - ss * b;
        const spk_f2 nq = ss * a + cc * b;
        wp[i] = np.x; vp[i] = np.y;
        wq[i] = nq.x; vq[i] = nq.y;
    }
    return g2 > 1.0e-12f * ab;
}

// Cyclic one-sided Jacobi: sweep until a sweep starts with every coupling of every lane below 1e-6 (wave-uniform
// branch).  Undeformed particles take 1 sweep, MPM-sized strains 3-4.
template <int MAX_SWEEPS>
SPK_DEV Svd3 svd3(const M3& a) {
    Svd3 r;
    r.w = a;
    r.v = m3_identity();
#pragma unroll 1
    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        bool more = jacobi_pair(r.w.m + 0, r.w.m + 3, r.v.m + 0, r.v.m + 3);
        more |= jacobi_pair(r.w.m + 0, r.w.m + 6, r.v.m + 0, r.v.m + 6);
        more |= jacobi_pair(r.w.m + 3, r.w.m + 6, r.v.m + 3, r.v.m + 6);
        if (!__any(more)) break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) r.s[j] = fast_sqrt(r.w.m[3 * j] * r.w.m[3 * j] + r.w.m[3 * j + 1] * r.w.m[3 * j + 1] + r.w.m[3 * j + 2] * r.w.m[3 * j + 2]);
    return r;
}
#ifndef SPK_SVD_SWEEPS
#define SPK_SVD_SWEEPS 8
#endif

// sum_j g_j * w_j v_j^T  ==  U diag(g_j * sigma_j) V^T
SPK_DEV M3 svd_compose_scaled(const Svd3& d, const float g[3]) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            r.m[i + 3 * c] = g[0] * d.w.m[i] * d.v.m[c] + g[1] * d.w.m[i + 3] * d.v.m[c + 3] + g[2] * d.w.m[i + 6] * d.v.m[c + 6];
    return r;
}

// ------------------------------------------------------------------------------------------- models
// Flattened device copy of SparklParticleModel (built on the host by models_set).
struct DevModel {
    int ctag;         // SPARKL_CONSTITUTIVE_*
    int ptag;         // SPARKL_PLASTIC_* or -1
    int is_fluid;
    int split_stress_on_failure;
    float cfl_coeff, lambda, mu;            // elastic models
    float pressure0, viscosity, max_neg_pressure; int gamma;  // Monaghan
    float h0, h1, h2, h3, p_lambda, p_mu, volume_correction; int only_active_when_failed;  // Drucker-Prager
    float p_k;        // (3 p_lambda + 2 p_mu) / (2 p_mu)
    // the plastic models the 3-D examples do not use (snow / Rankine / NACC) share one parameter block:
    //   snow: x0 = min_epsilon, x1 = max_epsilon, x2 = hardening_coeff
    //   Rankine: x0 = mu, x1 = lambda, x2 = tensile_strength, x3 = softening_rate
    //   NACC: x0 = mu, x1 = kappa, x2 = hardening_factor, x3 = cohesion, x4 = friction, xi = hardening_enabled
    float x0, x1, x2, x3, x4; int xi;
    int has_failure;  // MaximumStressFailure
    float max_principal_stress, max_shear_stress;
};

// Compile-time view of a model table whose every entry is a corotated-linear solid with Drucker-Prager plasticity, always active
// (the sand of sand3 / cube_through_sand3 and of the 10M benchmark column): the tags are constants, so the kernels instantiated on it
// carry no model dispatch; the parameters are still read per particle from the table.  Chosen by the host (models_set).
struct DpSandModel {
    static constexpr int ctag = SPARKL_CONSTITUTIVE_COROTATED_LINEAR;
    static constexpr int ptag = SPARKL_PLASTIC_DRUCKER_PRAGER;
    static constexpr int is_fluid = 0;
    static constexpr int only_active_when_failed = 0;
    static constexpr int has_failure = 0;
    static constexpr float pressure0 = 0.f, viscosity = 0.f, max_neg_pressure = 0.f;
    static constexpr int gamma = 0;
    const int& split_stress_on_failure;
    const float &cfl_coeff, &lambda, &mu, &h0, &h1, &h2, &h3, &p_lambda, &p_mu, &volume_correction, &p_k;
    __device__ __host__ __forceinline__ explicit DpSandModel(const DevModel& m)
        : split_stress_on_failure(m.split_stress_on_failure), cfl_coeff(m.cfl_coeff), lambda(m.lambda), mu(m.mu), h0(m.h0), h1(m.h1), h2(m.h2), h3(m.h3),
          p_lambda(m.p_lambda), p_mu(m.p_mu), volume_correction(m.volume_correction), p_k(m.p_k) {}
};

// ... and of one whose every entry is a neo-Hookean solid without plasticity or failure model (BASELINE configs[1], the elastic cube).
struct NeoHookeanOnlyModel {
    static constexpr int ctag = SPARKL_CONSTITUTIVE_NEO_HOOKEAN;
    static constexpr int ptag = -1;
    static constexpr int is_fluid = 0;
    static constexpr int only_active_when_failed = 0;
    static constexpr int has_failure = 0;
    static constexpr int split_stress_on_failure = 0;
    static constexpr float pressure0 = 0.f, viscosity = 0.f, max_neg_pressure = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, p_lambda = 0.f, p_mu = 0.f, volume_correction = 1.f, p_k = 0.f;
    static constexpr int gamma = 0;
    const float &cfl_coeff, &lambda, &mu;
    __device__ __host__ __forceinline__ explicit NeoHookeanOnlyModel(const DevModel& m) : cfl_coeff(m.cfl_coeff), lambda(m.lambda), mu(m.mu) {}
};

// ... and of one whose every entry is a corotated-linear solid with snow plasticity and no failure model (the snow block of configs "snow1m"): the
// kernels instantiated on it carry the snow return map behind its elastic predictor and nothing else of the rare code (no Rankine / NACC, no failure
// test, no particle projection, no penalty kick: g2p_update<RARE = 2>).
struct SnowOnlyModel {
    static constexpr int ctag = SPARKL_CONSTITUTIVE_COROTATED_LINEAR;
    static constexpr int ptag = SPARKL_PLASTIC_SNOW;
    static constexpr int is_fluid = 0;
    static constexpr int only_active_when_failed = 0;
    static constexpr int has_failure = 0;
    static constexpr float pressure0 = 0.f, viscosity = 0.f, max_neg_pressure = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, p_lambda = 0.f, p_mu = 0.f, volume_correction = 1.f, p_k = 0.f;
    static constexpr float x3 = 0.f, x4 = 0.f;
    static constexpr int gamma = 0, xi = 0;
    const int& split_stress_on_failure;
    const float &cfl_coeff, &lambda, &mu, &x0, &x1, &x2;
    __device__ __host__ __forceinline__ explicit SnowOnlyModel(const DevModel& m)
        : split_stress_on_failure(m.split_stress_on_failure), cfl_coeff(m.cfl_coeff), lambda(m.lambda), mu(m.mu), x0(m.x0), x1(m.x1), x2(m.x2) {}
};

// Compile-time view of a model table whose every entry is a plain fluid (Monaghan equation of state, no plastic or failure model:
// fluids3.rs): the FLUID_ONLY kernels carry no solid code at all.  Chosen by the host (models_set: all_plain_fluid).
struct PlainFluidModel {
    static constexpr int ctag = SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH;
    static constexpr int ptag = -1;
    static constexpr int is_fluid = 1;
    static constexpr int only_active_when_failed = 0;
    static constexpr int has_failure = 0;
    static constexpr int split_stress_on_failure = 0;
    static constexpr float cfl_coeff = 0.f, lambda = 0.f, mu = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, p_lambda = 0.f, p_mu = 0.f, volume_correction = 1.f, p_k = 0.f;
    const float &pressure0, &viscosity, &max_neg_pressure;
    const int& gamma;
    __device__ __host__ __forceinline__ explicit PlainFluidModel(const DevModel& m)
        : pressure0(m.pressure0), viscosity(m.viscosity), max_neg_pressure(m.max_neg_pressure), gamma(m.gamma) {}
};

// x^n for the small non-negative exponents of the Monaghan equation of state (gamma = 7 in fluids3.rs): the reference's `powi` multiplies
// by squaring as well (core::intrinsics::powif32 -> llvm.powi); four multiplications for n = 7 instead of a seven-trip loop
SPK_DEV float powi_f(float x, int n) {   // n >= 0: the exponent is MonaghanSphEos::gamma, a u32 in the reference
    float r = 1.f, b = x;
    unsigned e = (unsigned)n;
#pragma unroll 1
    while (e) {
        if (e & 1u) r *= b;
        e >>= 1;
        if (e) b *= b;
    }
    return r;
}

// tau for the corotated model.  phase == 1 (every particle unless a damage model is active) only needs
// the rotation R: U (S - I) V^T = F - R.
template <class MO>
SPK_DEV M3 corotated_kirchhoff_e(const MO& mo, float hardening, const M3& F);
template <class MO>
SPK_DEV M3 corotated_kirchhoff(const MO& mo, float phase, float hardening, const M3& F) {
    const float j = m3_det(F);
    const float sph = mo.lambda * hardening * (j - 1.f) * j;
    M3 tau;
    if (phase == 1.f) {
        return corotated_kirchhoff_e(mo, hardening, F);
    } else {
        const Svd3 d = svd3<SPK_SVD_SWEEPS>(F);
        float gp[3], gn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = d.s[k] - 1.f, is = inv_exact(d.s[k]);
            gp[k] = fmaxf(e, 0.f) * is;
            gn[k] = fminf(e, 0.f) * is;
        }
        const float k2 = 2.f * mo.mu * hardening;
        M3 pos = m3_mul_bt(svd_compose_scaled(d, gp), F);
        M3 neg = m3_mul_bt(svd_compose_scaled(d, gn), F);
#pragma unroll
        for (int i = 0; i < 9; ++i) { pos.m[i] *= k2; neg.m[i] *= k2; }
        if (j < 1.f) { neg.m[0] += sph; neg.m[4] += sph; neg.m[8] += sph; }
        else { pos.m[0] += sph; pos.m[4] += sph; pos.m[8] += sph; }
        const float pc = (mo.split_stress_on_failure && phase == 0.f) ? 0.f : 1.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) tau.m[i] = pos.m[i] * pc + neg.m[i];
    }
    return tau;
}

SPK_DEV float nh_phase_coeff(float phase) { return (1.f - 0.001f) * phase * phase + 0.001f; }

template <class MO>
SPK_DEV M3 neo_hookean_kirchhoff(const MO& mo, float phase, float hardening, const M3& F) {
    const float pc = nh_phase_coeff(phase);
    const float j = m3_det(F);
    const float k = 2.f / 3.f * mo.mu * hardening + mo.lambda * hardening;
    M3 cg = m3_mul_bt(F, F);
    const float sphp = (cg.m[0] + cg.m[4] + cg.m[8]) * (1.f / 3.f);
    cg.m[0] -= sphp; cg.m[4] -= sphp; cg.m[8] -= sphp;
    // J^(-2/3) = exp2(-2/3 log2 J) on the hardware exp / log units (J > 0 for any particle that has not failed; 1e-7 relative)
    const float kd = (mo.mu * hardening) * (j > 0.f ? __builtin_amdgcn_exp2f((-2.f / 3.f) * __builtin_amdgcn_logf(j)) : powf(j, -2.f / 3.f));
    const float vol = k / 2.f * (j * j - 1.f);
    M3 tau;
    if (j >= 1.f) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tau.m[i] = (kd * cg.m[i]) * pc;
        tau.m[0] = (kd * cg.m[0] + vol) * pc; tau.m[4] = (kd * cg.m[4] + vol) * pc; tau.m[8] = (kd * cg.m[8] + vol) * pc;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) tau.m[i] = (kd * cg.m[i]) * pc;
        tau.m[0] += vol; tau.m[4] += vol; tau.m[8] += vol;
    }
    return tau;
}

template <class MO>
SPK_DEV float monaghan_pressure(const MO& mo, float mass, float volume0, float density_fluid) {
    const float density0 = mass / volume0;
    return fmaxf(mo.pressure0 * (powi_f(density_fluid / density0, mo.gamma) - 1.f), -mo.max_neg_pressure);
}
// the same with the density ratio rho / rho0 = 1 / F00 given directly (the reference forms (m / V0) / F00 / (m / V0): the same
// number to 1e-7, three divisions)
template <class MO>
SPK_DEV float monaghan_pressure_ratio(const MO& mo, float ratio) {
    return fmaxf(mo.pressure0 * (powi_f(ratio, mo.gamma) - 1.f), -mo.max_neg_pressure);
}
// NB: the reference glue passes plastic_deformation_gradient_det as `fluid_j`
// (src/dynamics/models/eos_monaghan_sph.rs:12-20)
template <class MO>
SPK_DEV M3 monaghan_kirchhoff(const MO& mo, float mass, float volume0, float f00, float fluid_j, const M3& C) {
    const float p = -monaghan_pressure_ratio(mo, 1.0f / f00) * fluid_j;
    M3 tau = m3_zero();
    tau.m[0] = tau.m[4] = tau.m[8] = p;
    if (mo.viscosity != 0.f) {
        M3 sr;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) sr.m[i + 3 * j] = (C.m[i + 3 * j] + C.m[j + 3 * i]) * 0.5f;
        const float sph = (sr.m[0] + sr.m[4] + sr.m[8]) * (1.f / 3.f);
        sr.m[0] -= sph; sr.m[4] -= sph; sr.m[8] -= sph;
        const float k = 2.f * mo.viscosity * fluid_j;
#pragma unroll
        for (int i = 0; i < 9; ++i) tau.m[i] += k * sr.m[i];
    }
    return tau;
}

SPK_DEV float sound_speed_bound(float alpha, float bulk, float shear, float density0, float vnorm, float h) {
    const float c_dir = sqrtf((bulk + 4.f / 3.f * shear) * 1.f / density0);
    return alpha * h / fmaxf(vnorm, c_dir);
}

// model.constitutive_model.timestep_bound(particle, cell_width)
template <class MO>
SPK_DEV float model_timestep_bound(const MO& mo, float mass, float volume0, float eh, float f00, float vx, float vy, float vz, float h) {
    const float vn2 = vx * vx + vy * vy + vz * vz;
    if (mo.ctag == SPARKL_CONSTITUTIVE_EOS_MONAGHAN_SPH) {
        const float j = f00, density0 = mass / volume0;
        const float density_fluid = density0 / f00;
        const float pressure = -monaghan_pressure(mo, mass, volume0, density_fluid);
        const float single = (h / j) * sqrtf(density0 * (j - 1.f) / (6.f * pressure * 3.f));
        const float cfl = h / sqrtf(fmaxf(vn2, 1.f) / 0.1f);
        return fminf(single, cfl);
    }
    const float bulk = mo.lambda + 2.f * mo.mu / 3.f;
    return sound_speed_bound(mo.cfl_coeff, bulk * eh, mo.mu * eh, mass / volume0, sqrtf(vn2), h);
}

// ln(s) for s near 1 through 2 atanh((s-1)/(s+1)): relative error ~3e-8 for |s-1| < 0.3
SPK_DEV float log_near_one(float sgm) {
    const float e = sgm - 1.f;
    if (fabsf(e) < 0.3f) {
        const float z = e * fast_rcp(sgm + 1.f), z2 = z * z;
        return 2.f * z * (1.f + z2 * (1.f / 3.f + z2 * (1.f / 5.f + z2 * (1.f / 7.f + z2 * (1.f / 9.f)))));
    }
    return logf(sgm);
}

// sin(x) by its Taylor series to x^13 on |x| < 1.6 (friction angles), libm otherwise
SPK_DEV float sin_small(float x) {
    SPK_HOT(fabsf(x) < 1.6f);
    if (fabsf(x) < 1.6f) {
        const float x2 = x * x;
        return x * (1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f + x2 * (-1.f / 5040.f + x2 * (1.f / 362880.f + x2 * (-1.f / 39916800.f + x2 * (1.f / 6227020800.f)))))));
    }
    return sinf(x);
}

template <class MO>
SPK_DEV float dp_alpha(const MO& mo, float q) {
    const float angle = mo.h0 + (mo.h1 * q - mo.h3) * __expf(-mo.h2 * q);
    const float s = sin_small(angle);
    return 0.81649658092772603f * (2.f * s) * fast_rcp(3.f - s);   // sqrt(2/3)
}

// e^x - 1 without the cancellation of expf(x) - 1 (plastic strains are ~1e-4: exp rounds 1 + x to ~1e-7 absolute, i.e. 1e-3
// of the strain)
// The polynomial alone, for callers that bound |x| themselves (the series return map: |x| < 0.2, truncation x^6 / 5040 < 1e-8 relative).
// expm1_small's libm side is only three instructions, which the compiler lays out exec-masked WITHOUT a skip: every call then issues
// v_mul + v_exp (quarter rate) + v_add with no lane enabled -- six calls per particle in the return map.
SPK_DEV float expm1_poly(float x) { return x * (1.f + x * (0.5f + x * (1.f / 6.f + x * (1.f / 24.f + x * (1.f / 120.f + x * (1.f / 720.f)))))); }
SPK_DEV float expm1_small(float x) {
    SPK_HOT(fabsf(x) < 0.1f);
    if (fabsf(x) < 0.1f) return x * (1.f + x * (0.5f + x * (1.f / 6.f + x * (1.f / 24.f + x * (1.f / 120.f + x * (1.f / 720.f))))));
    return __expf(x) - 1.f;
}

// ------------------------------------------------------------------------------ strain-form arithmetic (E = F - I)
// Why: the stiff examples sit at elastic strains of ~1e-4 (E = 1e7 sand under its own weight).  A singular value of F
// computed in f32 carries ~1e-7 of absolute error whatever the algorithm (the one-sided Jacobi above: <= 5e-7), i.e.
// 1e-3..5e-3 of sigma - 1, and the stress is proportional to sigma - 1: that, not the f32 STORAGE of F, is what separates an
// f32 MPM step from the f64 one (measured on the CPU restatement: f64 arithmetic with F rounded to f32 after every substep
// stays within 5e-5 of pure f64 on the sand case; f32 arithmetic is at 4.6e-4).  Everything below works on the small
// quantities themselves -- E = F - I (exact in f32 for entries of F near 1), S = F^T F - I = E + E^T + E^T E, its
// eigen-decomposition S = V diag(lam) V^T (sigma^2 - 1 = lam), ln sigma = log1p(lam) / 2 -- so that the strain keeps ~1e-7
// RELATIVE precision.  U is never formed:  F' = U Sigma' V^T = F V diag(sigma'/sigma) V^T,  F - R = F V diag(1 - 1/sigma) V^T.
SPK_DEV float log1p_small(float x) {   // ln(1 + x) = 2 atanh(x / (2 + x)); relative error ~1e-7 for |x| < 0.35
    SPK_HOT(fabsf(x) < 0.35f);
    if (fabsf(x) < 0.35f) {
        const float z = x * fast_rcp(2.f + x), z2 = z * z;
        return 2.f * z * (1.f + z2 * (1.f / 3.f + z2 * (1.f / 5.f + z2 * (1.f / 7.f + z2 * (1.f / 9.f)))));
    }
    return log1pf(x);
}
// det(I + E) - 1 = tr E + (sum of the principal 2x2 minors) + det E
SPK_DEV float det_minus_one(const M3& E) {
    const float tr = E.m[0] + E.m[4] + E.m[8];
    const float c2 = (E.m[0] * E.m[4] - E.m[3] * E.m[1]) + (E.m[0] * E.m[8] - E.m[6] * E.m[2]) + (E.m[4] * E.m[8] - E.m[7] * E.m[5]);
    return tr + c2 + m3_det(E);
}
// S = E + E^T + E^T E as (xx, yy, zz, xy, xz, yz)
SPK_DEV void strain_metric(const M3& E, float S[6]) {
    const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int a = ii[e], b = jj[e];
        S[e] = (E.m[a + 3 * b] + E.m[b + 3 * a]) + (E.m[3 * a] * E.m[3 * b] + E.m[3 * a + 1] * E.m[3 * b + 1] + E.m[3 * a + 2] * E.m[3 * b + 2]);
    }
}
#ifndef SPK_EIG_TOL_ABS
#define SPK_EIG_TOL_ABS 1.0e-6f
#endif
#ifndef SPK_EIG_FLOOR
#define SPK_EIG_FLOOR 1.0e-10f   // off-diagonal strains below this are noise: (f32 rounding of F)^2 is 1e-14
#endif
struct Eig3 {
    M3 v;          // eigenvectors (columns)
    float lam[3];
};
// One Jacobi rotation of the symmetric 3x3 (pp, qq, pq; rp, rq couple the third index) and of the columns vp, vq of V.
SPK_DEV void jacobi_sym(float& pp, float& qq, float& pq, float& rp, float& rq, float* vp, float* vq, float skip) {
    const bool rot = fabsf(pq) > skip;
    if (!__any(rot)) return;              // wave-uniform: the converged tail costs three compares per sweep
    // branch-free per lane (a lane that does not rotate gets t = 0: c = 1, s = 0): no exec-masked copies of the twelve outputs
    // t = tan(theta) = sgn(tau) 2 pq / (|tau| + sqrt(tau^2 + 4 pq^2)),  tau = qq - pp: one v_sqrt and one v_rcp (the textbook
    // zeta = tau / (2 pq) form takes two reciprocals)
    const float tau = qq - pp, p2 = pq + pq;
    float t = copysignf(p2, tau * pq) * fast_rcp(fabsf(tau) + fast_sqrt(tau * tau + p2 * p2));
    t = rot ? t : 0.f;
    const float c = fast_rsq(1.f + t * t), sn = c * t;
    pp -= t * pq; qq += t * pq;
    pq = rot ? 0.f : pq;
    const float nrp = c * rp - sn * rq;
    rq = sn * rp + c * rq;
    rp = nrp;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a = vp[i], b = vq[i];
        vp[i] = c * a - sn * b;
        vq[i] = sn * a + c * b;
    }
}
// Cyclic Jacobi on S.  Stops after the sweep that leaves every off-diagonal entry of every lane below the tolerance
// (wave-uniform branch; the test comes AFTER the sweep, so a converged matrix is not swept once more).
// What the tolerance has to be: the callers take the part of every matrix function that is LINEAR in S from S itself and
// only the remainder (second order in the strain) from V diag(.) V^T, so an off-diagonal residue r -- a misalignment of V by
// r / gap -- costs r / gap x (second-order term) ~ r in RELATIVE terms whatever the strain.  r <= 1e-6 absolute is therefore
// enough (1e-3 of a 1e-3 strain: two sweeps; 1e-5 of a 0.1 strain: three or four), never looser than 1e-2 of the matrix.
SPK_DEV Eig3 sym_eig3(const float S[6]) {
    Eig3 r;
    r.v = m3_identity();
    float s00 = S[0], s11 = S[1], s22 = S[2], s01 = S[3], s02 = S[4], s12 = S[5];
    const float scale = fmaxf(fmaxf(fmaxf(fabsf(s00), fabsf(s11)), fmaxf(fabsf(s22), fabsf(s01))), fmaxf(fabsf(s02), fabsf(s12)));
    // no squares of the thresholds (they underflow for the strains of a particle that barely left F = I).  The floor: a
    // particle in free fall carries S = (rounding of F)^2 ~ 1e-12 of arbitrary structure -- three sweeps for nothing
    const float tol = fmaxf(fminf(1.0e-2f * scale, fmaxf(SPK_EIG_TOL_ABS, 1.0e-5f * scale)), SPK_EIG_FLOOR), skip = 1.0e-3f * tol;
    bool more = fmaxf(fabsf(s01), fmaxf(fabsf(s02), fabsf(s12))) > tol;
#pragma unroll 1   // rolled: eight unrolled sweeps of three instances were 25 KB of a kernel that is 60 KB long
    for (int sweep = 0; sweep < SPK_SVD_SWEEPS && __any(more); ++sweep) {
        jacobi_sym(s00, s11, s01, s02, s12, r.v.m + 0, r.v.m + 3, skip);
        jacobi_sym(s00, s22, s02, s01, s12, r.v.m + 0, r.v.m + 6, skip);
        jacobi_sym(s11, s22, s12, s01, s02, r.v.m + 3, r.v.m + 6, skip);
        more = fmaxf(fabsf(s01), fabsf(s02)) > tol;    // s12 was just annihilated
    }
    r.lam[0] = s00; r.lam[1] = s11; r.lam[2] = s22;
    return r;
}
// V diag(g) V^T as (xx, yy, zz, xy, xz, yz)
SPK_DEV void sym_compose(const M3& v, const float g[3], float out[6]) {
    const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) out[e] = g[0] * v.m[ii[e]] * v.m[jj[e]] + g[1] * v.m[3 + ii[e]] * v.m[3 + jj[e]] + g[2] * v.m[6 + ii[e]] * v.m[6 + jj[e]];
}
SPK_DEV M3 sym6_to_m3(const float t[6]) { return M3{{t[0], t[3], t[4], t[3], t[1], t[5], t[4], t[5], t[2]}}; }
// x -> e^x - 1 - x and x -> ln(1 + x) - x: the second-order remainders, without the cancellation of the differences
SPK_DEV float expm1_rem(float x) {
    if (fabsf(x) < 0.1f) return x * x * (0.5f + x * (1.f / 6.f + x * (1.f / 24.f + x * (1.f / 120.f + x * (1.f / 720.f)))));
    return __expf(x) - 1.f - x;
}
SPK_DEV float log1p_rem(float x) {
    if (fabsf(x) < 0.1f) return x * x * (-0.5f + x * (1.f / 3.f + x * (-0.25f + x * (0.2f + x * (-1.f / 6.f + x * (1.f / 7.f + x * -0.125f))))));
    return log1pf(x) - x;
}
// Hencky strain of F = I + E,  eps = ln(F^T F) / 2,  in two parts: the matrix S / 2 (first order, exact) and, in the eigenbasis V of
// S, the remainders re_k = (log1p(lam_k) - lam_k) / 2; e_k = ln sigma_k = lam_k / 2 + re_k.  eps = S / 2 + V diag(re) V^T is never
// composed on its own: every consumer adds its own first-order matrix and its own remainders and composes once (see sym_eig3 for
// why the split matters).
SPK_DEV Eig3 log_stretch_s(const float S[6], float e[3], float re[3]) {
    const Eig3 d = sym_eig3(S);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lam = fmaxf(d.lam[k], -0.99999988f);
        re[k] = 0.5f * log1p_rem(lam);
        e[k] = 0.5f * lam + re[k];
    }
    return d;
}
SPK_DEV Eig3 log_stretch(const M3& E, float S[6], float e[3], float re[3]) {
    strain_metric(E, S);
    return log_stretch_s(S, e, re);
}

// ------------------------------------------------------------------------------ series path (no eigenvectors)
// While the Frobenius norm of S = F^T F - I (a bound on its eigenvalues sigma^2 - 1) and the pending volume gain stay below
// SPK_SERIES_MAX -- stiff materials for most of their life: the E = 1e7 sand column of the benchmark passes it at its foot
// ~0.1 s after release -- every matrix function the return map and the corotated stress need is a short polynomial in a SYMMETRIC matrix that
// commutes with S:   ln(I + S) / 2 = S/2 - S^2/4 + S^3/6 - S^4/8,   exp(D) - I = D + D^2/2 + D^3/6,   I - exp(-e) = e - e^2/2 + e^3/6.
// The part linear in the strain is the strain matrix itself, exactly as in the eigenbasis form above, so the relative precision of
// the strain is kept (truncation: |S|^4 / 5 and |e|^3 / 24 relative, < 2e-7 at the switch-over); two Jacobi sweeps with their
// eigenvector accumulation (2 x 128 instructions, 18 transcendentals) and two eigenbasis compositions become seven symmetric
// products of 18 FMAs.  Only the positive-energy bookkeeping needs eigenVALUES (sum of max(sigma_k - 1, 0)^2): closed form below.
// Second tier, SPK_SERIES_MAX <= |S| < SPK_SERIES_MAX2 (the flowing sand: a third of the waves of the 10M column 2 000 substeps in):
// the logarithm to S^7 -- one more product, Paterson-Stockmeyer: (S/2 - S^2/4 + S^3/6) + S^4 (-1/8 + S/10 - S^2/12 + S^3/14) -- next
// term |S|^7 / 8 relative, 1.3e-8 at 0.1; the exponentials are quartic on both tiers (x^2 (x/6 + x^2/24): the same two products as the
// cubic), next term |x|^4 / 120 relative with |x| <= |S| / 2: 5e-8.
#ifndef SPK_SERIES_MAX
#define SPK_SERIES_MAX 0.03f
#endif
#ifndef SPK_SERIES_MAX2
#define SPK_SERIES_MAX2 0.1f
#endif
// a * b for two commuting symmetric matrices (the product is symmetric), (xx, yy, zz, xy, xz, yz)
template <class T>
SPK_DEV void sym_mul(const T a[6], const T b[6], T o[6]) {
    o[0] = a[0] * b[0] + a[3] * b[3] + a[4] * b[4];
    o[1] = a[3] * b[3] + a[1] * b[1] + a[5] * b[5];
    o[2] = a[4] * b[4] + a[5] * b[5] + a[2] * b[2];
    o[3] = a[0] * b[3] + a[3] * b[1] + a[4] * b[5];
    o[4] = a[0] * b[4] + a[3] * b[5] + a[4] * b[2];
    o[5] = a[3] * b[4] + a[1] * b[5] + a[5] * b[2];
}
// Eigenvalues of a symmetric 3x3 in closed form (trigonometric solution of the characteristic cubic of the deviator; the
// trisection cos(acos(r) / 3) as two Newton steps on 4 c^3 - 3 c = |r| from the chord, c in [cos(pi/6), 1] where the root is
// simple; r < 0 is the mirrored problem).  Errors: ~1e-7 |A| on the isolated eigenvalue; a near-degenerate pair is split
// symmetrically by up to ~5e-4 |dev A| -- second order in every symmetric function of the pair, which is all the callers form.
SPK_DEV void sym_eigvals(const float a[6], float e[3]) {
    const float q = (a[0] + a[1] + a[2]) * (1.f / 3.f);
    const float d0 = a[0] - q, d1 = a[1] - q, d2 = a[2] - q;
    const float p2 = (d0 * d0 + d1 * d1 + d2 * d2 + 2.f * (a[3] * a[3] + a[4] * a[4] + a[5] * a[5])) * (1.f / 6.f);
    if (!(p2 > 1.0e-36f)) { e[0] = e[1] = e[2] = q; return; }
    const float ip = fast_rsq(p2), p = p2 * ip;
    const float det = d0 * (d1 * d2 - a[5] * a[5]) - a[3] * (a[3] * d2 - a[5] * a[4]) + a[4] * (a[3] * a[5] - d1 * a[4]);
    const float r = fminf(fmaxf(0.5f * det * (ip * ip * ip), -1.f), 1.f);
    const float ar = fabsf(r);
    float c = 0.8660254f + 0.1339746f * ar;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const float c2 = c * c;
        c -= ((4.f * c2 - 3.f) * c - ar) * fast_rcp(12.f * c2 - 3.f);
    }
    const float sn = fast_sqrt(fmaxf(1.f - c * c, 0.f)) * 0.8660254f;   // (sqrt(3) / 2) sin(phi)
    const float sg = r < 0.f ? -p : p;
    e[0] = q + sg * (2.f * c);
    e[1] = q + sg * (-c + 2.f * sn);
    e[2] = q + sg * (-c - 2.f * sn);
}
// eps = ln(I + S) / 2 by its series: quartic (the next term is |S|^4 / 5 of the first), or to S^7 (`hi`, wave-uniform)
template <class T>
SPK_DEV void hencky_series(const T S[6], T eps[6], bool hi = false) {
    // Paterson-Stockmeyer on P2 = S^2 with coefficients linear in S (every factor is a polynomial in S: they commute):
    //   quartic:  (S/2 - P2/4) + P2 (S/6 - P2/8)                                            two products
    //   to S^7:   (S/2 - P2/4) + P2 ((S/6 - P2/8) + P2 ((S/10 - P2/12) + P2 (S/14)))        four; at most S, P2 and one running factor live
    T P2[6], V[6], U[6];
    sym_mul(S, S, P2);
    if (hi) {
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] = (1.f / 14.f) * S[i];
        sym_mul(P2, V, U);
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] = U[i] + (0.1f * S[i] - (1.f / 12.f) * P2[i]);
        sym_mul(P2, V, U);
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] = U[i] + ((1.f / 6.f) * S[i] - 0.125f * P2[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) V[i] = (1.f / 6.f) * S[i] - 0.125f * P2[i];
    }
    sym_mul(P2, V, U);
#pragma unroll
    for (int i = 0; i < 6; ++i) eps[i] = U[i] + (0.5f * S[i] - 0.25f * P2[i]);
}
// x - x^2/2 + x^3/6 - x^4/24 = I - exp(-x) to |x|^4 / 120 relative
template <class T>
SPK_DEV void one_minus_exp_neg_series(const T x[6], T m[6]) {
    T X2[6], Q[6], U[6];
    sym_mul(x, x, X2);
#pragma unroll
    for (int i = 0; i < 6; ++i) Q[i] = (1.f / 6.f) * x[i] - (1.f / 24.f) * X2[i];
    sym_mul(X2, Q, U);
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = x[i] + (U[i] - 0.5f * X2[i]);
}
// squared Frobenius norm: bounds every eigenvalue of the symmetric matrix
template <class T>
SPK_DEV T sym_frob2(const T S[6]) {
    return (S[0] * S[0] + S[1] * S[1] + S[2] * S[2]) + 2.f * (S[3] * S[3] + S[4] * S[4] + S[5] * S[5]);
}
SPK_DEV bool series_ok(const float S[6], float extra = 0.f) { return fmaxf(sym_frob2(S), extra * extra) < SPK_SERIES_MAX * SPK_SERIES_MAX; }
SPK_DEV bool series2_ok(const float S[6], float extra = 0.f) { return fmaxf(sym_frob2(S), extra * extra) < SPK_SERIES_MAX2 * SPK_SERIES_MAX2; }

// Corotated Kirchhoff stress (phase == 1) of the particle's final F = U diag(sigma') V^T, formed by the return map itself from
// the decomposition it holds (the fused kernel scatters it one substep later; see corotated_tau_e).  Symmetric:
// (xx, yy, zz, xy, xz, yz).
struct SymStress {
    float t[6];
    float jm1;     // det(F) - 1 of the same F (the callers' guards and the energy bookkeeping need it too)
    bool valid;
};
SPK_DEV M3 sym_to_m3(const SymStress& s) { return M3{{s.t[0], s.t[3], s.t[4], s.t[3], s.t[1], s.t[5], s.t[4], s.t[5], s.t[2]}}; }

// DruckerPragerPlasticity::update_particle.  Updates F, Jp, q, lvg in place and returns the singular
// values of the (possibly projected) F through sv[] for the pos-energy bookkeeping.
// `T` (want_stress): the corotated stress of the final F for the fused kernel (eh = elastic hardening), left invalid when the model / phase do not qualify.
// Corotated Kirchhoff stress (phase == 1) of F = I + E:
//   2 mu (F - R) F^T + lambda (J - 1) J I  =  2 mu F M F^T + lambda (J - 1) J I,   M = V diag((sigma - 1)/sigma) V^T = I - exp(-eps)
// (elasticity_corotated_linear.rs:31-78; R = U V^T, U = F V Sigma^-1).  The caller hands M over as (xx, yy, zz, xy, xz, yz):
// first order of eps as a matrix, remainder -expm1_rem(-e_k) in the eigenbasis.  J - 1 = det(I + E) - 1 without cancellation.
// `park` (fused kernel): the six components go straight to the LDS, element k of this lane at park[64 k], as each is formed -- this is
// where the kernel's register pressure peaks (E, M, A = F M and everything the particle update still holds), and the components are not
// needed again before the scatter ~2 000 instructions later: kept in registers they were spilled to scratch (kernels.h, lds_fp).
#ifdef SPK_HOST_BUILD_NO_AS
typedef volatile float* lds_fp;
#else
typedef volatile __attribute__((address_space(3))) float* lds_fp;
#endif
// The slot as the math functions take it: whether there is one is a fact of the CALL SITE (`on`, a constant after inlining), never a test of the
// address -- LDS offset 0 is a valid address and compares equal to a null pointer of that address space: with the array the slots live in placed
// first in the workgroup's LDS, lane 0 of wave 0 took "no slot", kept its stress in registers that the caller then ignored, and scattered what the
// column held before (found with the 512-thread k_g2p2g, whose layout put the array there).
struct ParkSlot {
    lds_fp p;
    bool on;
    SPK_DEV ParkSlot() : p(nullptr), on(false) {}
    SPK_DEV explicit ParkSlot(lds_fp q) : p(q), on(true) {}
};
// `jm1_known`: det(F) - 1 when the caller already holds it (the series return map: expm1 of the trace of the mapped Hencky strain).
template <class MO>
SPK_DEV SymStress corotated_tau_e(const MO& mo, float hardening, const M3& E, const float m6[6], ParkSlot park = ParkSlot(), const float* jm1_known = nullptr) {
    const M3 M = sym6_to_m3(m6);
    M3 A;                                // A = F M = M + E M, each entry one FMA chain started from M
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) A.m[i + 3 * j] = M.m[i + 3 * j] + (E.m[i] * M.m[3 * j] + (E.m[i + 3] * M.m[3 * j + 1] + E.m[i + 6] * M.m[3 * j + 2]));
    const float jm1 = jm1_known != nullptr ? *jm1_known : det_minus_one(E);
    const float sph = mo.lambda * hardening * jm1 * (1.f + jm1);
    const float k2 = 2.f * mo.mu * hardening;
    SymStress S;
    const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int a = ii[k], b = jj[k];
        // (A F^T)_ab = A_ab + sum_c A_ac E_bc
        const float t = k2 * (A.m[a + 3 * b] + (A.m[a] * E.m[b] + A.m[a + 3] * E.m[b + 3] + A.m[a + 6] * E.m[b + 6])) + (k < 3 ? sph : 0.f);
        if (park.on) { park.p[64 * k] = t; S.t[k] = 0.f; }
        else S.t[k] = t;
    }
    S.jm1 = jm1;
    S.valid = true;
    return S;
}

// S_l = F F^T - I = E + E^T + E E^T as (xx, yy, zz, xy, xz, yz): the LEFT metric (same eigenvalues, hence the same Frobenius norm, as strain_metric's)
SPK_DEV void strain_metric_left(const M3& E, float S[6]) {
    const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int a = ii[e], b = jj[e];
        S[e] = (E.m[a + 3 * b] + E.m[b + 3 * a]) + (E.m[a] * E.m[b] + E.m[a + 3] * E.m[b + 3] + E.m[a + 6] * E.m[b + 6]);
    }
}
// The same stress through the LEFT stretch, for the series tiers.  With F = V R (V = sqrt(F F^T) symmetric):  (F - R) F^T = V^2 - V, so
//   tau = 2 mu (I + S_l - sqrt(I + S_l)) + lambda (J - 1) J I,
//   I + x - sqrt(I + x) = x/2 + x^2/8 - x^3/16 + 5 x^4/128 - 7 x^5/256 + 21 x^6/1024 - 33 x^7/2048 + ...
// a polynomial in the symmetric S_l alone (Paterson-Stockmeyer on P2 = x^2 like hencky_series: two products to x^4 -- next term 0.055 |x|^4
// relative, 4e-8 at SPK_SERIES_MAX --, four to x^7 -- 0.026 |x|^7, 3e-9 at SPK_SERIES_MAX2): no M = I - exp(-eps), no A = F M, no A F^T
// (two 27-FMA products and one series product less than corotated_tau_e, and E is the only matrix live on entry).  The linear part is S_l / 2
// itself, so the relative precision of the strain is kept as in the right-stretch form.
template <class MO>
SPK_DEV SymStress corotated_tau_left(const MO& mo, float hardening, const M3& E, bool hi, ParkSlot park = ParkSlot(), const float* jm1_known = nullptr) {
    float x[6], P2[6], V[6], U[6];
    strain_metric_left(E, x);
    sym_mul(x, x, P2);
    if (hi) {
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = (-33.f / 2048.f) * x[k];
        sym_mul(P2, V, U);
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = U[k] + ((-7.f / 256.f) * x[k] + (21.f / 1024.f) * P2[k]);
        sym_mul(P2, V, U);
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = U[k] + ((-1.f / 16.f) * x[k] + (5.f / 128.f) * P2[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = (-1.f / 16.f) * x[k] + (5.f / 128.f) * P2[k];
    }
    sym_mul(P2, V, U);
    const float jm1 = jm1_known != nullptr ? *jm1_known : det_minus_one(E);
    const float sph = mo.lambda * hardening * jm1 * (1.f + jm1);
    const float k2 = 2.f * mo.mu * hardening;
    SymStress S;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float t = k2 * (U[k] + (0.5f * x[k] + 0.125f * P2[k])) + (k < 3 ? sph : 0.f);
        if (park.on) { park.p[64 * k] = t; S.t[k] = 0.f; }
        else S.t[k] = t;
    }
    S.jm1 = jm1;
    S.valid = true;
    return S;
}

// phase == 1 corotated stress of F = I + E
template <class MO>
SPK_DEV M3 corotated_kirchhoff_fe(const MO& mo, float hardening, const M3& E) {
    float S[6], e[3], re[3], g[3], m6[6];
#ifndef SPK_STRESS_RIGHT
    {   // the tier test on the left metric (same norm), which corotated_tau_left forms anyway
        float Sl[6];
        strain_metric_left(E, Sl);
        if (SPK_SERIES_MAX > 0.f && __all(series2_ok(Sl))) return sym_to_m3(corotated_tau_left(mo, hardening, E, !__all(series_ok(Sl))));   // wave-uniform, and so is the tier
    }
#endif
    strain_metric(E, S);
    if (SPK_SERIES_MAX > 0.f && __all(series2_ok(S))) {   // wave-uniform; series path: M = I - (I + S)^(-1/2)
        // = S/2 - 3/8 S^2 + 5/16 S^3 - 35/128 S^4 + 63/256 S^5 - 231/1024 S^6 + 429/2048 S^7 (next term 0.39 |S|^7 relative), evaluated like
        // hencky_series: P2 = S^2, coefficients linear in S, Horner in P2 -- four products, S, P2 and one running factor live
        float P2[6], V[6], U[6];
        sym_mul(S, S, P2);
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = 0.20947265625f * S[k];
        sym_mul(P2, V, U);
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = U[k] + (0.24609375f * S[k] - 0.2255859375f * P2[k]);
        sym_mul(P2, V, U);
#pragma unroll
        for (int k = 0; k < 6; ++k) V[k] = U[k] + (0.3125f * S[k] - 0.2734375f * P2[k]);
        sym_mul(P2, V, U);
#pragma unroll
        for (int k = 0; k < 6; ++k) m6[k] = U[k] + (0.5f * S[k] - 0.375f * P2[k]);
        return sym_to_m3(corotated_tau_e(mo, hardening, E, m6));
    }
    const Eig3 d = log_stretch_s(S, e, re);
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = re[k] - expm1_rem(-e[k]);
    sym_compose(d.v, g, m6);
#pragma unroll
    for (int k = 0; k < 6; ++k) m6[k] += 0.5f * S[k];
    return sym_to_m3(corotated_tau_e(mo, hardening, E, m6));
}
template <class MO>
SPK_DEV M3 corotated_kirchhoff_e(const MO& mo, float hardening, const M3& F) {
    M3 E = F;
    E.m[0] -= 1.f; E.m[4] -= 1.f; E.m[8] -= 1.f;
    return corotated_kirchhoff_fe(mo, hardening, E);
}

// DruckerPragerPlasticity::update_particle (plasticity_drucker_prager.rs:68-104) on E = F_trial - I, in place.
// `sm1` returns sigma' - 1 of the mapped F (for the positive-energy bookkeeping), T the corotated stress of the mapped F.
// The same return map on the series path (see "series path" above): `S` = F_trial^T F_trial - I with every entry, and the pending
// log_vol_gain, below SPK_SERIES_MAX.  K = ln(Sigma' / Sigma) = c I + D with D = g dev(eps) traceless, so
// exp(K) - I = expm1(c) I + e^c (D + D^2/2 + D^3/6).
template <bool WANT_STRESS, class MO>
SPK_DEV void dp_update_series(const MO& mo, float phase, float eh, const float S[6], M3& E, float& Jp, float& q, float& lvg, float sm1[3], bool& sv_valid,
                              SymStress& T, ParkSlot park = ParkSlot(), bool hi = false) {
    // (arguments of expm1_poly below: |S| < 0.1 => |eps| < 0.053, |tr eps| < 0.092; on the cone  -0.1 < lvg <= -tr eps  (dp_update_e's tier test), so
    // |tr eps + lvg| < 0.2 and |c| = |lvg| / 3 < 0.034; tensile: c = -tr eps / 3; eigenvalues of eps' < 0.12)
    float eps[6];
    SPK_MARK("dp_series_log");
    hencky_series(S, eps, hi);
    SPK_MARK("dp_alpha");
    const float alpha = dp_alpha(mo, q);
    SPK_MARK("dp_cases");
    const float tre = eps[0] + eps[1] + eps[2], em = tre * (1.f / 3.f);
    const float tr = tre + lvg;
    const float dev[6] = {eps[0] - em, eps[1] - em, eps[2] - em, eps[3], eps[4], eps[5]};
    const float dn2 = (dev[0] * dev[0] + dev[1] * dev[1] + dev[2] * dev[2]) + 2.f * (dev[3] * dev[3] + dev[4] * dev[4] + dev[5] * dev[5]);
    float c = 0.f, g = 0.f, dq = 0.f;
    bool moved = false;
    // det(F) - 1 = expm1(tr eps) comes with the volume bookkeeping below (truncation of the series: a few 1e-8 of the strain); the stress
    // and the caller's guards take it instead of the minors of E'
    float jm1 = expm1_poly(tre);
    if (dn2 == 0.f || tr > 0.f) {            // tensile: sigma' = 1, K = -eps
        dq = fast_sqrt(dn2 + tr * tr * (1.f / 3.f));
        c = -em; g = -1.f;
        moved = true;
    } else {
        const float dn = fast_sqrt(dn2);
        const float gamma = dn + mo.p_k * tr * alpha;
        if (gamma > 0.f) {                   // onto the cone: K = (lvg / 3) I - gdn dev(eps)
            c = lvg * (1.f / 3.f); g = -gamma * fast_rcp(dn);
            dq = gamma;
            moved = true;
        }
    }
    if (moved) {
        const float so = tre, sn = tre + 3.f * c;
        const float pm1 = jm1;
        float nm1 = expm1_poly(sn), lnew = sn;
        jm1 = nm1;                       // det F' = prod sigma' = exp(tr eps + 3 c), before the volume correction touches the bookkeeping
        const float diff = nm1 - pm1;
        SPK_HOT(mo.volume_correction == 1.f);
        if (!(diff > 0.f) && mo.volume_correction != 1.f) { nm1 = pm1 + diff * mo.volume_correction; lnew = log1p_small(nm1); }
        Jp *= (1.f + pm1) * fast_rcp(1.f + nm1);
        lvg += so - lnew;
        q += dq;
        float D[6], D2[6], Q[6], D34[6], G6[6];
        SPK_MARK("dp_exp");
#pragma unroll
        for (int i = 0; i < 6; ++i) D[i] = g * dev[i];
        sym_mul(D, D, D2);
#pragma unroll
        for (int i = 0; i < 6; ++i) Q[i] = (1.f / 6.f) * D[i] + (1.f / 24.f) * D2[i];
        sym_mul(D2, Q, D34);           // D^3/6 + D^4/24
        const float xc = expm1_poly(c), ec = 1.f + xc;
#pragma unroll
        for (int i = 0; i < 6; ++i) G6[i] = ec * (D[i] + (0.5f * D2[i] + D34[i]));
        G6[0] += xc; G6[1] += xc; G6[2] += xc;
        const M3 G = sym6_to_m3(G6);
        M3 En;                           // E' = E + (G + E G): the small part as one FMA chain started from G, ONE rounding at the size of E
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) En.m[i + 3 * j] = E.m[i + 3 * j] + (G.m[i + 3 * j] + (E.m[i] * G.m[3 * j] + (E.m[i + 3] * G.m[3 * j + 1] + E.m[i + 6] * G.m[3 * j + 2])));
        E = En;
#pragma unroll
        for (int i = 0; i < 6; ++i) eps[i] += D[i];
        eps[0] += c; eps[1] += c; eps[2] += c;
    }
    float e[3];
    SPK_MARK("dp_eigvals");
    sym_eigvals(eps, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) sm1[i] = expm1_poly(e[i]);
    sv_valid = true;
    SPK_MARK("dp_stress");
    if (WANT_STRESS && mo.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR && phase == 1.f) {
#ifdef SPK_STRESS_RIGHT
        float m6[6];
        one_minus_exp_neg_series(eps, m6);
        T = corotated_tau_e(mo, eh, E, m6, park, &jm1);
#else
        // of the mapped F itself, through its left stretch (the mapped strain stays inside the tier of the trial strain: the deviator only shrinks,
        // and so does |tr| as long as log_vol_gain >= 0; the tier bound covers |lvg| for the rest)
        T = corotated_tau_left(mo, eh, E, hi, park, &jm1);
#endif
    }
}

template <bool WANT_STRESS, class MO>
SPK_DEV void dp_update_e(const MO& mo, float phase, float eh, M3& E, float& Jp, float& q, float& lvg, float sm1[3], bool& sv_valid, SymStress& T,
                         ParkSlot park = ParkSlot()) {
    sv_valid = false;
    T.valid = false;
    if (mo.only_active_when_failed && phase != 0.f) return;
    // eps = lin + V diag(dv) V^T throughout: `lin` the first-order matrix, `dv` the remainders in the eigenbasis, e_k its eigenvalues
    float lin[6], e[3], dv[3];
    strain_metric(E, lin);
    // The pending volume gain only enters the tier test when NEGATIVE (possible with volume_correction != 1 only).  A positive one never reaches a
    // series: either tr eps + lvg > 0 -- tensile, sigma' = 1, no function of lvg is evaluated -- or lvg <= -tr eps, which |S| already bounds.
    // (Particles that have flown apart carry log_vol_gain of order 1 for the rest of the run: tested as |lvg| they sent every wave they sit
    // in down the eigenbasis path.)
#ifdef SPK_TIER_ABS_LVG
    const float lvg_t = lvg;
#else
    const float lvg_t = fminf(lvg, 0.f);
#endif
    SPK_HOT(__all(series2_ok(lin, lvg_t)));
    if (SPK_SERIES_MAX > 0.f && __all(series2_ok(lin, lvg_t))) {   // wave-uniform, and so is the tier
        dp_update_series<WANT_STRESS>(mo, phase, eh, lin, E, Jp, q, lvg, sm1, sv_valid, T, park, !__all(series_ok(lin, lvg_t)));
        return;
    }
#ifndef SPK_HOT_ONLY   // (analysis build: the eigenbasis path compiled out, see SPK_HOT)
    const Eig3 d = log_stretch_s(lin, e, dv);
#pragma unroll
    for (int i = 0; i < 6; ++i) lin[i] *= 0.5f;
    const float alpha = dp_alpha(mo, q);
    // strain = eps + (lvg / 3) I.  Its invariants from the eigenvalues: tr and |dev|^2 miss only the SQUARE of the off-diagonal residue
    const float tre = e[0] + e[1] + e[2], em = tre * (1.f / 3.f);
    const float tr = tre + lvg;
    const float dn2 = (e[0] - em) * (e[0] - em) + (e[1] - em) * (e[1] - em) + (e[2] - em) * (e[2] - em);
    // three outcomes: tensile (sigma' = 1), inside the yield surface (F stays), projection onto the cone; folded into one straight
    // line: K = ln(Sigma' / Sigma) = klin + V diag(kd) V^T with eigenvalues k, `moved`
    float klin[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, kd[3] = {0.f, 0.f, 0.f}, k[3] = {0.f, 0.f, 0.f}, dq = 0.f;
    bool moved = false;
    if (dn2 == 0.f || tr > 0.f) {
        dq = fast_sqrt(dn2 + tr * tr * (1.f / 3.f));     // |strain|^2 = |dev eps|^2 + (tr eps + lvg)^2 / 3
#pragma unroll
        for (int i = 0; i < 6; ++i) klin[i] = -lin[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { kd[i] = -dv[i]; k[i] = -e[i]; }
        moved = true;
    } else {
        const float dn = fast_sqrt(dn2);
        const float gamma = dn + mo.p_k * tr * alpha;   // p_k = (3 lambda + 2 mu) / (2 mu), precomputed on the host
        if (gamma > 0.f) {                              // otherwise inside the yield surface
            const float gdn = gamma * fast_rcp(dn);
            // ln sigma' = strain - gdn dev(strain)  =>  K = (lvg / 3) I - gdn dev(eps)
            const float lm = (lin[0] + lin[1] + lin[2]) * (1.f / 3.f), dm = (dv[0] + dv[1] + dv[2]) * (1.f / 3.f), c = lvg * (1.f / 3.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) { klin[i] = c - gdn * (lin[i] - lm); kd[i] = -gdn * (dv[i] - dm); k[i] = c - gdn * (e[i] - em); }
#pragma unroll
            for (int i = 3; i < 6; ++i) klin[i] = -gdn * lin[i];
            dq = gamma;
            moved = true;
        }
    }
    if (moved) {
        // prev_det = prod sigma = exp(tr eps), new_det = prod sigma' = exp(tr eps + tr K)
        const float so = tre, sn = tre + (k[0] + k[1] + k[2]);
        const float pm1 = expm1_small(so);
        float nm1 = expm1_small(sn), lnew = sn;
        const float diff = nm1 - pm1;
        if (!(diff > 0.f) && mo.volume_correction != 1.f) { nm1 = pm1 + diff * mo.volume_correction; lnew = log1p_small(nm1); }
        Jp *= (1.f + pm1) * fast_rcp(1.f + nm1);
        lvg += so - lnew;
        q += dq;
        // F' = F V diag(sigma'/sigma) V^T = F (I + G),  G = exp(K) - I = klin + V diag(kd + expm1_rem(k)) V^T  =>  E' = E + G + E G
        float gr[3], g6[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) gr[i] = kd[i] + expm1_rem(k[i]);
        sym_compose(d.v, gr, g6);
#pragma unroll
        for (int i = 0; i < 6; ++i) { g6[i] += klin[i]; lin[i] += klin[i]; }
        const M3 G = sym6_to_m3(g6);
        const M3 EG = m3_mul(E, G);
#pragma unroll
        for (int i = 0; i < 9; ++i) E.m[i] += G.m[i] + EG.m[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { e[i] += k[i]; dv[i] += kd[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) sm1[i] = expm1_small(e[i]);
    sv_valid = true;
    if (WANT_STRESS && mo.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR && phase == 1.f) {
        // M = I - exp(-eps') = eps' - V diag(expm1_rem(-e')) V^T = lin + V diag(dv - expm1_rem(-e')) V^T: one composition
        float mr[3], m6[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) mr[i] = dv[i] - expm1_rem(-e[i]);
        sym_compose(d.v, mr, m6);
#pragma unroll
        for (int i = 0; i < 6; ++i) m6[i] += lin[i];
        T = corotated_tau_e(mo, eh, E, m6, park);
    }
#endif
}

// SnowPlasticity::update_particle (plasticity_snow.rs:26-48)
template <class MO>
SPK_DEV void snow_update(const MO& mo, M3& F, float& eh, float& Jp) {
    const Svd3 d = svd3<SPK_SVD_SWEEPS>(F);
    float g[3], prod_old = 1.f, prod_new = 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float ns = fminf(fmaxf(d.s[i], 1.0f - mo.x0), 1.0f + mo.x1);
        prod_old *= d.s[i]; prod_new *= ns;
        g[i] = d.s[i] == 0.f ? 0.f : ns / d.s[i];
    }
    Jp *= prod_old / prod_new;
    eh = expf(mo.x2 * (1.0f - Jp));
    F = svd_compose_scaled(d, g);
}
// RankinePlasticity::update_particle (plasticity_rankine.rs:34-78); q = plastic_hardening
SPK_DEV void rankine_update(const DevModel& mo, M3& F, float& q) {
    const float mu = mo.x0, lambda = mo.x1;
    const Svd3 d = svd3<SPK_SVD_SWEEPS>(F);
    float e[3] = {logf(d.s[0]), logf(d.s[1]), logf(d.s[2])};
    const float p0 = e[0], p1 = e[1], p2 = e[2];
    // indices of the largest / middle / smallest principal strain, without dynamic register indexing
    const int i1 = (e[0] >= e[1]) ? ((e[0] >= e[2]) ? 0 : 2) : ((e[1] >= e[2]) ? 1 : 2);
    const int i3 = (e[0] < e[1]) ? ((e[0] < e[2]) ? 0 : 2) : ((e[1] < e[2]) ? 1 : 2);
    const int i2 = 3 - i1 - i3;
    const float e1 = i1 == 0 ? e[0] : (i1 == 1 ? e[1] : e[2]), e2 = i2 == 0 ? e[0] : (i2 == 1 ? e[1] : e[2]), e3 = i3 == 0 ? e[0] : (i3 == 1 ? e[1] : e[2]);
    const float sum = e[0] + e[1] + e[2];
    const float soft = mo.x2 - (q - 1.0f);
    if (lambda * sum + 2.0f * mu * e1 <= soft) return;
    float n1 = e1, n2 = e2, n3 = e3;
    if ((2.0f * mu + lambda) * e2 + lambda * (sum - e1) <= soft) {
        n1 = (soft - lambda * (sum - e1)) / (2.0f * mu + lambda);
    } else if ((2.0f * mu + 3.0f * lambda) * e3 <= soft) {
        n1 = n2 = (soft - lambda * (sum - e1 - e2)) / (2.0f * mu + 2.0f * lambda);
    } else {
        n1 = n2 = n3 = soft / (2.0f * mu + 3.0f * lambda);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) e[i] = (i == i1) ? n1 : ((i == i2) ? n2 : n3);
    const float d0 = p0 - e[0], d1 = p1 - e[1], d2 = p2 - e[2];
    q = fminf(q + mo.x3 * sqrtf(d0 * d0 + d1 * d1 + d2 * d2), mo.x2);
    float g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = d.s[i] == 0.f ? 0.f : expf(e[i]) / d.s[i];
    F = svd_compose_scaled(d, g);
}
// NaccPlasticity::project_deformation_gradient (plasticity_nacc.rs:62-152), d = 3
SPK_DEV void nacc_update(const DevModel& mo, M3& F, float& alpha) {
    const float mu = mo.x0, kappa = mo.x1, xi = mo.x2, beta = mo.x3, M = mo.x4;
    const bool hard = mo.xi != 0;
    const Svd3 d = svd3<SPK_SVD_SWEEPS>(F);
    const float sq[3] = {d.s[0] * d.s[0], d.s[1] * d.s[1], d.s[2] * d.s[2]};
    const float sq_tr = sq[0] + sq[1] + sq[2];
    const float p0 = kappa * (1.0e-5f + sinhf(xi * fmaxf(-alpha, 0.f)));
    const float j_tr = d.s[0] * d.s[1] * d.s[2];
    const float sc = mu * powf(j_tr, -2.0f / 3.0f);
    const float s_tr[3] = {sc * (sq[0] - sq_tr / 3.0f), sc * (sq[1] - sq_tr / 3.0f), sc * (sq[2] - sq_tr / 3.0f)};
    const float p_tr = -(kappa / 2.0f * (j_tr - 1.0f / j_tr)) * j_tr;
    float ns[3];
    const bool tip_a = p_tr > p0, tip_b = !tip_a && p_tr < -beta * p0;
    if (tip_a || tip_b) {
        const float j_n1 = sqrtf((tip_a ? -2.0f * p0 : 2.0f * beta * p0) / kappa + 1.0f);
        ns[0] = ns[1] = ns[2] = powf(j_n1, 1.0f / 3.0f);
        if (hard) alpha += logf(j_tr / j_n1);
    } else {
        const float y0 = (1.0f + 2.0f * beta) * 1.5f;
        const float y1 = M * M * (p_tr + beta * p0) * (p_tr - p0);
        const float s_n2 = s_tr[0] * s_tr[0] + s_tr[1] * s_tr[1] + s_tr[2] * s_tr[2];
        if (y0 * s_n2 + y1 < 1.0e-4f) return;   // inside the yield surface
        const float sn = sqrtf(s_n2);
        if (hard && p0 > 1.0e-4f && p_tr < p0 - 1.0e-4f && p_tr > -beta * p0 + 1.0e-4f) {
            const float p_c = (1.0f - beta) * p0 / 2.0f;
            const float q_tr = sqrtf(1.5f) * sn;
            float dx = p_c - p_tr, dy = -q_tr;
            const float dn = sqrtf(dx * dx + dy * dy);
            dx /= dn; dy /= dn;
            const float c = M * M * (p_c + beta * p0) * (p_c - p0);
            const float b = M * M * dx * (2.0f * p_c - p0 + beta * p0);
            const float a = M * M * dx * dx + (1.0f + 2.0f * beta) * dy * dy;
            const float discr = sqrtf(b * b - 4.0f * a * c);
            const float q1 = p_c + (-b + discr) / (2.0f * a) * dx, q2 = p_c + (-b - discr) / (2.0f * a) * dx;
            const float p_x = (p_tr - p_c) * (q1 - p_c) > 0.f ? q1 : q2;
            const float j_x = sqrtf(fabsf(-2.0f * p_x / kappa + 1.0f));
            if (j_x > 1.0e-4f) alpha += logf(j_tr / j_x);
        }
        const float k = sqrtf(-y1 / y0) * (powf(j_tr, 2.0f / 3.0f) / mu);
#pragma unroll
        for (int i = 0; i < 3; ++i) ns[i] = sqrtf(k * (s_tr[i] / sn) + sq_tr / 3.0f);
    }
    float g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = d.s[i] == 0.f ? 0.f : ns[i] / d.s[i];
    F = svd_compose_scaled(d, g);
}
// Elastic predictor test (round 6): can the plastic model possibly act on F = I + E?  A bound from the strain metric S = F^T F - I alone, without the SVD
// both models start with: every eigenvalue lambda_k = sigma_k^2 - 1 of S lies within +-|S|_F.
//   snow (plasticity_snow.rs:27-49): nothing is clamped while every sigma_k is inside [1 - min_epsilon, 1 + max_epsilon]  <=  |S|_F below the nearer of
//        1 - (1 - min_epsilon)^2 and (1 + max_epsilon)^2 - 1.  The reference still runs its SVD and recomposes the same F (to rounding) and multiplies
//        Jp by 1; skipping leaves F as it is -- one rounding closer to the f64 trajectory.
//   Rankine (plasticity_rankine.rs:34-47): returns at once when lambda tr(eps) + 2 mu eps_1 <= tensile strength left; eps_k = ln(1 + lambda_k) / 2
//        <= |S|_F / 2 bounds the left side by (3 lambda + 2 mu) |S|_F / 2 for lambda, mu >= 0: the same early return, F untouched like there.
// `false` = the model has nothing to do for this particle (no SVD).  A falling snow block is elastic everywhere but in the zone the impact has reached.
template <class MO>
SPK_DEV bool rare_plastic_may_act(const MO& mo, const M3& E, float q) {
    if (mo.ptag != SPARKL_PLASTIC_SNOW && mo.ptag != SPARKL_PLASTIC_RANKINE) return true;
    float S6[6];
    strain_metric_left(E, S6);   // (the LEFT metric F F^T - I: the same eigenvalues and Frobenius norm as F^T F - I, and the one the stress and the energy below form too)
    const float n2 = S6[0] * S6[0] + S6[1] * S6[1] + S6[2] * S6[2] + 2.f * (S6[3] * S6[3] + S6[4] * S6[4] + S6[5] * S6[5]);
    if (!(n2 == n2)) return true;
    if (mo.ptag == SPARKL_PLASTIC_SNOW) {
        const float lo = 1.f - (1.f - mo.x0) * (1.f - mo.x0), hi = (1.f + mo.x1) * (1.f + mo.x1) - 1.f;
        const float r = 0.98f * fminf(lo, hi);
        return !(r > 0.f && n2 < r * r);
    }
    const float mu = mo.x0, lambda = mo.x1, soft = mo.x2 - (q - 1.0f);
    if (!(mu >= 0.f && lambda >= 0.f && soft > 0.f)) return true;
    const float nS = sqrtf(n2);
    if (nS > 0.5f) return true;                                   // (ln(1 + x) / 2 <= x / 2 only needs x > -1; keep well away from it)
    return !(0.5f * 1.02f * (3.f * lambda + 2.f * mu) * nS <= soft);
}
// Snow / Rankine / NACC dispatch (only instantiated in the RARE kernels, see g2p_update)
template <class MO>
SPK_DEV void rare_plastic_update(const MO& mo, M3& F, float& Jp, float& q, float& eh, float& alpha) {
    if constexpr (std::is_same<MO, DevModel>::value) {
        if (mo.ptag == SPARKL_PLASTIC_SNOW) snow_update(mo, F, eh, Jp);
        else if (mo.ptag == SPARKL_PLASTIC_RANKINE) rankine_update(mo, F, q);
        else if (mo.ptag == SPARKL_PLASTIC_NACC) nacc_update(mo, F, alpha);
    } else {
        static_assert(MO::ptag == SPARKL_PLASTIC_SNOW, "the only rare plastic model with a compile-time view");
        snow_update(mo, F, eh, Jp);
    }
}
// MaximumStressFailure::particle_failed (failure_maximum_stress.rs:20-28).  nalgebra's iterative symmetric eigensolver
// (third party) -> closed-form eigenvalues of the symmetric 3x3 (trigonometric form), evaluated in double.
SPK_DEV bool maximum_stress_failed(const DevModel& mo, const M3& t) {
    const double a00 = t.m[0], a11 = t.m[4], a22 = t.m[8];
    const double a01 = 0.5 * ((double)t.m[3] + t.m[1]), a02 = 0.5 * ((double)t.m[6] + t.m[2]), a12 = 0.5 * ((double)t.m[7] + t.m[5]);
    const double qm = (a00 + a11 + a22) / 3.0;
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double p2 = (a00 - qm) * (a00 - qm) + (a11 - qm) * (a11 - qm) + (a22 - qm) * (a22 - qm) + 2.0 * p1;
    double mx, mn;
    if (p2 <= 0.0) { mx = mn = qm; }
    else {
        const double pp = sqrt(p2 / 6.0), ip = 1.0 / pp;
        const double b00 = (a00 - qm) * ip, b11 = (a11 - qm) * ip, b22 = (a22 - qm) * ip, b01 = a01 * ip, b02 = a02 * ip, b12 = a12 * ip;
        double r = 0.5 * (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02));
        r = fmin(fmax(r, -1.0), 1.0);
        // cos(acos(|r|) / 3) as Newton steps on 4 c^3 - 3 c = |r| from the chord (c in [cos(pi/6), 1]: a simple root, quadratic convergence from
        // an error <= 1e-2: four steps reach double precision); r < 0 is the mirrored problem.  The libm acos / cos this replaces (round 6) are two
        // dozen double-precision polynomial coefficients that the compiler hoisted out of the particle loop of the RARE kernels and SPILLED: the
        // whole 104 B per lane of scratch of k_g2p2g<RARE> / k_g2p<RARE> (profiles/r06_kernel_experiments.txt, 3).
        const double ar = fabs(r);
        double c = (double)(0.8660254f + 0.1339746f * (float)ar);   // (the start in f32: a 64-bit literal here is hoisted out of the particle loop and spilled too)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double c2 = c * c;
            c -= ((4.0 * c2 - 3.0) * c - ar) / (12.0 * c2 - 3.0);
        }
        const double sn = sqrt(fmax(3.0 - 3.0 * (c * c), 0.0));   // sqrt(3) sin(phi')
        // eigenvalues of the normalised deviator: r >= 0: {2 c, -c + sn, -c - sn};  r < 0: their negatives
        mx = qm + pp * (r >= 0.0 ? 2.0 * c : c + sn);
        mn = qm + pp * (r >= 0.0 ? -c - sn : -2.0 * c);
    }
    return mx > (double)mo.max_principal_stress || (mx - mn) * 0.5 > (double)mo.max_shear_stress;
}

// constitutive_model.pos_energy(particle)
// `jm1` = det(F) - 1 (det_minus_one(E)), which the caller holds anyway
template <class MO>
SPK_DEV float model_pos_energy(const MO& mo, float phase, float eh, const M3& F, const M3& E, const float sm1[3], bool sv_valid, float jm1) {
    if (mo.ctag == SPARKL_CONSTITUTIVE_COROTATED_LINEAR) {
        // elasticity_corotated_linear.rs:105-118 on the small quantities: sigma_k - 1 (`sm1` when the return map already holds
        // them) and J - 1; E = F - I
        float s[3];
        SPK_HOT(sv_valid);
        if (sv_valid) { s[0] = sm1[0]; s[1] = sm1[1]; s[2] = sm1[2]; }
        else {
            float S6[6], eps[3], re[3];
            strain_metric_left(E, S6);   // (eigenvalues and norm are those of F^T F - I; formed once for the predictor of the rare models, this sum and the stress)
            if (SPK_SERIES_MAX > 0.f && __all(series2_ok(S6))) {   // wave-uniform; small strains
                // sigma_k - 1 from the eigenvalues of S = F^T F - I themselves:  sigma^2 - 1 = lambda  =>  sigma - 1 = lambda / (1 + sqrt(1 + lambda)), an
                // identity without cancellation -- no logarithm series in front of the eigenvalues and no expm1 behind them (round 6: the Hencky strain is
                // what the return map needs, not this sum; 110 - 220 instructions per particle of every corotated model without Drucker-Prager)
                sym_eigvals(S6, eps);
#pragma unroll
                for (int k = 0; k < 3; ++k) s[k] = eps[k] * fast_rcp(1.f + fast_sqrt(1.f + eps[k]));
            } else {
                (void)log_stretch_s(S6, eps, re);
                s[0] = expm1_small(eps[0]); s[1] = expm1_small(eps[1]); s[2] = expm1_small(eps[2]);
            }
        }
        float nsq = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float e = fmaxf(s[i], 0.f); nsq += e * e; }
        const float pos_dev = mo.mu * eh * nsq;
        const float sph = mo.lambda * eh / 2.f * jm1 * jm1;
        return jm1 < 0.f ? pos_dev : pos_dev + sph;
    }
    if (mo.ctag == SPARKL_CONSTITUTIVE_NEO_HOOKEAN) {
        // elasticity_neo_hookean.rs:118-141:  dev = mu/2 (tr(F F^T) J^(-2/3) - 3),  vol = k/2 ((J^2 - 1)/2 - ln J).  Both are SECOND order in the
        // strain and the reference forms them as differences of O(1) numbers: in f32 the result carries 3e-7 mu of absolute error, which is the
        // whole value at strains of 1e-3 (the f32 oracle's psi_pos is 17 x its f64 value on the spinning cube).  Strain form (round 6): with
        // S = F F^T - I, its invariants s1, s2, s3 and d = J^2 - 1 = s1 + s2 + s3,
        //   tr(F F^T) J^(-2/3) - 3 = (3 + s1)(1 + a) - 3,   a = (1 + d)^(-1/3) - 1 = -d/3 + r(d),   r = 2 d^2/9 - 14 d^3/81 + ...
        //                          = -(s2 + s3) + 3 r + s1 a           (the first-order terms s1 - d/3 * 3 cancel ANALYTICALLY)
        //   (J^2 - 1)/2 - ln J     = (d - log1p(d)) / 2 = d^2/4 - d^3/6 + d^4/8 - ...
        // while |d| and |s1| stay below 0.1 (wave-uniform); beyond, the reference's form has nothing left to cancel.
        const float pc = nh_phase_coeff(phase);
        const float k = 2.f / 3.f * mo.mu * eh + mo.lambda * eh;
        float x[6];
        strain_metric_left(E, x);
        const float s1 = x[0] + x[1] + x[2];
        const float d = jm1 * (2.f + jm1);
        float dev_core, vol_core;
        if (__all(fabsf(d) < 0.1f && fabsf(s1) < 0.1f)) {
            const float s2 = (x[0] * x[1] - x[3] * x[3]) + (x[0] * x[2] - x[4] * x[4]) + (x[1] * x[2] - x[5] * x[5]);
            const float s3 = x[0] * (x[1] * x[2] - x[5] * x[5]) - x[3] * (x[3] * x[2] - x[5] * x[4]) + x[4] * (x[3] * x[5] - x[1] * x[4]);
            const float d2 = d * d;
            const float r = d2 * (2.f / 9.f + d * (-14.f / 81.f + d * (35.f / 243.f + d * (-91.f / 729.f + d * (728.f / 6561.f + d * (-1976.f / 19683.f + d * 0.09203f))))));
            const float a = r - d * (1.f / 3.f);
            dev_core = (3.f * r - (s2 + s3)) + s1 * a;
            vol_core = d2 * (0.25f + d * (-1.f / 6.f + d * (0.125f + d * (-0.1f + d * (1.f / 12.f + d * (-1.f / 14.f + d * 0.0625f))))));
        } else {
            const float j = 1.f + jm1;
            dev_core = (3.f + s1) * powf(j, -2.f / 3.f) - 3.f;
            vol_core = (d * 0.5f - logf(j));
        }
        const float dev = eh * mo.mu / 2.f * dev_core;
        const float vol = k / 2.f * vol_core;
        return jm1 < 0.f ? dev * pc : (dev + vol) * phase;
    }
    return 0.f;
}

}  // namespace spk
