// Host build of sparkl_amd/csrc/device_math.h for CPU tests of the DEVICE arithmetic (tests/test_device_math_host.py).
// The header is compiled unchanged by the ROCm clang++ as plain C++: the gfx950 builtins it uses are mapped to libm, the
// wave votes to the single "lane" of a host call.  Two builds: the default one and -DSPK_SERIES_MAX=0.f (eigenbasis path only).
#include <cmath>
#include <cstdint>
#include <cstring>
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) (1.0f / std::sqrt(x))
#define __builtin_amdgcn_sqrtf(x) (std::sqrt(x))
#define __builtin_amdgcn_exp2f(x) (std::exp2(x))
#define __builtin_amdgcn_logf(x) (std::log2(x))
#define __expf(x) (std::exp(x))
#define __any(x) (x)
#define __all(x) (x)
#define __device__
#define SPK_HOST_BUILD_NO_AS 1
#define __forceinline__ inline
#include "../../sparkl_amd/csrc/device_math.h"

using namespace spk;

static DevModel make_model(const float* mp) {
    // mp: lambda, mu, h0, h1, h2, h3, p_lambda, p_mu, volume_correction, has_plasticity
    DevModel mo;
    std::memset(&mo, 0, sizeof(mo));
    mo.ctag = SPARKL_CONSTITUTIVE_COROTATED_LINEAR;
    mo.ptag = mp[9] != 0.f ? SPARKL_PLASTIC_DRUCKER_PRAGER : -1;
    mo.cfl_coeff = 0.5f; mo.lambda = mp[0]; mo.mu = mp[1];
    mo.h0 = mp[2]; mo.h1 = mp[3]; mo.h2 = mp[4]; mo.h3 = mp[5]; mo.p_lambda = mp[6]; mo.p_mu = mp[7]; mo.volume_correction = mp[8];
    mo.p_k = (3.f * mo.p_lambda + 2.f * mo.p_mu) / (2.f * mo.p_mu);
    return mo;
}

extern "C" {
// Drucker-Prager return map on E = F_trial - I (column-major).  state = {Jp, q, lvg} in/out; out = E'[9], sm1[3], tau6[6], psi.
void devmath_dp_update(const float* mp, float phase, float eh, const float* E_in, float* state, float* E_out, float* sm1, float* tau6, float* psi, int* flags) {
    const DevModel mo = make_model(mp);
    M3 E;
    for (int i = 0; i < 9; ++i) E.m[i] = E_in[i];
    float s[3] = {0.f, 0.f, 0.f};
    bool sv_valid = false;
    SymStress T;
    T.valid = false;
    dp_update_e<true>(mo, phase, eh, E, state[0], state[1], state[2], s, sv_valid, T);
    for (int i = 0; i < 9; ++i) E_out[i] = E.m[i];
    for (int i = 0; i < 3; ++i) sm1[i] = s[i];
    for (int i = 0; i < 6; ++i) tau6[i] = T.valid ? T.t[i] : 0.f;
    M3 F = E;
    F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    *psi = model_pos_energy(mo, phase, eh, F, E, s, sv_valid, det_minus_one(E));
    *flags = (sv_valid ? 1 : 0) | (T.valid ? 2 : 0);
}
// phase == 1 corotated Kirchhoff stress of F = I + E (column-major 3x3 out) and the positive energy without a return map
void devmath_corotated(const float* mp, float eh, const float* E_in, float* tau9, float* psi) {
    const DevModel mo = make_model(mp);
    M3 E;
    for (int i = 0; i < 9; ++i) E.m[i] = E_in[i];
    const M3 t = corotated_kirchhoff_fe(mo, eh, E);
    for (int i = 0; i < 9; ++i) tau9[i] = t.m[i];
    M3 F = E;
    F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    const float s[3] = {0.f, 0.f, 0.f};
    *psi = model_pos_energy(mo, 1.f, eh, F, E, s, false, det_minus_one(E));
}
// neo-Hookean positive energy of F = I + E (column-major) at phase = 1: mp[0] = lambda, mp[1] = mu
float devmath_neo_hookean_energy(const float* mp, float eh, const float* E_in) {
    DevModel mo = make_model(mp);
    mo.ctag = SPARKL_CONSTITUTIVE_NEO_HOOKEAN;
    M3 E;
    for (int i = 0; i < 9; ++i) E.m[i] = E_in[i];
    M3 F = E;
    F.m[0] += 1.f; F.m[4] += 1.f; F.m[8] += 1.f;
    const float s[3] = {0.f, 0.f, 0.f};
    return model_pos_energy(mo, 1.f, eh, F, E, s, false, det_minus_one(E));
}
void devmath_sym_eigvals(const float* a6, float* e3) { sym_eigvals(a6, e3); }
float devmath_series_max(void) { return SPK_SERIES_MAX; }
float devmath_series_max2(void) { return SPK_SERIES_MAX2; }
}
