"""The DEVICE arithmetic of the constitutive path, run on the CPU: sparkl_amd/csrc/device_math.h is compiled unchanged as plain
C++ (tests/cpp/device_math_host.cpp maps the gfx950 builtins to libm) and held against the f64 oracle on random trial states.

Two builds: the default one, whose Drucker-Prager return map / corotated stress take the series path (polynomials in
S = F^T F - I, no eigenvectors; two tiers) while |S| < SPK_SERIES_MAX2, and -DSPK_SERIES_MAX=0.f, which always takes the eigenbasis path
(cyclic Jacobi).  Both must reproduce plasticity_drucker_prager.rs:36-104 / elasticity_corotated_linear.rs:31-118 as the oracle
restates them, to a few 1e-7 of the TRIAL strain (the strain-form bound of DESIGN.md section 5); the series path must also hand over
to the eigenbasis path without a step at the switch-over."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import oracle
from sparkl_amd.models import CorotatedLinearElasticity, DruckerPragerPlasticity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "device_math_host.cpp")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
f32 = np.float32


def _build(tag, extra):
    out = os.path.join(ROOT, "tests", "cpp", f"libdevmath_{tag}.so")
    deps = [SRC, os.path.join(ROOT, "sparkl_amd", "csrc", "device_math.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call([CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
                              + extra + [SRC, "-o", out])
    return C.CDLL(out)


@pytest.fixture(scope="module")
def libs():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang++ not found")
    L = {"series": _build("series", []), "jacobi": _build("jacobi", ["-DSPK_SERIES_MAX=0.f"])}
    for l in L.values():
        l.devmath_series_max.restype = C.c_float
        l.devmath_series_max2.restype = C.c_float
    assert L["series"].devmath_series_max() > 0.0 and L["jacobi"].devmath_series_max() == 0.0
    return L


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * K @ K


_DPM, _CMM = DruckerPragerPlasticity(1.0e7, 0.2), CorotatedLinearElasticity(1.0e7, 0.2)
_MP = np.array([_CMM.lambda_, _CMM.mu, _DPM.h0, _DPM.h1, _DPM.h2, _DPM.h3, _DPM.lambda_, _DPM.mu, 1.0, 1.0], dtype=f32)
_p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731


def _sym(t):
    return np.array([[t[0], t[3], t[4]], [t[3], t[1], t[5]], [t[4], t[5], t[2]]], dtype=np.float64)


def _dp(L, E, lvg=0.0, mp=_MP):
    Ecm = np.ascontiguousarray(E.T.reshape(9), dtype=f32)
    st = np.array([1.0, 1.0, lvg], dtype=f32)
    Eo, sm1, tau = np.zeros(9, f32), np.zeros(3, f32), np.zeros(6, f32)
    psi, fl = C.c_float(0), C.c_int(0)
    L.devmath_dp_update(_p(mp), C.c_float(1.0), C.c_float(1.0), _p(Ecm), _p(st), _p(Eo), _p(sm1), _p(tau), C.byref(psi), C.byref(fl))
    assert fl.value == 3
    return Eo.reshape(3, 3).T.astype(np.float64), st.astype(np.float64), _sym(tau), float(psi.value)


def _trial(rng, scale, theta, kind):
    A = rng.normal(size=(3, 3))
    sym = 0.5 * (A + A.T) * scale
    if kind == 0:
        sym -= np.eye(3) * abs(rng.normal()) * scale * 2           # compressive: inside / onto the cone
    elif kind == 1:
        sym = np.diag([-2.0, 0.9, 1.1]) * scale + 0.05 * sym       # uniaxial compression, nearly equal lateral strains (the column at rest)
    F = _rot(rng.normal(size=3), theta * rng.normal()) @ (np.eye(3) + sym)
    E32 = (F - np.eye(3)).astype(f32)
    return E32.astype(np.float64)


@pytest.mark.parametrize("scale", [1e-6, 1e-4, 3e-3, 1e-2, 5e-2])
def test_drucker_prager_return_map_matches_the_f64_oracle(libs, scale):
    """F', q, log_vol_gain, the corotated stress of F' and psi_pos against the oracle, in units of the trial strain."""
    dp = _DPM.to_core_model().u.drucker_prager
    cm = _CMM.to_core_model()
    rng = np.random.default_rng(7)
    worst = {k: np.zeros(4) for k in libs}
    for theta in (0.0, 1e-2):
        for trial in range(120):
            E = _trial(rng, scale, theta, trial % 3)
            F = E + np.eye(3)
            lvg0 = 0.0 if trial % 5 else scale * rng.normal()
            F2, Jp, q2, lvg = oracle.dp_update(dp, F, Jp=1.0, q=1.0, lvg=lvg0, f64=True)
            tau_ref = oracle.kirchhoff_stress(cm, F2, f64=True)
            psi_ref = oracle.pos_energy(cm, F2, f64=True)
            strain = max(np.abs(np.linalg.svd(F, compute_uv=False) - 1).max(), 1e-12)
            for k, L in libs.items():
                Eo, st, T, psi = _dp(L, E, lvg=lvg0)
                err = [np.abs(Eo + np.eye(3) - F2).max() / strain, np.abs(T - tau_ref).max() / (2 * _CMM.mu * strain),
                       abs(st[2] - lvg) / strain, abs(psi - psi_ref) / (_CMM.mu * strain ** 2)]
                worst[k] = np.maximum(worst[k], err)
    print(scale, {k: [f"{x:.1e}" for x in w] for k, w in worst.items()})
    # a particle rotated by theta (here up to ~3e-2 rad) carries E = O(theta): its f32 storage alone is 1e-7 theta absolute
    # ("Limits")
    bound = 3e-6 + 6e-9 / scale
    for k, w in worst.items():
        assert (w[:3] < bound).all() and w[3] < 10 * bound, (k, scale, w, bound)
    if scale <= 3e-3:   # inside the series domain: never worse than the eigenbasis path by more than rounding
        # (F' and log_vol_gain.  The stress of the series path is that of the STORED F' -- its left stretch, corotated_tau_left: what the
        # reference's next P2G evaluates -- so it carries the f32 storage of a rotated F', 1e-7 theta / strain, which `bound` above covers;
        # the eigenbasis path forms it from the unrounded mapped strain it still holds.  Trajectories do not tell them apart:
        # profiles/r04_parity.jsonl before / after.)
        assert (worst["series"][[0, 2]] <= worst["jacobi"][[0, 2]] + 5e-7).all(), worst


@pytest.mark.parametrize("which", ["devmath_series_max", "devmath_series_max2"])
def test_series_path_hands_over_without_a_step(libs, which):
    """Across each switch-over (|S|_F = SPK_SERIES_MAX: quartic -> S^7 logarithm; SPK_SERIES_MAX2: series -> eigenbasis) the series build
    agrees with the eigenbasis build to the truncation bound of the series (< 1e-6 of the strain)."""
    thr = getattr(libs["series"], which)()
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(200):
        E = _trial(rng, 0.25 * thr * (0.8 + 0.4 * rng.random()), 0.0, trial % 3)   # |S|_F ~ 2 |eps|: straddles the threshold
        a = _dp(libs["series"], E)
        b = _dp(libs["jacobi"], E)
        strain = np.abs(np.linalg.svd(E + np.eye(3), compute_uv=False) - 1).max()
        worst = max(worst, np.abs(a[0] - b[0]).max() / strain, np.abs(a[2] - b[2]).max() / (2 * _CMM.mu * strain))
    print("hand-over", worst)
    assert worst < 1.5e-6


def test_corotated_stress_without_plasticity(libs):
    """corotated_kirchhoff_fe / model_pos_energy (the elastic block of sand3.rs) on both paths against the oracle."""
    cm = _CMM.to_core_model()
    mp = _MP.copy()
    mp[9] = 0.0
    rng = np.random.default_rng(11)
    for scale in (1e-5, 1e-3, 8e-3, 4e-2):
        worst = {k: np.zeros(2) for k in libs}
        for trial in range(100):
            E = _trial(rng, scale, 1e-2, trial % 3)
            F = E + np.eye(3)
            tau_ref = oracle.kirchhoff_stress(cm, F, f64=True)
            psi_ref = oracle.pos_energy(cm, F, f64=True)
            strain = np.abs(np.linalg.svd(F, compute_uv=False) - 1).max()
            for k, L in libs.items():
                tau, psi = np.zeros(9, f32), C.c_float(0)
                L.devmath_corotated(_p(mp), C.c_float(1.0), _p(np.ascontiguousarray(E.T.reshape(9), dtype=f32)), _p(tau), C.byref(psi))
                worst[k] = np.maximum(worst[k], [np.abs(tau.reshape(3, 3).T - tau_ref).max() / (2 * _CMM.mu * strain), abs(psi.value - psi_ref) / (_CMM.mu * strain ** 2)])
        print(scale, worst)
        bound = 3e-6 + 6e-9 / scale
        for k, w in worst.items():
            assert w[0] < bound and w[1] < 10 * bound, (k, scale, w)


def test_large_positive_volume_gain_stays_on_the_series_path(libs):
    """A particle that has flown apart carries log_vol_gain of order 1.  No series of the return map ever sees a positive lvg (tensile: sigma' = 1;
    on the cone lvg <= -tr eps), so the tier test of dp_update_e ignores it: the series build must agree with the eigenbasis build and the f64
    oracle on tensile AND compressive trial states with lvg up to 2 (before round 4's change these went down the eigenbasis path by |lvg|)."""
    dp = _DPM.to_core_model().u.drucker_prager
    cm = _CMM.to_core_model()
    rng = np.random.default_rng(21)
    worst = {k: np.zeros(3) for k in libs}
    for trial in range(150):
        scale = 10.0 ** rng.uniform(-5, -2.3)
        E = _trial(rng, scale, 1e-2, trial % 3)
        F = E + np.eye(3)
        lvg0 = float(rng.uniform(0.0, 2.0)) if trial % 4 else float(rng.uniform(0.0, 3.0)) * scale   # of order 1, or comparable to the strain (the cone case)
        F2, Jp, q2, lvg = oracle.dp_update(dp, F, Jp=1.0, q=1.0, lvg=lvg0, f64=True)
        tau_ref = oracle.kirchhoff_stress(cm, F2, f64=True)
        strain = max(np.abs(np.linalg.svd(F, compute_uv=False) - 1).max(), 1e-12)
        for k, L in libs.items():
            Eo, st, T, psi = _dp(L, E, lvg=lvg0)
            err = [np.abs(Eo + np.eye(3) - F2).max() / strain, np.abs(T - tau_ref).max() / (2 * _CMM.mu * strain), abs(st[2] - lvg) / max(strain, abs(lvg))]
            worst[k] = np.maximum(worst[k], err)
    print(worst)
    for k, w in worst.items():
        assert (w < 2e-4).all(), (k, w)     # the storage bound of a rotated F at the smallest strains here (1e-7 theta / strain, strain >= 1e-5)
    assert (np.abs(worst["series"] - worst["jacobi"]) < 5e-5).all(), worst


def test_closed_form_eigenvalues(libs):
    """sym_eigvals against numpy: isolated eigenvalues to 1e-6 |A|, a near-degenerate pair split symmetrically (its sum exact)."""
    rng = np.random.default_rng(5)
    L = libs["series"]
    for trial in range(2000):
        A = rng.normal(size=(3, 3))
        A = 0.5 * (A + A.T) * 10.0 ** rng.uniform(-6, 0)
        if trial % 4 == 0:
            Q = _rot(rng.normal(size=3), rng.normal())
            A = Q @ np.diag([1.0, 1.0 + 1e-6 * rng.normal(), -0.7]) @ Q.T * 10.0 ** rng.uniform(-6, 0)
        a6 = np.array([A[0, 0], A[1, 1], A[2, 2], A[0, 1], A[0, 2], A[1, 2]], dtype=f32)
        e = np.zeros(3, f32)
        L.devmath_sym_eigvals(_p(a6), _p(e))
        ref = np.linalg.eigvalsh(_sym(a6.astype(np.float64)))
        got = np.sort(e.astype(np.float64))
        nrm = np.abs(ref).max()
        assert abs(got.sum() - ref.sum()) < 2e-6 * nrm
        assert abs((got ** 2).sum() - (ref ** 2).sum()) < 5e-6 * nrm ** 2       # symmetric functions: second order in the pair split
        assert np.abs(got - ref).max() < 2e-3 * nrm
        if trial % 4:
            gaps = np.diff(ref)
            if gaps.min() > 0.05 * nrm:
                assert np.abs(got - ref).max() < 3e-6 * nrm


@pytest.mark.parametrize("scale", [1e-5, 1e-3, 3e-2, 0.3])
def test_neo_hookean_positive_energy_keeps_the_strain_precision(libs, scale):
    """elasticity_neo_hookean.rs:118-141 forms psi_pos as differences of O(1) numbers (tr(F F^T) J^(-2/3) - 3, (J^2 - 1)/2 - ln J): in f32 that is
    3e-7 mu of absolute error on a value of mu * strain^2.  The device form (device_math.h, model_pos_energy) cancels the first order analytically:
    against the f64 formula on random stretches behind a rotation of the order of the strain (E = F - I is O(rotation): a particle that has turned far
    carries the F-form's precision again), to 1e-4 of the value at every strain scale (the reference's f32 form: tens of per cent
    at strains of 1e-3)."""
    L = libs["series"]
    L.devmath_neo_hookean_energy.restype = C.c_float
    rng = np.random.default_rng(5)
    lam, mu = 5.77e5, 3.85e5
    mp = _MP.copy(); mp[0], mp[1] = lam, mu
    worst = 0.0
    for t in range(200):
        A = rng.normal(size=(3, 3))
        F = _rot(rng.normal(size=3), 3.0 * scale * rng.normal()) @ (np.eye(3) + 0.5 * (A + A.T) * scale)
        E32 = (F - np.eye(3)).astype(f32)
        F64 = np.eye(3) + E32.astype(np.float64)
        j = np.linalg.det(F64)
        if j <= 0.2:
            continue
        k = 2.0 / 3.0 * mu + lam
        dev = mu / 2.0 * (np.trace(F64 @ F64.T) * j ** (-2.0 / 3.0) - 3.0)
        vol = k / 2.0 * ((j * j - 1.0) / 2.0 - math.log(j))
        want = dev if j < 1.0 else dev + vol
        got = float(L.devmath_neo_hookean_energy(_p(mp), C.c_float(1.0), _p(np.ascontiguousarray(E32.T.reshape(9)))))
        worst = max(worst, abs(got - want) / max(abs(want), 1e-300))
    assert worst < 1e-4, worst
