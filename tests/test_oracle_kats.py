"""Analytic known-answer tests of the oracle (SURVEY 8(c) "What pins the oracle instead").

The reference has no dynamics test, so these closed-form cases pin the restated math: B-spline weights, SVD, the three
constitutive laws, the Drucker-Prager return map, the dt bound, the cuboid projection and the APIC transfer properties."""
import math

import numpy as np
import pytest

import oracle
from sparkl_amd import abi, scenes
from sparkl_amd.models import (CorotatedLinearElasticity, DruckerPragerPlasticity, MonaghanSphEos, NeoHookeanElasticity, ParticleModel,
                               ParticleModelSet, lame_lambda_mu)
from sparkl_amd.particles import new_particles

RNG = np.random.default_rng(12345)


def test_quadratic_kernel_moments():
    # kernel.rs:15-21: partition of unity, first moment 0, second moment h^2/4 (= 1/inv_d) for any x in [0.5, 1.5]
    for x in np.linspace(0.5, 1.5, 41):
        w = oracle.eval_all(x, f64=True)
        nodes = np.array([0.0, 1.0, 2.0])
        assert abs(w.sum() - 1.0) < 1e-14
        assert abs((w * (nodes - x)).sum()) < 1e-14
        assert abs((w * (nodes - x) ** 2).sum() - 0.25) < 1e-14
    assert np.allclose(oracle.eval_all(1.0), [0.125, 0.75, 0.125])


def test_svd_against_numpy():
    for trial in range(2000):
        A = RNG.normal(size=(3, 3))
        if trial % 5 == 0:
            A = np.eye(3) + 1e-3 * A                      # near identity: repeated singular values
        if trial % 7 == 0:
            A[:, 2] = A[:, 0] * 1e-9 + A[:, 1]            # nearly rank deficient
        U, S, V = oracle.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12 * max(1.0, np.abs(A).max()))
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-10) and np.allclose(V.T @ V, np.eye(3), atol=1e-10)
        assert (S >= 0).all()
        assert np.allclose(np.sort(S), np.sort(np.linalg.svd(A, compute_uv=False)), atol=1e-12 * max(1.0, S.max()))
    U, S, V = oracle.svd3(np.diag([2.0, 2.0, 2.0]))
    assert np.allclose(S, 2.0)
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    U, S, V = oracle.svd3(R)
    assert np.allclose(S, 1.0) and np.allclose(U @ V.T, R)


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K


def test_elastic_stress_closed_forms():
    E, nu = 1.0e7, 0.2
    lam, mu = lame_lambda_mu(E, nu)
    co = CorotatedLinearElasticity(E, nu).to_core_model()
    nh = NeoHookeanElasticity(E, nu).to_core_model()
    I = np.eye(3)
    for cm in (co, nh):
        assert np.abs(oracle.kirchhoff_stress(cm, I, f64=True)).max() == 0.0          # F = I  =>  tau = 0
        assert np.abs(oracle.kirchhoff_stress(cm, _rot([1, 2, 3], 0.7), f64=True)).max() < 1e-7   # rigid rotation
    # corotated, diagonal F: tau = 2 mu (S - I) S + lam (J - 1) J I
    s = np.array([1.1, 0.9, 1.05])
    J = s.prod()
    want = np.diag(2 * mu * (s - 1) * s + lam * (J - 1) * J)
    assert np.allclose(oracle.kirchhoff_stress(co, np.diag(s), f64=True), want, rtol=1e-12)
    assert np.allclose(oracle.kirchhoff_stress(co, np.diag(s), f64=False), want, rtol=2e-5)
    # ... and objectivity: tau(R F) = R tau(F) R^T
    R = _rot([0.3, -1, 2], 1.1)
    assert np.allclose(oracle.kirchhoff_stress(co, R @ np.diag(s), f64=True), R @ want @ R.T, atol=1e-5)
    # neo-Hookean, F = s I: tau = kappa/2 (s^6 - 1) I, kappa = 2/3 mu + lam  
    for sc in (1.05, 0.93):
        kappa = 2.0 / 3.0 * mu + lam
        got = oracle.kirchhoff_stress(nh, sc * I, f64=True)
        assert np.allclose(got, kappa / 2 * (sc ** 6 - 1) * I, rtol=1e-12, atol=1e-6)
    # hardening scales the stress linearly
    assert np.allclose(oracle.kirchhoff_stress(co, np.diag(s), hardening=2.0, f64=True), 2 * want, rtol=1e-12)


def test_monaghan_eos():
    cm = MonaghanSphEos(1.0e6, 7, 1.01e-3).to_core_model()
    I = np.eye(3)
    # J = F00 = 1: p = 0
    assert np.abs(oracle.kirchhoff_stress(cm, I, mass=1.0, volume0=1.0e-3, fluid_j=1.0, f64=True)).max() == 0.0
    # compression F00 = 0.98: p = p0 ((1/F00)^7 - 1), tau = -p * fluid_j * I
    F = I.copy(); F[0, 0] = 0.98
    p = 1.0e6 * ((1 / 0.98) ** 7 - 1)
    assert np.allclose(oracle.kirchhoff_stress(cm, F, mass=2.0, volume0=2.0e-3, fluid_j=1.0, f64=True), -p * I, rtol=1e-12)
    # expansion clamps at -max_neg_pressure = -1
    F[0, 0] = 1.5
    assert np.allclose(oracle.kirchhoff_stress(cm, F, mass=2.0, volume0=2.0e-3, fluid_j=1.0, f64=True), 1.0 * I)
    # viscous part: 2 nu fluid_j dev(sym C)
    F[0, 0] = 1.0
    Cm = np.array([[0.0, 2.0, 0], [0, 0, 0], [0, 0, 0]])
    tau = oracle.kirchhoff_stress(cm, F, mass=1.0, volume0=1.0e-3, fluid_j=1.0, Cmat=Cm, f64=True)
    want = 2 * 1.01e-3 * np.array([[0, 1.0, 0], [1.0, 0, 0], [0, 0, 0]])
    assert np.allclose(tau, want, atol=1e-15)


def test_drucker_prager_return_map():
    dp = DruckerPragerPlasticity(1.0e7, 0.2).to_core_model().u.drucker_prager
    lam, mu = dp.lambda_, dp.mu
    # alpha(q) with the defaults (35, 9, 0.2, 10 degrees)
    q = 1.0
    ang = math.radians(35) + (math.radians(9) * q - math.radians(10)) * math.exp(-0.2 * q)
    assert oracle.dp_alpha(dp, q, f64=True) == pytest.approx(math.sqrt(2 / 3) * 2 * math.sin(ang) / (3 - math.sin(ang)), rel=1e-6)
    alpha = oracle.dp_alpha(dp, q, f64=True)
    # sigma = (1,1,1): zero strain -> "dev == 0" branch: sigma' = 1, dq = 0
    sv, dq = oracle.dp_project(dp, [1, 1, 1], 0.0, alpha, f64=True)
    assert np.allclose(sv, 1.0) and dq == 0.0
    # isotropic expansion: tr > 0 -> sigma' = 1, dq = |eps|
    sv, dq = oracle.dp_project(dp, [1.02, 1.02, 1.02], 0.0, alpha, f64=True)
    assert np.allclose(sv, 1.0) and dq == pytest.approx(math.sqrt(3) * math.log(1.02))
    # compressed + sheared, inside the cone -> None ; outside -> hand-computed gamma
    k = (3 * lam + 2 * mu) / (2 * mu)
    for s in ([0.95, 0.951, 0.949], [1.05, 0.85, 1.0]):
        eps = np.log(s)
        tr = eps.sum()
        dev = eps - tr / 3
        gamma = np.linalg.norm(dev) + k * tr * alpha
        res = oracle.dp_project(dp, s, 0.0, alpha, f64=True)
        if gamma <= 0:
            assert res is None
        else:
            H = eps - gamma * dev / np.linalg.norm(dev)
            assert np.allclose(res[0], np.exp(H), rtol=1e-12) and res[1] == pytest.approx(gamma)
    assert oracle.dp_project(dp, [0.95, 0.951, 0.949], 0.0, alpha, f64=True) is None
    assert oracle.dp_project(dp, [1.05, 0.85, 1.0], 0.0, alpha, f64=True) is not None
    # update_particle on a rotated F keeps U, V and projects S; Jp and log_vol_gain follow the determinants
    R1, R2 = _rot([1, 0, 2], 0.4), _rot([0, 1, 1], -0.9)
    s = np.array([1.05, 0.85, 1.0])
    F = R1 @ np.diag(s) @ R2.T
    F2, Jp, q2, lvg = oracle.dp_update(dp, F, Jp=1.0, q=1.0, lvg=0.0, f64=True)
    s2, dq = oracle.dp_project(dp, s, 0.0, alpha, f64=True)
    assert np.allclose(np.sort(np.linalg.svd(F2, compute_uv=False)), np.sort(s2), rtol=1e-10)
    assert np.allclose(F2, R1 @ np.diag(s2) @ R2.T, atol=1e-10)
    assert Jp == pytest.approx(s.prod() / s2.prod()) and lvg == pytest.approx(math.log(s.prod()) - math.log(s2.prod()))
    assert q2 == pytest.approx(1.0 + dq)


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * K @ K


def test_snow_plasticity_clamps_singular_values():
    """plasticity_snow.rs:26-48: sigma clamped to [1 - 2.5e-2, 1 + 4.5e-3], Jp and the hardening follow."""
    from sparkl_amd.models import SnowPlasticity
    pm = SnowPlasticity().to_core_model()
    U, V = _rot([1, 2, 3], 0.7), _rot([-1, 0.5, 2], -1.1)
    sig = np.array([1.02, 0.99, 0.95])
    F, st = oracle.plastic_update(pm, U @ np.diag(sig) @ V.T, Jp=1.0, f64=True)
    want = np.clip(sig, 1 - float(np.float32(2.5e-2)), 1 + float(np.float32(4.5e-3)))   # the parameters are f32
    assert np.allclose(np.sort(np.linalg.svd(F, compute_uv=False)), np.sort(want), atol=1e-12)
    assert np.allclose(F, U @ np.diag(want) @ V.T, atol=1e-12)
    jp = sig.prod() / want.prod()
    assert st["Jp"] == pytest.approx(jp, rel=1e-12)
    assert st["elastic_hardening"] == pytest.approx(math.exp(10.0 * (1.0 - jp)), rel=1e-12)
    # inside the box: untouched (up to the SVD round trip), hardening = exp(10 (1 - Jp))
    F2, st2 = oracle.plastic_update(pm, U @ np.diag([1.001, 0.999, 1.0]) @ V.T, Jp=0.98, f64=True)
    assert np.allclose(F2, U @ np.diag([1.001, 0.999, 1.0]) @ V.T, atol=1e-12) and st2["Jp"] == pytest.approx(0.98, rel=1e-12)
    assert st2["elastic_hardening"] == pytest.approx(math.exp(10.0 * 0.02), rel=1e-12)


def test_rankine_plasticity_branches():
    """plasticity_rankine.rs:34-78: the four return cases on the principal Hencky strains."""
    from sparkl_amd.models import RankinePlasticity
    rp = RankinePlasticity(1.0e4, 0.25, 50.0, 2.0)
    lam, mu = rp.lambda_, rp.mu
    pm = rp.to_core_model()
    U, V = _rot([0.3, 1, -0.5], 0.4), _rot([1, 1, 0], 0.9)

    def run(eps, hard=1.0):
        F, st = oracle.plastic_update(pm, U @ np.diag(np.exp(eps)) @ V.T, plastic_hardening=hard, f64=True)
        return np.log(np.linalg.svd(F, compute_uv=False)), st["plastic_hardening"], F

    # 1. below the tensile strength: nothing happens
    eps = np.array([0.002, 0.0, -0.001])
    assert lam * eps.sum() + 2 * mu * eps.max() <= 50.0
    e, h, F = run(eps)
    assert np.allclose(np.sort(e), np.sort(eps), atol=1e-12) and h == 1.0
    # 2. only the largest principal stress yields: it is pulled back onto sigma_1 = strength
    eps = np.array([0.02, 0.0, -0.001])
    e, h, F = run(eps)
    e1 = (50.0 - lam * (eps.sum() - eps[0])) / (2 * mu + lam)
    assert np.allclose(np.sort(e), np.sort([e1, 0.0, -0.001]), atol=1e-12)
    assert h == pytest.approx(min(1.0 + 2.0 * abs(eps[0] - e1), 50.0), rel=1e-12)
    assert np.allclose(F, U @ np.diag(np.exp([e1, 0.0, -0.001])) @ V.T, atol=1e-12)
    # 3. two principal stresses yield
    eps = np.array([0.02, 0.019, -0.001])
    e, h, _ = run(eps)
    ne = (50.0 - lam * eps[2]) / (2 * mu + 2 * lam)
    assert np.allclose(np.sort(e), np.sort([ne, ne, -0.001]), atol=1e-12)
    # 4. all three yield: hydrostatic apex
    eps = np.array([0.02, 0.019, 0.018])
    e, h, _ = run(eps)
    assert np.allclose(e, 50.0 / (2 * mu + 3 * lam), atol=1e-12)
    # softening lowers the effective strength: strength - (hardening - 1)
    e, h, _ = run(np.array([0.004, 0.0, 0.0]), hard=31.0)     # effective strength 20
    assert np.sort(e)[-1] == pytest.approx((20.0 - 0.0) / (2 * mu + lam), rel=1e-10)


def test_nacc_plasticity_cases():
    """plasticity_nacc.rs:62-152 against a direct numpy evaluation of the same formulas (d = 3)."""
    from sparkl_amd.models import NaccPlasticity
    n = NaccPlasticity(1.0e5, 0.3, cohesion=0.05, hardening_enabled=True, hardening_factor=1.5, friction_angle=math.radians(35.0))
    pm = n.to_core_model()
    mu, kappa, xi, beta, M = n.mu, n.kappa, 1.5, float(np.float32(0.05)), float(np.float32(n.friction))
    # the friction coefficient of `new` in f32 (plasticity_nacc.rs:38-40)
    sf = math.sin(math.radians(35.0))
    assert M == pytest.approx(math.sqrt(2 / 3) * 2 * sf / (3 - sf) * 3 / math.sqrt(2 / 3), rel=1e-6)
    U, V = _rot([1, 0, 1], 0.3), _rot([0, 1, 1], -0.8)

    def ref(sig, alpha):
        sig = np.asarray(sig, dtype=np.float64)
        sq, tr = sig ** 2, (sig ** 2).sum()
        p0 = kappa * (1e-5 + math.sinh(xi * max(-alpha, 0.0)))
        J = sig.prod()
        s_tr = mu * J ** (-2 / 3) * (sq - tr / 3)
        p_tr = -(kappa / 2 * (J - 1 / J)) * J
        if p_tr > p0:
            Jn = math.sqrt(-2 * p0 / kappa + 1)
            return np.full(3, Jn ** (1 / 3)), alpha + math.log(J / Jn), "tipA"
        if p_tr < -beta * p0:
            Jn = math.sqrt(2 * beta * p0 / kappa + 1)
            return np.full(3, Jn ** (1 / 3)), alpha + math.log(J / Jn), "tipB"
        y0 = (1 + 2 * beta) * 1.5
        y1 = M * M * (p_tr + beta * p0) * (p_tr - p0)
        if y0 * (s_tr ** 2).sum() + y1 < 1e-4:
            return sig, alpha, "inside"
        if p0 > 1e-4 and p_tr < p0 - 1e-4 and p_tr > -beta * p0 + 1e-4:
            p_c = (1 - beta) * p0 / 2
            q_tr = math.sqrt(1.5) * np.linalg.norm(s_tr)
            d = np.array([p_c - p_tr, -q_tr]); d /= np.linalg.norm(d)
            c = M * M * (p_c + beta * p0) * (p_c - p0)
            b = M * M * d[0] * (2 * p_c - p0 + beta * p0)
            a = M * M * d[0] ** 2 + (1 + 2 * beta) * d[1] ** 2
            disc = math.sqrt(b * b - 4 * a * c)
            p1, p2 = p_c + (-b + disc) / (2 * a) * d[0], p_c + (-b - disc) / (2 * a) * d[0]
            p_x = p1 if (p_tr - p_c) * (p1 - p_c) > 0 else p2
            Jx = math.sqrt(abs(-2 * p_x / kappa + 1))
            if Jx > 1e-4:
                alpha += math.log(J / Jx)
        b_e = math.sqrt(-y1 / y0) * (J ** (2 / 3) / mu) * s_tr / np.linalg.norm(s_tr) + tr / 3
        return np.sqrt(b_e), alpha, "surface"

    seen = set()
    for sig, alpha in [((0.9, 0.9, 0.9), -0.01), ((1.05, 1.05, 1.05), -0.01), ((1.0002, 0.9998, 1.0), -0.5), ((0.96, 0.90, 0.88), -0.05),
                       ((0.97, 0.94, 0.91), -0.2), ((1.08, 0.96, 0.93), -0.05)]:
        want, want_alpha, case = ref(sig, alpha)
        seen.add(case)
        F, st = oracle.plastic_update(pm, U @ np.diag(sig) @ V.T, nacc_alpha=alpha, f64=True)
        assert np.allclose(F, U @ np.diag(want) @ V.T, atol=1e-10), case
        assert st["nacc_alpha"] == pytest.approx(want_alpha, rel=1e-10, abs=1e-12), case
    assert seen == {"tipA", "tipB", "inside", "surface"}


def test_maximum_stress_failure():
    """failure_maximum_stress.rs:20-28: principal stress and maximum shear criteria on the symmetric eigenvalues."""
    from sparkl_amd.models import MaximumStressFailure
    fm = MaximumStressFailure(100.0, 40.0).to_core_model().u.maximum_stress
    Q = _rot([1, 2, -1], 1.2)
    assert not oracle.maximum_stress_failed(fm, Q @ np.diag([90.0, 50.0, 20.0]) @ Q.T)       # max 90 <= 100, shear 35 <= 40
    assert oracle.maximum_stress_failed(fm, Q @ np.diag([101.0, 50.0, 30.0]) @ Q.T)          # principal
    assert oracle.maximum_stress_failed(fm, Q @ np.diag([90.0, 50.0, 5.0]) @ Q.T)            # shear 42.5
    assert not oracle.maximum_stress_failed(fm, Q @ np.diag([-500.0, -480.0, -470.0]) @ Q.T)  # compression never fails on max principal


def test_pos_energy():
    co = CorotatedLinearElasticity(1.0e7, 0.2).to_core_model()
    lam, mu = lame_lambda_mu(1.0e7, 0.2)
    s = np.array([1.1, 0.9, 1.05])
    J = s.prod()
    want = mu * ((s - 1).clip(min=0) ** 2).sum() + (lam / 2 * (J - 1) ** 2 if J >= 1 else 0.0)
    assert oracle.pos_energy(co, np.diag(s), f64=True) == pytest.approx(want, rel=1e-12)
    assert oracle.pos_energy(co, np.eye(3), f64=True) == 0.0


def test_timestep_bound_matches_survey_number():
    # E = 1e7, nu = 0.2, rho = 2700, h = 0.2: c = sqrt((K + 4G/3)/rho) ~ 64.15 m/s, dt = 0.9 h / c ~ 2.806e-3
    sc = scenes.sand_column(4, 4, 4)
    o = oracle.Oracle(sc.cell_width, f64=True)
    o.load_scene(sc)
    K, G = 1.0e7 / (3 * (1 - 0.4)), 1.0e7 / (2 * 1.2)
    c = math.sqrt((K + 4 * G / 3) / 2700.0)
    assert c == pytest.approx(64.15, abs=0.01)
    assert o.estimate_dt(1.0) == pytest.approx(0.9 * 0.2 / c, rel=1e-5)
    assert o.estimate_dt(1.0) == pytest.approx(2.806e-3, rel=1e-3)
    assert o.estimate_dt(1.0e-3) == 1.0e-3   # max_dt caps it (timestep_estimator.rs:25)


def test_cuboid_projection():
    c = abi.CuboidColliderC()
    c.half_extents[:] = [1.0, 2.0, 3.0]
    c.rotation[:] = [0, 0, 0, 1]
    c.translation[:] = [10.0, 0, 0]
    pt, inside = oracle.cuboid_project(c, [12.5, 0.5, 0.0])       # outside: clamp
    assert not inside and np.allclose(pt, [11.0, 0.5, 0.0])
    pt, inside = oracle.cuboid_project(c, [10.9, 0.0, 0.0])       # inside: closest face (solid = false)
    assert inside and np.allclose(pt, [11.0, 0.0, 0.0])
    pt, inside = oracle.cuboid_project(c, [10.0, -1.9, 0.0])
    assert inside and np.allclose(pt, [10.0, -2.0, 0.0])
    # rotated 90 degrees about z: local x extent maps to world y
    c.rotation[:] = [0, 0, math.sin(math.pi / 4), math.cos(math.pi / 4)]
    pt, inside = oracle.cuboid_project(c, [10.0, 1.5, 0.0])
    assert not inside and np.allclose(pt, [10.0, 1.0, 0.0], atol=1e-6)


def test_heightfield_projection():
    """parry HeightField::project_local_point: brute force over the triangles (p00,p10,p01),(p10,p11,p01) of every cell."""
    # a tilted plane y = 0.25 x' + 0.5 z' (x', z' in [0, 1]) -> every triangle is coplanar, so the answer is the plane projection
    n = 9
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")   # i: row (z), j: column (x)
    heights = 0.25 * j / (n - 1) + 0.5 * i / (n - 1)
    hf = abi.heightfield(heights, (4.0, 2.0, 8.0), translation=(1.0, -1.0, 3.0))
    # local surface: y = 2 (0.25 (x/4 + 0.5) + 0.5 (z/8 + 0.5)) = 0.125 x + 0.125 z + 0.75
    nrm = np.array([-0.125, 1.0, -0.125]); nrm /= np.linalg.norm(nrm)
    rng = np.random.default_rng(3)
    for _ in range(50):
        lp = rng.uniform([-1.5, -1.0, -3.0], [1.5, 3.0, 3.0])
        d = (lp[1] - (0.125 * lp[0] + 0.125 * lp[2] + 0.75)) * nrm[1]
        want = lp - d * nrm
        if abs(want[0]) > 2.0 or abs(want[2]) > 4.0:
            continue
        pt, inside = oracle.heightfield_project(hf, lp + [1.0, -1.0, 3.0])
        assert not inside                                          # CPU path: a heightfield has no inside
        assert np.allclose(pt, want + [1.0, -1.0, 3.0], atol=1e-5)
    # off the side: closest point is on the border edge
    pt, _ = oracle.heightfield_project(hf, [1.0 + 5.0, 0.0, 3.0])
    assert pt[0] == pytest.approx(3.0, abs=1e-6)
    # the triangle split matters on a saddle cell: h00 = h11 = 0, h10 = h01 = 1 ; diagonal (p10, p01) is the ridge
    sad = abi.heightfield([[0.0, 1.0], [1.0, 0.0]], (1.0, 1.0, 1.0))
    pt, _ = oracle.heightfield_project(sad, [0.0, 5.0, 0.0])       # straight above the centre -> ridge at y = 1
    assert np.allclose(pt, [0.0, 1.0, 0.0], atol=1e-6)
    # below_is_solid (CUDA path rule): under the surface and inside the footprint
    hs = abi.heightfield(heights, (4.0, 2.0, 8.0), translation=(1.0, -1.0, 3.0), below_is_solid=True)
    assert oracle.heightfield_project(hs, [1.0, -3.0, 3.0])[1] and not oracle.heightfield_project(hs, [1.0, 3.0, 3.0])[1]
    assert not oracle.heightfield_project(hs, [9.0, -3.0, 3.0])[1]


def test_trimesh_projection_matches_the_box_it_triangulates():
    """parry TriMesh::project_local_point on a closed oriented box mesh == Cuboid::project_point (clamp outside, closest face
    inside); without ORIENTED the mesh is a two-sided surface (is_inside never set); flip_interior inverts."""
    he = [1.0, 2.0, 1.5]
    v, i = scenes.box_mesh(he)
    q = [0.0, 0.0, math.sin(0.3), math.cos(0.3)]
    t = [3.0, -1.0, 0.5]
    box = abi.CuboidColliderC()
    box.half_extents[:] = he
    box.rotation[:] = q
    box.translation[:] = t
    solid = abi.trimesh(v, i, translation=t, rotation=q, oriented=True)
    sheet = abi.trimesh(v, i, translation=t, rotation=q, oriented=False)
    flipped = abi.trimesh(v, i, translation=t, rotation=q, oriented=True, flip_interior=True)
    rng = np.random.default_rng(11)
    n_in = 0
    for _ in range(300):
        pt = rng.uniform(-3.0, 3.0, 3) + t
        want, want_inside = oracle.cuboid_project(box, pt)
        got, inside = oracle.trimesh_project(solid, pt)
        assert inside == want_inside
        assert np.linalg.norm(got - pt) == pytest.approx(np.linalg.norm(want - pt), abs=1e-9)   # the closest point may be another face at equal distance
        if np.linalg.norm(want - pt) > 1e-3 and not want_inside:
            assert np.allclose(got, want, atol=1e-9)
        got2, inside2 = oracle.trimesh_project(sheet, pt)
        assert not inside2 and np.allclose(got2, got, atol=1e-12)
        assert oracle.trimesh_project(flipped, pt)[1] == (not want_inside)
        n_in += want_inside
    assert 5 < n_in < 295
    # closest feature is a vertex / an edge of a convex corner: the pseudo-normals still say "outside"
    for local in ([1.5, 2.5, 2.0], [1.5, 2.5, 0.0], [-1.2, 0.3, -1.9]):
        c, s_ = math.cos(0.6), math.sin(0.6)
        world = np.array([c * local[0] - s_ * local[1], s_ * local[0] + c * local[1], local[2]]) + t
        assert not oracle.trimesh_project(solid, world)[1]


def test_heightfield_scene_sand_comes_to_rest_on_the_surface():
    sc = scenes.sand_on_heightfield(6, 4, 6)
    o = oracle.Oracle(sc.cell_width)
    o.load_scene(sc)
    y0 = sc.particles.position["point"][:, 1].min()
    for _ in range(600):
        o.substep_fixed(sc.params, sc.gravity, 5.0e-4)
    c = o.get_particles()
    surf = oracle.heightfield_project(sc.heightfields[0], [float(c.position["point"][:, 0].mean()), 0.0, float(c.position["point"][:, 2].mean())])[0][1]
    assert c.position["point"][:, 1].min() < y0 - 0.05           # it fell (contact starts one cell above the surface) ...
    assert c.position["point"][:, 1].min() > surf - 0.5 * sc.cell_width   # ... and did not go through
    assert c.status["failed"].sum() == 0


def _elastic_cloud(n=14):
    models = ParticleModelSet()
    mid = models.insert(ParticleModel.new(NeoHookeanElasticity(1.0e6, 0.3)))
    h, r = 0.2, 0.05
    pos = scenes.cube_positions([1.0, 1.0, 1.0], n, n, n, r, jitter=True)
    return h, models, new_particles(mid, pos, r, 1000.0)


def test_p2g_conserves_mass_and_momentum_and_apic_reproduces_affine_fields():
    h, models, comps = _elastic_cloud()
    x = comps.position["point"].astype(np.float64)
    c = x.mean(0)
    A = np.array([[0.1, -0.3, 0.2], [0.4, 0.05, -0.1], [-0.2, 0.3, -0.15]])   # uniform affine velocity field
    v0 = np.array([0.3, -0.2, 0.5])
    comps.velocity["vector"] = (v0 + (x - c) @ A.T).astype(np.float32)
    comps.velocity_gradient["m"] = np.tile(A.T.reshape(9), (len(comps), 1)).astype(np.float32)  # column-major
    o = oracle.Oracle(h, f64=True)
    o.set_models(models); o.set_cuboids([]); o.set_particles(comps)
    o.sort(); o.p2g(1.0e-4, (0, 0, 0))
    nodes = o.nodes()
    m = comps.volume["mass"].astype(np.float64)
    assert nodes["mass"].astype(np.float64).sum() == pytest.approx(m.sum(), rel=1e-6)
    pmom = (m[:, None] * comps.velocity["vector"].astype(np.float64)).sum(0)
    assert np.allclose(nodes["momentum"].astype(np.float64).sum(0), pmom, rtol=1e-5, atol=1e-7)
    # APIC: the grid carries the affine field exactly, so interior nodes have v(x_n) = v0 + A (x_n - c) ...
    xn = nodes["cell"].astype(np.float64) * h
    vn = v0 + (xn - c) @ A.T
    heavy = nodes["mass"] > 0.5 * nodes["mass"].max()
    assert np.allclose(nodes["velocity"][heavy], vn[heavy], atol=2e-5)
    # ... and G2P returns v and C of interior particles unchanged (F = I: no stress)
    o.grid_update(1.0e-4, abi.BOUNDARY_NONE, 0)
    o.g2p(1.0e-4)
    got = o.get_particles()
    lo, hi = x.min(0) + 1.2 * h, x.max(0) - 1.2 * h
    interior = ((x > lo) & (x < hi)).all(1)
    assert interior.sum() > 10
    assert np.allclose(got.velocity["vector"][interior], comps.velocity["vector"][interior], atol=3e-5)
    assert np.allclose(got.velocity_gradient["m"][interior], comps.velocity_gradient["m"][interior], atol=3e-4)


def test_rigid_translation_is_exact_to_rounding():
    h, models, comps = _elastic_cloud(5)
    comps.velocity["vector"][:] = np.array([0.5, -0.25, 0.125], np.float32)
    o = oracle.Oracle(h)
    o.set_models(models); o.set_cuboids([]); o.set_particles(comps)
    params = abi.SolverParameters(boundary_handling=abi.BOUNDARY_NONE)
    for _ in range(5):
        o.substep_fixed(params, (0, 0, 0), 1.0e-3)
    got = o.get_particles()
    assert np.allclose(got.velocity["vector"], [0.5, -0.25, 0.125], atol=2e-6)
    assert np.allclose(got.position["point"], comps.position["point"] + 5.0e-3 * np.array([0.5, -0.25, 0.125]), atol=2e-6)
    assert np.abs(got.volume["deformation_gradient"] - np.eye(3).reshape(9)).max() < 2e-6


def test_step_driver_substep_rule():
    # mls_solver.rs:33-59,125: dt clamped up to dt/max_num_substeps while more than that remains; frame of 1/60 s -> 6 substeps
    sc = scenes.sand_column(6, 6, 6)
    o = oracle.Oracle(sc.cell_width)
    o.load_scene(sc)
    t = o.step(sc.params, sc.gravity)
    dts = o.last_dts()
    assert t == pytest.approx(1 / 60, rel=1e-6) and len(dts) == 6
    assert dts[:5] == pytest.approx([2.806e-3] * 5, rel=1e-3) and dts.sum() == pytest.approx(1 / 60, rel=1e-5)
    p = abi.SolverParameters(stop_after_one_substep=1)
    assert o.step(p, sc.gravity) == pytest.approx(dts[0], rel=1e-3)


def test_modified_eigenerosion_threshold_kat():
    """grid_to_particle.rs:57-77: with the same psi_pos = c on every particle the node ratio psi_momentum / psi_mass is c at every
    node that received anything, the gather (partition of unity) returns c, and crack_energy = factor * h * c: a threshold just
    below cracks every particle, one just above none."""
    from oracle import Oracle
    from sparkl_amd import abi, scenes
    sc = scenes.elastic_cube(6)
    sc.particles.velocity["vector"][:] = 0
    c, h = 3.0, sc.cell_width
    sc.particles.phase["psi_pos"][:] = c
    sc.params.damage_model = abi.DAMAGE_MODIFIED_EIGENEROSION
    n = len(sc.particles)
    for factor, thr, want in ((2.0, 2.0 * h * c * 0.999, 0.0), (2.0, 2.0 * h * c * 1.001, 1.0), (0.0, 0.0, 1.0)):
        o = Oracle(sc.cell_width, f64=True)
        o.load_scene(sc)
        o.set_fracture(np.tile(np.float32([factor, thr]), (n, 1)))
        o.substep_fixed(sc.params, sc.gravity, 1.0e-5)
        ph = o.get_particles().phase["phase"]
        o.close()
        assert (ph == want).all(), (factor, thr, np.unique(ph))


def test_parallel_stable_sort_is_the_sequential_stable_order():
    """ParticleSet::sort uses rayon's `par_sort_by_key` (particle_set.rs:142), a STABLE sort: the accumulation order of P2G follows from
    it.  The oracle's parallel merge sort (co-ranked merges) must give exactly numpy's stable argsort, ties included, at every thread
    count and for sizes around the chunk boundaries."""
    import ctypes as C
    from oracle import lib
    L = lib()
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 4096 * 2 - 1, 4096 * 8 + 3, 200_003):
        for nkeys in (3, 1000, 1 << 40):
            keys = rng.integers(0, nkeys, size=n, dtype=np.uint64)
            want = np.argsort(keys, kind="stable").astype(np.uint64)
            for threads in (1, 2, 3, 8, 16):
                out = np.empty(n, dtype=np.uint64)
                L.oracle_debug_parallel_sort(keys.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_int(threads), out.ctypes.data_as(C.c_void_p))
                assert np.array_equal(out, want), (n, nkeys, threads)
