"""Parity metrics of SURVEY 8(d): particles matched by caller index."""
import numpy as np


def particle_errors(got, ref, h):
    """eps_x = max |dx| / h ; eps_v = max |dv| / max |v_ref| ; eps_F = max Frobenius diff."""
    dx = np.linalg.norm(got.position["point"].astype(np.float64) - ref.position["point"].astype(np.float64), axis=1)
    dv = np.linalg.norm(got.velocity["vector"].astype(np.float64) - ref.velocity["vector"].astype(np.float64), axis=1)
    vmax = max(float(np.linalg.norm(ref.velocity["vector"].astype(np.float64), axis=1).max()), 1e-30)
    dF = np.linalg.norm(got.volume["deformation_gradient"].astype(np.float64) - ref.volume["deformation_gradient"].astype(np.float64), axis=1)
    dC = np.linalg.norm(got.velocity_gradient["m"].astype(np.float64) - ref.velocity_gradient["m"].astype(np.float64), axis=1)
    # C is a velocity gradient: C h is a velocity difference across one cell, so its scale is never below vmax / h (a body in free fall has
    # C = 0 to rounding, and "relative to max |C|" would then measure rounding noise against rounding noise)
    cmax = max(float(np.linalg.norm(ref.velocity_gradient["m"].astype(np.float64), axis=1).max()), vmax / h)
    out = {
        "eps_x": float(dx.max() / h), "eps_x_mean": float(dx.mean() / h), "eps_x_p999": float(np.quantile(dx, 0.999) / h),
        "eps_v": float(dv.max() / vmax), "eps_v_mean": float(dv.mean() / vmax), "eps_v_p999": float(np.quantile(dv, 0.999) / vmax),
        "eps_F": float(dF.max()), "eps_C": float(dC.max() / cmax), "vmax": vmax,
        "failed_mismatch": int((got.status["failed"] != ref.status["failed"]).sum()),
    }
    # the per-particle scalars the models carry (round 6: gated, not only logged).  Absolute where the quantity is O(1) by construction
    # (hardening factors, log-volume gain, det of the plastic part, NACC alpha: divided by max(1, max |ref|) so that a scene that drives one
    # of them far from 1 is still held to a RELATIVE 1e-4); psi_pos is an energy density (its scale is the scene's: E times strain squared)
    # and is taken relative to its maximum over the particles.
    def scal(a, b, floor):
        a = a.astype(np.float64); b = b.astype(np.float64)
        return float(np.abs(a - b).max() / max(float(np.abs(b).max()), floor)) if len(a) else 0.0
    out["eps_hard"] = scal(got.data["plastic_hardening"], ref.data["plastic_hardening"], 1.0)
    out["eps_ehard"] = scal(got.data["elastic_hardening"], ref.data["elastic_hardening"], 1.0)
    out["eps_lvg"] = scal(got.data["log_vol_gain"], ref.data["log_vol_gain"], 1.0)
    out["eps_nacc"] = scal(got.data["nacc_alpha"], ref.data["nacc_alpha"], 1.0)
    out["eps_Jp"] = scal(got.volume["plastic_deformation_gradient_det"], ref.volume["plastic_deformation_gradient_det"], 1.0)
    # psi_pos is the running maximum of the positive elastic energy DENSITY: relative to its maximum over the particles, and never below the
    # kinetic energy density of the scene, rho vmax^2 / 2 (a lattice at rest holds psi_pos = 0 to rounding; the scale of an energy density
    # that matters to anything in a moving scene is that of its motion)
    rho = (ref.volume["mass"].astype(np.float64) / np.maximum(ref.volume["volume0"].astype(np.float64), 1e-300))
    out["eps_psi"] = scal(got.phase["psi_pos"], ref.phase["psi_pos"], max(0.5 * float(rho.max() if len(rho) else 0.0) * vmax * vmax, 1e-30))
    out["phase_mismatch"] = int((got.phase["phase"] != ref.phase["phase"]).sum())
    out["phase_zero"] = int((ref.phase["phase"] == 0).sum())
    return out


# every quantity of the gate: the two kinematic ones of north_star, then what round 6 added
GATED = ("eps_x", "eps_v", "eps_F", "eps_C", "eps_hard", "eps_ehard", "eps_lvg", "eps_nacc", "eps_Jp", "eps_psi")
# psi_pos is quadratic in the strain: a relative strain error of TOL is a relative energy error of 2 TOL
TOL_FACTOR = {"eps_psi": 2.0}


def save_components(path, comps, **extra):
    """Every component array of a ParticleComponents (one npz key per slot) + whatever else the caller wants to keep: what a slab rank hands to
    the test that gates the UNION of the ranks (the gate reads every component since round 6)."""
    np.savez(path, **{f"comp_{k}": getattr(comps, k) for k in comps.__slots__}, **extra)


def union_components(parts, caller):
    """The ranks' component arrays (save_components) in caller order; `caller` = caller index of every concatenated record."""
    from sparkl_amd.particles import ParticleComponents
    inv = np.argsort(caller)
    got = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        setattr(got, k, np.concatenate([q[f"comp_{k}"] for q in parts])[inv])
    return got


TOL = 1.0e-4   # BASELINE.json north_star: <= 1e-4 relative particle-state divergence over 100 steps


TOLERANCE_STATEMENT = ("arbiter = the reference's algorithm evaluated in f64 (oracle/): eps_x (in h) and eps_v (relative to max |v|) of the HIP path "
                       "vs that trajectory <= 1e-4, at every size, no storage allowance (positions are held relative to the home block since round 5); "
                       "vs the reference's own f32 arithmetic (f32 oracle): eps <= eps(f32 oracle vs f64 oracle) + 1e-4 for positions and velocities alike -- "
                       "the f32 oracle itself sits 1e-4..6e-4 from the f64 trajectory on stiff sand and 3.9e-4 in eps_x on the 10.8 m cube (f32 world coordinates); "
                       "2 M-particle fluid: eps_v vs the f32 oracle (the f32 reference is 4.6e-3 from f64 there); "
                       "since round 6 the same two bounds are asserted for F (max Frobenius distance), C (relative to max |C|), plastic / elastic hardening, "
                       "log_vol_gain, det Fp, nacc_alpha (relative to max(1, max |ref|)) and psi_pos (relative to its maximum), and the phase flags must agree")


def parity_verdict(got, ref32, ref64, h, arbiter="f64", spread=()):
    """The parity gate.  The arbiter is the oracle evaluated in f64 (the exact trajectory of the reference's algorithm):

        eps(HIP vs f64 oracle)  <=  1e-4      for positions (in h) and for velocities (relative to max |v|), no exceptions.

    This arbiter is the BUILDER's choice (TOLERANCE_STATEMENT, repeated in bench.py's JSON line and the README): the reference's
    own f32 arithmetic (the f32 oracle) does not meet 1e-4 against it on the stiff sand cases (3e-4 .. 6e-4 in eps_v: a singular
    value of F in f32 carries 1e-7 of absolute error against a strain of 1e-4 .. 1e-3), while the HIP path works on F - I
    (strain-form arithmetic, sparkl_amd/csrc/device_math.h) and sits at 1e-6 .. 3e-5.  So that drifting away from the reference's
    arithmetic still fails a test (a bug shared by the HIP kernels and the f64 restatement would otherwise pass), the distance to
    the f32 oracle is gated as well, for positions and velocities alike:

        eps(HIP vs f32 oracle)  <=  eps(f32 oracle vs f64 oracle) + 1e-4      (the triangle through the arbiter, no slack beyond it)

    Positions (round 5): the particle streams hold positions relative to the home block, in cells (sparkl_amd/csrc/kernels.h,
    cell_frac), so the HIP path no longer loses what `nsub` additions x += v dt lose in an f32 WORLD coordinate -- the reference's own
    arithmetic does (3.9e-4 h on BASELINE configs[1]: coordinates up to 10.8 m, h = 0.2, 100 substeps), which is why the position bound
    against the f32 oracle is the triangle and not the plain 1e-4.  The `x_storage_steps` allowance of rounds 3 - 4 is gone: the bound
    against the arbiter is the plain 1e-4 at every size.

    `arbiter="either"` (the 2 M-particle fluid case only): the velocity bound holds against the f64 OR the f32 oracle.  A weakly
    compressible fluid turns the f32 rounding of the node masses (1e-7) into pressure noise (p0 gamma = 7e6 times that), and after
    100 substeps of the 22 m deep dam the reference's own f32 arithmetic is 4.6e-3 from its f64 trajectory in eps_v; the HIP path
    shares that arithmetic (7.7e-5 from the f32 oracle) and no f32 implementation can sit within 1e-4 of f64 there.  The bound
    eps_v(HIP vs f32) <= eps_v(f32 vs f64) + 1e-4 stays in force.

    Round 6: the same two bounds hold for EVERY quantity in `GATED` -- the deformation gradient (max Frobenius distance), the APIC matrix
    (relative to max |C|), plastic / elastic hardening, log-volume gain, det of the plastic deformation gradient, NACC alpha, psi_pos
    (relative to its maximum) -- and the phase flags must agree (see particle_errors).  Rounds 1 - 5 computed eps_F and eps_C and asserted neither.

    (Round 6: in this mode the quantities added in round 6 are held to the `spread` bound below -- C, the most sensitive of them, sits 1e-3 from
    f64 in the reference's own f32 arithmetic on the small two-slab dam, 1.9e-4 here.)

    `spread` (round 6; the snow / NACC tests only, each naming its quantities): quantities whose bound against the f64 arbiter is
    eps(HIP vs f64) <= 2 eps(f32 oracle vs f64 oracle) + tol instead of the plain tol.  Those models run the reference's own F-form f32 arithmetic
    here too (no strain form), and a yield surface that snaps amplifies rounding: where the reference's f32 trajectory itself sits 1e-3 .. 6e-3 from
    its f64 trajectory no f32 implementation can sit within 1e-4 of f64.  The bound against the f32 oracle (the triangle) stays as it is.

    Returns the three error dictionaries (max, mean and p99.9 in each) and a list of violations."""
    vs32 = particle_errors(got, ref32, h)
    vs64 = particle_errors(got, ref64, h)
    floor = particle_errors(ref32, ref64, h)
    bad = []
    if vs64["failed_mismatch"] or vs32["failed_mismatch"]:
        bad.append(f"failed flags differ ({vs32['failed_mismatch']} vs f32, {vs64['failed_mismatch']} vs f64)")
    # phase is 0 or 1 (failure models, eigenerosion): a particle that cracked in one trajectory and not in the other is a count, not a distance.
    # Where the two oracles disagree among themselves (a threshold crossed within f32 rounding) the HIP path may side with either.
    # A particle sitting exactly on the threshold may flip in any arithmetic: the bulk must agree (1 % of the cracked particles, at least 2).
    flips = floor["phase_mismatch"] + max(2, vs64["phase_zero"] // 100)
    if vs64["phase_mismatch"] > flips and vs32["phase_mismatch"] > flips:
        bad.append(f"phase differs on {vs32['phase_mismatch']} particles vs f32, {vs64['phase_mismatch']} vs f64 (the oracles differ on {floor['phase_mismatch']})")
    for k in GATED:
        tol = TOL * TOL_FACTOR.get(k, 1.0)
        tol64 = tol + (2.0 * floor[k] if (k in spread or (arbiter == "either" and k not in ("eps_x", "eps_v"))) else 0.0)
        if not vs64[k] <= tol64 and not (arbiter == "either" and vs32[k] <= tol):
            bad.append(f"{k} vs f64 oracle {vs64[k]:.3e} > {tol64:.3e} (f32 oracle vs f64: {floor[k]:.3e}; vs f32 oracle: {vs32[k]:.3e})")
        if not vs32[k] <= floor[k] + tol:
            bad.append(f"{k} vs f32 oracle {vs32[k]:.3e} > {k}(f32 vs f64 oracle) {floor[k]:.3e} + {tol:.0e}")
    return {"vs_f32": vs32, "vs_f64": vs64, "oracle_f32_vs_f64": floor, "tol_x": TOL}, bad


def node_errors(got, ref):
    """Match sparse node lists by integer cell; relative errors on mass / momentum / velocity."""
    def key(a):
        c = a["cell"].astype(np.int64)
        return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    kg, kr = key(got), key(ref)
    og, orr = np.argsort(kg), np.argsort(kr)
    got, ref, kg, kr = got[og], ref[orr], kg[og], kr[orr]
    common, ig, ir = np.intersect1d(kg, kr, return_indices=True)
    out = {"n_got": len(got), "n_ref": len(ref), "n_common": len(common)}
    g, r = got[ig], ref[ir]
    out["mass_rel"] = float(np.abs(g["mass"].astype(np.float64) - r["mass"]).max() / np.abs(r["mass"]).max())
    pm = max(float(np.abs(r["momentum"]).max()), 1e-30)
    out["momentum_rel"] = float(np.abs(g["momentum"].astype(np.float64) - r["momentum"]).max() / pm)
    vm = max(float(np.abs(r["velocity"]).max()), 1e-30)
    out["velocity_rel"] = float(np.abs(g["velocity"].astype(np.float64) - r["velocity"]).max() / vm)
    return out
