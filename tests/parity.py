"""Parity metrics of SURVEY 8(d): particles matched by caller index."""
import numpy as np


def particle_errors(got, ref, h):
    """eps_x = max |dx| / h ; eps_v = max |dv| / max |v_ref| ; eps_F = max Frobenius diff."""
    dx = np.linalg.norm(got.position["point"].astype(np.float64) - ref.position["point"].astype(np.float64), axis=1)
    dv = np.linalg.norm(got.velocity["vector"].astype(np.float64) - ref.velocity["vector"].astype(np.float64), axis=1)
    vmax = max(float(np.linalg.norm(ref.velocity["vector"].astype(np.float64), axis=1).max()), 1e-30)
    dF = np.linalg.norm(got.volume["deformation_gradient"].astype(np.float64) - ref.volume["deformation_gradient"].astype(np.float64), axis=1)
    dC = np.linalg.norm(got.velocity_gradient["m"].astype(np.float64) - ref.velocity_gradient["m"].astype(np.float64), axis=1)
    cmax = max(float(np.linalg.norm(ref.velocity_gradient["m"].astype(np.float64), axis=1).max()), 1e-30)
    return {
        "eps_x": float(dx.max() / h), "eps_x_mean": float(dx.mean() / h), "eps_x_p999": float(np.quantile(dx, 0.999) / h),
        "eps_v": float(dv.max() / vmax), "eps_v_mean": float(dv.mean() / vmax), "eps_v_p999": float(np.quantile(dv, 0.999) / vmax),
        "eps_F": float(dF.max()), "eps_C": float(dC.max() / cmax), "vmax": vmax,
        "failed_mismatch": int((got.status["failed"] != ref.status["failed"]).sum()),
    }


TOL = 1.0e-4   # BASELINE.json north_star: <= 1e-4 relative particle-state divergence over 100 steps


def parity_verdict(got, ref32, ref64, h):
    """The parity gate.  The arbiter is the oracle evaluated in f64 (the exact trajectory of the reference's algorithm):

        eps(HIP vs f64 oracle)  <=  1e-4      for positions (in h) and for velocities (relative to max |v|), no exceptions.

    The reference's own f32 arithmetic (the f32 oracle) does NOT meet this on the stiff sand cases (3e-4 .. 6e-4: a singular value
    of F in f32 carries 1e-7 of absolute error against a strain of 1e-4 .. 1e-3); its distance from the f64 trajectory is
    reported beside ours (`oracle_f32_vs_f64`) but is no longer part of the gate: the HIP path works on F - I (strain-form
    arithmetic, sparkl_amd/csrc/device_math.h) and sits at 1e-6 .. 3e-5 in every case of the suite.
    Returns the three error dictionaries (max, mean and p99.9 in each) and a list of violations."""
    vs32 = particle_errors(got, ref32, h)
    vs64 = particle_errors(got, ref64, h)
    floor = particle_errors(ref32, ref64, h)
    bad = []
    if vs64["failed_mismatch"] or vs32["failed_mismatch"]:
        bad.append(f"failed flags differ ({vs32['failed_mismatch']} vs f32, {vs64['failed_mismatch']} vs f64)")
    for k in ("eps_x", "eps_v"):
        if not vs64[k] <= TOL:
            bad.append(f"{k} vs f64 oracle {vs64[k]:.3e} > 1e-4 (f32 oracle vs f64: {floor[k]:.3e})")
    return {"vs_f32": vs32, "vs_f64": vs64, "oracle_f32_vs_f64": floor}, bad


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
