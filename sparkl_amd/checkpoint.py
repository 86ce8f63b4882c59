"""Checkpoint / serialisation of particle + model state.

Two formats:

* **native** (`save` / `load`): one `.npz` holding the seven SoA component arrays exactly as the C ABI exchanges them,
  the `SparklParticleModel` array as raw bytes and a JSON header (cell width, solver parameters, simulated time).
  Loss-free at the precision of the ABI (the deformation gradient crosses it as f32 F; the device keeps F - I, which is
  rounded once when F is formed for the read-back); `load` + `MpmPipeline.upload_particles` resumes a run (the sorted
  order is rebuilt by the next re-bin, so the continuation matches an uninterrupted run to f32 rounding, not bit for bit).

* **sparkl serde layout** (`dump_bincode` / `load_bincode`): what `bincode 1` (default options: little endian,
  fixed-width integers, u64 sequence lengths, u32 enum variant index, u8 `Option` tag) writes for the reference's
  `#[derive(Serialize)]` types --
      `Vec<Particle>`                    src/dynamics/particle.rs:5-51   (field order as declared)
      `Vec<ParticleModel>` as tuples     src/dynamics/particle_model.rs:20-40
         `(CoreConstitutiveModel, Option<CorePlasticModel>, Option<CoreFailureModel>)`
  with nalgebra 0.32's serde for fixed-size storage (`ArrayStorage` = `serialize_seq(Some(R * C))`, i.e. a u64 length
  followed by the column-major elements) and rapier's arena `Index { index: u32, generation: u32 }` as the model handle.
  **UNVERIFIED AGAINST SPARKL.**  No byte of this format has ever been produced or read by a real sparkl build: there is no
  rustc in this image, the crates are absent, and the reference holds no serialized fixture.  The layout is restated from
  the crates' published serde implementations and only round-trip tested against itself here; treat a file written by
  `dump_bincode` as a proposal to be checked against `bincode::serialize(&particles)` on a machine with the reference
  toolchain before relying on it.  The native format above is the one the resume tests exercise (against the oracle).
"""
import io
import json
import struct

import numpy as np

from . import abi
from .particles import ParticleComponents

NATIVE_VERSION = 2      # 2: carries `has_fracture` / the ParticleFracture component; version 1 files (none) still load
_LOADABLE_VERSIONS = (1, 2)


# ------------------------------------------------------------------------------------------------ native
def save(path, components, model_set_or_array, cell_width, params=None, simulated_time=0.0, extra=None, fracture=None):
    """`fracture`: the optional `ParticleFracture` component (abi.FRACTURE_DTYPE, `MpmPipeline.read_fracture()`), in the particles'
    order.  `DamageModel::ModifiedEigenerosion` only runs its crack sweep while that component is uploaded, so a checkpoint of such
    a run WITHOUT it would resume with damage silently off: refused."""
    models = model_set_or_array.to_c_array() if hasattr(model_set_or_array, "to_c_array") else model_set_or_array
    n_models = len(model_set_or_array)
    if fracture is None and params is not None and int(params.damage_model) == abi.DAMAGE_MODIFIED_EIGENEROSION:
        raise ValueError("checkpoint.save: params.damage_model is ModifiedEigenerosion but no `fracture` component was given "
                         "(pass MpmPipeline.read_fracture(): the crack sweep needs it after the resume)")
    header = {"format": "sparkl_amd-checkpoint", "version": NATIVE_VERSION, "cell_width": float(cell_width), "n_models": int(n_models),
              "simulated_time": float(simulated_time), "extra": extra or {}, "has_fracture": fracture is not None}
    arrays = {k: getattr(components, k) for k in ParticleComponents.__slots__}
    if fracture is not None:
        fracture = np.ascontiguousarray(fracture, dtype=abi.FRACTURE_DTYPE)
        if len(fracture) != len(components):
            raise ValueError(f"checkpoint.save: {len(fracture)} fracture records for {len(components)} particles")
        arrays["fracture"] = fracture
    arrays["models"] = np.frombuffer(bytes(models), dtype=np.uint8)[: n_models * abi.C.sizeof(abi.ParticleModelC)].copy()
    if params is not None:
        arrays["params"] = np.frombuffer(bytes(params), dtype=np.uint8).copy()
    arrays["header"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
    with open(path, "wb") as f:
        np.savez(f, **arrays)


class ModelArray:
    """A loaded `SparklParticleModel[]`: quacks like ParticleModelSet for `MpmPipeline.set_models` / `Oracle.set_models`."""

    def __init__(self, arr, n):
        self.arr, self.n = arr, n

    def __len__(self):
        return self.n

    def to_c_array(self):
        return self.arr


def load(path):
    """-> (ParticleComponents, ModelArray, header dict, SolverParameters or None).  header["fracture"] is the `ParticleFracture`
    component (abi.FRACTURE_DTYPE array) when the checkpoint carries one, else None: hand it to `MpmPipeline.set_fracture` after
    `upload_particles` (`resume` below does)."""
    z = np.load(path)
    header = json.loads(bytes(z["header"]).decode())
    if header.get("format") != "sparkl_amd-checkpoint" or header.get("version") not in _LOADABLE_VERSIONS:
        raise ValueError(f"{path}: not a sparkl_amd checkpoint of version {_LOADABLE_VERSIONS}")
    c = ParticleComponents(0)
    for k in ParticleComponents.__slots__:
        a = z[k]
        want = getattr(ParticleComponents(0), k).dtype
        if a.dtype != want:
            raise ValueError(f"{path}: component {k} has dtype {a.dtype}, expected {want}")
        setattr(c, k, a)
    header["fracture"] = None
    if header.get("has_fracture"):
        fr = z["fracture"]
        if fr.dtype != abi.FRACTURE_DTYPE or len(fr) != len(c):
            raise ValueError(f"{path}: fracture component has dtype {fr.dtype} / length {len(fr)}, expected {abi.FRACTURE_DTYPE} / {len(c)}")
        header["fracture"] = fr
    n = header["n_models"]
    arr = (abi.ParticleModelC * max(n, 1))()
    raw = bytes(z["models"])
    abi.C.memmove(arr, raw, len(raw))
    params = None
    if "params" in z:
        params = abi.SolverParameters()
        abi.C.memmove(abi.C.byref(params), bytes(z["params"]), abi.C.sizeof(params))
    return c, ModelArray(arr, n), header, params


def save_pipeline(path, pipe, model_set_or_array, params=None, simulated_time=0.0, extra=None):
    """Checkpoint of a live `MpmPipeline`: particle state, and the `ParticleFracture` component if the caller had uploaded one."""
    # the component travels whenever the pipeline HOLDS one, whatever its values (an all-default upload is state too: it selects the unfused
    # schedule and is what ModifiedEigenerosion resumes with); a pipeline that never got one saves none -- under ModifiedEigenerosion the
    # defaults are then written out explicitly, so that the resumed run sweeps cracks exactly like the original
    fr = pipe.read_fracture() if pipe.has_fracture() else None
    if fr is None and params is not None and int(params.damage_model) == abi.DAMAGE_MODIFIED_EIGENEROSION:
        fr = pipe.read_fracture()
    save(path, pipe.read_particles(), model_set_or_array, pipe.cell_width, params=params, simulated_time=simulated_time, extra=extra, fracture=fr)


def resume(pipe, path):
    """Load `path` into `pipe` (models, particles, fracture component); colliders are the caller's.  -> (header, params)"""
    c, models, header, params = load(path)
    pipe.set_models(models)
    pipe.upload_particles(c)
    if header["fracture"] is not None:
        pipe.set_fracture(header["fracture"])
    return header, params


# ------------------------------------------------------------------------------------- sparkl serde layout (bincode 1)
_CONSTITUTIVE_VARIANT = {abi.CONSTITUTIVE_COROTATED_LINEAR: 0, abi.CONSTITUTIVE_NEO_HOOKEAN: 1, abi.CONSTITUTIVE_EOS_MONAGHAN_SPH: 2,
                         abi.CONSTITUTIVE_CUSTOM: 3}      # declaration order of CoreConstitutiveModel (constitutive_model.rs)
_PLASTIC_VARIANT = {abi.PLASTIC_DRUCKER_PRAGER: 0, abi.PLASTIC_NACC: 1, abi.PLASTIC_RANKINE: 2, abi.PLASTIC_SNOW: 3, abi.PLASTIC_CUSTOM: 4}


def _seq(w, values):
    """nalgebra ArrayStorage: u64 length + elements (column-major)."""
    v = np.asarray(values, dtype="<f4").ravel()
    w.write(struct.pack("<Q", len(v)))
    w.write(v.tobytes())


def _model_bytes(w, m):
    cm = m.constitutive_model
    w.write(struct.pack("<I", _CONSTITUTIVE_VARIANT[cm.tag]))
    if cm.tag == abi.CONSTITUTIVE_COROTATED_LINEAR:
        c = cm.u.corotated      # { split_stress_on_failure: bool, cfl_coeff, lambda, mu }
        w.write(struct.pack("<?fff", bool(c.split_stress_on_failure), c.cfl_coeff, c.lambda_, c.mu))
    elif cm.tag == abi.CONSTITUTIVE_NEO_HOOKEAN:
        c = cm.u.neo_hookean    # { cfl_coeff, lambda, mu }
        w.write(struct.pack("<fff", c.cfl_coeff, c.lambda_, c.mu))
    elif cm.tag == abi.CONSTITUTIVE_EOS_MONAGHAN_SPH:
        c = cm.u.monaghan       # { pressure0, gamma: u32, viscosity, max_neg_pressure }
        w.write(struct.pack("<fIff", c.pressure0, c.gamma, c.viscosity, c.max_neg_pressure))
    else:
        w.write(struct.pack("<I", cm.u.custom_id))
    w.write(struct.pack("<B", 1 if m.has_plastic_model else 0))
    if m.has_plastic_model:
        pm = m.plastic_model
        w.write(struct.pack("<I", _PLASTIC_VARIANT[pm.tag]))
        if pm.tag == abi.PLASTIC_DRUCKER_PRAGER:
            d = pm.u.drucker_prager   # { h0, h1, h2, h3, lambda, mu, only_active_when_failed: bool, volume_correction }
            w.write(struct.pack("<ffffff?f", d.h0, d.h1, d.h2, d.h3, d.lambda_, d.mu, bool(d.only_active_when_failed), d.volume_correction))
        elif pm.tag == abi.PLASTIC_NACC:
            d = pm.u.nacc             # { mu, kappa, hardening_enabled: bool, hardening_factor, cohesion, friction }
            w.write(struct.pack("<ff?fff", d.mu, d.kappa, bool(d.hardening_enabled), d.hardening_factor, d.cohesion, d.friction))
        elif pm.tag == abi.PLASTIC_RANKINE:
            d = pm.u.rankine          # { mu, lambda, tensile_strength, softening_rate }
            w.write(struct.pack("<ffff", d.mu, d.lambda_, d.tensile_strength, d.softening_rate))
        elif pm.tag == abi.PLASTIC_SNOW:
            d = pm.u.snow             # { min_epsilon, max_epsilon, hardening_coeff }
            w.write(struct.pack("<fff", d.min_epsilon, d.max_epsilon, d.hardening_coeff))
        else:
            w.write(struct.pack("<I", pm.u.custom_id))
    w.write(struct.pack("<B", 1 if m.has_failure_model else 0))
    if m.has_failure_model:
        fm = m.failure_model
        w.write(struct.pack("<I", 0 if fm.tag == abi.FAILURE_MAXIMUM_STRESS else 1))
        if fm.tag == abi.FAILURE_MAXIMUM_STRESS:
            w.write(struct.pack("<ff", fm.u.maximum_stress.max_principal_stress, fm.u.maximum_stress.max_shear_stress))
        else:
            w.write(struct.pack("<I", fm.u.custom_id))


def dump_bincode(components, model_set_or_array):
    """-> (particles_bytes, models_bytes): `bincode::serialize(&Vec<Particle>)`, `bincode::serialize(&Vec<ParticleModel>)`."""
    c = components
    n = len(c)
    w = io.BytesIO()
    w.write(struct.pack("<Q", n))
    zero3 = np.zeros(3, dtype="<f4")
    for i in range(n):
        st, vol, ph, da = c.status[i], c.volume[i], c.phase[i], c.data[i]
        w.write(struct.pack("<II", int(st["model_index"]) & 0xFFFFFFFF, 0))                  # arena Index { index, generation }
        w.write(struct.pack("<fff", vol["mass"], vol["volume0"], vol["radius0"]))
        _seq(w, c.position["point"][i])
        _seq(w, c.velocity["vector"][i])
        if st["kinematic_vel_enabled"]:
            w.write(b"\x01")
            _seq(w, st["kinematic_vel"])
        else:
            w.write(b"\x00")
        _seq(w, c.velocity_gradient["m"][i])
        _seq(w, vol["deformation_gradient"])
        w.write(struct.pack("<f", vol["plastic_deformation_gradient_det"]))
        w.write(struct.pack("<Q??", 0, bool(st["failed"]), bool(st["is_static"])))            # grid_index (rebuilt by sort), failed, is_static
        psi = float(ph["psi_pos"])
        w.write(struct.pack("<fffff", psi, psi * float(vol["mass"]), float(vol["mass"]), 0.0, 0.0))   # psi_pos, parameter1, parameter2, crack_*
        _seq(w, zero3)                                                                        # boundary_normal
        w.write(struct.pack("<f", 0.0))                                                       # boundary_dist
        w.write(struct.pack("<fff", 0.0, 0.0, ph["phase"]))                                   # m_c, g, phase
        _seq(w, zero3)                                                                        # phase_buf
        w.write(struct.pack("<ffff", da["nacc_alpha"], da["plastic_hardening"], da["elastic_hardening"], da["log_vol_gain"]))
        w.write(struct.pack("<Qf", 0, 0.0))                                                   # user_data, debug_val
    models = model_set_or_array.to_c_array() if hasattr(model_set_or_array, "to_c_array") else model_set_or_array
    wm = io.BytesIO()
    wm.write(struct.pack("<Q", len(model_set_or_array)))
    for k in range(len(model_set_or_array)):
        _model_bytes(wm, models[k])
    return w.getvalue(), wm.getvalue()


class _Reader:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, fmt):
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += struct.calcsize(fmt)
        return v

    def seq(self, n):
        (ln,) = self.take("<Q")
        if ln != n:
            raise ValueError(f"expected a sequence of {n} floats, found {ln}")
        v = np.frombuffer(self.b, dtype="<f4", count=n, offset=self.o).copy()
        self.o += 4 * n
        return v


def load_bincode(particles_bytes, models_bytes):
    r = _Reader(particles_bytes)
    (n,) = r.take("<Q")
    c = ParticleComponents(n)
    for i in range(n):
        idx, _gen = r.take("<II")
        c.status["model_index"][i] = idx
        c.volume["mass"][i], c.volume["volume0"][i], c.volume["radius0"][i] = r.take("<fff")
        c.position["point"][i] = r.seq(3)
        c.velocity["vector"][i] = r.seq(3)
        (tag,) = r.take("<B")
        if tag:
            c.status["kinematic_vel_enabled"][i] = 1
            c.status["kinematic_vel"][i] = r.seq(3)
        c.velocity_gradient["m"][i] = r.seq(9)
        c.volume["deformation_gradient"][i] = r.seq(9)
        (c.volume["plastic_deformation_gradient_det"][i],) = r.take("<f")
        _gi, failed, static = r.take("<Q??")
        c.status["failed"][i], c.status["is_static"][i] = failed, static
        psi, _p1, _p2, _cf, _ct = r.take("<fffff")
        c.phase["psi_pos"][i] = psi
        r.seq(3)
        r.take("<f")
        _mc, _g, c.phase["phase"][i] = r.take("<fff")
        r.seq(3)
        c.data["nacc_alpha"][i], c.data["plastic_hardening"][i], c.data["elastic_hardening"][i], c.data["log_vol_gain"][i] = r.take("<ffff")
        r.take("<Qf")
    if r.o != len(particles_bytes):
        raise ValueError("trailing bytes after Vec<Particle>")
    m = _Reader(models_bytes)
    (nm,) = m.take("<Q")
    arr = (abi.ParticleModelC * max(nm, 1))()
    cons = {v: k for k, v in _CONSTITUTIVE_VARIANT.items()}
    plas = {v: k for k, v in _PLASTIC_VARIANT.items()}
    for k in range(nm):
        pmodel = arr[k]
        (var,) = m.take("<I")
        pmodel.constitutive_model.tag = cons[var]
        u = pmodel.constitutive_model.u
        if cons[var] == abi.CONSTITUTIVE_COROTATED_LINEAR:
            s, u.corotated.cfl_coeff, u.corotated.lambda_, u.corotated.mu = m.take("<?fff")
            u.corotated.split_stress_on_failure = int(s)
        elif cons[var] == abi.CONSTITUTIVE_NEO_HOOKEAN:
            u.neo_hookean.cfl_coeff, u.neo_hookean.lambda_, u.neo_hookean.mu = m.take("<fff")
        elif cons[var] == abi.CONSTITUTIVE_EOS_MONAGHAN_SPH:
            u.monaghan.pressure0, u.monaghan.gamma, u.monaghan.viscosity, u.monaghan.max_neg_pressure = m.take("<fIff")
        else:
            (u.custom_id,) = m.take("<I")
        (has,) = m.take("<B")
        pmodel.has_plastic_model = has
        if has:
            (var,) = m.take("<I")
            pmodel.plastic_model.tag = plas[var]
            pu = pmodel.plastic_model.u
            if plas[var] == abi.PLASTIC_DRUCKER_PRAGER:
                d = pu.drucker_prager
                d.h0, d.h1, d.h2, d.h3, d.lambda_, d.mu, oa, d.volume_correction = m.take("<ffffff?f")
                d.only_active_when_failed = int(oa)
            elif plas[var] == abi.PLASTIC_NACC:
                d = pu.nacc
                d.mu, d.kappa, he, d.hardening_factor, d.cohesion, d.friction = m.take("<ff?fff")
                d.hardening_enabled = int(he)
            elif plas[var] == abi.PLASTIC_RANKINE:
                d = pu.rankine
                d.mu, d.lambda_, d.tensile_strength, d.softening_rate = m.take("<ffff")
            elif plas[var] == abi.PLASTIC_SNOW:
                d = pu.snow
                d.min_epsilon, d.max_epsilon, d.hardening_coeff = m.take("<fff")
            else:
                (pu.custom_id,) = m.take("<I")
        (has,) = m.take("<B")
        pmodel.has_failure_model = has
        if has:
            (var,) = m.take("<I")
            pmodel.failure_model.tag = abi.FAILURE_MAXIMUM_STRESS if var == 0 else abi.FAILURE_CUSTOM
            if var == 0:
                pmodel.failure_model.u.maximum_stress.max_principal_stress, pmodel.failure_model.u.maximum_stress.max_shear_stress = m.take("<ff")
            else:
                (pmodel.failure_model.u.custom_id,) = m.take("<I")
    if m.o != len(models_bytes):
        raise ValueError("trailing bytes after Vec<ParticleModel>")
    return c, ModelArray(arr, nm)
