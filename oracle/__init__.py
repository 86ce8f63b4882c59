"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper of oracle/liboracle.so (the C++ restatement of sparkl's CPU dynamics path,
oracle/mpm_oracle.hpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; nothing under sparkl_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from sparkl_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mpm_oracle.cpp", "mpm_oracle.hpp")] + [os.path.join(_HERE, "..", "include", "sparkl_hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_float, C.c_uint32]
        for name in ("oracle_step", "oracle_substep_fixed", "oracle_estimate_dt", "oracle_pos_energy", "oracle_dp_alpha",
                     "oracle_model_timestep_bound", "oracle_timed_substeps"):
            getattr(L, name).restype = C.c_double
        for name in ("oracle_num_particles", "oracle_get_nodes", "oracle_num_regions", "oracle_num_particle_regions", "oracle_num_bins",
                     "oracle_pack", "oracle_packed_add", "oracle_morton_encode3", "oracle_packed_nbh_shift",
                     "oracle_packed_region_shift", "oracle_check_pack_roundtrip", "oracle_check_packed_add", "oracle_get_region_ids"):
            getattr(L, name).restype = C.c_uint64
        L.oracle_pack.argtypes = [C.c_uint32] * 3
        L.oracle_morton_encode3.argtypes = [C.c_uint32] * 3
        L.oracle_packed_add.argtypes = [C.c_uint64] * 2
        L.oracle_unpack.argtypes = [C.c_uint64, C.c_void_p]
        L.oracle_region_color.argtypes = [C.c_uint64]
        L.oracle_check_pack_roundtrip.argtypes = [C.c_uint32] * 6
        L.oracle_check_packed_add.argtypes = [C.c_uint32] * 6
        _LIB = L
    return _LIB


def _g(g):
    return (C.c_float * 3)(*[float(x) for x in g])


class Oracle:
    """One CPU solver instance (f32 by default; f64=True for the double-precision variant)."""

    def __init__(self, cell_width, f64=False, domain_log2_bytes=30, threads=1, penalty_force=False):
        self.L = lib()
        self.f64 = bool(f64)
        self.h = self.L.oracle_create(int(f64), C.c_float(cell_width), domain_log2_bytes)
        if not self.h:
            raise MemoryError("oracle mmap failed")
        self.h = C.c_void_p(self.h)
        self.L.oracle_set_num_threads(self.h, int(threads))
        # the reference has the collider penalty force under `if false` (particle_updater.rs:204): off unless a test asks for the intended rule
        self.L.oracle_set_penalty_force(self.h, int(bool(penalty_force)))
        self._keep = []

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state
    def set_particles(self, comps):
        self.L.oracle_set_particles(self.h, C.c_uint64(len(comps)), abi.ptr(comps.status, abi.STATUS_DTYPE),
                                    abi.ptr(comps.position, abi.POSITION_DTYPE), abi.ptr(comps.velocity, abi.VELOCITY_DTYPE),
                                    abi.ptr(comps.volume, abi.VOLUME_DTYPE), abi.ptr(comps.phase, abi.PHASE_DTYPE),
                                    abi.ptr(comps.data, abi.DATA_DTYPE), abi.ptr(comps.velocity_gradient, abi.VELGRAD_DTYPE))

    def set_fracture(self, fracture):
        """`ParticleFracture` per particle (caller order): numpy (n, 2) float32 = crack_propagation_factor, crack_threshold."""
        f = np.ascontiguousarray(fracture, dtype=np.float32)
        assert f.shape == (int(self.L.oracle_num_particles(self.h)), 2)
        self.L.oracle_set_fracture(self.h, C.c_uint64(len(f)), f.ctypes.data_as(C.c_void_p))

    def get_particles(self):
        from sparkl_amd.particles import ParticleComponents
        n = int(self.L.oracle_num_particles(self.h))
        c = ParticleComponents(n)
        self.L.oracle_get_particles(self.h, abi.ptr(c.status, abi.STATUS_DTYPE), abi.ptr(c.position, abi.POSITION_DTYPE),
                                    abi.ptr(c.velocity, abi.VELOCITY_DTYPE), abi.ptr(c.volume, abi.VOLUME_DTYPE),
                                    abi.ptr(c.phase, abi.PHASE_DTYPE), abi.ptr(c.data, abi.DATA_DTYPE),
                                    abi.ptr(c.velocity_gradient, abi.VELGRAD_DTYPE))
        return c

    def get_state_f64(self):
        n = int(self.L.oracle_num_particles(self.h))
        pos, vel = np.zeros((n, 3)), np.zeros((n, 3))
        F, Cm = np.zeros((n, 9)), np.zeros((n, 9))
        self.L.oracle_get_state_f64(self.h, pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p),
                                    F.ctypes.data_as(C.c_void_p), Cm.ctypes.data_as(C.c_void_p))
        return pos, vel, F, Cm

    def set_models(self, model_set):
        arr = model_set.to_c_array()
        self.L.oracle_set_models(self.h, C.c_uint32(len(model_set)), arr)

    def set_cuboids(self, cuboids):
        arr = (abi.CuboidColliderC * max(len(cuboids), 1))(*cuboids)
        self.L.oracle_set_cuboids(self.h, C.c_uint32(len(cuboids)), arr)

    def set_heightfields(self, heightfields):
        arr = (abi.HeightFieldColliderC * max(len(heightfields), 1))(*heightfields)
        self._keep.append(heightfields)
        self.L.oracle_set_heightfields(self.h, C.c_uint32(len(heightfields)), arr)

    def set_trimeshes(self, trimeshes):
        arr = (abi.TriMeshColliderC * max(len(trimeshes), 1))(*trimeshes)
        self._keep.append(trimeshes)
        self.L.oracle_set_trimeshes(self.h, C.c_uint32(len(trimeshes)), arr)

    def set_polylines(self, polylines):
        arr = (abi.PolylineColliderC * max(len(polylines), 1))(*polylines)
        self._keep.append(polylines)
        self.L.oracle_set_polylines(self.h, C.c_uint32(len(polylines)), arr)

    def set_collider_options(self, kind, options):
        arr = (abi.ColliderOptionsC * max(len(options), 1))(*options)
        self.L.oracle_set_collider_options(self.h, C.c_int({"cuboid": 0, "heightfield": 1, "trimesh": 2, "polyline": 3}[kind]), C.c_uint32(len(options)), arr)

    def load_scene(self, scene):
        self.set_models(scene.models)
        self.set_cuboids(scene.colliders)
        self.set_heightfields(getattr(scene, "heightfields", []))
        self.set_trimeshes(getattr(scene, "trimeshes", []))
        self.set_polylines(getattr(scene, "polylines", []))
        for kind, opts in getattr(scene, "collider_options", {}).items():
            self.set_collider_options(kind, opts)
        self.set_particles(scene.particles)

    # ---- stepping
    def step(self, params, gravity):
        return float(self.L.oracle_step(self.h, C.byref(params), _g(gravity)))

    def last_dts(self):
        n = int(self.L.oracle_last_num_substeps(self.h))
        out = np.zeros(n)
        self.L.oracle_last_dts(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def substep_fixed(self, params, gravity, dt):
        return float(self.L.oracle_substep_fixed(self.h, C.byref(params), _g(gravity), C.c_double(dt)))

    def timed_substeps(self, params, gravity, dt, n):
        return float(self.L.oracle_timed_substeps(self.h, C.byref(params), _g(gravity), C.c_double(dt), int(n)))

    def sort(self):
        self.L.oracle_sort(self.h)

    def recompute_fluids_volumes(self):
        self.L.oracle_recompute_fluids_volumes(self.h)

    def estimate_dt(self, max_dt):
        return float(self.L.oracle_estimate_dt(self.h, C.c_double(max_dt)))

    def p2g(self, dt, gravity):
        self.L.oracle_p2g(self.h, C.c_double(dt), _g(gravity))

    def grid_update(self, dt, boundary_handling, dofs=0):
        self.L.oracle_grid_update(self.h, C.c_double(dt), int(boundary_handling), C.c_uint32(dofs))

    def g2p(self, dt):
        self.L.oracle_g2p(self.h, C.c_double(dt))

    def nodes(self):
        n = int(self.L.oracle_get_nodes(self.h, None, C.c_uint64(0)))
        out = np.zeros(n, abi.GRID_NODE_DTYPE)
        self.L.oracle_get_nodes(self.h, abi.ptr(out, abi.GRID_NODE_DTYPE), C.c_uint64(n))
        return out

    def region_ids(self):
        n = int(self.L.oracle_get_region_ids(self.h, None, C.c_uint64(0)))
        out = np.zeros(n, np.uint64)
        self.L.oracle_get_region_ids(self.h, out.ctypes.data_as(C.c_void_p), C.c_uint64(n))
        return out

    def get_block_nodes(self, region_id):
        out = np.zeros((64, 4), np.float32)
        self.L.oracle_get_block_nodes(self.h, C.c_uint64(int(region_id)), out.ctypes.data_as(C.c_void_p))
        return out

    def add_block_nodes(self, region_id, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.float32)
        return int(self.L.oracle_add_block_nodes(self.h, C.c_uint64(int(region_id)), nodes.ctypes.data_as(C.c_void_p)))

    def refresh_velocities(self, dt, gravity):
        self.L.oracle_refresh_velocities(self.h, C.c_double(dt), _g(gravity))

    def num_regions(self):
        return int(self.L.oracle_num_regions(self.h))

    def num_particle_regions(self):
        return int(self.L.oracle_num_particle_regions(self.h))

    def order(self):
        out = np.zeros(int(self.L.oracle_num_particles(self.h)), np.uint64)
        self.L.oracle_get_order(self.h, out.ctypes.data_as(C.c_void_p))
        return out

    def grid_indices(self):
        out = np.zeros(int(self.L.oracle_num_particles(self.h)), np.uint64)
        self.L.oracle_get_grid_indices(self.h, out.ctypes.data_as(C.c_void_p))
        return out


# ---- unit-level helpers (KATs)
def _d(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(n)
    return a, a.ctypes.data_as(C.c_void_p)


def svd3(A):
    """A is 3x3 row-major numpy; returns U, S, V (A = U diag(S) V^T)."""
    Acm, pa = _d(np.asarray(A, dtype=np.float64).T, 9)  # column-major
    U, S, V = np.zeros(9), np.zeros(3), np.zeros(9)
    lib().oracle_svd3(pa, U.ctypes.data_as(C.c_void_p), S.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p))
    return U.reshape(3, 3).T, S, V.reshape(3, 3).T


def kirchhoff_stress(cm, F, phase=1.0, hardening=1.0, mass=1.0, volume0=1.0, fluid_j=1.0, Cmat=None, f64=False):
    _, pf = _d(np.asarray(F, dtype=np.float64).T, 9)
    cmat = np.zeros((3, 3)) if Cmat is None else np.asarray(Cmat, dtype=np.float64)
    _, pc = _d(cmat.T, 9)
    out = np.zeros(9)
    lib().oracle_kirchhoff_stress(int(f64), C.byref(cm), C.c_double(phase), C.c_double(hardening), pf, C.c_double(mass),
                                  C.c_double(volume0), C.c_double(fluid_j), pc, out.ctypes.data_as(C.c_void_p))
    return out.reshape(3, 3).T


def pos_energy(cm, F, phase=1.0, hardening=1.0, f64=False):
    _, pf = _d(np.asarray(F, dtype=np.float64).T, 9)
    return float(lib().oracle_pos_energy(int(f64), C.byref(cm), C.c_double(phase), C.c_double(hardening), pf))


def dp_alpha(dp, q, f64=False):
    return float(lib().oracle_dp_alpha(int(f64), C.byref(dp), C.c_double(q)))


def dp_update(dp, F, Jp=1.0, q=1.0, lvg=0.0, phase=1.0, f64=False):
    f, pf = _d(np.asarray(F, dtype=np.float64).T.copy(), 9)
    st = np.array([Jp, q, lvg], dtype=np.float64)
    lib().oracle_dp_update(int(f64), C.byref(dp), C.c_double(phase), pf, st.ctypes.data_as(C.c_void_p))
    return f.reshape(3, 3).T.copy(), st[0], st[1], st[2]


def plastic_update(pm, F, Jp=1.0, plastic_hardening=1.0, log_vol_gain=0.0, elastic_hardening=1.0, nacc_alpha=-0.01, phase=1.0, f64=False):
    """`CorePlasticModel::update_particle` on one particle: returns (F, dict of the updated scalars)."""
    f, pf = _d(np.asarray(F, dtype=np.float64).T.copy(), 9)
    st = np.array([Jp, plastic_hardening, log_vol_gain, elastic_hardening, nacc_alpha], dtype=np.float64)
    lib().oracle_plastic_update(int(f64), C.byref(pm), C.c_double(phase), pf, st.ctypes.data_as(C.c_void_p))
    return f.reshape(3, 3).T.copy(), {"Jp": st[0], "plastic_hardening": st[1], "log_vol_gain": st[2], "elastic_hardening": st[3], "nacc_alpha": st[4]}


def maximum_stress_failed(fm, stress):
    _, ps = _d(np.asarray(stress, dtype=np.float64).T.copy(), 9)
    return bool(lib().oracle_maximum_stress_failed(C.byref(fm), ps))


def dp_project(dp, sv, lvg, alpha, f64=False):
    _, ps = _d(sv, 3)
    out = np.zeros(4)
    r = lib().oracle_dp_project(int(f64), C.byref(dp), ps, C.c_double(lvg), C.c_double(alpha), out.ctypes.data_as(C.c_void_p))
    return (out[:3].copy(), out[3]) if r else None


def eval_all(x, f64=False):
    out = np.zeros(3)
    lib().oracle_eval_all(int(f64), C.c_double(x), out.ctypes.data_as(C.c_void_p))
    return out


def heightfield_project(hf, pt):
    _, pp = _d(pt, 3)
    out = np.zeros(3)
    inside = C.c_int(0)
    lib().oracle_heightfield_project(C.byref(hf), pp, out.ctypes.data_as(C.c_void_p), C.byref(inside))
    return out, bool(inside.value)


def trimesh_project(tm, pt):
    _, pp = _d(pt, 3)
    out = np.zeros(3)
    inside = C.c_int(0)
    lib().oracle_trimesh_project(C.byref(tm), pp, out.ctypes.data_as(C.c_void_p), C.byref(inside))
    return out, bool(inside.value)


def cuboid_project(cuboid, pt):
    _, pp = _d(pt, 3)
    out = np.zeros(3)
    inside = C.c_int(0)
    lib().oracle_cuboid_project(C.byref(cuboid), pp, out.ctypes.data_as(C.c_void_p), C.byref(inside))
    return out, bool(inside.value)
