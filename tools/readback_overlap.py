"""Frame loop at N particles: step only / step + blocking reads / step + overlapped readback."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sparkl_amd import abi, scenes                      # noqa: E402
from sparkl_amd.pipeline import MpmPipeline             # noqa: E402

n_target = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0e6
substeps, frames = 64, 6
side = int(round((n_target / 1.25) ** (1.0 / 3.0) / 8.0)) * 8
sc = scenes.sand_column(side, int(side * 1.25), side)
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes) as p:
    p.load_scene(sc)
    n = len(p)
    pos, vel, ph = p.host_alloc(n, abi.POSITION_DTYPE), p.host_alloc(n, abi.VELOCITY_DTYPE), p.host_alloc(n, abi.PHASE_DTYPE)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 16)
    p.sync()
    out = {"particles": n, "substeps_per_frame": substeps, "bytes_per_frame": n * 32}

    def loop(mode):
        p.load_scene(sc)                                # same physical state for every mode (the collapse changes the cost per frame)
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 16)
        p.readback_begin(pos, vel, ph)
        p.readback_wait()
        p.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            p.substeps(sc.params, sc.gravity, sc.fixed_dt, substeps)
            if mode == "blocking":
                p.read_positions(); p.read_velocities(); p.read_phases()
            elif mode == "overlapped":
                p.readback_wait()                       # previous frame's copy must have landed before its buffers are reused
                p.readback_begin(pos, vel, ph)
        p.readback_wait()
        p.sync()
        return 1e3 * (time.perf_counter() - t0) / frames

    p.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        p.readback_begin(pos, vel, ph)
        p.readback_wait()
    out["readback_alone_ms"] = 1e3 * (time.perf_counter() - t0) / frames
    for mode in ("step_only", "blocking", "overlapped"):
        out[mode + "_ms_per_frame"] = loop(mode)
    print(json.dumps(out))
