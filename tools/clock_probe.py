"""Shader clock (clock64 / wall_clock64) sampled by a one-wave kernel on a second stream while 10M-sand substeps run."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "clock_probe.so"))
lib.probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
sc = scenes.sand_column(200, 250, 200)
n = 7000
period = 2000  # 20 us at 100 MHz
st = torch.cuda.Stream()
with MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, stream=st.cuda_stream) as p:
    p.load_scene(sc)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 4)
    p.sync()
    out = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    s2 = torch.cuda.Stream()
    assert lib.probe_launch(out.data_ptr(), n, period, s2.cuda_stream) == 0
    import time; time.sleep(0.01)          # ~10 ms idle samples first
    mode = sys.argv[1] if len(sys.argv) > 1 else "none"
    if mode == "matmul":                    # does a different heavy load settle the clock for the substeps that follow?
        with torch.cuda.stream(st):
            a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
            t_end = time.perf_counter() + 0.06
            while time.perf_counter() < t_end:
                for _ in range(4):
                    c = a @ b
                torch.cuda.current_stream().synchronize()
    elif mode == "substeps":
        p.substeps(sc.params, sc.gravity, sc.fixed_dt, 100)
    print("load starts at", mode)
    p.substeps(sc.params, sc.gravity, sc.fixed_dt, 60)
    p.sync()
    s2.synchronize()
o = out.cpu().numpy().reshape(n, 2)
w, c = o[:, 0].astype(np.float64), o[:, 1].astype(np.float64)
f = np.diff(c) / np.diff(w) * 100.0   # MHz
t = (w[1:] - w[0]) / 100.0 / 1000.0   # ms
for lo in range(0, len(f), 100):
    print(f"t={t[lo]:7.2f} ms  shader clock {f[lo:lo+100].mean():8.1f} MHz  (min {f[lo:lo+100].min():8.1f} max {f[lo:lo+100].max():8.1f})")
