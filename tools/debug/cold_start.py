"""Per-launch k_g2p2g time over the first substeps after upload (cold start): is the driver-style `--steps 20 --warmup 5` run slower
because of clocks, first-touch, or the re-bin?"""
import sys, time
sys.path.insert(0, ".")
import torch
from sparkl_amd import scenes
from sparkl_amd.pipeline import MpmPipeline
spin = len(sys.argv) > 1 and sys.argv[1] == "spin"
sc = scenes.sand_column(200, 250, 200)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
p = MpmPipeline(sc.cell_width, domain_log2_bytes=sc.domain_log2_bytes, stream=stream.cuda_stream)
p.load_scene(sc)
if spin:   # keep the device busy for ~0.4 s on unrelated work first
    a = torch.randn(4096, 4096, device="cuda"); t0 = time.time()
    while time.time() - t0 < 0.4: a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
out = []
for k in range(12):
    p.clock_enable(True, stride=1); p.clock_read(reset=True); p.sync()
    t0 = time.perf_counter(); p.substeps(sc.params, sc.gravity, sc.fixed_dt, 5); p.sync(); dt = time.perf_counter() - t0
    c = p.clock_read()
    out.append((1e3 * dt / 5, c.g2p2g_ms / max(c.g2p2g_launches, 1)))
print("spin" if spin else "cold", " ".join(f"{a:.3f}/{b:.3f}" for a, b in out))
