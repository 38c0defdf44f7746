"""How long does the host need to ENQUEUE one frame (no GPU wait)?  Separates host-bound from GPU-bound."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import rend3_amd as r3, rend3_amd.scenes
import bench
r = r3.Renderer(r3.host.RIGHT, np.float32(3840 / 2160))
info = r3.scenes.bistro_like(r, r3.host, r3.material_record)
base = r3.BaseRenderGraph(r)
def frame(k):
    r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
    r.render(3840, 2160, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
for k in range(10): frame(k)
r.sync()
t0 = time.perf_counter()
for k in range(10, 70): frame(k)
t1 = time.perf_counter()
r.sync()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/60:.3f} ms/frame; total {1e3*(t2-t0)/60:.3f} ms/frame")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(70, 100): frame(k)
pr.disable(); r.sync()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
