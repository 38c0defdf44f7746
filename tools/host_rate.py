#!/usr/bin/env python3
"""Is the bench host-bound?  Time to ENQUEUE K frames (no synchronisation) against the time until the GPU has finished them."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import rend3_amd as r3, rend3_amd.scenes as S
import bench
r = r3.Renderer(r3.host.RIGHT, np.float32(bench.WIDTH) / np.float32(bench.HEIGHT))
info = S.bistro_like(r, r3.host, r3.material_record, textured=True)
base = r3.BaseRenderGraph(r)
views = [bench.camera_path(r3.host, info["camera"][0], k) for k in range(400)]
def frame(k):
    r.set_camera_data(views[k], info["camera"][1])
    r.render(bench.WIDTH, bench.HEIGHT, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
for k in range(10): frame(k)
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for k in range(K): frame(10 + k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{'node-by-node (R3N_FRAME_NODES=1)' if r.frame_nodes else 'one call (r3n_render_frame)'}: enqueue {1e3*(t1-t0)/K:.3f} ms/frame, until done {1e3*(t2-t0)/K:.3f} ms/frame")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(50): frame(200 + k)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
