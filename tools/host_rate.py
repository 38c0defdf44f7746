#!/usr/bin/env python3
"""Host cost of a frame.  (1) Sustained: time to ENQUEUE K frames against the time until the GPU has finished them -- the host runs
ahead of the GPU until a pinned frame image it wants to reuse is still in flight (4 frames), so the sustained enqueue time
converges to the GPU's frame time whenever the host is the faster side.  (2) Unthrottled: bursts of 3 frames after a full
synchronisation (nothing to wait for): the host's own time per frame -- Python glue + the C++ frame + the HIP runtime's launches."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import rend3_amd as r3, rend3_amd.scenes as S
import bench
r = r3.Renderer(r3.host.RIGHT, np.float32(bench.WIDTH) / np.float32(bench.HEIGHT))
info = S.bistro_like(r, r3.host, r3.material_record, textured=True)
base = r3.BaseRenderGraph(r)
views = [bench.camera_path(r3.host, info["camera"][0], k) for k in range(600)]
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
mode = 'node-by-node (R3N_FRAME_NODES=1)' if r.frame_nodes else 'one call (r3n_render_frame)'
print(f"{mode}: sustained enqueue {1e3*(t1-t0)/K:.3f} ms/frame, until done {1e3*(t2-t0)/K:.3f} ms/frame")
bursts = []
k = 120
for b in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(3):
        frame(k); k += 1
    bursts.append((time.perf_counter() - t0) / 3)
bursts.sort()
print(f"{mode}: unthrottled host time per frame: median {1e3*bursts[len(bursts)//2]:.3f} ms, min {1e3*bursts[0]:.3f} ms (bursts of 3 frames after a sync)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for j in range(50): frame(k); k += 1
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
