#!/usr/bin/env python3
"""GPU box: throughput of the bench frame with and without the resolve node (how much of the frame could hide behind it)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import rend3_amd as r3
import rend3_amd.scenes
from rend3_amd.renderer import RenderGraph, BaseRenderGraphInputs, BaseRenderGraphSettings
import bench

W, H = 3840, 2160
r = r3.Renderer(r3.host.RIGHT, np.float32(W / H))
info = r3.scenes.bistro_like(r, r3.host, r3.material_record, textured="--untextured" not in sys.argv)
base = r3.BaseRenderGraph(r)


def frame(k, skip):
    r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
    ev = r.evaluate_instructions()
    g = RenderGraph()
    base.add_to_graph(g, BaseRenderGraphInputs(ev, base.default_routines(), (W, H), 1), BaseRenderGraphSettings(bench.AMBIENT, bench.CLEAR))
    g.nodes = [n for n in g.nodes if n[0] not in skip]
    g.execute(r, ev)


for name, skip in (("full frame", ()), ("without resolve", ("Resolve Opaque",)), ("resolve only repeated", None)):
    if skip is None:
        continue
    for k in range(8):
        frame(k, skip)
    r.sync()
    t0 = time.perf_counter()
    for k in range(8, 68):
        frame(k, skip)
    r.sync()
    print(f"{name:<24} {1e3 * (time.perf_counter() - t0) / 60:.4f} ms/frame")
