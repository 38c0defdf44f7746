#!/usr/bin/env python3
"""GPU box: work items (> 8x8 px triangles split into <= 32x32 px cells) and drawn triangles per r3n_forward call of a bench
workload -- how many items a big triangle costs, per camera.   usage: python tools/item_stats.py [--config 4 | --bistro-v2]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rend3_amd as r3  # noqa: E402
from rend3_amd import _ffi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--bistro-v2", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(scene=None, config=a.config, bistro_v2=a.bistro_v2, objects=3000, tris=2_800_000, untextured=False, instanced=False, samples=1)
r = r3.Renderer(r3.host.RIGHT, np.float32(3840 / 2160))
info = bench.build_workload(args, r, r3.host, r3.material_record)
base = r3.BaseRenderGraph(r)
for k in range(4):
    r.set_camera_data(bench.camera_path(r3.host, info["camera"][0], k), info["camera"][1])
    r.render(3840, 2160, ambient=info["ambient"], clear_color=info["clear"], readback=False, base=base)
st = np.zeros(64, dtype=np.uint32)
r.lib.r3n_readback_raster_stats(r.ctx, _ffi.ptr(st))
calls = np.zeros((6, 5), dtype=np.uint32)
r.lib.r3n_readback_draw_calls(r.ctx, 0xFFFFFFFF, _ffi.ptr(calls))
print("workload:", info["workload"][:90])
print("viewport: work items per forward call (pass 1 opaque, cutout, pass 2 opaque, cutout, ...):", st[:8].tolist(), " drawn triangles predicted / residual per key:", (calls[:, 0] // 3).tolist())
for lane in range(2):
    print(f"shadow lane {lane + 1}: work items per forward call (view a opaque, cutout, view b opaque, cutout):", st[16 + 12 * lane:16 + 12 * lane + 6].tolist())
for si in range(4):
    r.lib.r3n_readback_draw_calls(r.ctx, si, _ffi.ptr(calls))
    print("shadow view", si, "drawn triangles per key:", (calls[:3, 0] // 3).tolist())
