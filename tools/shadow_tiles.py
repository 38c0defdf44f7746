#!/usr/bin/env python3
"""Histogram of the shadow views' tile lists on the bench scene (diagnostics: how uneven is the binning?)."""
import sys
sys.path.insert(0, '.')
import ctypes
import numpy as np
import rend3_amd as r3, rend3_amd.scenes as S
import bench
r = r3.Renderer(r3.host.RIGHT, np.float32(bench.WIDTH) / np.float32(bench.HEIGHT))
info = S.bistro_like(r, r3.host, r3.material_record, textured=False)
for k in range(3):
    r.set_camera_data(bench.camera_path(r3.host, info["camera"][0], k), info["camera"][1])
    r.render(bench.WIDTH, bench.HEIGHT, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False)
for v in range(4):
    counts = np.zeros(4096, dtype=np.uint32)
    tx = ctypes.c_uint32(0)
    r._check(r.lib.r3n_readback_shadow_tile_counts(r.ctx, v, r3._ffi.ptr(counts), len(counts), ctypes.byref(tx)), "tile counts")
    n = tx.value ** 2
    c = counts[:n]
    print(f"view {v}: tiles {n} total refs {int(c.sum())} empty {(c == 0).sum()} mean {c.mean():.0f} median {np.median(c):.0f} "
          f"p90 {np.percentile(c, 90):.0f} p99 {np.percentile(c, 99):.0f} max {c.max()} >512: {(c > 512).sum()} >2048: {(c > 2048).sum()} >=8192: {(c >= 8192).sum()}")
