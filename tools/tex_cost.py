#!/usr/bin/env python3
"""GPU box: resolve time of the textured bench scene with subsets of the material maps (which map costs what)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import rend3_amd as r3
import rend3_amd.scenes
import bench

W, H = 3840, 2160


def run(name, strip, nearest=False):
    def mk(**kw):
        for k in strip:
            kw.pop(k, None)
        if "albedo_texture" not in kw and str(kw.get("albedo_mode", "")).startswith("texture"):
            kw["albedo_mode"] = "value"
        if nearest:
            kw["nearest"] = True
        return r3.material_record(**kw)
    r = r3.Renderer(r3.host.RIGHT, np.float32(W / H))
    info = r3.scenes.bistro_like(r, r3.host, mk, textured=True)
    base = r3.BaseRenderGraph(r)
    r.set_multi_stream(False)
    for k in range(4):
        r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
        r.render(W, H, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
    r.sync(); r.timing_enable(True); r.stage_times(reset=True)
    for k in range(4, 14):
        r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
        r.render(W, H, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
    r.sync()
    st = r.stage_times(reset=True)
    print(f"{name:<28} shade {st['shade'][0] / 10 * 1e3:7.1f} us")
    r.close()


run("all three maps", [])
run("albedo + normal", ["aomr"])
run("albedo + orm", ["normal_texture"])
run("albedo only", ["aomr", "normal_texture"])
run("normal only", ["aomr", "albedo_texture"])
run("no maps (textured kernel)", ["aomr", "normal_texture", "albedo_texture"])
run("all three, nearest", [], nearest=True)
