#!/usr/bin/env python3
"""Cull-rate measurement on the cull-centric configs of BASELINE.json: configs[1] (scifi-like, 20 000 objects, 1080p)
and configs[3] (emerald-like, 1 048 576 objects, 4K).  Prints one JSON line per config with objects/s and Mtri/s of
the bake + object + triangle-cull stages (HIP events) and the whole-frame time."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import rend3_amd as r3, rend3_amd.scenes as S

def run(name, build, w, h, rh, steps=20):
    r = r3.Renderer(r3.host.RIGHT if rh else r3.host.LEFT, np.float32(w) / np.float32(h))
    t0 = time.perf_counter(); info = build(r); build_s = time.perf_counter() - t0
    base = r3.BaseRenderGraph(r)
    view0 = info["camera"][0]
    def frame(k):
        r.set_camera_data(r3.host.mat4_mul(r3.host.rotation_y(0.004 * k), view0), info["camera"][1])
        r.render(w, h, ambient=(0.1, 0.1, 0.1, 1), readback=False, base=base)
    for k in range(4): frame(k)
    r.sync(); t0 = time.perf_counter()
    for k in range(4, 4 + steps): frame(k)
    r.sync(); dt = (time.perf_counter() - t0) / steps
    r.timing_enable(True); r.stage_times(reset=True)
    for k in range(4 + steps, 14 + steps): frame(k)
    r.sync(); st = r.stage_times(reset=True); r.timing_enable(False)
    cams = 1 + len(r.dir_lights)
    cull_ms = sum(st[s][0] for s in ("bake", "object_cull", "triangle_cull")) / 10
    print(json.dumps({"config": name, "objects": info["objects"], "triangles": info["triangles"], "resolution": [w, h], "cameras": cams,
                      "frame_ms": round(1e3 * dt, 3), "cull_ms_per_frame": round(cull_ms, 4),
                      "culled_objects_per_s": round(info["objects"] * cams / (cull_ms * 1e-3)),
                      "culled_mtris_per_s": round(info["triangles"] * cams / (cull_ms * 1e-3) / 1e6, 1),
                      "stage_ms": {k: round(v[0] / 10, 4) for k, v in st.items()}, "scene_build_s": round(build_s, 1)}))
    r.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg2", "cfg4"]
    if "cfg2" in which:
        run("configs[1] scifi_like", lambda r: S.scifi_like(r, r3.host, r3.material_record), 1920, 1080, True)
    if "cfg4" in which:
        run("configs[3] emerald_like", lambda r: S.emerald_like(r, r3.host, r3.material_record), 3840, 2160, False)
