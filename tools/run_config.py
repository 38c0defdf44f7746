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

def run_cfg5(n=50000, steps=20):
    """BASELINE.json configs[4]: 50 000 instances of a 160-ish vertex / 2-joint rig (RiggedSimple.glb shape), skinning only."""
    from rend3_amd.scenes import skinned_cylinder
    r = r3.Renderer(r3.host.LEFT, np.float32(16 / 9))
    pos, idx, nrm, tang, ji, jw = skinned_cylinder(2)
    mesh = r.add_mesh(pos, idx, normals=nrm, tangents=tang, joint_indices=ji, joint_weights=jw)
    rng = np.random.Generator(np.random.PCG64(0x5141))
    def poses():
        m = np.tile(r3.host.identity(), (n, 2, 1)).astype(np.float32)
        ang = rng.uniform(-0.8, 0.8, (n, 2))
        m[:, :, 0] = np.cos(ang); m[:, :, 1] = np.sin(ang); m[:, :, 4] = -np.sin(ang); m[:, :, 5] = np.cos(ang)
        return m
    sks = r.add_skeletons_bulk(mesh, list(poses()))
    sk_in, _ = r.skinning_buffers()
    verts = int(sk_in[:, 9].sum())
    mats = [np.ascontiguousarray(poses().reshape(-1, 16)) for _ in range(4)]
    def skin(k):
        m = mats[k % 4]
        r._check(r.lib.r3n_skinning(r.ctx, r3._ffi.ptr(sk_in), len(sk_in), r3._ffi.ptr(m), len(m)), "r3n_skinning")
    for k in range(3): skin(k)
    r.sync(); r.timing_enable(True); r.stage_times(reset=True)
    t0 = time.perf_counter()
    for k in range(steps): skin(k)
    r.sync(); wall = (time.perf_counter() - t0) / steps
    st = r.stage_times(reset=True)
    kern_ms = st["skinning"][0] / steps
    bytes_v = 60 + 36 + 24  # p,n,t 36 + joints 8 + weights 16 read; p,n,t 36 written (60 B + 36 B) -- tangent run included
    print(json.dumps({"config": "configs[4] skinning", "skeletons": n, "vertices": verts, "joints_per_skeleton": 2,
                      "kernel_ms": round(kern_ms, 4), "wall_ms_incl_matrix_upload": round(1e3 * wall, 3),
                      "vertices_per_s": round(verts / (kern_ms * 1e-3)),
                      "algorithmic_GBps": round(verts * 96 / (kern_ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000,
                      "frac_of_hbm_roofline": round(verts * 96 / (kern_ms * 1e-3) / 1e9 / 8000, 4)}))
    r.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg2", "cfg4", "cfg5"]
    if "cfg5" in which:
        run_cfg5()
    if "cfg2" in which:
        run("configs[1] scifi_like", lambda r: S.scifi_like(r, r3.host, r3.material_record), 1920, 1080, True)
    if "cfg4" in which:
        run("configs[3] emerald_like", lambda r: S.emerald_like(r, r3.host, r3.material_record), 3840, 2160, False)
