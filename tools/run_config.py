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

def run_cfg5(n=50000, steps=20, mfma=False):
    """BASELINE.json configs[4]: 50 000 instances of a 160-ish vertex / 2-joint rig (RiggedSimple.glb shape), skinning only.
    mfma: the matrix-core variant (R3N_SKIN_MFMA) instead of the vector kernel -- the A/B BASELINE.json's config names."""
    from rend3_amd.scenes import skinned_cylinder
    r = r3.Renderer(r3.host.LEFT, np.float32(16 / 9))
    if mfma:
        r.set_skinning_mode(1)
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
    print(json.dumps({"config": "configs[4] skinning" + (" (R3N_SKIN_MFMA: v_mfma_f32_16x16x4_f32)" if mfma else " (vector kernel)"), "skeletons": n, "vertices": verts, "joints_per_skeleton": 2,
                      "kernel_ms": round(kern_ms, 4), "wall_ms_incl_matrix_upload": round(1e3 * wall, 3),
                      "vertices_per_s": round(verts / (kern_ms * 1e-3)),
                      "algorithmic_GBps": round(verts * 96 / (kern_ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000,
                      "frac_of_hbm_roofline": round(verts * 96 / (kern_ms * 1e-3) / 1e9 / 8000, 4)}))
    r.close()


def run_cfg5_anim(n=50000, joints=24, steps=20):
    """configs[4] with rend3-anim in front of the skinning pass: n instances of a `joints`-joint chain rig, every instance at
    its own time of an 8-key clip (rotation + translation channels per joint); poses evaluated on the GPU (csrc/anim.hip),
    no joint matrices cross PCIe (16 B per instance and frame: clip, time, matrix base)."""
    from rend3_amd.scenes import skinned_cylinder
    from rend3_amd import anim as pa
    r = r3.Renderer(r3.host.LEFT, np.float32(16 / 9))
    pos, idx, nrm, tang, ji, jw = skinned_cylinder(2)
    ji = (ji.astype(np.int64) * (joints - 1)).astype(np.uint16)  # bind the two ends to the first / last joint of the chain
    mesh = r.add_mesh(pos, idx, normals=nrm, tangents=tang, joint_indices=ji, joint_weights=jw)
    rng = np.random.Generator(np.random.PCG64(0x5142))
    sks = r.add_skeletons_bulk(mesh, [np.tile(r3.host.identity(), (joints, 1))] * n)
    ident = r3.host.identity()
    nodes = [dict(parent=(j - 1 if j else None), local_transform=ident, objects=[], skin=None, skeletons=[]) for j in range(joints)]
    nodes[0]["skin"], nodes[0]["skeletons"] = 0, sks
    keys = 8
    channels = {}
    for j in range(joints):
        q = rng.normal(size=(keys, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        t = np.linspace(0, 2, keys).astype(np.float32)
        channels[j] = dict(rotation=(t, q.astype(np.float32)), translation=(t, rng.normal(size=(keys, 3)).astype(np.float32)))
    inst = dict(nodes=nodes, skins=[dict(joints=list(range(joints)), inverse_bind_matrices=np.tile(ident, (joints, 1)))], topological_order=list(range(joints)))
    pa.AnimationData.from_gltf_scene(r, [dict(channels=channels, duration=np.float32(2.0))], inst)
    sk_in, _ = r.skinning_buffers()
    verts = int(sk_in[:, 9].sum())
    rq = np.zeros(n, dtype=[("clip", np.uint32), ("time", np.float32), ("base", np.uint32), ("pad", np.uint32)])
    rq["base"] = sk_in[:, 8]
    def step(k):
        rq["time"] = ((np.arange(n) * 0.00004 + 0.01 * k) % 2.0).astype(np.float32)
        r._check(r.lib.r3n_pose_skeletons(r.ctx, r3._ffi.ptr(rq), n), "r3n_pose_skeletons")
        r._check(r.lib.r3n_skinning(r.ctx, r3._ffi.ptr(sk_in), len(sk_in), None, n * joints), "r3n_skinning")
    for k in range(3): step(k)
    r.sync(); r.timing_enable(True); r.stage_times(reset=True)
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    r.sync(); wall = (time.perf_counter() - t0) / steps
    st = r.stage_times(reset=True)
    pose_ms, skin_ms = st["pose"][0] / steps, st["skinning"][0] / steps
    # algorithmic bytes per joint: 80 B track + 80 B joint record + 64 B matrix out (keys are shared by all instances: cache)
    print(json.dumps({"config": "configs[4] skinning + rend3-anim poses on the GPU", "skeletons": n, "joints_per_skeleton": joints,
                      "keys_per_channel": keys, "pose_kernel_ms": round(pose_ms, 4), "skinning_kernel_ms": round(skin_ms, 4),
                      "wall_ms_per_frame": round(1e3 * wall, 3), "joint_matrices_per_s": round(n * joints / (pose_ms * 1e-3)),
                      "pose_algorithmic_GBps": round(n * joints * 224 / (pose_ms * 1e-3) / 1e9, 1), "vertices": verts}))
    r.close()


def run_cfg5_asset(n=5000, steps=20):
    """configs[4], "34-joint / 2324-vertex variant" (SURVEY section 8d): the reference's animation example character
    (tests/golden/animation-character.glb) x n skeleton instances, each at its own time of the asset's clip: GPU poses + skinning."""
    import os
    from rend3_amd.gltf import Gltf, instance_scene, load_animations
    from rend3_amd import anim as pa
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    r = r3.Renderer(r3.host.LEFT, np.float32(16 / 9))
    g = Gltf(os.path.join(root, "animation-character.glb"))
    inst = instance_scene(g, r, r3.host, r3.material_record)
    anims = load_animations(g)
    first = inst["skeletons"][0]
    mesh = r.skeletons[first]["mesh"]
    joints = len(r.skeletons[first]["matrices"])
    more = r.add_skeletons_bulk(mesh, [np.tile(r3.host.identity(), (joints, 1))] * (n - 1))
    node = next(nd for nd in inst["nodes"] if nd["skin"] == 0)
    node["skeletons"] += more
    data = pa.AnimationData.from_gltf_scene(r, anims, inst)
    sks = data.skin_skeletons[0]
    sk_in, _ = r.skinning_buffers()
    verts = int(sk_in[:, 9].sum())
    rq = np.zeros(len(sks), dtype=[("clip", np.uint32), ("time", np.float32), ("base", np.uint32), ("pad", np.uint32)])
    rq["clip"] = data.clip_base
    rq["base"] = sk_in[sks, 8]
    dur = float(anims[0]["duration"])
    def step(k):
        rq["time"] = ((np.arange(len(sks)) * 0.0023 + 0.016 * k) % dur).astype(np.float32)
        r._check(r.lib.r3n_pose_skeletons(r.ctx, r3._ffi.ptr(rq), len(rq)), "r3n_pose_skeletons")
        r._check(r.lib.r3n_skinning(r.ctx, r3._ffi.ptr(sk_in), len(sk_in), None, len(sks) * joints), "r3n_skinning")
    for k in range(3): step(k)
    r.sync(); r.timing_enable(True); r.stage_times(reset=True)
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    r.sync(); wall = (time.perf_counter() - t0) / steps
    st = r.stage_times(reset=True)
    pose_ms, skin_ms = st["pose"][0] / steps, st["skinning"][0] / steps
    print(json.dumps({"config": "configs[4] 34-joint asset (animation example character) x instances", "skeletons": len(sks),
                      "joints_per_skeleton": joints, "vertices": verts, "pose_kernel_ms": round(pose_ms, 4),
                      "skinning_kernel_ms": round(skin_ms, 4), "wall_ms_per_frame": round(1e3 * wall, 3),
                      "joint_matrices_per_s": round(len(sks) * joints / (pose_ms * 1e-3)),
                      "vertices_per_s": round(verts / (skin_ms * 1e-3)),
                      "skinning_algorithmic_GBps": round(verts * 96 / (skin_ms * 1e-3) / 1e9, 1)}))
    r.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg2", "cfg4", "cfg5"]
    if "cfg5" in which:
        run_cfg5()
    if "cfg5mfma" in which:
        run_cfg5(mfma=True)
    if "cfg5anim" in which:
        run_cfg5_anim()
    if "cfg5asset" in which:
        run_cfg5_asset()
    if "cfg2" in which:
        run("configs[1] scifi_like", lambda r: S.scifi_like(r, r3.host, r3.material_record), 1920, 1080, True)
    if "cfg4" in which:
        run("configs[3] emerald_like", lambda r: S.emerald_like(r, r3.host, r3.material_record), 3840, 2160, False)
