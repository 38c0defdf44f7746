"""Row J: the scene-viewer harness (rend3_amd/scene_viewer.py, tools/scene_viewer.py, bench.py --scene) -- the reference's
examples/src/scene_viewer/mod.rs:336-751 over the C ABI -- on the three real assets the reference tree ships
(tests/golden/: static_gltf data.glb, skinning RiggedSimple.glb, the animation example's character).  CPU: flag parsing, the
camera formula, the oracle through the same builder.  GPU: HIP == oracle through the harness, and bench.py --scene end to end."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import host as oh
from oracle.world import OracleRenderer
from oracle.world import material_record as omk
from rend3_amd import scene_viewer as sv

f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ASSETS = {
    # file, flags (the assets carry no lights of their own), camera x,y,z,pitch,yaw
    "static_gltf": ("static_gltf-data.glb", ["--directional-light", "-1,-4,2", "--directional-light-intensity", "4", "--shadow-distance", "20",
                                               "--camera", "3,3,5,-0.55,-0.5"]),
    "skinning": ("skinning-RiggedSimple.glb", ["--directional-light", "-1,-4,2", "--directional-light-intensity", "10", "--shadow-distance", "40",
                                                "--camera", "0,0,14,0,0", "--ambient", "0.2"]),
    "animation": ("animation-character.glb", ["--directional-light", "-1,-4,2", "--directional-light-intensity", "5", "--shadow-distance", "30",
                                               "--camera", "0,1.5,5,0,0", "--msaa", "4"]),
}


def settings_for(name, extra=()):
    file, flags = ASSETS[name]
    ap = sv.add_arguments(argparse.ArgumentParser())
    ap.add_argument("file")
    return sv.settings_from(ap.parse_args(sv.normalize_argv([os.path.join(GOLD, file)] + list(flags) + list(extra))))


def test_flags_and_defaults_follow_the_reference():
    """SceneViewer::default + from_args (mod.rs:300-431) and the Bistro test's flags (:727-751)."""
    ap = sv.add_arguments(argparse.ArgumentParser())
    ap.add_argument("file")
    d = sv.settings_from(ap.parse_args(["x.glb"]))
    assert d["samples"] == 1 and d["ambient"] == 0.1 and d["scale"] == 1.0 and d["shadow_distance"] == 100.0 and d["shadow_resolution"] == 2048
    assert d["enable_directional"] and not d["normal_y_down"] and d["directional_light"] is None and d["directional_light_intensity"] == 1.0
    assert d["camera"] == (-2.9936655, 2.189423, 5.308956, -0.08869916, 5.899576)
    b = sv.settings_from(ap.parse_args(sv.normalize_argv(sv.BISTRO_FLAGS + ["bistro.gltf"])))
    assert b["samples"] == 4 and b["normal_y_down"] and not b["enable_directional"] and b["directional_light"] == (1.0, -5.0, -1.0)
    assert b["directional_light_intensity"] == 15.0 and b["camera"] == (-17.174278, 3.715882, -4.631997, 0.04430086, 4.6065736)


def test_camera_view_formula():
    """handle_redraw (mod.rs:640-641): view = from_euler(XYZ, -pitch, -yaw, 0) * T(-location); identical through both host mirrors."""
    from rend3_amd import host as ph
    cam = (-17.174278, 3.715882, -4.631997, 0.04430086, 4.6065736)
    v = sv.camera_view(oh, cam)
    want = oh.mat4_mul(oh.from_euler_xyz(f32(-0.04430086), f32(-4.6065736), f32(0.0)), oh.translation((17.174278, -3.715882, 4.631997)))
    assert np.array_equal(v.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(sv.camera_view(ph, cam).view(np.uint32), v.view(np.uint32))
    loc = oh.mat4_inverse(v)[12:15]
    assert np.allclose(loc, cam[:3], atol=1e-4)


@pytest.mark.parametrize("name", sorted(ASSETS))
def test_oracle_renders_the_assets_through_the_harness(name):
    s = settings_for(name)
    w, h = 160, 90
    o = OracleRenderer(oh.RIGHT, f32(w) / f32(h))
    info = sv.build(o, oh, omk, s)
    assert info["objects"] >= 1 and info["triangles"] > 100 and len(o.dir_lights) == 1
    out = o.render(w, h, samples=info["samples"], ambient=info["ambient"], clear_color=info["clear"])
    covered = (out["vis"] != 0).reshape(h, w, -1).any(axis=2)
    assert 0.02 < covered.mean() < 0.9, covered.mean()          # the asset is in view, not filling it
    assert out["rgba8"][covered][:, :3].max() > 60               # and lit
    assert (out["atlas"] != 0).any()                             # and casts into its shadow view


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ASSETS))
def test_gpu_assets_through_the_harness_match_the_oracle(name):
    """The same settings build the oracle's world and the HIP renderer's; three frames along bench.py's camera dolly (history,
    Hi-Z cull, residual pass), everything compare_frames checks -- sets, keys, atlas, HDR bit-identical."""
    import torch
    assert torch.cuda.is_available()
    import bench
    import rend3_amd as r3
    from test_gpu_parity import compare_frames
    s = settings_for(name)
    w, h = 640, 360
    o, p = OracleRenderer(oh.RIGHT, f32(w) / f32(h)), r3.Renderer(oh.RIGHT, f32(w) / f32(h))
    io, ip = sv.build(o, oh, omk, s), sv.build(p, r3.host, r3.material_record, s)
    assert io["triangles"] == ip["triangles"] and io["objects"] == ip["objects"]
    if name == "animation":  # posed by rend3-anim: the oracle on the host, the product on the GPU
        from oracle import anim as oa
        from rend3_amd import anim as pa
        from rend3_amd.gltf import load_animations
        anims = load_animations(io["gltf"])
        data = pa.AnimationData.from_gltf_scene(p, load_animations(ip["gltf"]), ip["instance"])
        oa.pose_animation_frame(o, io["instance"], anims, 0, 1.25)
        pa.pose_animation_frame(p, ip["instance"], data, 0, 1.25)
    for k in range(3):
        o.set_camera_data(bench.camera_path(oh, io["camera"][0], k), io["camera"][1])
        p.set_camera_data(bench.camera_path(r3.host, ip["camera"][0], k), ip["camera"][1])
        kw = dict(samples=io["samples"], ambient=io["ambient"], clear_color=io["clear"])
        fo, fp = o.render(w, h, **kw), p.render(w, h, **kw)
        compare_frames(fo, fp, f"scene viewer {name} frame {k}")
        assert (fo["vis"] != 0).sum() > 500 and fo["pass"].sum() > 50
    p.close()


@pytest.mark.gpu
def test_gpu_bench_scene_line():
    """bench.py --scene on a real asset: the same JSON line as the synthetic workload (data "asset"), with the parity verdict of
    the benchmarked frame and the CPU baseline of the same asset."""
    file, flags = ASSETS["static_gltf"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", os.path.join(GOLD, file), "--resolution", "1280x720", "--steps", "6",
           "--warmup", "2", "--cpu-sample-frames", "2"] + flags
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["data"] == "asset" and "static_gltf-data.glb" in d["config"]["workload"] and d["n_gpus"] == 1 and d["steps"] == 6
    assert d["parity"]["ok"], d["parity"]
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_gpu_scene_viewer_tool(tmp_path):
    """tools/scene_viewer.py end to end: loads the asset, renders, writes the PNG."""
    from PIL import Image
    file, flags = ASSETS["skinning"]
    out = str(tmp_path / "view.png")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scene_viewer.py"), os.path.join(GOLD, file), "--resolution", "320x180",
                          "--out", out, "--json"] + flags, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert d["objects"] == 1 and d["triangles"] == 188 and d["covered_px"] > 100
    img = np.array(Image.open(out))
    assert img.shape == (180, 320, 4) and img[..., :3].max() > 60
