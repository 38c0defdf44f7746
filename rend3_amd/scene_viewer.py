"""The reference's scene-viewer example as a harness over the C ABI: load a glTF / GLB file, instance it, add the flagged
directional light, place the camera -- the inputs the named BASELINE.json configs are quoted on (scifi-base.glb, Bistro, Emerald
Square: `examples/src/scene_viewer`).  Follows (reference file:line):

  SceneViewer::default / from_args        examples/src/scene_viewer/mod.rs:300-330, 336-431 (flags and their defaults)
  setup: the flagged light + load_gltf    :463-520  (light: colour 1, `--directional-light-intensity`, distance =
                                           `--shadow-distance`, resolution 2048)
  handle_redraw: camera, settings         :640-646 (view = euler XYZ(-pitch, -yaw, 0) * T(-location), Perspective{60, 0.1}),
                                           :678-681 (ambient = (a, a, a, 1), clear (0, 0, 0, 1))
  App::HANDEDNESS = Right                 :434
  the Bistro test                          :727-751 (flags + camera of BASELINE.json configs[2])

`build(renderer, host_module, material_record, settings)` works on anything with the Renderer's world-edit API -- the HIP
renderer and, in the tests / bench.py's cpu_baseline leg, the oracle -- so a real asset runs through exactly the parity and
measurement code the synthetic stand-ins do.  This module never imports the oracle.
"""
import argparse
import os

import numpy as np

RIGHT = 1


def _vec3(s):
    v = [float(x) for x in s.split(",")]
    if len(v) != 3:
        raise argparse.ArgumentTypeError("expected x,y,z")
    return tuple(v)


def _camera(s):
    v = [float(x) for x in s.split(",")]
    if len(v) != 5:
        raise argparse.ArgumentTypeError("expected x,y,z,pitch,yaw")
    return tuple(v)


VECTOR_FLAGS = ("--directional-light", "--camera")


def normalize_argv(argv):
    """`--directional-light -1,-4,2` as the reference's parser (pico-args) accepts it: argparse would take the value for a flag
    because it starts with '-', so flag and value are joined with '=' first."""
    out, i = [], 0
    argv = list(argv)
    while i < len(argv):
        if argv[i] in VECTOR_FLAGS and i + 1 < len(argv):
            out.append(argv[i] + "=" + argv[i + 1])
            i += 2
        else:
            out.append(argv[i])
            i += 1
    return out


def add_arguments(ap):
    """The scene-viewer flags that reach the hot path (mod.rs:355-405); windowing / backend / control flags have no meaning here."""
    ap.add_argument("--msaa", type=int, choices=(1, 4), default=1, help="SampleCount (mod.rs:352: --msaa)")
    ap.add_argument("--normal-y-down", action="store_true", help="NormalTextureYDirection::Down (Bistro)")
    ap.add_argument("--directional-light", type=_vec3, default=None, metavar="X,Y,Z", help="add a directional light with this direction")
    ap.add_argument("--directional-light-intensity", type=float, default=1.0)
    ap.add_argument("--ambient", type=float, default=0.1, help="ambient light level (default 0.1)")
    ap.add_argument("--scale", type=float, default=1.0, help="GltfLoadSettings::scale")
    ap.add_argument("--shadow-distance", type=float, default=100.0, help="GltfLoadSettings::directional_light_shadow_distance")
    ap.add_argument("--shadow-resolution", type=int, default=2048, help="GltfLoadSettings::directional_light_resolution (lights of the file)")
    ap.add_argument("--gltf-disable-directional-lights", action="store_true", help="ignore KHR_lights_punctual lights of the file")
    ap.add_argument("--camera", type=_camera, default=None, metavar="X,Y,Z,PITCH,YAW",
                    help="camera location and angles (default: the default scene's, mod.rs:320-322)")
    return ap


# SceneViewer::default(): camera of the default scene
DEFAULT_CAMERA = (-2.9936655, 2.189423, 5.308956, -0.08869916, 5.899576)
# examples/src/scene_viewer/mod.rs:727-751: the Bistro test's flags
BISTRO_FLAGS = ["--msaa", "4", "--normal-y-down", "--gltf-disable-directional-lights", "--directional-light", "1,-5,-1",
                "--directional-light-intensity", "15", "--camera", "-17.174278,3.715882,-4.631997,0.04430086,4.6065736"]


def settings_from(args):
    """argparse namespace (add_arguments) -> plain settings dict."""
    return dict(file=getattr(args, "scene", None) or getattr(args, "file", None), samples=args.msaa, normal_y_down=args.normal_y_down,
                directional_light=args.directional_light, directional_light_intensity=args.directional_light_intensity,
                ambient=args.ambient, scale=args.scale, shadow_distance=args.shadow_distance,
                shadow_resolution=args.shadow_resolution, enable_directional=not args.gltf_disable_directional_lights,
                camera=args.camera or DEFAULT_CAMERA)


def default_settings(**over):
    s = settings_from(add_arguments(argparse.ArgumentParser()).parse_args([]))
    s.update(over)
    return s


def camera_view(hm, camera):
    """handle_redraw (mod.rs:640-641): view = Mat4::from_euler(XYZ, -pitch, -yaw, 0) * T(-location)."""
    x, y, z, pitch, yaw = (np.float32(v) for v in camera)
    return hm.mat4_mul(hm.from_euler_xyz(-pitch, -yaw, np.float32(0.0)), hm.translation((-x, -y, -z)))


PROJECTION = ("perspective", 60.0, 0.1)  # mod.rs:645
CLEAR = (0.0, 0.0, 0.0, 1.0)             # mod.rs:681


def build(r, hm, mk, settings):
    """setup + the first handle_redraw of the example on renderer `r` (right-handed): the flagged light, the file through
    rend3-gltf's load + instance path (rend3_amd/gltf.py), rend3-anim tables when the file has animations and the renderer
    poses on the GPU, the camera.  Returns dict(instance, animations, camera=(view, projection), ambient, clear, samples,
    objects, triangles)."""
    from . import gltf
    assert r.handedness == RIGHT, "scene_viewer is right-handed (App::HANDEDNESS, mod.rs:434)"
    light = None
    if settings["directional_light"] is not None:  # setup (mod.rs:463-472)
        light = r.add_directional_light(color=(1.0, 1.0, 1.0), intensity=settings["directional_light_intensity"],
                                        direction=settings["directional_light"], distance=settings["shadow_distance"], resolution=2048)
    g = gltf.Gltf(settings["file"])
    inst = gltf.instance_scene(g, r, hm, mk, scale=settings["scale"], enable_directional=settings["enable_directional"],
                               directional_light_shadow_distance=settings["shadow_distance"],
                               directional_light_resolution=settings["shadow_resolution"], normal_y_down=settings["normal_y_down"])
    view = camera_view(hm, settings["camera"])
    r.set_camera_data(view, PROJECTION)
    a = settings["ambient"]
    meshes = r.meshes
    tris = int(sum(meshes[m["mesh"]].index_count // 3 for m in r.object_meta.values() if m["enabled"]))
    return dict(instance=inst, gltf=g, light=light, camera=(view, PROJECTION), ambient=(a, a, a, 1.0), clear=CLEAR,
                samples=settings["samples"], objects=len(inst["objects"]), triangles=tris, file=os.path.basename(settings["file"]))
