#!/usr/bin/env python3
"""scene_viewer: the reference's examples/src/scene_viewer (mod.rs:336-751) over the MI355X path.

  python tools/scene_viewer.py FILE.glb|FILE.gltf [--msaa 4] [--directional-light x,y,z --directional-light-intensity f]
         [--shadow-distance d] [--shadow-resolution r] [--ambient a] [--scale s] [--camera x,y,z,pitch,yaw]
         [--gltf-disable-directional-lights] [--normal-y-down] [--bistro]
         [--resolution WxH] [--frames N] [--time t] [--out image.png] [--json]

Loads the file through the product's rend3-gltf path, renders `--frames` frames (the first ones build the temporal history the
two-pass culling wants) and writes the last one as a PNG; --json prints objects / triangles / per-frame time.  `--bistro` sets
the flags of the reference's Bistro test (mod.rs:727-751).  `bench.py --scene FILE ...` runs the same scene through the
benchmark (timed frames, roofline, parity against the CPU oracle, cpu_baseline)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    from rend3_amd import scene_viewer as sv
    ap = sv.add_arguments(argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter))
    ap.add_argument("file")
    ap.add_argument("--bistro", action="store_true", help="the flags of the reference's Bistro test (camera, light, MSAA x4, normal Y down)")
    ap.add_argument("--resolution", default="1280x720")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--time", type=float, default=None, help="pose the file's first animation at this time (rend3-anim, on the GPU)")
    ap.add_argument("--out", default=None, help="PNG of the last frame")
    ap.add_argument("--json", action="store_true")
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--bistro" in argv:
        argv = sv.BISTRO_FLAGS + argv  # explicit flags after them win
    args = ap.parse_args(sv.normalize_argv(argv))
    w, h = (int(v) for v in args.resolution.lower().split("x"))
    import numpy as np
    import rend3_amd as r3
    settings = sv.settings_from(args)
    r = r3.Renderer(r3.host.RIGHT, np.float32(w) / np.float32(h))
    t0 = time.perf_counter()
    info = sv.build(r, r3.host, r3.material_record, settings)
    load_s = time.perf_counter() - t0
    if args.time is not None:
        from rend3_amd import anim, gltf
        anims = gltf.load_animations(info["gltf"])
        if anims:
            data = anim.AnimationData.from_gltf_scene(r, anims, info["instance"])
            anim.pose_animation_frame(r, info["instance"], data, 0, np.float32(args.time))
    times = []
    out = None
    for f in range(max(args.frames, 1)):
        last = f == max(args.frames, 1) - 1
        t0 = time.perf_counter()
        out = r.render(w, h, samples=info["samples"], ambient=info["ambient"], clear_color=info["clear"], readback=last and bool(args.out or args.json))
        r.sync()
        times.append(time.perf_counter() - t0)
    if args.out:
        from PIL import Image
        Image.fromarray(out["rgba8"]).save(args.out)
    if args.json:
        print(json.dumps({"file": info["file"], "objects": info["objects"], "triangles": info["triangles"], "resolution": [w, h],
                          "samples": info["samples"], "load_s": round(load_s, 3), "frame_ms": [round(1e3 * t, 3) for t in times],
                          "covered_px": int((out["vis"] != 0).sum()) if out is not None else None, "out": args.out}))
    r.close()


if __name__ == "__main__":
    main()
