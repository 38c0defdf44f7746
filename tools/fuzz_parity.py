#!/usr/bin/env python3
"""GPU box: randomised parity campaign -- the HIP path against the oracle on scenes the fixed tests do not hold.

Every case draws its own parameters from its seed: handedness, an arbitrary (often odd) target size, 1 or 4 samples, the scene
builder (factor-only with optional cutout materials, or textured in the RGBA8 / block-compressed / float-decoded formats), object
count, 0-3 directional lights with shadow views of different sizes, point lights, translucent objects, perspective (random field
of view / near plane) or orthographic projection, a random eye and target; then three frames with a moving camera, one object
moved, one removed and one added in between, so the temporal two-pass culling, Hi-Z and the growth / removal paths engage.  The
comparison is tests/test_gpu_parity.py::compare_frames (sets, keys, atlas, HDR bit-exact; framebuffer within 1e-3).

    python tools/fuzz_parity.py --seconds 240 --first-seed 1000        # prints one line per case, a summary, exit code 1 on a mismatch
"""
import argparse
import math
import os
import sys
import time
import traceback

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "32")  # the oracle's scenes here are a few hundred rows: a 256-thread team per loop only costs fork / join time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import scenes  # noqa: E402
from oracle import host as oh  # noqa: E402
from oracle.world import OracleRenderer  # noqa: E402
from oracle.world import material_record as omk  # noqa: E402
from test_gpu_parity import compare_frames  # noqa: E402

f32 = np.float32


def draw_case(seed):
    rng = scenes.Pcg32(seed ^ 0x5EED5EED)
    c = {"seed": seed}
    c["handedness"] = oh.LEFT if rng.uniform() < 0.5 else oh.RIGHT
    c["w"] = 48 + rng.randint(340)
    c["h"] = 40 + rng.randint(220)
    c["samples"] = 4 if rng.uniform() < 0.3 else 1
    kind = rng.randint(5)
    c["builder"] = ("random", "random_cutout", "textured", "textured_encoded", "textured_float")[kind]
    c["objects"] = 20 + rng.randint(260)
    c["lights"] = rng.randint(4)
    c["shadow_res"] = (64, 128, 256, 512)[rng.randint(4)]
    c["shadow_distance"] = rng.uniform(15.0, 80.0)
    c["point_lights"] = rng.randint(3)
    c["blend"] = rng.uniform() < 0.35
    c["ortho"] = rng.uniform() < 0.15
    c["vfov"] = rng.uniform(25.0, 110.0)
    c["near"] = (0.01, 0.1, 0.5, 2.0)[rng.randint(4)]
    c["eye"] = (rng.uniform(-12, 12), rng.uniform(-2, 6), rng.uniform(-12, 12))
    c["target"] = (rng.uniform(-10, 10), rng.uniform(-3, 3), rng.uniform(-10, 10))
    c["ambient"] = (0.1, 0.1, 0.1, 1.0) if rng.uniform() < 0.5 else (0.0, 0.0, 0.0, 0.0)
    c["move"] = (rng.uniform(-4, 4), rng.uniform(-1, 2), rng.uniform(-4, 4))
    return c


def build(r, hm, mk, c):
    b = c["builder"]
    if b.startswith("random"):
        hs = scenes.build_random_scene(r, hm, mk, c["objects"], c["seed"], handedness=c["handedness"], lights=c["lights"], shadow_res=c["shadow_res"],
                                       shadow_distance=c["shadow_distance"], with_cutout=b.endswith("cutout"))
    else:
        enc = {"textured": False, "textured_encoded": True, "textured_float": "float"}[b]
        hs = scenes.build_textured_scene(r, hm, mk, min(c["objects"], 120), c["seed"], handedness=c["handedness"], lights=c["lights"],
                                         shadow_res=c["shadow_res"], shadow_distance=c["shadow_distance"], encoded=enc)
    prng = scenes.Pcg32(c["seed"] + 77)
    for _ in range(c["point_lights"]):
        r.add_point_light((prng.uniform(-8, 8), prng.uniform(0, 4), prng.uniform(-8, 8)), (prng.uniform(0.2, 1), prng.uniform(0.2, 1), prng.uniform(0.2, 1)),
                          prng.uniform(1.0, 8.0), prng.uniform(2.0, 12.0))
    if c["blend"]:
        scenes.add_blend_objects(r, hm, mk, c["seed"] + 5, n=8, textured=not b.startswith("random"))
    return hs


def camera(c, f):
    look = oh.look_at_lh if c["handedness"] == oh.LEFT else oh.look_at_rh
    eye = tuple(e + 0.35 * f * m for e, m in zip(c["eye"], c["move"]))
    tgt = c["target"]
    if max(abs(a - b) for a, b in zip(eye, tgt)) < 0.5:
        tgt = (tgt[0] + 3.0, tgt[1], tgt[2] + 3.0)
    view = look(eye, tgt, (0, 1, 0))
    proj = ("orthographic", (12.0, 12.0 * c["h"] / c["w"], 60.0)) if c["ortho"] else ("perspective", c["vfov"], c["near"])
    return view, proj


def run_case(r3, c):
    o = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
    p = r3.Renderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
    try:
        ho = build(o, oh, omk, c)
        hp = build(p, r3.host, r3.material_record, c)
        for f in range(3):
            view, proj = camera(c, f)
            for r in (o, p):
                r.set_camera_data(view, proj)
            if f == 1 and len(ho) > 8:
                for r, hs in ((o, ho), (p, hp)):
                    r.set_object_transform(hs[3], oh.mat4_mul(oh.translation(c["move"]), oh.scale((1.5, 0.5, 2.0))))
                    r.remove_object(hs[6])
            if f == 2 and len(ho) > 8:
                for r, hs, mk in ((o, ho, omk), (p, hp, r3.material_record)):
                    r.add_object(scenes.cube_mesh(r), scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), oh.translation(tuple(0.5 * t for t in c["target"])))
            kw = dict(samples=c["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
            t1 = time.time()
            fo = o.render(c["w"], c["h"], **kw)
            t2 = time.time()
            fp = p.render(c["w"], c["h"], **kw)
            if os.environ.get("FUZZ_TIMES"):
                print(f"  frame {f}: oracle {t2 - t1:.2f} s, HIP (with read-back) {time.time() - t2:.2f} s", flush=True)
            compare_frames(fo, fp, f"frame {f}")
        return int(fo["visible"].sum()), int(fo["pass"].sum()), int((fo["vis"] != 0).sum())
    finally:
        p.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--first-seed", type=int, default=1000)
    ap.add_argument("--max-cases", type=int, default=10000)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a drawn parameter (bisecting a case), e.g. --set blend=0 --set lights=1")
    a = ap.parse_args()
    import rend3_amd as r3
    t0 = time.time()
    ok, bad, covered = 0, [], {}
    seed = a.first_seed
    while time.time() - t0 < a.seconds and ok + len(bad) < a.max_cases:
        c = draw_case(seed)
        for kv in a.set:
            k, v = kv.split("=", 1)
            c[k] = type(c[k])(int(v)) if isinstance(c[k], (bool, int)) else (float(v) if isinstance(c[k], float) else v)
        tag = f"seed {seed}: {'LH' if c['handedness'] == oh.LEFT else 'RH'} {c['w']}x{c['h']} s{c['samples']} {c['builder']} objects {c['objects']} lights {c['lights']}@{c['shadow_res']} " \
              f"points {c['point_lights']} blend {int(c['blend'])} {'ortho' if c['ortho'] else 'vfov %.0f near %g' % (c['vfov'], c['near'])}"
        t1 = time.time()
        try:
            vis, tris, px = run_case(r3, c)
            ok += 1
            covered[c["builder"]] = covered.get(c["builder"], 0) + 1
            print(f"ok   {tag}: visible objects {vis}, pass triangles {tris}, covered samples {px} ({time.time() - t1:.1f} s)", flush=True)
        except AssertionError as e:
            bad.append((seed, str(e)))
            print(f"FAIL {tag}: {e}", flush=True)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, repr(e)))
            print(f"ERR  {tag}: {e!r}\n{traceback.format_exc()}", flush=True)
        seed += 1
    print(f"\n{ok} cases bit-exact, {len(bad)} failed, seeds {a.first_seed}..{seed - 1}, {time.time() - t0:.0f} s; cases per builder: {covered}")
    for s, why in bad:
        print(f"  seed {s}: {why[:300]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
