#!/usr/bin/env python3
"""GPU box: randomised parity campaign -- the HIP path against the oracle on scenes the fixed tests do not hold.

Every case draws its own parameters from its seed: handedness, an arbitrary (often odd) target size, 1 or 4 samples, the scene
builder (factor-only with optional cutout materials, or textured in the RGBA8 / block-compressed / float-decoded formats), object
count, 0-3 directional lights with shadow views of different sizes, point lights, translucent objects, perspective (random field
of view / near plane) or orthographic projection, a random eye and target; then three frames with a moving camera, one object
moved, one removed and one added in between, so the temporal two-pass culling, Hi-Z and the growth / removal paths engage.  The
comparison is tests/test_gpu_parity.py::compare_frames (sets, keys, atlas, HDR bit-exact; framebuffer within 1e-3).

--mutate: five frames and, between them, up to four world edits drawn from the case's seed -- objects moved / removed / added
one by one or in bulk (the object buffer doubles, freed handles are reused), a material rewritten or moved to ANOTHER TRANSPARENCY
KEY (the frame after is compared like every other: tests/test_key_flip.py), directional lights turned / resized / added (the shadow atlas is laid out again), point lights moved / added, the target
resized, the sample count switched, new meshes / materials / textures created between frames.

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


class oracle_threads:
    """`with oracle_threads(32):` -- the oracle's OpenMP team for the block.  OMP_NUM_THREADS only counts when it is set before the
    OpenMP runtime loads (inside a pytest session torch has loaded it long before); omp_set_num_threads works at any time.  The
    cases here are a few hundred rows: on the GPU box's 256 hardware threads a team per loop costs more in fork / join than it
    saves (0.25 s per case instead of 0.05)."""

    def __init__(self, n):
        self.n, self.old, self.omp = n, None, None

    def __enter__(self):
        import ctypes
        try:
            self.omp = ctypes.CDLL("libgomp.so.1")
            self.old = self.omp.omp_get_max_threads()
            self.omp.omp_set_num_threads(min(self.n, self.old))
        except OSError:
            self.omp = None
        return self

    def __exit__(self, *exc):
        if self.omp is not None and self.old:
            self.omp.omp_set_num_threads(self.old)
        return False


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


def mutate(rng, c, st, pair, f):
    """Apply up to four random world edits to BOTH renderers (pair = ((renderer, host module, material_record), ...));
    st: the case's mutable state (live handles, own material / mesh handles, target size, samples).  Returns what it did."""
    done = []
    for _ in range(rng.randint(5)):
        kind = rng.randint(15)
        live = st["live"]
        if kind == 0 and live:  # move
            h = live[rng.randint(len(live))]
            pos = (rng.uniform(-10, 10), rng.uniform(-2, 4), rng.uniform(-10, 10))
            sc = math.exp(rng.uniform(math.log(0.3), math.log(3.0)))
            ang = rng.uniform(0, 6.28)
            for r, hm, _mk in pair:
                r.set_object_transform(h, hm.mat4_mul(hm.mat4_mul(hm.translation(pos), hm.rotation_y(ang)), hm.scale((sc, sc, sc))))
            done.append(f"move {h}")
        elif kind == 1 and len(live) > 4:  # remove
            h = live.pop(rng.randint(len(live)))
            for r, _hm, _mk in pair:
                r.remove_object(h)
            done.append(f"remove {h}")
        elif kind in (2, 3):  # add one (near the target, so it is usually in view)
            pos = tuple(0.5 * t + rng.uniform(-3, 3) for t in c["target"])
            sc = rng.uniform(0.3, 2.5)
            hs = [r.add_object(st["mesh"][i][kind - 2], st["mat"][i][0],
                               hm.mat4_mul(hm.translation(pos), hm.scale((sc, sc, sc)))) for i, (r, hm, _mk) in enumerate(pair)]
            assert hs[0] == hs[1], f"handles diverge: {hs}"
            live.append(hs[0])
            done.append(f"add {hs[0]}")
        elif kind == 4:  # add in bulk: the object buffer grows, freed handles are reused first
            n = 1 + rng.randint(max(len(live), 1) + 40)
            xs = np.zeros((n, 16), dtype=f32)
            for k in range(n):
                pos = (rng.uniform(-14, 14), rng.uniform(-3, 5), rng.uniform(-14, 14))
                sc = rng.uniform(0.2, 1.5)
                xs[k] = np.asarray(oh.mat4_mul(oh.translation(pos), oh.scale((sc, sc, sc))), dtype=f32).reshape(16)
            which = rng.randint(2)
            hs = [r.add_objects_bulk([st["mesh"][i][which]] * n, [st["mat"][i][k % 2] for k in range(n)], xs) for i, (r, _hm, _mk) in enumerate(pair)]
            assert list(hs[0]) == list(hs[1]), "handles diverge (bulk)"
            live.extend(int(h) for h in hs[0])
            done.append(f"bulk +{n}")
        elif kind == 5:  # rewrite a material (colour, roughness) in place, its transparency key kept (key changes: flip_key below)
            col = (rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), 1.0)
            rough = rng.uniform(0.2, 0.9)
            for i, (r, _hm, mk) in enumerate(pair):
                r.update_material(st["mat"][i][1], mk(albedo=col, albedo_mode="value", roughness=rough))
            done.append("material rewritten")
        elif kind == 6 and st["dir"] > 0:  # turn / resize a directional light
            h = rng.randint(st["dir"])
            ch = dict(direction=(rng.uniform(-1, 1), -rng.uniform(0.5, 3.0), rng.uniform(-1, 1)))
            if rng.uniform() < 0.5:
                ch["resolution"] = (64, 128, 256, 512)[rng.randint(4)]
            if rng.uniform() < 0.3:
                ch["distance"] = rng.uniform(10.0, 90.0)
            for r, _hm, _mk in pair:
                r.update_directional_light(h, **ch)
            done.append(f"light {h} {sorted(ch)}")
        elif kind == 7 and st["dir"] < 4:  # one more shadow view: the atlas is laid out again
            ch = dict(color=(1, 1, 1), intensity=rng.uniform(0.5, 4.0), direction=(rng.uniform(-1, 1), -rng.uniform(0.5, 3.0), rng.uniform(-1, 1)),
                      distance=rng.uniform(15.0, 70.0), resolution=(64, 128, 256, 512)[rng.randint(4)])
            for r, _hm, _mk in pair:
                r.add_directional_light(**ch)
            st["dir"] += 1
            done.append("light added")
        elif kind == 8:  # point lights
            pos = (rng.uniform(-8, 8), rng.uniform(0, 4), rng.uniform(-8, 8))
            if st["point"] and rng.uniform() < 0.5:
                h = rng.randint(st["point"])
                for r, _hm, _mk in pair:
                    r.update_point_light(h, position=pos)
            else:
                intensity, radius = rng.uniform(1.0, 6.0), rng.uniform(2.0, 10.0)
                for r, _hm, _mk in pair:
                    r.add_point_light(pos, (1.0, 0.8, 0.6), intensity, radius)
                st["point"] += 1
            done.append("point light")
        elif kind == 9:  # resize the target
            st["w"], st["h"] = 48 + rng.randint(340), 40 + rng.randint(220)
            done.append(f"resize {st['w']}x{st['h']}")
        elif kind == 10 and rng.uniform() < 0.5:
            st["samples"] = 5 - st["samples"]  # 1 <-> 4
            done.append(f"samples {st['samples']}")
        elif kind in (11, 12, 13):  # a NEW mesh / material / texture between frames (the mesh buffer, the material table and the texel
            # pool grow while the previous frame's resolve may still be reading them) and an object that uses it
            pos = tuple(0.5 * t + rng.uniform(-3, 3) for t in c["target"])
            sc = rng.uniform(0.4, 2.0)
            col = (rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), 1.0)
            sub = rng.randint(3)
            tex_seed = c["seed"] * 31 + f * 7 + len(done)
            hs = []
            for i, (r, hm, mk) in enumerate(pair):
                mesh, mat = st["mesh"][i][0], st["mat"][i][0]
                if kind in (11, 13):
                    mp, mi, mn = scenes.icosphere(sub)
                    if c["handedness"] == oh.LEFT:
                        mi = mi.reshape(-1, 3)[:, ::-1].reshape(-1)
                    uv = (mp[:, :2] * np.float32(1.5) + np.float32(0.5)).astype(np.float32)
                    mesh = r.add_mesh(mp, mi, normals=mn, uv0=uv)
                if kind == 12:
                    mat = scenes.lit(r, mk, col)
                if kind == 13:
                    img = np.random.default_rng(tex_seed).integers(0, 256, (16 << sub, 16, 4), dtype=np.uint8)
                    t = r.add_texture_2d(img, srgb=True, mip_count="maximum", mip_source="generated")
                    mat = r.add_material(mk(albedo_mode="texture_value", albedo_texture=t, albedo=col, roughness=0.5), scenes.OPAQUE)
                hs.append(r.add_object(mesh, mat, hm.mat4_mul(hm.translation(pos), hm.scale((sc, sc, sc)))))
            assert hs[0] == hs[1], f"handles diverge: {hs}"
            live.append(hs[0])
            done.append(("mesh", "newmat", "texture")[kind - 11] + f" {hs[0]}")
        elif kind == 14:
            done.append(flip_key(st, pair))
    if st["flip_rng"].uniform() < 0.3:
        done.append(flip_key(st, pair))
    return done


def flip_key(st, pair):
    """Renderer::update_material with another transparency (renderer/mod.rs:256-266; the archetype stays: material.rs:163-188): ANY
    material of the scene moves to a drawn key, alpha_cutout following it (pbr/material.rs:562-565).  The frame after is the one in
    which last frame's predicted triangles sit in the OLD key's draw range (tests/test_key_flip.py).  Drawn from a stream of its
    own, so the schedule of the other edits is what it was before flips joined the campaign."""
    rng = st["flip_rng"]
    o = pair[0][0]
    m = rng.randint(len(o.materials))
    key = (o.materials[m][1] + 1 + rng.randint(2)) % 3
    cutout = rng.uniform(0.1, 0.9) if key == scenes.CUTOUT else 0.0
    for r, _hm, _mk in pair:
        rec = np.array(r.materials[m][0], dtype=f32, copy=True)
        rec[50] = f32(cutout)
        r.update_material(m, rec, key=key)
    return f"material {m} -> key {key}"


def run_mutating_case(r3, c):
    o = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
    p = r3.Renderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
    log = []
    try:
        ho = build(o, oh, omk, c)
        hp = build(p, r3.host, r3.material_record, c)
        assert ho == hp
        pair = ((o, oh, omk), (p, r3.host, r3.material_record))
        st = dict(live=list(ho), w=c["w"], h=c["h"], samples=c["samples"], dir=c["lights"], point=c["point_lights"], mesh=[], mat=[])
        for r, hm, mk in pair:
            pos, idx, nrm = scenes.icosphere(1)
            if c["handedness"] == oh.LEFT:
                idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1)
            st["mesh"].append((scenes.cube_mesh(r), r.add_mesh(pos, idx, normals=nrm)))
            st["mat"].append((scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), scenes.lit(r, mk, (0.3, 0.6, 0.9, 1.0))))
        rng = scenes.Pcg32(c["seed"] * 7919 + 13)
        st["flip_rng"] = scenes.Pcg32(c["seed"] * 104729 + 71)
        for f in range(5):
            view, proj = camera(c, f)
            for r in (o, p):
                r.set_camera_data(view, proj)
            if f:
                log.append((f, mutate(rng, c, st, pair, f)))
            kw = dict(samples=st["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
            fo = o.render(st["w"], st["h"], **kw)
            fp = p.render(st["w"], st["h"], **kw)
            compare_frames(fo, fp, f"frame {f} after {log}")
        return int(fo["visible"].sum()), int(fo["pass"].sum()), int((fo["vis"] != 0).sum())
    finally:
        p.close()


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
    ap.add_argument("--mutate", action="store_true", help="five frames with random world edits in between (see the module docstring)")
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
            vis, tris, px = (run_mutating_case if a.mutate else run_case)(r3, c)
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
