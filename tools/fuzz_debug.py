#!/usr/bin/env python3
"""GPU box: one case of tools/fuzz_parity.py in detail -- after every frame, which of the frame's read-backs differ between the oracle
and the HIP path (inputs first: object / material / light buffers, headers; then sets, keys, atlas, HDR), and for triangle sets whose
triangles they are.    usage: python tools/fuzz_debug.py SEED [--mutate] [key=value ...]"""
import os
import sys

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import fuzz_parity as F  # noqa: E402
import scenes  # noqa: E402
from oracle import host as oh  # noqa: E402
from oracle.world import OracleRenderer, material_record as omk  # noqa: E402
import rend3_amd as r3  # noqa: E402

seed = int(sys.argv[1])
mutating = "--mutate" in sys.argv
c = F.draw_case(seed)
for kv in sys.argv[2:]:
    if "=" in kv:
        k, v = kv.split("=", 1)
        c[k] = type(c[k])(int(v)) if isinstance(c[k], (bool, int)) else (float(v) if isinstance(c[k], float) else v)
print(c)
f32 = np.float32
o = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
p = r3.Renderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
ho = F.build(o, oh, omk, c)
hp = F.build(p, r3.host, r3.material_record, c)
pair = ((o, oh, omk), (p, r3.host, r3.material_record))
st = dict(live=list(ho), w=c["w"], h=c["h"], samples=c["samples"], dir=c["lights"], point=c["point_lights"], mesh=[], mat=[])
if mutating:
    for r, hm, mk in pair:
        pos, idx, nrm = scenes.icosphere(1)
        if c["handedness"] == oh.LEFT:
            idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1)
        st["mesh"].append((scenes.cube_mesh(r), r.add_mesh(pos, idx, normals=nrm)))
        st["mat"].append((scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), scenes.lit(r, mk, (0.3, 0.6, 0.9, 1.0))))
rng = scenes.Pcg32(c["seed"] * 7919 + 13)


def report(fo, fp):
    for k in sorted(fo):
        a, b = fo[k], fp.get(k)
        if k in ("shadows", "blend_list", "shadow_descs") or b is None:
            continue
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b
        if k in ("pass", "residual", "visible", "tri_base", "baked", "objects"):
            b = b[:len(a)]
        if a.shape != b.shape:
            print(f"  {k}: shapes {a.shape} / {b.shape}")
            continue
        d = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0] if a.ndim else np.zeros(0, int)
        if len(d):
            extra = ""
            if k in ("pass", "residual"):
                obj = np.searchsorted(fo["tri_base"], d, side="right") - 1
                extra = f"; objects {sorted(set(obj.tolist()))[:16]}; oracle {fo[k][d[:8]].tolist()} hip {fp[k][d[:8]].tolist()}"
            elif a.ndim == 1 or k in ("objects", "materials", "dir_buf", "point_buf"):
                extra = f"; rows {d[:10].tolist()}"
            elif k in ("hdr16", "rgba8", "vis"):
                ys = np.unique(d)
                extra = f"; rows y {ys[:6].tolist()}..{ys[-1]}"
            print(f"  {k}: {len(d)} rows differ{extra}")
    for si, (so, sp) in enumerate(zip(fo["shadows"], fp["shadows"])):
        for name in ("visible", "pass"):
            a, b = so[name], sp[name][:len(so[name])]
            if not np.array_equal(a, b):
                print(f"  shadow {si} {name}: {(a != b).sum()} differ")


for f in range(5 if mutating else 3):
    view, proj = F.camera(c, f)
    for r in (o, p):
        r.set_camera_data(view, proj)
    if mutating and f:
        print(f"frame {f} edits:", F.mutate(rng, c, st, pair, f))
    elif not mutating:
        if f == 1 and len(ho) > 8:
            for r, hs in ((o, ho), (p, hp)):
                r.set_object_transform(hs[3], oh.mat4_mul(oh.translation(c["move"]), oh.scale((1.5, 0.5, 2.0))))
                r.remove_object(hs[6])
        if f == 2 and len(ho) > 8:
            for r, hs, mk in ((o, ho, omk), (p, hp, r3.material_record)):
                r.add_object(scenes.cube_mesh(r), scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), oh.translation(tuple(0.5 * t for t in c["target"])))
    kw = dict(samples=st["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
    fo = o.render(st["w"], st["h"], **kw)
    fp = p.render(st["w"], st["h"], **kw)
    print(f"frame {f}: {st['w']}x{st['h']} s{st['samples']} capacity {fo['capacity']} / {fp['capacity']}, point lights {len(o.point_lights)} / {len(p.point_lights)}, "
          f"directional {len(o.dir_lights)} / {len(p.dir_lights)}")
    report(fo, fp)
p.close()
