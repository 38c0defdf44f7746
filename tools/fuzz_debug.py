#!/usr/bin/env python3
"""GPU box: one case of tools/fuzz_parity.py in detail -- which triangles' pass / residual bits differ, and whose they are."""
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
c = F.draw_case(seed)
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    c[k] = type(c[k])(int(v)) if isinstance(c[k], (bool, int)) else (float(v) if isinstance(c[k], float) else v)
print(c)
f32 = np.float32
o = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
p = r3.Renderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
ho = F.build(o, oh, omk, c)
hp = F.build(p, r3.host, r3.material_record, c)
print("handles equal:", ho == hp, "n", len(ho), "capacity", o.capacity if hasattr(o, "capacity") else None, p.capacity)
for f in range(3):
    view, proj = F.camera(c, f)
    for r in (o, p):
        r.set_camera_data(view, proj)
    if f == 1 and len(ho) > 8:
        for r, hs in ((o, ho), (p, hp)):
            r.set_object_transform(hs[3], oh.mat4_mul(oh.translation(c["move"]), oh.scale((1.5, 0.5, 2.0))))
            r.remove_object(hs[6])
    if f == 2 and len(ho) > 8:
        added = []
        for r, hs, mk in ((o, ho, omk), (p, hp, r3.material_record)):
            added.append(r.add_object(scenes.cube_mesh(r), scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), oh.translation(tuple(0.5 * t for t in c["target"]))))
        print("frame 2 adds handle", added, "removed", ho[6], hp[6])
    kw = dict(samples=c["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
    fo = o.render(c["w"], c["h"], **kw)
    fp = p.render(c["w"], c["h"], **kw)
    n = len(fo["pass"])
    print(f"frame {f}: capacity {fo['capacity']} / {fp['capacity']}, triangle slots {n} / {len(fp['pass'])}, tri_base equal {np.array_equal(fo['tri_base'], fp['tri_base'][:len(fo['tri_base'])]) if 'tri_base' in fp else 'n/a'}")
    for name in ("visible", "pass", "residual"):
        a, b = fo[name], fp[name][:len(fo[name])]
        d = np.nonzero(a != b)[0]
        if len(d):
            if name == "visible":
                print(f"  {name}: {len(d)} differ: slots {d[:20].tolist()} oracle {a[d[:20]].tolist()} hip {b[d[:20]].tolist()}")
            else:
                obj = np.searchsorted(fo["tri_base"], d, side="right") - 1
                print(f"  {name}: {len(d)} triangles differ, objects {sorted(set(obj.tolist()))[:20]}; oracle bits {a[d[:12]].tolist()} hip {b[d[:12]].tolist()}; "
                      f"pass there: oracle {fo['pass'][d[:12]].tolist()} hip {fp['pass'][d[:12]].tolist()}")
    for si, (so, sp) in enumerate(zip(fo["shadows"], fp["shadows"])):
        for name in ("visible", "pass"):
            a, b = so[name], sp[name][:len(so[name])]
            if not np.array_equal(a, b):
                print(f"  shadow {si} {name}: {(a != b).sum()} differ")
    print("  keys differ px:", int((fo["vis"] != fp["vis"]).sum()), " atlas differ:", int((fo["atlas"].view(np.uint32) != fp["atlas"].view(np.uint32)).sum()))
p.close()
