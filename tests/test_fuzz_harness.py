"""CPU: the randomised parity campaign's harness (tools/fuzz_parity.py) against itself.

On the GPU box the tool compares the HIP path with the oracle; here two ORACLE renderers take its place, so that the case generator
and the world-edit schedule stay runnable (every edit must reach both renderers with the same values -- the first version drew a new
point light's intensity once per renderer and reported 700 false mismatches) and the oracle stays deterministic under them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import fuzz_parity as F  # noqa: E402
import scenes  # noqa: E402
from oracle import host as oh  # noqa: E402
from oracle.world import OracleRenderer  # noqa: E402
from oracle.world import material_record as omk  # noqa: E402

f32 = np.float32


def test_case_parameters_are_a_function_of_the_seed():
    assert F.draw_case(1234) == F.draw_case(1234)
    drawn = [F.draw_case(s) for s in range(2000, 2200)]
    assert {c["builder"] for c in drawn} == {"random", "random_cutout", "textured", "textured_encoded", "textured_float"}
    assert {c["samples"] for c in drawn} == {1, 4} and any(c["ortho"] for c in drawn) and any(c["blend"] for c in drawn)


def test_edit_schedule_reaches_both_renderers_alike():
    kinds, flips = set(), 0
    for seed in range(5000, 5016):
        c = F.draw_case(seed)
        o = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
        p = OracleRenderer(c["handedness"], f32(c["w"]) / f32(c["h"]))
        ho, hp = F.build(o, oh, omk, c), F.build(p, oh, omk, c)
        assert ho == hp
        pair = ((o, oh, omk), (p, oh, omk))
        st = dict(live=list(ho), w=c["w"], h=c["h"], samples=c["samples"], dir=c["lights"], point=c["point_lights"], mesh=[], mat=[])
        for r, _hm, mk in pair:
            pos, idx, nrm = scenes.icosphere(1)
            if c["handedness"] == oh.LEFT:
                idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1)
            st["mesh"].append((scenes.cube_mesh(r), r.add_mesh(pos, idx, normals=nrm)))
            st["mat"].append((scenes.lit(r, mk, (0.8, 0.7, 0.2, 1.0)), scenes.lit(r, mk, (0.3, 0.6, 0.9, 1.0))))
        rng = scenes.Pcg32(c["seed"] * 7919 + 13)
        st["flip_rng"] = scenes.Pcg32(c["seed"] * 104729 + 71)
        for f in range(4):
            view, proj = F.camera(c, f)
            for r in (o, p):
                r.set_camera_data(view, proj)
            if f:
                edits = F.mutate(rng, c, st, pair, f)
                kinds.update(e.split(" ")[0] for e in edits)
                flips += sum("-> key" in e for e in edits)
            kw = dict(samples=st["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
            fo, fp = o.render(st["w"], st["h"], **kw), p.render(st["w"], st["h"], **kw)
            for k in ("vis", "hdr16", "pass", "residual", "visible", "point_buf", "dir_buf", "objects", "materials", "material_keys"):
                assert np.array_equal(np.asarray(fo[k]), np.asarray(fp[k])), f"seed {seed} frame {f}: {k}"
    assert {"move", "remove", "add", "bulk", "material", "light", "point", "resize", "mesh", "newmat", "texture"} <= kinds
    assert flips >= 8, "key flips are part of the schedule (VERDICT r5 item 2)"


def test_oracle_thread_limit_is_scoped():
    import ctypes
    omp = ctypes.CDLL("libgomp.so.1")
    before = omp.omp_get_max_threads()
    with F.oracle_threads(2):
        assert omp.omp_get_max_threads() == min(2, before)
    assert omp.omp_get_max_threads() == before
