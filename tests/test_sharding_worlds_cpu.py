"""CPU: the multi-GPU algorithm at the world sizes the process-based tests do not reach, on drawn scenes.

tests/test_multi_rank_gloo.py runs one fixed scene with 2 and 4 processes over gloo.  Here N "ranks" are N threads of one process,
each with its own sharded oracle renderer, and the exchange steps are the collectives' definitions applied to the ranks' arrays
between two barriers (element-wise MAX of the depth plane / of the 64-bit keys; a shadow view's rectangle copied from its owner):
no process group, so seven or eight ranks cost nothing, and the scenes come from tools/fuzz_parity.py's generator.  What is checked
is what the native exchanges (r3n_comm_*) and rend3_amd/parallel.py implement: with the viewport split by object ranges, by owner
bytes (the spatial partition) or by rows, and the shadow views split by view, every rank ends every frame with the unsharded frame's keys (its own rows under the row
split), atlas, Hi-Z pyramid and image, and its triangle sets are the unsharded ones restricted to what it owns."""
import os
import sys
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))

import fuzz_parity as F  # noqa: E402
from oracle import host as oh  # noqa: E402
from oracle.world import OracleRenderer  # noqa: E402
from oracle.world import material_record as omk  # noqa: E402
from rend3_amd import parallel  # noqa: E402

f32 = np.float32


class Ranks:
    """The exchange sites of OracleRenderer.render for N ranks that are threads: publish, barrier, merge into a private copy,
    barrier, write back -- nobody reads an array somebody else is writing."""

    def __init__(self, world):
        self.world = world
        self.slot = [None] * world
        self.barrier = threading.Barrier(world)
        self.sites = set()

    def exchange_for(self, rank, rows_mode):
        def exchange(what, arr, shadows=None):
            self.sites.add(what)
            self.slot[rank] = arr
            self.barrier.wait()
            if what == "shadow":
                merged = arr.copy()
                for v, sh in enumerate(shadows):
                    x, y, s = int(sh["offset"][0]), int(sh["offset"][1]), int(sh["size"])
                    merged[y:y + s, x:x + s] = self.slot[parallel.shadow_view_owner(v, self.world)][y:y + s, x:x + s]
            elif what == "pass2" and rows_mode is True:
                merged = None  # sort-first: nothing is exchanged after pass 2
            elif arr.dtype == np.uint64:  # keys: depth bits << 32 | triangle, depth >= 0: unsigned MAX
                merged = np.maximum.reduce([self.slot[k] for k in range(self.world)])
            else:  # f32 depth plane, depth >= 0 (clear value 0.0 never wins)
                merged = np.maximum.reduce([self.slot[k] for k in range(self.world)])
            self.barrier.wait()
            if merged is not None:
                arr[...] = merged
        return exchange


def _run(world, rows_mode, seed):
    c = F.draw_case(seed)
    c["objects"] = min(c["objects"], 120)
    c["lights"] = max(c["lights"], 1)  # at least one shadow view to own
    w, h = c["w"], c["h"]
    full = OracleRenderer(c["handedness"], f32(w) / f32(h))
    F.build(full, oh, omk, c)
    shards = []
    for rank in range(world):
        s = OracleRenderer(c["handedness"], f32(w) / f32(h))
        F.build(s, oh, omk, c)
        s.shadow_views_owned = {v for v in range(len(s.dir_lights)) if parallel.shadow_view_owner(v, world) == rank}
        shards.append(s)
    counts = (full.objects[:, 21] // 3) * (full.objects[:, 29] != 0)
    ranges = parallel.partition_objects(counts, world)
    rows = parallel.row_ranges(h, world)
    spheres = full.objects[:, 16:20].view(np.float32)
    owners = parallel.partition_objects_spatial(spheres[:, :3], counts, world)
    for rank, s in enumerate(shards):
        if rows_mode == "spatial":
            s.object_owners = (owners, rank)  # owner bytes (Morton-order partition) instead of a slot range
        elif rows_mode:
            s.row_band = rows[rank]
        else:
            s.object_range = ranges[rank]
    _recs, mat_keys = full.material_buffers()
    is_blend = mat_keys[np.minimum(full.objects[:, 22], len(mat_keys) - 1)] == 2  # TransparencyType::Blend
    ranks = Ranks(world)
    kw = dict(samples=c["samples"], ambient=c["ambient"], clear_color=(0.02, 0.03, 0.05, 1.0))
    for f in range(3):
        view, proj = F.camera(c, f)
        for r in [full] + shards:
            r.set_camera_data(view, proj)
        ref = full.render(w, h, **kw)
        got, errors = [None] * world, []

        def work(rank):
            try:
                got[rank] = shards[rank].render(w, h, exchange=ranks.exchange_for(rank, rows_mode), **kw)
            except BaseException as e:  # noqa: BLE001  (a rank that dies must not leave the others at a barrier)
                errors.append((rank, repr(e)))
                ranks.barrier.abort()

        threads = [threading.Thread(target=work, args=(rank,)) for rank in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        tri_obj = np.searchsorted(ref["tri_base"], np.arange(len(ref["pass"])), side="right") - 1
        union = np.zeros(len(ref["pass"]), dtype=bool)
        for rank in range(world):
            g, tag = got[rank], f"seed {seed} world {world} {'rows' if rows_mode is True else (rows_mode or 'objects')} frame {f} rank {rank}"
            assert np.array_equal(ref["atlas"].view(np.uint32), g["atlas"].view(np.uint32)), tag + ": atlas"
            assert np.array_equal(ref["hiz"].view(np.uint32), g["hiz"].view(np.uint32)), tag + ": Hi-Z"
            r0, r1 = rows[rank]
            if rows_mode is True:
                assert np.array_equal(ref["vis"][r0:r1], g["vis"][r0:r1]), tag + ": own rows of the keys"
                assert np.array_equal(ref["rgba8"][r0:r1], g["rgba8"][r0:r1]), tag + ": own rows of the image"
                for k in ("visible", "pass", "residual"):
                    assert np.array_equal(ref[k], g[k]), tag + f": {k} (sort-first: every rank culls everything)"
            else:
                assert np.array_equal(ref["vis"], g["vis"]), tag + ": keys"
                assert np.array_equal(ref["rgba8"], g["rgba8"]), tag + ": image"
                b, e = ranges[rank]
                mine = is_blend.copy()  # translucent objects are culled and drawn by EVERY rank (ordered blending is not a MAX merge)
                if rows_mode == "spatial":
                    mine |= owners == rank
                else:
                    mine[b:e] = True
                assert np.array_equal(g["visible"].astype(bool), ref["visible"].astype(bool) & mine), tag + ": L1"
                assert np.array_equal(g["pass"].astype(bool), ref["pass"].astype(bool) & mine[tri_obj]), tag + ": L2 pass"
                assert np.array_equal(g["residual"].astype(bool), ref["residual"].astype(bool) & mine[tri_obj]), tag + ": L2 residual"
                union |= g["pass"].astype(bool)
        if rows_mode is not True:
            assert np.array_equal(union, ref["pass"].astype(bool)), f"seed {seed} frame {f}: the ranks' pass sets tile the unsharded one"
    assert {"shadow", "pass2"} <= ranks.sites and ({"pass1_depth", "pass1"} & ranks.sites), ranks.sites
    return int(ref["pass"].sum())


@pytest.mark.parametrize("world", [3, 5, 7, 8])
@pytest.mark.parametrize("rows_mode", [False, True, "spatial"], ids=["objects", "rows", "spatial"])
def test_every_rank_ends_with_the_unsharded_frame(world, rows_mode):
    ran, drawn = 0, 0
    with F.oracle_threads(4):  # N renderers run side by side
        for seed in range(9000 + 10 * world, 9000 + 10 * world + 5):
            drawn += _run(world, rows_mode, seed)
            ran += 1
    assert ran == 5 and drawn > 0, "the drawn cases must include scenes with something in view"
