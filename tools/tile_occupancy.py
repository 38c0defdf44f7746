#!/usr/bin/env python3
"""Measured screen occupancy of the object partitions (one GPU): for N = 2 / 4 / 8 and both partitions (contiguous slot ranges,
Morton-order spatial) every rank's share of the bench scene is rendered alone -- no exchange -- and the 64 x 64 px tiles its
visibility keys touch are counted (a tile is dirty when any pixel of it holds a key), after pass 1 + pass 2 of a steady-state
frame.  `dirty_tile_fraction` is what a tile-sparse exchange would have to move relative to whole targets; `row_fraction` is what
the implemented row-limited exchange could skip at best (rows with no key at all).  Feeds DESIGN.md section 6.
   python tools/tile_occupancy.py [--config 4]"""
import json
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import bench
    import rend3_amd as r3
    import rend3_amd.scenes as S
    from rend3_amd import parallel as P
    cfg4 = "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "4"
    W, H, T = 3840, 2160, 64

    def make():
        r = r3.Renderer(r3.host.LEFT if cfg4 else r3.host.RIGHT, np.float32(W) / np.float32(H))
        info = S.emerald_like(r, r3.host, r3.material_record, n_objects=1 << 17) if cfg4 else \
            S.bistro_like(r, r3.host, r3.material_record, textured=False, unique=False)
        r.evaluate_instructions()
        return r, info

    r, info = make()
    counts = np.zeros(r.capacity, dtype=np.int64)
    spheres = np.zeros((r.capacity, 4))
    for h, m in r.object_meta.items():
        counts[h] = r.meshes[m["mesh"]].index_count // 3
        spheres[h] = m["sphere"]
    r.close()
    out = {"scene": "emerald_like (131 072 objects)" if cfg4 else "bistro_like", "resolution": [W, H], "tile": T}
    for world in (2, 4, 8):
        owners_by = {"spatial": P.partition_objects_spatial(spheres[:, :3], counts, world)}
        slot_owner = np.zeros(len(counts), dtype=np.uint8)
        for k, (b, e) in enumerate(P.partition_objects(counts, world)):
            slot_owner[b:e] = k
        owners_by["slots"] = slot_owner
        res = {}
        for name, owners in owners_by.items():
            tiles, rows = [], []
            for rank in range(world):
                rr, inf = make()
                rr.set_object_owners(owners, rank)
                for k in range(3):
                    rr.set_camera_data(bench.camera_path(r3.host, inf["camera"][0], k), inf["camera"][1])
                    rr.render(W, H, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False)
                vis = np.zeros((H, W), dtype=np.uint64)
                rr._check(rr.lib.r3n_readback_visibility(rr.ctx, vis.ctypes.data), "readback_visibility")
                rr.close()
                d = vis != 0
                th, tw = (H + T - 1) // T, (W + T - 1) // T
                pad = np.zeros((th * T, tw * T), dtype=bool)
                pad[:H, :W] = d
                tiles.append(float(pad.reshape(th, T, tw, T).any(axis=(1, 3)).mean()))
                rows.append(float(d.any(axis=1).mean()))
            res[name] = {"dirty_tile_fraction_mean": round(float(np.mean(tiles)), 3), "per_rank": [round(t, 3) for t in tiles],
                         "row_fraction_mean": round(float(np.mean(rows)), 3)}
        out[f"N{world}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
