#!/usr/bin/env python3
"""Row extents of the spatial partition on the benchmark scenes (host-side arithmetic only, no GPU): for N = 2 / 4 / 8 ranks the
fraction of the target's rows inside each rank's conservative screen extent (parallel.partition_row_extents) along bench.py's
camera path -- what the row-limited pass-1 / pass-2 exchanges move relative to whole-target collectives -- next to the same figure
for contiguous slot ranges (partition_objects) with the same box machinery.  Feeds the cost model of DESIGN.md section 6.
  python tools/partition_extents.py [bistro|emerald]        (builds the scene in the CPU oracle's world model: spheres only)"""
import json
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main(which):
    import bench
    from oracle import host as oh                     # tools may not import the oracle for PRODUCT work; this is scene statistics
    from oracle.world import OracleRenderer, material_record as omk
    import rend3_amd.scenes as S
    from rend3_amd import parallel as P
    W, H = 3840, 2160
    o = OracleRenderer(oh.RIGHT if which == "bistro" else oh.LEFT, np.float32(W) / np.float32(H))
    info = S.bistro_like(o, oh, omk, textured=False) if which == "bistro" else S.emerald_like(o, oh, omk)
    counts = ((o.objects[:, 21] // 3) * (o.objects[:, 29] != 0)).astype(np.int64)
    spheres = o.objects[:, 16:20].view(np.float32).astype(np.float64)
    out = {"scene": which, "objects": int((counts > 0).sum()), "triangles": int(counts.sum())}
    for world in (2, 4, 8):
        owners = P.partition_objects_spatial(spheres[:, :3], counts, world)
        slot_owner = np.zeros(len(counts), dtype=np.uint8)
        for r, (b, e) in enumerate(P.partition_objects(counts, world)):
            slot_owner[b:e] = r
        rows = {}
        for name, own in (("spatial", owners), ("slots", slot_owner)):
            bounds = P.partition_bounds(own, spheres[:, :3], spheres[:, 3], counts, world)
            fr = []
            for k in range(0, 60, 6):
                o.set_camera_data(bench.camera_path(oh, info["camera"][0], k), info["camera"][1])
                ext = P.partition_row_extents(bounds, o.camera.view_proj, H)
                fr.append([(b - a) / H for a, b in ext])
            fr = np.array(fr)
            rows[name] = {"mean_row_fraction": round(float(fr.mean()), 3), "per_rank": [round(float(x), 3) for x in fr.mean(axis=0)]}
        out[f"N{world}"] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bistro")
