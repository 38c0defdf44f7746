#!/usr/bin/env python3
"""GPU box: screen-space bounding-box statistics of the triangles each camera of the bench scene draws (what the
rasteriser's small/big split and the per-lane scan loops see).  numpy on read-back data; analysis only."""
import sys
sys.path.insert(0, '.')
import numpy as np
import rend3_amd as r3
import rend3_amd.scenes
import bench

W, H = 3840, 2160
r = r3.Renderer(r3.host.RIGHT, np.float32(W / H))
info = r3.scenes.bistro_like(r, r3.host, r3.material_record)
base = r3.BaseRenderGraph(r)
for k in range(3):
    r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
    out = r.render(W, H, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=(k == 2), base=base)

handles = sorted(h for h, m in r.object_meta.items() if m["enabled"])
recs = np.stack([r._object_record(h) for h in handles]).astype(np.int64)
tris = recs[:, 21] // 3
tri_base = np.concatenate([[0], np.cumsum(tris)])[:-1]
total = int(tris.sum())
obj_of = np.repeat(np.arange(len(handles)), tris)
tri_of = np.arange(total) - np.repeat(tri_base, tris)
words = r.readback_mesh_words(0, int(r.mesh_cursor))


def stats(name, cam, bits, res):
    sel = np.nonzero(bits)[0]
    o, t = obj_of[sel], tri_of[sel]
    first = recs[o, 20] + 3 * t
    idx = np.stack([words[first + k] for k in range(3)], axis=1).astype(np.int64)
    pos_w = (recs[o, 23] // 4)[:, None] + 3 * idx
    P = np.stack([words[pos_w + c].view(np.float32) for c in range(3)], axis=2)  # n,3,3
    mvp = cam["baked"][np.asarray(handles)[o], 16:32].reshape(-1, 4, 4)  # column-major: m[c][r]
    clip = (mvp[:, None, 0, :] * P[:, :, 0:1] + mvp[:, None, 1, :] * P[:, :, 1:2] + mvp[:, None, 2, :] * P[:, :, 2:3]
            + mvp[:, None, 3, :])
    w = clip[:, :, 3]
    ok = (w > 0).all(axis=1)
    ndc = clip[:, :, :2] / np.where(w > 0, w, 1)[:, :, None]
    sx = (ndc[:, :, 0] + 1) * res[0] / 2
    sy = (1 - ndc[:, :, 1]) * res[1] / 2
    x0, x1 = np.clip(np.floor(sx.min(1) + 0.5), 0, res[0]), np.clip(np.ceil(sx.max(1) - 0.5), -1, res[0] - 1)
    y0, y1 = np.clip(np.floor(sy.min(1) + 0.5), 0, res[1]), np.clip(np.ceil(sy.max(1) - 0.5), -1, res[1] - 1)
    bw, bh = np.maximum(x1 - x0 + 1, 0), np.maximum(y1 - y0 + 1, 0)
    area = bw * bh
    m = np.maximum(bw, bh)
    print(f"{name}: {len(sel)} triangles, {(~ok).sum()} cross w=0; bbox max-dim histogram (w>0 only):")
    edges = [0, 1, 2, 3, 4, 6, 8, 12, 16, 32, 64, 128, 1 << 20]
    for a, b in zip(edges[:-1], edges[1:]):
        s = ok & (m > a) & (m <= b)
        print(f"   ({a:>4},{b:>7}]  tris {s.sum():>8}  bbox texels {int(area[s].sum()):>11}  mean area {area[s].mean() if s.any() else 0:8.1f}")
    s = ok & (m == 0)
    print(f"   empty bbox: {s.sum()}")
    big = ok & (m > 8)
    if big.any():
        cnt = (np.ceil(bw[big] / 32) * np.ceil(bh[big] / 32))
        grp = np.add.reduceat(big.astype(np.int64), np.arange(0, len(big), 64))
        # per list-order group of 64 triangles (what one wave of k_raster_small holds): emit-loop trip count = max cnt
        cnt_full = np.where(big, np.ceil(bw / 32) * np.ceil(bh / 32), 0)
        gmax = np.maximum.reduceat(cnt_full, np.arange(0, len(big), 64))
        a_full = np.where(ok & (m <= 8), area, 0)
        amax = np.maximum.reduceat(a_full, np.arange(0, len(big), 64))
        print(f"   big (>8): {big.sum()} triangles, items {int(cnt.sum())}, items/triangle mean {cnt.mean():.2f} p90 {np.percentile(cnt, 90):.0f} p99 {np.percentile(cnt, 99):.0f} max {cnt.max():.0f};"
              f" per-64 group: big lanes mean {grp.mean():.1f}, emit trips (max cnt) mean {gmax.mean():.1f} p90 {np.percentile(gmax, 90):.0f} max {gmax.max():.0f};"
              f" scan trips (max small area) mean {amax.mean():.1f}; sum of small areas / 64 = {a_full.sum() / len(amax) / 64:.1f}")
    small = ok & (m <= 8) & (m > 0)
    if small.any():
        a = area[small]
        print(f"   small (<=8): mean area {a.mean():.1f}, p50 {np.percentile(a, 50):.0f}, p90 {np.percentile(a, 90):.0f}, p99 {np.percentile(a, 99):.0f}, max {a.max():.0f};"
              f" mean of per-64 max {np.mean([a[i:i + 64].max() for i in range(0, len(a), 64)]):.1f}")


for si, sh in enumerate(out["shadows"]):
    stats(f"shadow {si}", sh, sh["pass"], (2048, 2048))
stats("viewport (pass set)", out, out["pass"], (W, H))
