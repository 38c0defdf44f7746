#!/usr/bin/env python3
"""GPU box, diagnostics build only (python tools/variants.py build trace=-DR3N_WAVE_TRACE; R3N_LIB=variants/lib_trace.so):
per-wave lifetimes of the shadow views' k_raster_big launches on the bench scene -- how much of a launch is load
imbalance (kernel span vs mean wave lifetime) and what a work item costs (fit of lifetime against items and scan steps)."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
import rend3_amd as r3
import rend3_amd.scenes
import bench

W, H = 3840, 2160
r = r3.Renderer(r3.host.RIGHT, np.float32(W / H))
info = r3.scenes.bistro_like(r, r3.host, r3.material_record, textured=False)
base = r3.BaseRenderGraph(r)
for k in range(4):
    r.set_camera_data(bench.camera_path(r3.host, info['camera'][0], k), info['camera'][1])
    r.render(W, H, ambient=bench.AMBIENT, clear_color=bench.CLEAR, readback=False, base=base)
r.sync()
buf = np.zeros((4, 32768, 4), np.uint32)
fn = r.lib.r3n_debug_wave_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert fn(r.ctx, buf.ctypes.data) == 0
for q in range(4):
    t = buf[q]
    live = t[(t[:, 1] != 0)]
    if not len(live):
        continue
    t0 = live[:, 0].astype(np.int64)
    t1 = live[:, 1].astype(np.int64)
    base_t = t0.min()
    life = (t1 - t0) / 100.0  # us
    items, steps = live[:, 2].astype(np.float64), live[:, 3].astype(np.float64)
    span = (t1.max() - base_t) / 100.0
    A = np.stack([np.ones_like(items), items, steps], axis=1)
    coef, *_ = np.linalg.lstsq(A, life, rcond=None)
    print(f"cascade quadrant {q}: {len(live)} waves, span {span:.1f} us, start spread {(t0.max() - base_t) / 100.0:.1f} us, "
          f"lifetime mean {life.mean():.1f} p50 {np.percentile(life, 50):.1f} p90 {np.percentile(life, 90):.1f} "
          f"p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
    print(f"    items/wave mean {items.mean():.1f} max {items.max():.0f} (total {items.sum():.0f}); steps/wave mean {steps.mean():.1f} "
          f"p90 {np.percentile(steps, 90):.0f} max {steps.max():.0f} (total {steps.sum():.0f}, {steps.sum() / max(items.sum(), 1):.2f} per item)")
    print(f"    fit lifetime = {coef[0]:.1f} us + {coef[1]:.3f} us x items + {coef[2]:.3f} us x steps")
    end_rel = (t1 - base_t) / 100.0
    hist, edges = np.histogram(end_rel, bins=10, range=(0, span))
    print("    wave end-time histogram (tenths of the span):", hist.tolist())

# ---- the per-triangle pass (k_raster_small<depth>) of the same launches
sb = np.zeros((4, 8192, 6), np.uint32)
fn = r.lib.r3n_debug_small_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert fn(r.ctx, sb.ctypes.data) == 0
for q in range(4):
    t = sb[q]
    live = t[t[:, 2] != 0]
    if not len(live):
        continue
    t0, tp, t1 = (live[:, k].astype(np.int64) for k in range(3))
    base_t = t0.min()
    life, prep = (t1 - t0) / 100.0, (tp - t0) / 100.0
    tris, box, items = (live[:, k].astype(np.float64) for k in (3, 4, 5))
    span = (t1.max() - base_t) / 100.0
    print(f"per-triangle pass, quadrant {q}: {len(live)} waves, span {span:.1f} us, start spread {(t0.max() - base_t) / 100.0:.1f} us; lifetime mean "
          f"{life.mean():.1f} p50 {np.percentile(life, 50):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us; "
          f"of which until the last setup: mean {prep.mean():.1f} p90 {np.percentile(prep, 90):.1f} max {prep.max():.1f} us")
    print(f"    triangles/wave mean {tris.mean():.1f} max {tris.max():.0f} (total {tris.sum():.0f}); largest in-place box mean {box.mean():.1f} max {box.max():.0f} texels; "
          f"items/wave mean {items.mean():.2f} p99 {np.percentile(items, 99):.0f} max {items.max():.0f} (total {items.sum():.0f})")
    A = np.stack([np.ones_like(tris), box, items], axis=1)
    coef, *_ = np.linalg.lstsq(A, life - prep, rcond=None)
    print(f"    fit (lifetime - setup) = {coef[0]:.2f} us + {coef[1]:.4f} us x box texels + {coef[2]:.3f} us x items")
    end_rel = (t1 - base_t) / 100.0
    print("    wave end-time histogram (tenths of the span):", np.histogram(end_rel, bins=10, range=(0, span))[0].tolist())
    worst = np.argsort(-life)[:5]
    for w in worst:
        print(f"      slowest: start {(t0[w] - base_t) / 100.0:.1f} setup {prep[w]:.1f} life {life[w]:.1f} us, triangles {tris[w]:.0f} box {box[w]:.0f} items {items[w]:.0f}")
