#!/usr/bin/env python3
"""GPU box: the schedule the frame's streams actually run -- which rocprofv3's kernel trace cannot show, it serialises the
dispatches (the traced frame takes 1.69 ms clear to clear, one kernel at a time; the untraced frame 0.92 ms).  The library's timing
taps (two HIP events around every launch, r3n.hip Timed) are read back as (stage, stream, start, end) relative to the first launch
of the window (r3n_internal_read_timeline), for a few consecutive steady-state frames.

Printed: per stream the launches in start order; how long 0, 1, 2 ... launches were in flight; per stage the time in flight against
its stand-alone duration.  The events cost a few microseconds per launch (the frame under them is ~5 % slower than the untimed one).

usage: python tools/frame_timeline.py [--scene default|cfg4|v2] [--frames 3]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import rend3_amd as r3  # noqa: E402
import tune_caps  # noqa: E402

STAGES = ["bake", "object_cull", "triangle_cull", "hiz", "raster", "shade", "tonemap", "clear", "raster_big", "shadow_raster", "shadow_raster_big",
          "skinning", "vertex", "pose", "x_shadow", "x_depth", "x_rows", "x_keys", "raster_cut", "raster_big_cut"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="default")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--tune", default="")
    a = ap.parse_args()
    r, info, base = tune_caps.scene(a.scene)
    fn = r.lib.r3n_internal_set_tuning
    fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_char_p], ctypes.c_int
    assert fn(r.ctx, ("timed_pipeline=1 " + a.tune).encode()) == 0  # (frames stay in flight under the taps)
    read = r.lib.r3n_internal_read_timeline
    read.argtypes, read.restype = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    views = [bench.camera_path(r3.host, info["camera"][0], k) for k in range(40)]

    def frames(k0, n):
        for k in range(k0, k0 + n):
            r.set_camera_data(views[k % len(views)], info["camera"][1])
            r.render(3840, 2160, ambient=info["ambient"], clear_color=info["clear"], readback=False, base=base)

    frames(0, 10)
    r.sync()
    r.timing_enable(True)
    frames(10, 4)           # (pipeline filled under the taps)
    r.sync()
    r.stage_times(reset=True)
    frames(14, a.frames)
    buf = np.zeros((4096, 4), np.float32)
    n = read(r.ctx, buf.ctypes.data, len(buf))
    assert n > 0, n
    t = buf[:n]
    t[:, 2:] *= 1e3  # microseconds
    span = t[:, 3].max() - t[:, 2].min()
    print(f"scene {a.scene}: {a.frames} frames, {n} launches, {span:.1f} us from the first start to the last end ({span / a.frames:.1f} us per frame)")
    for s in sorted(set(t[:, 1].astype(int))):
        print(f"-- stream {s}")
        for row in sorted(t[t[:, 1] == s].tolist(), key=lambda x: x[2]):
            print(f"   {row[2]:8.1f} -> {row[3]:8.1f}  ({row[3] - row[2]:6.1f})  {STAGES[int(row[0])]}")
    ev = sorted([(x[2], 1) for x in t.tolist()] + [(x[3], -1) for x in t.tolist()])
    cur, last, hist = 0, ev[0][0], {}
    for when, d in ev:
        hist[cur] = hist.get(cur, 0.0) + (when - last)
        cur, last = cur + d, when
    print("launches in flight -> us:", {c: round(v, 1) for c, v in sorted(hist.items())})
    per = {}
    for row in t.tolist():
        per.setdefault(STAGES[int(row[0])], []).append(row[3] - row[2])
    print("stage: launches, mean us in flight, total us per frame")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {k:<18} {len(v):4d}  {sum(v) / len(v):8.1f}  {sum(v) / a.frames:8.1f}")
    r.close()


if __name__ == "__main__":
    main()
