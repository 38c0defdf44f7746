#!/usr/bin/env python3
"""GPU box: a search over the launch parameters that decide how the frame's kernels share the chip (r3n.hip Tune: dynamic-LDS
occupancy caps of the rasterisers / the triangle cull / the resolve, grids of the persistent rasterisers), in ONE process --
r3n_internal_set_tuning changes them between frames, so a setting costs ~0.1 s of frames instead of a 10 s bench process.

VERDICT r4 weak #5: "the round's last four gains came from occupancy caps found by hand A/B ... there is no search and no model".
Coordinate descent from the current defaults: every knob in turn over its candidate values, the best kept, until a full sweep
improves nothing (each measurement = `--frames` frames of the bench's camera dolly after a short warm-up, median of `--reps`
repeats; a candidate must win by 0.5 % to replace the incumbent).  Scenes: the default bench scene, `--config 4`, `--bistro-v2`;
the table per scene and the consensus go to stdout (profiles/r05_tune_caps.txt).

usage: python tools/tune_caps.py [--scenes default,cfg4,v2] [--frames 60] [--reps 3]"""
import argparse
import ctypes
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rend3_amd as r3  # noqa: E402

KNOBS = {
    "big_lds": [0, 16384, 32768, 40960, 49152, 65536],
    "vp_big_lds": [0, 16384, 32768, 40960, 49152, 65536],
    "small_lds": [0, 24576, 32768, 49152],
    "vp_small_lds": [0, 24576, 32768, 49152],
    "cut_big_lds": [0, 16384, 32768, 49152],
    "vp_cut_big_lds": [0, 16384, 32768, 49152],
    "cut_small_lds": [0, 32768],
    "vp_cut_small_lds": [0, 32768],
    "cull_lds": [0, 16384],
    "vp_cull_lds": [0, 16384],
    "resolve_lds": [0, 36864, 49152],   # four / three workgroups per CU (160 KB): below ~36 KB the cap does not bind
    "big_grid": [4096, 8192],
    "small_grid": [1024, 2048, 4096],
}
# (the library's defaults: r3n.hip Tune)
DEFAULTS = {"big_lds": 40960, "vp_big_lds": 49152, "small_lds": 24576, "vp_small_lds": 0, "cut_big_lds": 0, "vp_cut_big_lds": 0, "cut_small_lds": 0,
            "vp_cut_small_lds": 0, "cull_lds": 0, "vp_cull_lds": 0, "resolve_lds": 0, "big_grid": 8192, "small_grid": 2048}


def scene(name):
    args = argparse.Namespace(scene=None, config=4 if name == "cfg4" else 3, bistro_v2=name == "v2", objects=3000, tris=2_800_000, untextured=False,
                              instanced=False, samples=1)
    r = r3.Renderer(r3.host.RIGHT, np.float32(3840 / 2160))
    info = bench.build_workload(args, r, r3.host, r3.material_record)
    return r, info, r3.BaseRenderGraph(r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="default,cfg4,v2")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--knobs", default="", help="comma-separated subset of the knobs to search (default: all)")
    a = ap.parse_args()
    knobs = {k: v for k, v in KNOBS.items() if not a.knobs or k in a.knobs.split(",")}
    set_tuning = None
    results = {}
    for name in a.scenes.split(","):
        r, info, base = scene(name)
        if set_tuning is None:
            set_tuning = r.lib.r3n_internal_set_tuning
            set_tuning.argtypes, set_tuning.restype = [ctypes.c_void_p, ctypes.c_char_p], ctypes.c_int
        views = [bench.camera_path(r3.host, info["camera"][0], k) for k in range(8 + a.frames)]
        step = [0]

        def frames(n):
            for _ in range(n):
                r.set_camera_data(views[step[0] % len(views)], info["camera"][1])
                step[0] += 1
                r.render(3840, 2160, ambient=info["ambient"], clear_color=info["clear"], readback=False, base=base)

        def measure(setting):
            kv = " ".join(f"{k}={v}" for k, v in setting.items()).encode()
            assert set_tuning(r.ctx, kv) == 0, r.lib.r3n_last_error(r.ctx)
            times = []
            for _ in range(a.reps):
                frames(6)
                r.sync()
                t0 = time.perf_counter()
                frames(a.frames)
                r.sync()
                times.append((time.perf_counter() - t0) * 1e3 / a.frames)
            return statistics.median(times)

        frames(8)
        best = dict(DEFAULTS)
        best_ms = measure(best)
        print(f"== {name}: defaults {best_ms:.4f} ms/frame", flush=True)
        improved, sweeps = True, 0
        while improved and sweeps < 3:
            improved, sweeps = False, sweeps + 1
            for knob, values in knobs.items():
                row = {}
                for v in values:
                    row[v] = best_ms if v == best[knob] else measure({**best, knob: v})
                pick = min(row, key=row.get)
                print(f"   {knob:<13}" + "  ".join(f"{v}:{ms:.4f}" for v, ms in row.items()) + (f"   -> {pick}" if pick != best[knob] and row[pick] < 0.995 * best_ms else ""), flush=True)
                if pick != best[knob] and row[pick] < 0.995 * best_ms:
                    best[knob], best_ms, improved = pick, row[pick], True
            best_ms = measure(best)  # (re-measured: the sweep's own best can be a lucky run)
        print(f"   best {best_ms:.4f} ms/frame: " + " ".join(f"{k}={v}" for k, v in best.items()), flush=True)
        results[name] = (best, best_ms, measure(dict(DEFAULTS)))
        r.close()
    print("\nscene      defaults -> best (ms/frame)   setting")
    for name, (best, ms, ms0) in results.items():
        print(f"{name:<10} {ms0:.4f} -> {ms:.4f}   " + " ".join(f"{k}={v}" for k, v in best.items() if v != DEFAULTS[k]))


if __name__ == "__main__":
    main()
