#!/usr/bin/env python3
"""Kernel experiments: build the library several times with different -D flags and time each build on the bench scene.

  here (no GPU):  python tools/variants.py build  name=-DFLAG1,-DFLAG2  other=...
  on the GPU box: python tools/variants.py run [--steps K]      (every variants/lib_*.so; prints one line per build)

The variant libraries live in variants/ (git-ignored, shipped by gpurun).  rend3_amd/_ffi.py loads the one R3N_LIB names.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "variants")


def build(specs):
    sys.path.insert(0, ROOT)
    from rend3_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for f in os.listdir(VDIR):
        if f.endswith(".so"):
            os.remove(os.path.join(VDIR, f))
    for spec in specs:  # every variant compiles its translation units in parallel (rend3_amd/build.py)
        name, _, flags = spec.partition("=")
        try:
            b.build(force=True, extra=[f for f in flags.split(",") if f] or ["-DR3N_VARIANT_BASE"], out=os.path.join(VDIR, f"lib_{name}.so"),
                    obj_dir=os.path.join(VDIR, f"obj_{name}"))
            print(name, "ok")
        except RuntimeError as e:
            print(name, "FAILED\n" + str(e))


def run(steps):
    for f in sorted(os.listdir(VDIR)):
        if not f.endswith(".so"):
            continue
        env = dict(os.environ, R3N_LIB=os.path.join(VDIR, f))
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(res.stdout.strip().splitlines()[-1])
            st = {k: round(v * 1e3, 1) for k, v in d["stage_ms_per_frame"].items() if v}
            print(f"{f[4:-3]:<16} frame {d['ms_per_step']:.4f} ms  stages(us) {st}")
        except Exception:
            print(f, "FAILED", res.stdout[-400:], res.stderr[-800:])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40)
