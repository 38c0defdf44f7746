#!/usr/bin/env python3
"""Kernel experiments: build the library several times -- each from a PATCHED copy of rend3_amd/csrc (and / or with extra compiler
flags) -- and time each build on the bench scene.  The product kernels carry no experiment switches: an
experiment is a patch under profiles/patches/, kept with its measurement whether or not it was adopted.

  here (no GPU):  python tools/variants.py build  base=  early=profiles/patches/r05_resolve_early_shadow_lookups.patch  x=some.patch,-DFLAG
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
    import shutil
    for spec in specs:  # every variant compiles its translation units in parallel (rend3_amd/build.py)
        name, _, what = spec.partition("=")
        parts = [f for f in what.split(",") if f]
        patches, flags = [f for f in parts if f.endswith(".patch")], [f for f in parts if not f.endswith(".patch")]
        src = os.path.join(VDIR, f"src_{name}", "rend3_amd", "csrc")  # (the patches name rend3_amd/csrc/...: git diff from the root)
        shutil.rmtree(os.path.join(VDIR, f"src_{name}"), ignore_errors=True)
        shutil.copytree(b.CSRC, src, ignore=shutil.ignore_patterns("_obj*"))
        os.makedirs(os.path.join(VDIR, f"src_{name}", "include"), exist_ok=True)
        shutil.copy(os.path.join(ROOT, "include", "r3n.h"), os.path.join(VDIR, f"src_{name}", "include", "r3n.h"))
        try:
            for pt in patches:
                res = subprocess.run(["patch", "-p1", "-i", os.path.abspath(pt)], cwd=os.path.join(VDIR, f"src_{name}"), capture_output=True, text=True)
                if res.returncode != 0:
                    raise RuntimeError(f"{pt} does not apply:\n" + res.stdout + res.stderr)
            b.build(force=True, extra=flags or ["-DR3N_VARIANT_BASE"], out=os.path.join(VDIR, f"lib_{name}.so"),
                    obj_dir=os.path.join(VDIR, f"obj_{name}"), csrc=src)
            print(name, "ok")
        except RuntimeError as e:
            print(name, "FAILED\n" + str(e))


def run(steps):
    for f in sorted(os.listdir(VDIR)):
        if not f.endswith(".so"):
            continue
        env = dict(os.environ, R3N_LIB=os.path.join(VDIR, f))
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline"] + os.environ.get("VARIANT_FLAGS", "").split(),
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
