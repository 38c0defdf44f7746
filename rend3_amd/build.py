"""Builds librend3_amd.so (HIP kernels + C-ABI + host mirror) in-tree for gfx950 with hipcc.

-ffp-contract=off: every f32 operation rounds once (no FMA contraction) -- the arithmetic contract the
bit-exact visible-set parity rests on.  Division and sqrt stay correctly rounded
(hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librend3_amd.so")
SOURCES = ["r3n.hip", "blend_sort.hip", "texture_decode.hip", "anim.hip", "host.cpp"]
DEPS = SOURCES + ["layouts.h", "device_math.h", "texture.h", "kernels_cull.h", "kernels_raster.h", "bc7_tables.h", "../../include/r3n.h"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: rend3_amd needs ROCm's hipcc to build its gfx950 kernels")


def up_to_date():
    if not os.path.exists(SO):
        return False
    t = os.path.getmtime(SO)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return SO
    extra = os.environ.get("R3N_EXTRA_CXXFLAGS", "").split()
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", SO] + extra + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        sys.stderr.write(res.stderr)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
