"""Builds librend3_amd.so (HIP kernels + C-ABI + host mirror) in-tree for gfx950 with hipcc.

-ffp-contract=off: every f32 operation rounds once (no FMA contraction) -- the arithmetic contract the
bit-exact visible-set parity rests on.  Division and sqrt stay correctly rounded
(hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  The opt-in FAST shading variants (r3n_config.shade_mode)
are separate kernels inside shade.hip that enable contraction / approximate reciprocals locally.

Every translation unit is compiled on its own (in parallel) and the objects are linked; only the units whose
dependencies changed are recompiled.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SO = os.path.join(HERE, "librend3_amd.so")
COMMON = ["layouts.h", "device_math.h", "../../include/r3n.h"]
# translation unit -> the headers it includes (besides COMMON)
UNITS = {
    "r3n.hip": ["texture.h", "kernels_cull.h", "kernels_raster.h", "kernels_shade.h", "comm.h"],
    "shade.hip": ["texture.h", "kernels_shade.h"],
    "shade_cls.hip": ["texture.h", "kernels_shade.h"],
    "shade_ms.hip": ["texture.h", "kernels_shade.h"],
    "shade_blend.hip": ["texture.h", "kernels_shade.h"],
    "texture_decode.hip": ["bc7_tables.h", "bc6h_tables.h"],
    "anim.hip": [],
    "selftest.hip": ["exact_math.h"],
    "skin_mfma.hip": [],
    "host.cpp": [],
}
SOURCES = list(UNITS)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: rend3_amd needs ROCm's hipcc to build its gfx950 kernels")


def _deps(unit, csrc=None):
    return [os.path.join(csrc or CSRC, d) for d in [unit] + UNITS[unit] + COMMON]


def _obj(unit, obj_dir=OBJ):
    return os.path.join(obj_dir, unit.replace(".", "_") + ".o")


def _stale(unit, obj_dir=OBJ, csrc=None):
    o = _obj(unit, obj_dir)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(unit, csrc))


def up_to_date():
    if not os.path.exists(SO):
        return False
    t = os.path.getmtime(SO)
    return all(os.path.getmtime(d) <= t for u in UNITS for d in _deps(u))


def build(force=False, verbose=False, extra=None, out=SO, obj_dir=OBJ, csrc=None):
    """extra: additional compiler flags (variant builds, tools/variants.py; or R3N_EXTRA_CXXFLAGS): their objects go into a
    directory of their own (named by a hash of the flags) and, unless `out` is given, into librend3_amd.<hash>.so -- load it with
    R3N_LIB.  csrc: another source directory with the same layout (tools/variants.py: a patched copy of csrc/; `out` and
    `obj_dir` must then be given).  Returns the library's path."""
    src = csrc or CSRC
    if csrc is not None and (out == SO or obj_dir == OBJ):
        raise ValueError("a build from another source directory needs its own `out` and `obj_dir`")
    if extra is None:
        extra = os.environ.get("R3N_EXTRA_CXXFLAGS", "").split()
    if not force and not extra and out == SO and up_to_date():
        return SO
    if extra and obj_dir == OBJ:
        # variant flags never share the default object directory (or, unless the caller named an output, the default library):
        # a later plain build() would find "fresh" objects there and link the variant's kernels into the tested library
        import hashlib
        tag = hashlib.sha256(" ".join(extra).encode()).hexdigest()[:10]
        obj_dir = os.path.join(CSRC, "_obj_" + tag)
        if out == SO:
            out = os.path.join(HERE, f"librend3_amd.{tag}.so")
    os.makedirs(obj_dir, exist_ok=True)
    todo = [u for u in UNITS if force or extra or _stale(u, obj_dir, csrc)]

    def compile_unit(u):
        cmd = [hipcc()] + FLAGS + extra + ["-c", "-o", _obj(u, obj_dir), os.path.join(src, u)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        return u, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        results = list(pool.map(compile_unit, todo))
    for u, res in results:
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {u}:\n" + res.stdout + res.stderr)
        if verbose:
            sys.stderr.write(res.stderr)
    link = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + [_obj(u, obj_dir) for u in UNITS]
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
