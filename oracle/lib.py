"""
oracle/lib.py -- builds and loads the C oracle (oracle/r3o.c) through ctypes.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by rend3_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libr3o.so")
_SRC = os.path.join(_HERE, "r3o.c")
_SRCS = [_SRC, os.path.join(_HERE, "bcn.c")]
_DEPS = _SRCS + [os.path.join(_HERE, "bc7_tables.h"), os.path.join(_HERE, "bc6h_tables.h")]

u8p = ctypes.POINTER(ctypes.c_uint8)
vp = ctypes.c_void_p


def build(force=False):
    """gcc -O2 -ffp-contract=off: no FMA contraction so every f32 op rounds once."""
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= max(os.path.getmtime(d) for d in _DEPS):
        return _SO
    cmd = [
        "gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
        "-Wall", "-Wextra", "-o", _SO, *_SRCS, "-lm",
    ]
    subprocess.run(cmd, check=True)
    return _SO


_SO_TALLY = os.path.join(_HERE, "libr3o_tally.so")


def build_tally(force=False):
    """The same restatement with a counting f32 (oracle/tally.h, r3o_tally.cpp): SURVEY.md section 8(d)'s op tally.  C++ (the
    operators), single-threaded (no -fopenmp: plain counters), the same rounding flags as the oracle proper."""
    deps = _DEPS + [os.path.join(_HERE, "tally.h"), os.path.join(_HERE, "r3o_tally.cpp")]
    if not force and os.path.exists(_SO_TALLY) and os.path.getmtime(_SO_TALLY) >= max(os.path.getmtime(d) for d in deps):
        return _SO_TALLY
    obj = os.path.join(_HERE, "bcn_tally.o")
    subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-c", "-ffp-contract=off", "-fno-fast-math", "-o", obj, os.path.join(_HERE, "bcn.c")], check=True)
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fpermissive", "-w", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas",
                    "-o", _SO_TALLY, os.path.join(_HERE, "r3o_tally.cpp"), obj, "-lm"], check=True)
    os.remove(obj)
    return _SO_TALLY


class OracleLib:
    def __init__(self, tally=False):
        self.tally = tally
        if tally:
            self.c = ctypes.CDLL(build_tally())
            self.c.r3o_tally_reset.restype = None
            self.c.r3o_tally_reset.argtypes = []
            self.c.r3o_tally_read.restype = None
            self.c.r3o_tally_read.argtypes = [vp]
        else:
            build()
            self.c = ctypes.CDLL(_SO)
        c = self.c
        c.r3o_hiz_mip_count.restype = ctypes.c_uint32
        c.r3o_hiz_mip_count.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        c.r3o_hiz_mip_offset.restype = ctypes.c_uint64
        c.r3o_hiz_mip_offset.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        c.r3o_frustum_contains_sphere.restype = ctypes.c_int
        c.r3o_frustum_contains_sphere.argtypes = [vp, vp, ctypes.c_float]
        c.r3o_texture_level_bytes.restype = ctypes.c_uint64
        c.r3o_texture_level_bytes.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        c.r3o_texture_decode_level.restype = ctypes.c_int
        c.r3o_texture_decode_level.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
        c.r3o_texture_is_float.restype = ctypes.c_int
        c.r3o_texture_is_float.argtypes = [ctypes.c_uint32]
        c.r3o_texture_decode_level_f32.restype = ctypes.c_int
        c.r3o_texture_decode_level_f32.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
        c.r3o_float_to_half.restype = ctypes.c_uint16
        c.r3o_float_to_half.argtypes = [ctypes.c_float]
        c.r3o_texture_generates_mips_f32.restype = ctypes.c_int
        c.r3o_texture_generates_mips_f32.argtypes = [ctypes.c_uint32]
        c.r3o_generate_mips_f32.restype = ctypes.c_int
        c.r3o_generate_mips_f32.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp]
        c.r3o_bc6h_decode_level_half.restype = ctypes.c_int
        c.r3o_bc6h_decode_level_half.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
        c.r3o_f32_to_f16.restype = ctypes.c_uint16
        c.r3o_f32_to_f16.argtypes = [ctypes.c_float]
        c.r3o_f16_to_f32.restype = ctypes.c_float
        c.r3o_f16_to_f32.argtypes = [ctypes.c_uint16]
        for name, args in {
            "r3o_mat4_mul": [vp, vp, vp],
            "r3o_uniform_bake": [vp, vp, vp],
            "r3o_frustum_from_matrix": [vp, vp],
            "r3o_frustum_cull": [vp, vp, vp],
            "r3o_hiz_build": [vp, ctypes.c_uint32, ctypes.c_uint32],
            "r3o_cull_triangles": [vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp, vp],
            "r3o_raster_visibility": [vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint32, vp, vp, vp],
            "r3o_raster_depth": [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint64, vp, ctypes.c_uint32,
                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, ctypes.c_uint32, vp],
            "r3o_vis_to_depth": [vp, ctypes.c_uint64, ctypes.c_uint32, vp],
            "r3o_shade": [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32, vp,
                          ctypes.c_uint32, vp, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp, ctypes.c_uint32, vp,
                          vp, vp, ctypes.c_uint64, vp],
            "r3o_generate_mips": [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp],
            "r3o_srgb8_table": [vp],
            "r3o_tonemap": [vp, ctypes.c_uint64, vp, vp],
            "r3o_tonemap_format": [vp, ctypes.c_uint64, vp, vp, ctypes.c_uint32],
            "r3o_skinning": [vp, vp, ctypes.c_uint32, vp],
            "r3o_skinning_mfma_order": [vp, vp, ctypes.c_uint32, vp, vp],
            "r3o_set_snap_bits": [ctypes.c_int],
            "r3o_set_depth_mode": [ctypes.c_int],
        }.items():
            fn = getattr(c, name)
            fn.restype = None
            fn.argtypes = args

    @staticmethod
    def ptr(a):
        if a is None:
            return None
        assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need contiguous ndarray"
        return a.ctypes.data_as(ctypes.c_void_p)

    def __getattr__(self, name):
        return getattr(self.c, name)


    def tally_reset(self):
        self.c.r3o_tally_reset()

    def tally_read(self):
        """{class: count} since the last reset (oracle/tally.h), all scopes together, + "flops" = add + mul + div + sqrt +
        transcendental + 2 x fma; the same per scope under "fixed_function" / "vs_main" / "fs_main" (r3o.c R3O_SCOPE)"""
        out = np.zeros((3, 8), dtype=np.uint64)
        self.c.r3o_tally_read(self.ptr(out))
        names = ("add", "mul", "div", "sqrt", "transcendental", "fma", "minmax_cmp", "convert")

        def row(v):
            d = dict(zip(names, (int(x) for x in v)))
            d["flops"] = d["add"] + d["mul"] + d["div"] + d["sqrt"] + d["transcendental"] + 2 * d["fma"]
            return d
        d = row(out.sum(axis=0))
        for k, scope in enumerate(("fixed_function", "vs_main", "fs_main")):
            d[scope] = row(out[k])
        return d


_LIB = None


def get():
    global _LIB
    if _LIB is None:
        _LIB = OracleLib()
    return _LIB
