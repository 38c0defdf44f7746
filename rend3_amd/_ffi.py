"""ctypes binding of librend3_amd.so (include/r3n.h).  The library is built in-tree by rend3_amd/build.py;
if it is missing and cannot be built this module raises -- there is no CPU fallback for the product path."""
import ctypes
import os

import numpy as np

from . import build as _build

vp = ctypes.c_void_p
u32 = ctypes.c_uint32
u64 = ctypes.c_uint64
cint = ctypes.c_int
cfloat = ctypes.c_float

CAMERA_VIEWPORT = 0xFFFFFFFF
PASS_DEPTH, PASS_FORWARD = 0, 1
SOURCE_PREDICTED, SOURCE_RESIDUAL = 0, 1
KEY_OPAQUE, KEY_CUTOUT, KEY_BLEND = 0, 1, 2
STAGES = ["bake", "object_cull", "triangle_cull", "hiz", "raster", "shade", "tonemap", "clear", "raster_big",
          "shadow_raster", "shadow_raster_big", "skinning", "vertex", "pose", "exchange_shadow", "exchange_depth", "exchange_rows", "exchange_keys",
          "raster_cut", "raster_big_cut"]

COMM_ID_BYTES, COMM_IDS = 128, 3  # R3N_COMM_ID_BYTES, R3N_COMM_IDS

# every symbol include/r3n.h declares: (restype, argtypes)
SIGNATURES = {
    "r3n_create": (vp, [cint, vp]),
    "r3n_destroy": (None, [vp]),
    "r3n_last_error": (ctypes.c_char_p, [vp]),
    "r3n_create_error": (ctypes.c_char_p, []),
    "r3n_sync": (cint, [vp]),
    "r3n_stream": (vp, [vp]),
    "r3n_mesh_buffer_write": (cint, [vp, u64, vp, u64]),
    "r3n_objects_write": (cint, [vp, vp, vp, u32, u32]),
    "r3n_materials_write": (cint, [vp, vp, vp, vp, u32]),
    "r3n_textures_write": (cint, [vp, vp, u32, vp, u64]),
    "r3n_textures_write_encoded": (cint, [vp, vp, u32, vp, u64]),
    "r3n_animation_write": (cint, [vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32]),
    "r3n_pose_skeletons": (cint, [vp, vp, u32]),
    "r3n_set_output_format": (cint, [vp, u32]),
    "r3n_set_shade_mode": (cint, [vp, u32]),
    "r3n_set_skinning_mode": (cint, [vp, u32]),
    "r3n_blend_order_write": (cint, [vp, vp, u32]),
    "r3n_lights_write": (cint, [vp, vp, u64, vp, u64]),
    "r3n_frame_begin": (cint, [vp, vp, u32, u32, u32, vp, u32, u32]),
    "r3n_skinning": (cint, [vp, vp, u32, vp, u32]),
    "r3n_uniform_bake": (cint, [vp, u32, vp]),
    "r3n_cull": (cint, [vp, u32]),
    "r3n_hi_z": (cint, [vp]),
    "r3n_shadow_viewport": (cint, [vp, u32, u32, u32, u32]),
    "r3n_forward": (cint, [vp, u32, u32, u32, u32]),
    "r3n_resolve_opaque": (cint, [vp]),
    "r3n_tonemap": (cint, [vp, vp, u64]),
    "r3n_hdr_write": (cint, [vp, vp, u64, u64]),
    "r3n_frame_end": (cint, [vp]),
    "r3n_render_frame": (cint, [vp, vp]),
    "r3n_set_object_range": (cint, [vp, u32, u32]),
    "r3n_set_object_owners": (cint, [vp, vp, u32, u32]),
    "r3n_exchange_buffers": (cint, [vp, vp, vp, vp, vp]),
    "r3n_exchange_shadow_stream": (cint, [vp, vp, vp, vp]),
    "r3n_set_shard_mode": (cint, [vp, u32]),
    "r3n_comm_unique_id": (cint, [vp]),
    "r3n_comm_init": (cint, [vp, vp, u32, u32]),
    "r3n_comm_destroy": (cint, [vp]),
    "r3n_comm_set_split": (cint, [vp, u32]),
    "r3n_timing_overhead": (cint, [vp, vp]),
    "r3n_set_camera_object_range": (cint, [vp, u32, u32, u32]),
    "r3n_exchange_depth": (cint, [vp, vp, vp]),
    "r3n_set_row_range": (cint, [vp, u32, u32]),
    "r3n_output_buffer": (cint, [vp, vp, vp]),
    "r3n_output_buffer_async": (cint, [vp, vp, vp, vp]),
    "r3n_output_work_enqueued": (cint, [vp]),
    "r3n_readback_visible_objects": (cint, [vp, u32, vp, u32]),
    "r3n_readback_triangle_sets": (cint, [vp, u32, vp, vp, u64]),
    "r3n_readback_draw_calls": (cint, [vp, u32, vp]),
    "r3n_readback_raster_stats": (cint, [vp, vp]),
    "r3n_readback_baked": (cint, [vp, u32, vp, u32]),
    "r3n_readback_mesh": (cint, [vp, u64, vp, u64]),
    "r3n_readback_texels": (cint, [vp, u64, vp, u64]),
    "r3n_readback_joint_matrices": (cint, [vp, u32, vp, u32]),
    "r3n_readback_visibility": (cint, [vp, vp]),
    "r3n_readback_depth": (cint, [vp, vp]),
    "r3n_readback_hiz": (cint, [vp, vp, u64]),
    "r3n_readback_shadow_atlas": (cint, [vp, vp]),
    "r3n_readback_hdr": (cint, [vp, vp]),
    "r3n_readback_output": (cint, [vp, vp, vp]),
    "r3n_timing_enable": (cint, [vp, cint]),
    "r3n_stage_times": (cint, [vp, vp, vp, cint]),
    "r3n_set_multi_stream": (cint, [vp, cint]),
    "r3n_hbm_copy_rate": (cint, [vp, u64, u32, vp]),
    "r3n_selftest_exact_math": (cint, [cint, vp, vp]),
    "r3n_selftest_unorm8": (cint, [cint, vp]),
    "r3n_host_mat4_mul": (None, [vp, vp, vp]),
    "r3n_host_mat4_inverse": (None, [vp, vp]),
    "r3n_host_look_at": (None, [vp, vp, vp, cint, vp]),
    "r3n_host_projection": (None, [cint, vp, cint, cfloat, vp]),
    "r3n_host_frustum_from_matrix": (None, [vp, vp]),
    "r3n_host_frustum_contains_sphere": (cint, [vp, vp, cfloat]),
    "r3n_host_bounding_sphere_from_mesh": (None, [vp, u64, vp, vp]),
    "r3n_host_bounding_sphere_apply_transform": (None, [vp, cfloat, vp, vp, vp]),
    "r3n_host_build_object_records": (None, [u32, vp, vp, vp, vp, vp]),
    "r3n_host_calculate_normals": (None, [vp, u64, vp, u64, cint, vp]),
    "r3n_host_shadow_camera": (None, [vp, cfloat, u32, vp, cint, vp, vp]),
    "r3n_host_allocate_shadow_atlas": (u32, [vp, vp, u32, u32, vp, vp]),
    "r3n_host_evaluate_frame": (cint, [vp, vp, u32, u32, vp, u32, u32, u32, u32, vp]),
}

MAX_SHADOW_VIEWS = 64
EXCHANGE_SITES = ("shadow", "pass1", "pass2")  # R3N_EXCHANGE_*
FRAME_VIEWPORT_FIRST, FRAME_SHADOW_MASK = 1, 2
EXCHANGE_FN = ctypes.CFUNCTYPE(cint, vp, u32)  # r3n_exchange_fn


class ShadowView272(ctypes.Structure):
    """r3n_shadow_view272"""
    _fields_ = [("header", ctypes.c_uint8 * 240), ("x", u32), ("y", u32), ("size", u32), ("_pad", u32 * 5)]


class HostCamera144(ctypes.Structure):
    """r3n_host_camera144"""
    _fields_ = [("view", cfloat * 16), ("projection_kind", u32), ("handedness", u32), ("aspect_ratio", cfloat), ("_pad", u32),
                ("projection_params", cfloat * 16)]


class HostFrame(ctypes.Structure):
    """r3n_host_frame"""
    _fields_ = [("uniforms", ctypes.c_uint8 * 496), ("viewport_header", ctypes.c_uint8 * 240),
                ("shadow_atlas_width", u32), ("shadow_atlas_height", u32), ("n_shadow_views", u32), ("_pad0", u32),
                ("camera_location", cfloat * 3), ("_pad1", cfloat), ("view_proj", cfloat * 16),
                ("shadow_views", ShadowView272 * MAX_SHADOW_VIEWS), ("shadow_handles", u32 * MAX_SHADOW_VIEWS),
                ("directional_bytes", u64), ("directional_buffer", ctypes.c_uint8 * (16 + 128 * MAX_SHADOW_VIEWS))]


class FrameDesc(ctypes.Structure):
    """r3n_frame_desc"""
    _fields_ = [("struct_size", u32), ("flags", u32), ("width", u32), ("height", u32), ("samples", u32),
                ("shadow_atlas_width", u32), ("shadow_atlas_height", u32), ("n_shadow_views", u32), ("clear_color", cfloat * 4),
                ("uniforms", vp), ("viewport_header", vp), ("shadow_views", vp), ("shadow_view_mask", u64),
                ("directional_buffer", vp), ("directional_bytes", u64), ("point_buffer", vp), ("point_bytes", u64),
                ("skin_inputs", vp), ("n_skeletons", u32), ("n_joint_matrices", u32), ("joint_matrices", vp),
                ("exchange", EXCHANGE_FN), ("exchange_user", vp)]


# sizes the C side pins with static_asserts (rend3_amd/csrc/layouts.h)
assert ctypes.sizeof(ShadowView272) == 272 and ctypes.sizeof(HostCamera144) == 144 and ctypes.sizeof(FrameDesc) == 152
assert ctypes.sizeof(HostFrame) == 26712

_LIB = None


def library_path():
    return _build.SO


def lib():
    """Loads (building first if needed) the native library.  Raises if that is impossible."""
    global _LIB
    if _LIB is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/HSA runtime, and two HSA
        # runtimes cannot both open the KFD.  Importing torch first makes this library bind to the runtime
        # torch already loaded (torch is the plumbing for streams / torch.distributed in this harness).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # R3N_LIB: developer knob -- load another build of the same library (kernel experiments, tools/variants.py)
        path = os.environ.get("R3N_LIB") or _build.build()
        if not os.path.exists(path):
            raise RuntimeError("librend3_amd.so is missing and could not be built; the HIP path has no fallback")
        l = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here == the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = l
    return _LIB


def ptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "need a contiguous ndarray"
    return a.ctypes.data_as(vp)


class R3nError(RuntimeError):
    pass


def check(ctx, code, what):
    if code != 0:
        msg = lib().r3n_last_error(ctx)
        raise R3nError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
