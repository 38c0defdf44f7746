"""rend3_amd -- MI355X-native (gfx950, HIP) implementation of rend3's GPU-driven object pipeline:
uniform bake -> frustum/occlusion cull + indirect-draw compaction -> Hi-Z -> PBR opaque forward
(+ directional shadow views) -> tonemap, behind the C ABI in include/r3n.h.

The HIP library is the product; this package is the thin host-side mirror of the reference's routine
interface used by the standalone harness, bench.py and the tests.  There is no CPU fallback.
"""
from . import host  # noqa: F401
from ._ffi import R3nError, lib, library_path  # noqa: F401
from .renderer import (BLEND, CUTOUT, OPAQUE, BaseRenderGraph, BaseRenderGraphInputs,  # noqa: F401
                       BaseRenderGraphRoutines, BaseRenderGraphSettings, CameraSpecifier, ForwardRoutine, GpuCuller,
                       HiZRoutine, PbrRoutine, RenderGraph, Renderer, TonemappingRoutine, material_record)
