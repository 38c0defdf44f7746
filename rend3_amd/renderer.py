"""Host-side mirror of the reference's routine interface for the hot path, driving the HIP kernels through
the C ABI (include/r3n.h).  Names, argument meaning and the node order follow the reference:

  Renderer (world edits the path consumes)         rend3/src/renderer/mod.rs:126-424
  RenderGraph / node closures                      rend3/src/graph/graph.rs:81-519, graph/node.rs:59-213
  BaseRenderGraph::{new, add_to_graph}             rend3-routine/src/base.rs:110-186
  GpuCuller::{add_object_uniform_upload_to_graph,
              add_culling_to_graph}                rend3-routine/src/culling/culler.rs:661-713
  ForwardRoutine::add_forward_to_graph             rend3-routine/src/forward.rs:192-315
  PbrRoutine (5 forward routines + hi-z)           rend3-routine/src/pbr/routine.rs:35-133
  HiZRoutine::add_hi_z_to_graph                    rend3-routine/src/hi_z.rs:161-234
  TonemappingRoutine::add_to_graph                 rend3-routine/src/tonemapping.rs:108-147

The graph here is the fixed schedule only (the render-graph machinery itself is
out of scope); nodes are closures executed in declaration order by RenderGraph.execute.
This module never imports the oracle; without the native library it raises.
"""
import ctypes

import os

import numpy as np

from . import _ffi, host

f32 = np.float32
INVALID = 0xFFFFFFFF
OPAQUE, CUTOUT, BLEND = 0, 1, 2

# rend3-routine/shaders/src/material.wgsl:1-15
FLAGS_ALBEDO_ACTIVE = 0x0001
FLAGS_ALBEDO_BLEND = 0x0002
FLAGS_ALBEDO_VERTEX_SRGB = 0x0004
FLAGS_AOMR_COMBINED = 0x0040
FLAGS_AOMR_SPLIT = 0x0100
FLAGS_CC_GLTF_COMBINED = 0x0400
FLAGS_BICOMPONENT_NORMAL = 0x0008
FLAGS_SWIZZLED_NORMAL = 0x0010
FLAGS_YDOWN_NORMAL = 0x0020
FLAGS_AOMR_SWIZZLED_SPLIT = 0x0080
FLAGS_AOMR_BW_SPLIT = 0x0200
FLAGS_CC_GLTF_SPLIT = 0x0800
FLAGS_CC_BW_SPLIT = 0x1000
FLAGS_UNLIT = 0x2000
FLAGS_NEAREST = 0x4000



def material_texture_slots(ru, normal_texture, normal_mode, normal_y_down, aomr, reflectance_texture,
                           clearcoat_textures, emissive_texture, anisotropy_texture):
    """Texture ids (handle + 1) of slots 1..9 of the 208-byte record and the flags of NormalTexture / AoMRTextures /
    ClearcoatTextures::to_flags (pbr/material.rs:201-222, 305-317, 354-364).  aomr: None | ("combined", tex) |
    ("swizzled_split", ao, mr) | ("split", ao, mr) | ("bw_split", ao, m, r); clearcoat_textures: None |
    ("gltf_combined", tex) | ("gltf_split", cc, ccr) | ("bw_split", cc, ccr); any texture may be None."""
    def put(slot, tex):
        if tex is not None:
            ru[slot] = int(tex) + 1
    flags = 0
    if normal_texture is not None:
        put(1, normal_texture)
        flags |= {"tricomponent": 0, "bicomponent": FLAGS_BICOMPONENT_NORMAL,
                  "bicomponent_swizzled": FLAGS_BICOMPONENT_NORMAL | FLAGS_SWIZZLED_NORMAL}[normal_mode]
        if normal_y_down:
            flags |= FLAGS_YDOWN_NORMAL
    if aomr is None:
        flags |= FLAGS_AOMR_COMBINED  # "so shader only checks roughness texture, then bails"
    elif aomr[0] == "combined":
        flags |= FLAGS_AOMR_COMBINED
        put(2, aomr[1])
    elif aomr[0] in ("swizzled_split", "split"):
        flags |= FLAGS_AOMR_SWIZZLED_SPLIT if aomr[0] == "swizzled_split" else FLAGS_AOMR_SPLIT
        put(9, aomr[1])
        put(2, aomr[2])
    elif aomr[0] == "bw_split":
        flags |= FLAGS_AOMR_BW_SPLIT
        put(9, aomr[1])
        put(3, aomr[2])
        put(2, aomr[3])
    else:
        raise ValueError(aomr)
    put(4, reflectance_texture)
    if clearcoat_textures is None or clearcoat_textures[0] == "gltf_combined":
        flags |= FLAGS_CC_GLTF_COMBINED
        if clearcoat_textures is not None:
            put(5, clearcoat_textures[1])
    elif clearcoat_textures[0] in ("gltf_split", "bw_split"):
        flags |= FLAGS_CC_GLTF_SPLIT if clearcoat_textures[0] == "gltf_split" else FLAGS_CC_BW_SPLIT
        put(5, clearcoat_textures[1])
        put(6, clearcoat_textures[2])
    else:
        raise ValueError(clearcoat_textures)
    put(7, emissive_texture)
    put(8, anisotropy_texture)
    return flags


def material_record(albedo=(0, 0, 0, 1), albedo_mode="value", unlit=False, roughness=0.0, metallic=0.0,
                    reflectance=0.5, emissive=(0, 0, 0), ao=1.0, clear_coat=0.0, clear_coat_roughness=0.0,
                    cutout=None, vertex_srgb=True, albedo_texture=None, nearest=False, uv_transform0=None,
                    normal_texture=None, normal_mode="tricomponent", normal_y_down=False, aomr=None,
                    reflectance_texture=None, clearcoat_textures=None, emissive_texture=None, anisotropy_texture=None):
    """ShaderMaterial::from_material (rend3-routine/src/pbr/material.rs:548-583) behind the 48-byte texture-id prefix
    (rend3/src/managers/material.rs:25-29).  208 bytes as f32[52].  albedo_mode (AlbedoComponent, material.rs:60-140):
    "none" | "vertex" | "value" | "value_vertex", or with `albedo_texture` (texture handle) "texture" |
    "texture_vertex" | "texture_value" | "texture_vertex_value".  nearest = SampleType::Nearest; uv_transform0 = Mat3
    (row-major nested list)."""
    rec = np.zeros(52, dtype=f32)
    ru = rec.view(np.uint32)
    for base in (12, 24):  # uv_transform0/1 = identity mat3 (3 x vec4 columns)
        rec[base + 0] = rec[base + 5] = rec[base + 10] = 1.0
    flags = material_texture_slots(ru, normal_texture, normal_mode, normal_y_down, aomr, reflectance_texture,
                                   clearcoat_textures, emissive_texture, anisotropy_texture)
    if albedo_mode == "none":
        alb = (0.0, 0.0, 0.0, 1.0)
    elif albedo_mode == "vertex":
        flags |= FLAGS_ALBEDO_ACTIVE | FLAGS_ALBEDO_BLEND | (FLAGS_ALBEDO_VERTEX_SRGB if vertex_srgb else 0)
        alb = (1.0, 1.0, 1.0, 1.0)
    elif albedo_mode == "value":
        flags |= FLAGS_ALBEDO_ACTIVE
        alb = albedo
    elif albedo_mode == "value_vertex":
        flags |= FLAGS_ALBEDO_ACTIVE | FLAGS_ALBEDO_BLEND | (FLAGS_ALBEDO_VERTEX_SRGB if vertex_srgb else 0)
        alb = albedo
    elif albedo_mode in ("texture", "texture_value"):
        flags |= FLAGS_ALBEDO_ACTIVE
        alb = albedo if albedo_mode == "texture_value" else (1.0, 1.0, 1.0, 1.0)
    elif albedo_mode in ("texture_vertex", "texture_vertex_value"):
        flags |= FLAGS_ALBEDO_ACTIVE | FLAGS_ALBEDO_BLEND | (FLAGS_ALBEDO_VERTEX_SRGB if vertex_srgb else 0)
        alb = albedo if albedo_mode == "texture_vertex_value" else (1.0, 1.0, 1.0, 1.0)
    else:
        raise ValueError(albedo_mode)
    if albedo_mode.startswith("texture"):
        if albedo_texture is None:
            raise ValueError("texture albedo modes need albedo_texture")
        ru[0] = int(albedo_texture) + 1  # NonZeroU32 index into the bindless array
    if nearest:
        flags |= FLAGS_NEAREST
    if uv_transform0 is not None:
        m = np.asarray(uv_transform0, dtype=f32).reshape(3, 3)
        for col in range(3):
            rec[12 + 4 * col: 12 + 4 * col + 3] = m[:, col]
    if unlit:
        flags |= FLAGS_UNLIT
    rec[36:40] = alb
    rec[40:43] = emissive
    rec[43], rec[44], rec[45], rec[46], rec[47] = roughness, metallic, reflectance, clear_coat, clear_coat_roughness
    rec[49] = ao
    rec[50] = 0.0 if cutout is None else cutout
    ru[51] = flags
    return rec


class CameraSpecifier:
    """rend3-routine/src/common/camera.rs:3-35"""
    VIEWPORT = _ffi.CAMERA_VIEWPORT

    @staticmethod
    def shadow(i):
        assert i != 0xFFFFFFFF
        return i


class _Mesh:
    __slots__ = ("attr_off", "first_index", "index_count", "centre", "radius", "vertex_count", "joint_off", "weight_off")


class EvalOutput:
    """What Renderer::evaluate_instructions hands the graph (rend3/src/renderer/eval.rs:157-187), path subset."""

    def __init__(self, shadows, shadow_target_size):
        self.shadows = shadows  # [dict(offset=(x, y), size, handle[, camera])] in shadow-view order
        self.shadow_target_size = shadow_target_size


class Renderer:
    """World bookkeeping feeding the object/mesh/material/light buffers of the C ABI."""

    def __init__(self, handedness=host.LEFT, aspect_ratio=None, device=0):
        self.lib = _ffi.lib()
        self.ctx = self.lib.r3n_create(device, None)
        if not self.ctx:
            raise _ffi.R3nError("r3n_create failed: " + self.lib.r3n_create_error().decode())
        self.handedness = handedness
        self.aspect_ratio = aspect_ratio
        self.mesh_cursor = 0  # in u32 words
        self.meshes = []
        self.materials = []
        self._had_blend = False
        self._blend_cache = None
        self.tex_descs = np.zeros((0, 8), dtype=np.uint32)  # r3n_texture_desc32 rows
        self.tex_pool = np.zeros(4, dtype=np.uint8)   # every texture's levels in ITS format (bytes)
        self.tex_used = 0
        self._tex_dirty = False
        self._pose_state = {}      # skeleton handle -> (clip, time): rend3-anim poses re-evaluated in front of every skinning pass
        self._anim_sets = None     # concatenated rend3-anim tables of every AnimationData (animation_add)
        self.output_format = 0
        self.capacity = 16  # FreelistDerivedBuffer::STARTING_SIZE
        self.object_meta = {}
        self.free_handles, self.pending_free, self.deferred_removals = [], [], []
        self.next_handle = 0
        self.world_version = 0  # bumped by every object / skeleton edit
        self.dirty_objects = {}
        self.dir_lights, self.point_lights = [], []
        self._camera_inputs = (host.identity(), ("raw", host.identity()))
        self._camera = None
        self.object_range = None
        self.skeletons = []
        # R3N_FRAME_NODES=1: render() issues the frame node by node through the graph mirror (one C call per reference node, what
        # a Rust integration's node closures do); default: Renderer::evaluate's CPU work and the whole node list behind ONE C call
        # each (r3n_host_evaluate_frame, r3n_render_frame)
        self.frame_nodes = os.environ.get("R3N_FRAME_NODES", "0") == "1"
        self._fc = None  # persistent ctypes blocks of the one-call path
        self._lights_version = 0
        self._capacity_sent = None
        self._resolution = (0, 0)
        self._write_objects([], force_capacity=True)

    def close(self):
        if self.ctx:
            self.lib.r3n_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, code, what):
        _ffi.check(self.ctx, code, what)

    # ------------------------------------------------------------------ world edits
    def add_mesh(self, positions, indices=None, normals=None, colors=None, mesh_handedness=host.LEFT, tangents=None,
                 joint_indices=None, joint_weights=None, uv0=None):
        positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
        if indices is None:
            indices = np.arange(len(positions), dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if normals is None:  # MeshBuilder::build (rend3-types/src/lib.rs:501-504)
            normals = host.calculate_normals(positions, indices, mesh_handedness == host.LEFT)
        normals = np.ascontiguousarray(normals, dtype=f32).reshape(-1, 3)
        m = _Mesh()
        m.attr_off = [INVALID] * 6
        chunks = []
        cursor = self.mesh_cursor

        def push(words):
            nonlocal cursor
            start = cursor
            chunks.append(np.ascontiguousarray(words))
            cursor += len(words)
            return start

        m.attr_off[0] = 4 * push(positions.view(np.uint32).reshape(-1))
        m.attr_off[1] = 4 * push(normals.view(np.uint32).reshape(-1))
        if tangents is not None:
            m.attr_off[2] = 4 * push(np.ascontiguousarray(tangents, dtype=f32).reshape(-1).view(np.uint32))
        if uv0 is not None:  # VERTEX_ATTRIBUTE_TEXTURE_COORDINATES_0: vec2<f32>
            m.attr_off[3] = 4 * push(np.ascontiguousarray(uv0, dtype=f32).reshape(-1, 2).reshape(-1).view(np.uint32))
        if colors is not None:
            colors = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 4)
            m.attr_off[5] = 4 * push(colors.view(np.uint32).reshape(-1))
        m.vertex_count = len(positions)
        m.joint_off = m.weight_off = INVALID
        if joint_indices is not None:  # [u16; 4] per vertex + vec4<f32> weights (rend3-types/src/attribute.rs:97-135)
            m.joint_off = 4 * push(np.ascontiguousarray(joint_indices, dtype=np.uint16).reshape(-1, 4).view(np.uint32).reshape(-1))
            m.weight_off = 4 * push(np.ascontiguousarray(joint_weights, dtype=f32).reshape(-1, 4).view(np.uint32).reshape(-1))
        m.first_index = push(indices)
        m.index_count = len(indices)
        blob = np.concatenate(chunks)
        self._check(self.lib.r3n_mesh_buffer_write(self.ctx, 4 * self.mesh_cursor, _ffi.ptr(blob), blob.nbytes),
                    "r3n_mesh_buffer_write")
        self.mesh_cursor = cursor
        m.centre, m.radius = host.bounding_sphere_from_mesh(positions)
        self.meshes.append(m)
        return len(self.meshes) - 1

    # ---- skeletons (rend3/src/managers/skeleton.rs:67-163)
    def add_skeleton(self, mesh, joint_matrices):
        m = self.meshes[mesh]
        if m.joint_off == INVALID:
            raise ValueError("Mesh must have joint indices to be used in a skeleton")  # SkeletonCreationError
        out_off = [INVALID] * 3
        for a in range(3):  # private position / normal / tangent copies (skeleton.rs:110-113)
            if m.attr_off[a] != INVALID:
                out_off[a] = 4 * self.mesh_cursor
                zeros = np.zeros(3 * m.vertex_count, dtype=np.uint32)
                self._check(self.lib.r3n_mesh_buffer_write(self.ctx, out_off[a], _ffi.ptr(zeros), zeros.nbytes), "r3n_mesh_buffer_write")
                self.mesh_cursor += len(zeros)
        self.skeletons.append(dict(mesh=mesh, out_off=out_off, matrices=np.ascontiguousarray(joint_matrices, dtype=f32).reshape(-1, 16)))
        self._skin_inputs = None
        return len(self.skeletons) - 1

    def add_skeletons_bulk(self, mesh, joint_matrices_per_skeleton):
        """Many skeletons of one mesh (config 5): one zero-fill upload for all private output ranges."""
        m = self.meshes[mesh]
        n = len(joint_matrices_per_skeleton)
        n_attr = sum(1 for a in range(3) if m.attr_off[a] != INVALID)
        words = 3 * m.vertex_count
        zeros = np.zeros(n * n_attr * words, dtype=np.uint32)
        base = self.mesh_cursor
        self._check(self.lib.r3n_mesh_buffer_write(self.ctx, 4 * base, _ffi.ptr(zeros), zeros.nbytes), "r3n_mesh_buffer_write")
        self.mesh_cursor += len(zeros)
        first = len(self.skeletons)
        for i in range(n):
            out_off, k = [INVALID] * 3, 0
            for a in range(3):
                if m.attr_off[a] != INVALID:
                    out_off[a] = 4 * (base + (i * n_attr + k) * words)
                    k += 1
            self.skeletons.append(dict(mesh=mesh, out_off=out_off,
                                       matrices=np.ascontiguousarray(joint_matrices_per_skeleton[i], dtype=f32).reshape(-1, 16)))
        self._skin_inputs = None
        return list(range(first, first + n))

    def set_skeleton_joint_matrices(self, sk, joint_matrices):
        self.world_version += 1  # skinned vertices may leave the bounds the partition was built from
        self._pose_state.pop(sk, None)
        self.skeletons[sk]["matrices"] = np.ascontiguousarray(joint_matrices, dtype=f32).reshape(-1, 16)

    def animation_add(self, rigs, joints, clips, tracks, times, values):
        """Register one AnimationData's tables (anim.AnimationData builds them; record layouts in include/r3n.h).  The
        library holds ONE table set (r3n_animation_write), so the sets of all scene instances are concatenated here with
        their indices rebased.  Returns the index of the set's first clip."""
        sets = self._anim_sets
        if sets is None:
            sets = self._anim_sets = [[np.zeros(0, dtype=a.dtype) for a in (rigs, joints, clips, tracks)] + [np.zeros(0, f32), np.zeros(0, f32)]]
        cur = sets[0]
        rigs, clips, tracks = rigs.copy(), clips.copy(), tracks.copy()
        rigs["first"] += len(cur[1])
        clips["rig"] += len(cur[0])
        clips["track"] += len(cur[3])
        tracks["kf"] += np.where(tracks["kc"] > 0, len(cur[4]), 0).astype(np.uint32)
        tracks["vf"] += np.where(tracks["kc"] > 0, len(cur[5]), 0).astype(np.uint32)
        clip_base = len(cur[2])
        sets[0] = [np.concatenate([a, b]) for a, b in zip(cur, (rigs, joints, clips, tracks, np.asarray(times, f32), np.asarray(values, f32)))]
        arrs = [np.ascontiguousarray(a) for a in sets[0]]
        args = []
        for a, rec in zip(arrs, (16, 80, 16, 80, 4, 4)):
            args += [_ffi.ptr(a) if a.size else None, a.nbytes // rec]
        self._check(self.lib.r3n_animation_write(self.ctx, *args), "r3n_animation_write")
        return clip_base

    def pose_skeletons(self, requests):
        """rend3-anim poses: requests = [(clip, time, skeleton handle)].  The joint matrices are evaluated on the GPU
        (csrc/anim.hip) in front of every skinning pass, straight into the buffer the skinning kernel reads, until the
        skeleton gets another pose or explicit matrices (Renderer::set_skeleton_joint_matrices semantics: the last
        value set stays)."""
        self.world_version += 1
        for clip, time, sk in requests:
            self._pose_state[sk] = (int(clip), np.float32(time))

    def _pose_requests(self):
        state = self._pose_state
        rq = np.zeros(len(state), dtype=[("clip", np.uint32), ("time", np.float32), ("base", np.uint32), ("pad", np.uint32)])
        for i, (sk, (clip, time)) in enumerate(sorted(state.items())):
            rq[i] = (clip, time, int(self._skin_inputs[sk][8]), 0)
        return rq

    def skinning_buffers(self):
        """build_gpu_skinning_input_buffers (rend3-routine/src/skinning.rs:54-139)"""
        if getattr(self, "_skin_inputs", None) is None:
            inputs = np.zeros((len(self.skeletons), 10), dtype=np.uint32)
            base = 0
            for i, sk in enumerate(self.skeletons):
                m = self.meshes[sk["mesh"]]
                inputs[i] = [m.attr_off[0], m.attr_off[1], m.attr_off[2], m.joint_off, m.weight_off, sk["out_off"][0],
                             sk["out_off"][1], sk["out_off"][2], base, m.vertex_count]
                base += len(sk["matrices"])
            self._skin_inputs = inputs
        mats = np.ascontiguousarray(np.concatenate([sk["matrices"] for sk in self.skeletons]))
        return self._skin_inputs, mats

    def add_texture_2d(self, rgba8, srgb=True, mip_count=1, mip_source="uploaded"):
        """Renderer::add_texture_2d with Texture{data, format, size, mip_count, mip_source}: rgba8 = (H, W, 4) u8;
        format Rgba8UnormSrgb | Rgba8Unorm; mip_count int or "maximum"; mip_source "uploaded" | "generated".
        The whole bindless array is re-sent (r3n_textures_write_encoded).  Returns the texture handle (index)."""
        data, w, h, mips, stored = host.prepare_texture(rgba8, srgb, mip_count, mip_source)
        return self._append_texture(data, w, h, mips, 1 if srgb else 0, stored)

    def add_texture_2d_encoded(self, fmt, width, height, levels, generate_mips=False):
        """Renderer::add_texture_2d for the other formats rend3-gltf's loader produces (containers.py ids = R3N_TEXTURE_*):
        `levels` = the stored levels' bytes, largest first.  generate_mips: MipmapCount::Maximum + MipmapSource::Generated
        (single-level uncompressed files, rend3-gltf/src/lib.rs:1031-1036); the chain is then built from the expanded
        RGBA8 level (on the GPU, like the expansion itself), which gives every channel the values the format's own blit
        chain would.  Block-compressed data goes to the GPU as stored and is decoded there."""
        from . import containers
        if generate_mips:
            if not containers.generate_mips_allowed(fmt) or len(levels) != 1:
                raise ValueError("mips are generated for single-level 8-bit uncompressed textures only")
            mips = int(max(width, height)).bit_length()
            return self._append_texture(np.frombuffer(levels[0], dtype=np.uint8), width, height, mips, fmt, 1 if mips > 1 else 0)
        for k, lv in enumerate(levels):
            if len(lv) != containers.level_bytes(fmt, max(1, width >> k), max(1, height >> k)):
                raise ValueError(f"level {k}: wrong byte count for its extent")
        return self._append_texture(np.frombuffer(b"".join(levels), dtype=np.uint8), width, height, len(levels), fmt)

    def _append_texture(self, data_u8, w, h, mips, fmt, stored=0):
        start = (self.tex_used + 3) & ~3  # level 0 of every texture starts on a 4-byte boundary
        end = start + len(data_u8)
        if end > len(self.tex_pool):
            grown = np.zeros(max(2 * len(self.tex_pool), end), dtype=np.uint8)
            grown[: self.tex_used] = self.tex_pool[: self.tex_used]
            self.tex_pool = grown
        self.tex_pool[start:end] = data_u8
        self.tex_used = end
        desc = np.array([[start, w, h, mips, fmt, stored, 0, 0]], dtype=np.uint32)
        self.tex_descs = np.ascontiguousarray(np.concatenate([self.tex_descs, desc]))
        self._tex_dirty = True  # the array is re-sent once, when the next frame is evaluated (TextureManager::evaluate)
        return len(self.tex_descs) - 1

    def _flush_textures(self):
        if self._tex_dirty:
            self._check(self.lib.r3n_textures_write_encoded(self.ctx, _ffi.ptr(self.tex_descs), len(self.tex_descs),
                                                            _ffi.ptr(self.tex_pool), self.tex_used), "r3n_textures_write_encoded")
            self._tex_dirty = False

    def add_material(self, record, key=OPAQUE):
        idx = len(self.materials)
        self.materials.append((np.asarray(record, dtype=f32), key))
        slots = np.array([idx], dtype=np.uint32)
        rec = np.ascontiguousarray(record, dtype=f32).reshape(1, 52)
        keys = np.array([key], dtype=np.uint8)
        self._check(self.lib.r3n_materials_write(self.ctx, _ffi.ptr(slots), _ffi.ptr(rec), _ffi.ptr(keys), 1),
                    "r3n_materials_write")
        return idx

    def update_material(self, handle, record, key=None):
        """Renderer::update_material: rewrites the material's 208-B record in place (r3n_materials_write orders itself
        after a resolve still in flight)."""
        key = self.materials[handle][1] if key is None else key
        if key != self.materials[handle][1]:
            self._blend_cache = None  # the material moved into / out of the transparent pass
        self.materials[handle] = (np.asarray(record, dtype=f32), key)
        slots = np.array([handle], dtype=np.uint32)
        rec = np.ascontiguousarray(record, dtype=f32).reshape(1, 52)
        keys = np.array([key], dtype=np.uint8)
        self._check(self.lib.r3n_materials_write(self.ctx, _ffi.ptr(slots), _ffi.ptr(rec), _ffi.ptr(keys), 1), "r3n_materials_write")

    def update_directional_light(self, handle, **changes):
        """Renderer::update_directional_light with a DirectionalLightChange (only the given fields change)."""
        self.dir_lights[handle].update(changes)
        self._lights_version += 1

    def update_point_light(self, handle, **changes):
        self.point_lights[handle].update(changes)
        self._lights_version += 1

    def _alloc_handle(self):
        if self.free_handles:
            return self.free_handles.pop(0)
        h = self.next_handle
        self.next_handle += 1
        return h

    def _use_index(self, idx):
        cap = self.capacity
        if idx > cap:  # freelist/buffer.rs:48-52
            cap = 1 << (int(idx) - 1).bit_length()
        while cap <= idx:
            cap *= 2
        self.capacity = cap

    def _object_record(self, h):
        meta = self.object_meta[h]
        mesh = self.meshes[meta["mesh"]]
        rec = np.zeros(32, dtype=np.uint32)
        rf = rec.view(f32)
        rf[0:16] = meta["transform"]
        c, r = host.bounding_sphere_apply_transform(mesh.centre, mesh.radius, meta["transform"])
        rf[16:19] = c
        rf[19] = r
        rec[20], rec[21], rec[22] = mesh.first_index, mesh.index_count, meta["material"]
        rec[23:29] = mesh.attr_off
        if meta.get("skeleton") is not None:  # object.rs:250-258: skeleton ranges override the mesh's
            for a, off in enumerate(self.skeletons[meta["skeleton"]]["out_off"]):
                if off != INVALID:
                    rec[23 + a] = off
        rec[29] = 1 if meta["enabled"] else 0
        return rec

    def _mark(self, h, rec):
        self.world_version += 1  # an object was added / moved / removed (parallel.Exchange: host-side bounds are stale)
        self._blend_cache = None
        self._use_index(h)
        self.dirty_objects[h] = rec

    def _write_objects(self, items, force_capacity=False):
        if not items and not force_capacity:
            return
        slots = np.array([h for h, _ in items], dtype=np.uint32)
        recs = np.ascontiguousarray(np.stack([r for _, r in items]) if items else np.zeros((0, 32), dtype=np.uint32))
        self._check(self.lib.r3n_objects_write(self.ctx, _ffi.ptr(slots) if len(slots) else None,
                                               _ffi.ptr(recs) if len(slots) else None, len(slots), self.capacity),
                    "r3n_objects_write")

    def add_object(self, mesh, material, transform, skeleton=None):
        h = self._alloc_handle()
        if skeleton is not None:
            mesh = self.skeletons[skeleton]["mesh"]
        self.object_meta[h] = dict(mesh=mesh, material=material, transform=np.asarray(transform, dtype=f32).copy(),
                                   enabled=True, skeleton=skeleton)
        rec = self._object_record(h)
        self._mark(h, rec)
        # object.rs:273: a new object's sorting location is its transformed bounding-sphere centre
        self.object_meta[h]["location"] = rec.view(f32)[16:19].copy()
        self.object_meta[h]["sphere"] = rec.view(f32)[16:20].copy()  # world-space bounding sphere (multi-GPU partitioning)
        return h

    def add_objects_bulk(self, mesh_ids, material_ids, transforms):
        """Vectorised add_object for large synthetic scenes (same records as add_object, built by the C++ host mirror)."""
        n = len(mesh_ids)
        transforms = np.ascontiguousarray(transforms, dtype=f32).reshape(n, 16)
        desc = np.zeros((n, 4), dtype=f32)
        mu = np.zeros((n, 8), dtype=np.uint32)
        for i, mid in enumerate(mesh_ids):
            m = self.meshes[mid]
            desc[i, :3], desc[i, 3] = m.centre, m.radius
            mu[i, 0], mu[i, 1] = m.first_index, m.index_count
            mu[i, 2:8] = m.attr_off
        mats = np.ascontiguousarray(material_ids, dtype=np.uint32)
        recs = np.zeros((n, 32), dtype=np.uint32)
        self.lib.r3n_host_build_object_records(n, _ffi.ptr(transforms), _ffi.ptr(desc), _ffi.ptr(mu), _ffi.ptr(mats), _ffi.ptr(recs))
        handles = []
        for i in range(n):
            h = self._alloc_handle()
            self.object_meta[h] = dict(mesh=int(mesh_ids[i]), material=int(material_ids[i]), transform=transforms[i], enabled=True,
                                       location=recs[i].view(f32)[16:19].copy(), sphere=recs[i].view(f32)[16:20].copy())
            self._mark(h, recs[i])
            handles.append(h)
        return handles

    def set_object_transform(self, h, transform):
        self.object_meta[h]["transform"] = np.asarray(transform, dtype=f32).copy()
        rec = self._object_record(h)
        self.object_meta[h]["sphere"] = rec.view(f32)[16:20].copy()
        self._mark(h, rec)
        # object.rs:313: after a transform update the sorting location is the translation
        self.object_meta[h]["location"] = self.object_meta[h]["transform"][12:15].copy()

    def remove_object(self, h):
        self.object_meta[h]["enabled"] = False  # object.rs:330-342: disabled now, removed next frame
        self._mark(h, self._object_record(h))
        self.deferred_removals.append(h)

    def add_directional_light(self, color=(1, 1, 1), intensity=1.0, direction=(0, -1, 0), distance=100.0,
                              resolution=2048):
        self.dir_lights.append(dict(color=color, intensity=intensity, direction=direction, distance=distance,
                                    resolution=resolution))
        self._lights_version += 1
        return len(self.dir_lights) - 1

    def add_point_light(self, position, color=(1, 1, 1), intensity=1.0, radius=1.0):
        self.point_lights.append(dict(position=position, color=color, intensity=intensity, radius=radius))
        self._lights_version += 1
        return len(self.point_lights) - 1

    def set_camera_data(self, view, projection):
        """Renderer::set_camera_data: the CameraState (camera.rs:23-85) is derived when something asks for it -- in the one-call
        frame path that is r3n_host_evaluate_frame, in C++."""
        self._camera_inputs = (view, projection)
        self._camera = None

    @property
    def camera(self):
        if self._camera is None:
            self._camera = host.CameraState(self._camera_inputs[0], self._camera_inputs[1], self.handedness, self.aspect_ratio)
        return self._camera

    def current_view_proj(self):
        """view_proj of the camera the last evaluated frame used (f32[16], column-major)."""
        if self._fc is not None and not self.frame_nodes:
            return np.array(self._fc["frame"].view_proj[:], dtype=f32)
        return self.camera.view_proj

    def current_resolution(self):
        return self._resolution

    def set_object_owners(self, owners, rank):
        """Multi-GPU sharding by owner byte (parallel.partition_objects_spatial): this rank culls / draws the slots it owns."""
        owners = np.ascontiguousarray(owners, dtype=np.uint8)
        self._check(self.lib.r3n_set_object_owners(self.ctx, owners.ctypes.data if len(owners) else None, len(owners), rank), "r3n_set_object_owners")

    def comm_init(self, rank, world, share):
        """Multi-GPU, native (r3n_comm_init): the sort-first exchanges are issued by the library itself over RCCL inside
        r3n_render_frame -- no exchange object, no host-language collective on the frame path.  `share(ids or None) -> ids`: hands
        rank 0's communicator ids (bytes) to every rank (comm_init_torch passes them through torch.distributed).  Collective."""
        n = _ffi.COMM_IDS * _ffi.COMM_ID_BYTES
        ids = None
        if rank == 0:
            buf = (ctypes.c_uint8 * n)()
            for k in range(_ffi.COMM_IDS):
                self._check(self.lib.r3n_comm_unique_id(ctypes.byref(buf, k * _ffi.COMM_ID_BYTES)), "r3n_comm_unique_id")
            ids = bytes(buf)
        ids = share(ids)
        assert isinstance(ids, (bytes, bytearray)) and len(ids) == n
        self._check(self.lib.r3n_comm_init(self.ctx, (ctypes.c_uint8 * n).from_buffer_copy(ids), rank, world), "r3n_comm_init")
        self._comm = (int(rank), int(world))

    def comm_init_torch(self, group=None):
        import torch.distributed as dist

        def share(ids):
            box = [ids]
            dist.broadcast_object_list(box, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
            return box[0]
        self.comm_init(dist.get_rank(group), dist.get_world_size(group), share)

    def comm_set_split(self, by_objects):
        """The split the library's own exchanges implement (r3n_comm_set_split): False = sort-first rows (what comm_init selects),
        True = object ranges (set_object_range / owners): depth MAX all-reduce in front of Hi-Z + key MAX reduce-scatter behind
        pass 2, issued by r3n_render_frame."""
        self._check(self.lib.r3n_comm_set_split(self.ctx, 0 if by_objects else 1), "r3n_comm_set_split")

    def comm_destroy(self):
        self._check(self.lib.r3n_comm_destroy(self.ctx), "r3n_comm_destroy")
        self._comm = None

    def set_object_range(self, begin, end):
        """Multi-GPU sharding (not in the reference): this rank culls/draws object slots [begin, end)."""
        self._check(self.lib.r3n_set_object_range(self.ctx, begin, end), "r3n_set_object_range")

    def set_camera_object_range(self, camera, begin, end):
        """This camera's own object range (a shadow view owned whole by one rank draws every slot there)."""
        self._check(self.lib.r3n_set_camera_object_range(self.ctx, camera, begin, end), "r3n_set_camera_object_range")

    # ------------------------------------------------------------------ per-frame evaluation
    def evaluate_instructions(self):
        self._flush_textures()
        return self._evaluate_instructions()

    def _flush_objects(self):
        """ObjectManager::evaluate (object.rs:344-364): last frame's removals become real, dirty records are scattered."""
        for h in self.pending_free:
            self.dirty_objects[h] = np.zeros(32, dtype=np.uint32)  # unwrap_or_default(), object.rs:363
            self.object_meta.pop(h, None)
            self.free_handles.append(h)
        self.pending_free = self.deferred_removals
        self.deferred_removals = []
        if self.dirty_objects or self._capacity_sent != self.capacity:
            self._write_objects(sorted(self.dirty_objects.items()), force_capacity=True)
            self.dirty_objects = {}
            self._capacity_sent = self.capacity

    def _evaluate_instructions(self):
        """Renderer::evaluate_instructions (rend3/src/renderer/eval.rs:9-187), path subset: flush dirty objects,
        evaluate lights -> shadow cameras + atlas + light buffers."""
        self._flush_objects()
        size, shadows, dir_buf = host.evaluate_directional_lights(self.dir_lights, self.camera)
        point_buf = host.point_light_buffer(self.point_lights)
        self._check(self.lib.r3n_lights_write(self.ctx, _ffi.ptr(dir_buf), dir_buf.nbytes, _ffi.ptr(point_buf),
                                              point_buf.nbytes), "r3n_lights_write")
        self._dir_buf, self._point_buf = dir_buf, point_buf
        self._write_blend_order(self.camera.location)
        return EvalOutput(shadows, size)

    def _write_blend_order(self, camera_location):
        # the CPU batcher's back-to-front order of the blend-key objects (batching.rs:146-176), every frame
        if self._blend_cache is None:  # rebuilt only after objects / materials changed
            self._blend_cache = [h for h, m in sorted(self.object_meta.items())
                                 if m["enabled"] and self.materials[m["material"]][1] == BLEND]
        blend = self._blend_cache
        order = host.blend_draw_order(camera_location, blend, [self.object_meta[h]["location"] for h in blend]) if blend else []
        if order or self._had_blend:
            arr = np.asarray(order, dtype=np.uint32)
            self._check(self.lib.r3n_blend_order_write(self.ctx, _ffi.ptr(arr) if len(arr) else None, len(arr)), "r3n_blend_order_write")
        self._had_blend = bool(order)

    # ------------------------------------------------------------------ convenience: one whole frame
    def render(self, width, height, samples=1, ambient=(0, 0, 0, 0), clear_color=(0, 0, 0, 0), readback=True,
               base=None, exchange=None):
        """The reference's per-frame driver (rend3-test/src/runner.rs:121-169): evaluate, build the graph with
        BaseRenderGraph::add_to_graph, execute.  `readback` additionally pulls the parity taps."""
        self._resolution = (width, height)
        comm = getattr(self, "_comm", None)
        if comm is not None and (self.frame_nodes or exchange is not None):
            raise RuntimeError("comm_init: the library issues the exchanges inside r3n_render_frame (no exchange object, no per-node frame)")
        if not self.frame_nodes:
            eval_output = self.render_frame(width, height, samples, ambient, clear_color, exchange,
                                            viewport_first=bool(base is not None and base.viewport_first))
            if not readback:
                return None
            owned = None
            if exchange is not None and hasattr(exchange, "owns_shadow_view"):
                owned = {si for si in range(len(eval_output.shadows)) if exchange.owns_shadow_view(si)}
            elif comm is not None:
                owned = {si for si in range(len(eval_output.shadows)) if si % comm[1] == comm[0]}
            return self.readback_frame(eval_output, width, height, samples, owned)
        eval_output = self.evaluate_instructions()
        base = base or BaseRenderGraph(self)
        graph = RenderGraph()
        inputs = BaseRenderGraphInputs(eval_output, base.default_routines(), (width, height), samples)
        base.add_to_graph(graph, inputs, BaseRenderGraphSettings(ambient, clear_color), exchange=exchange)
        graph.execute(self, eval_output)
        if not readback:
            return None
        owned = None
        if exchange is not None and hasattr(exchange, "owns_shadow_view"):
            owned = {si for si in range(len(eval_output.shadows)) if exchange.owns_shadow_view(si)}
        return self.readback_frame(eval_output, width, height, samples, owned)

    # ------------------------------------------------------------------ the frame in two C calls
    def render_frame(self, width, height, samples=1, ambient=(0, 0, 0, 0), clear_color=(0, 0, 0, 0), exchange=None,
                     viewport_first=False):
        """evaluate_instructions + BaseRenderGraph::add_to_graph + RenderGraph::execute with the host out of the loop: the CPU
        half (camera state, shadow cameras + atlas, light buffer, frame uniforms, camera headers) is r3n_host_evaluate_frame, the
        node list r3n_render_frame.  Same inputs, same results as the node-by-node path (R3N_FRAME_NODES=1); returns the
        EvalOutput (shadow layout) the read-backs want."""
        import ctypes as ct
        lib = self.lib
        fc = self._fc
        if fc is None:
            fc = self._fc = dict(cam=_ffi.HostCamera144(), frame=_ffi.HostFrame(), desc=_ffi.FrameDesc(), lights=None, lights_version=-1,
                                 point=None, ambient=(ct.c_float * 4)(), cb=None, cb_for=None, error=None)
            d = fc["desc"]
            d.struct_size = ct.sizeof(_ffi.FrameDesc)
            fr = fc["frame"]
            base = ct.addressof(fr)
            d.uniforms = base + _ffi.HostFrame.uniforms.offset
            d.viewport_header = base + _ffi.HostFrame.viewport_header.offset
            d.shadow_views = base + _ffi.HostFrame.shadow_views.offset
            d.directional_buffer = base + _ffi.HostFrame.directional_buffer.offset
        self._flush_textures()
        self._flush_objects()
        cam, fr, d = fc["cam"], fc["frame"], fc["desc"]
        view, projection = self._camera_inputs
        ct.memmove(cam.view, np.ascontiguousarray(view, dtype=f32).ctypes.data, 64)
        kind = projection[0]
        cam.handedness = 1 if self.handedness == host.RIGHT else 0
        cam.aspect_ratio = 0.0 if self.aspect_ratio is None else float(self.aspect_ratio)
        if kind == "perspective":
            cam.projection_kind = 1
            cam.projection_params[0], cam.projection_params[1] = projection[1], projection[2]
        elif kind == "orthographic":
            cam.projection_kind = 0
            cam.projection_params[0], cam.projection_params[1], cam.projection_params[2] = projection[1]
        elif kind == "raw":
            cam.projection_kind = 2
            ct.memmove(cam.projection_params, np.ascontiguousarray(projection[1], dtype=f32).ctypes.data, 64)
        else:
            raise ValueError(kind)
        if fc["lights_version"] != self._lights_version:
            la = np.zeros((max(len(self.dir_lights), 1), 12), dtype=f32)
            for i, l in enumerate(self.dir_lights):
                if l is None:
                    continue
                la[i, 0:3], la[i, 3] = l["color"], l["intensity"]
                la[i, 4:7], la[i, 7] = l["direction"], l["distance"]
                la[i, 8:9].view(np.uint32)[0] = l["resolution"]
            fc["lights"] = la
            fc["point"] = np.ascontiguousarray(host.point_light_buffer(self.point_lights))
            fc["lights_version"] = self._lights_version
        amb = fc["ambient"]
        amb[0], amb[1], amb[2], amb[3] = ambient
        if lib.r3n_host_evaluate_frame(ct.byref(cam), fc["lights"].ctypes.data, len(self.dir_lights), 16384, amb, width, height, samples,
                                       self.capacity, ct.byref(fr)) != 0:
            raise _ffi.R3nError("r3n_host_evaluate_frame: more shadow-casting lights than R3N_MAX_SHADOW_VIEWS")
        n_views = fr.n_shadow_views
        shadows = [dict(offset=(fr.shadow_views[k].x, fr.shadow_views[k].y), size=fr.shadow_views[k].size, handle=fr.shadow_handles[k])
                   for k in range(n_views)]
        ev = EvalOutput(shadows, (fr.shadow_atlas_width, fr.shadow_atlas_height))
        if self._blend_cache is None or self._blend_cache or self._had_blend:
            self._write_blend_order(np.array(fr.camera_location[:], dtype=f32))
        d.flags = _ffi.FRAME_VIEWPORT_FIRST if viewport_first else 0
        d.width, d.height, d.samples = width, height, samples
        d.shadow_atlas_width, d.shadow_atlas_height, d.n_shadow_views = fr.shadow_atlas_width, fr.shadow_atlas_height, n_views
        d.clear_color[0], d.clear_color[1], d.clear_color[2], d.clear_color[3] = clear_color
        d.directional_bytes = fr.directional_bytes
        d.point_buffer, d.point_bytes = fc["point"].ctypes.data, fc["point"].nbytes
        d.shadow_view_mask = 0
        if exchange is not None and hasattr(exchange, "owns_shadow_view"):
            d.flags |= _ffi.FRAME_SHADOW_MASK
            d.shadow_view_mask = sum(1 << v for v in range(n_views) if exchange.owns_shadow_view(v))
        keep = None
        if self.skeletons:  # skinning (base.rs:145, skinning.rs:211-226)
            sk_in, sk_m = self.skinning_buffers()
            poses = self._pose_requests()
            if len(poses):
                self._check(lib.r3n_pose_skeletons(self.ctx, _ffi.ptr(np.ascontiguousarray(poses)), len(poses)), "r3n_pose_skeletons")
            keep = (sk_in, sk_m)
            d.skin_inputs, d.n_skeletons, d.joint_matrices, d.n_joint_matrices = sk_in.ctypes.data, len(sk_in), sk_m.ctypes.data, len(sk_m)
        else:
            d.skin_inputs, d.n_skeletons, d.joint_matrices, d.n_joint_matrices = None, 0, None, 0
        if exchange is not None:
            state = dict(ev=ev, samples=samples)
            fc["cb_state"] = state
            if fc["cb_for"] is not exchange:
                def callback(_user, site, _self=self, _fc=fc, _exchange=exchange):
                    try:
                        st = _fc["cb_state"]
                        _exchange(_ffi.EXCHANGE_SITES[site], _self, ev=st["ev"], samples=st["samples"])
                        return 0
                    except BaseException as e:  # never unwind through the C frames
                        _fc["error"] = e
                        return -1
                fc["cb"], fc["cb_for"] = _ffi.EXCHANGE_FN(callback), exchange
            d.exchange = fc["cb"]
        else:
            d.exchange = _ffi.EXCHANGE_FN()
        code = lib.r3n_render_frame(self.ctx, ct.byref(d))
        del keep
        if fc["error"] is not None:
            err, fc["error"] = fc["error"], None
            raise err
        self._check(code, "r3n_render_frame")
        return ev

    def readback_frame(self, eval_output, width, height, samples=1, owned_views=None):
        lib, ctx = self.lib, self.ctx
        cap = self.capacity
        out = {"capacity": cap, "shadows": []}
        total = int(sum(self.meshes[m["mesh"]].index_count // 3 for m in self.object_meta.values() if m["enabled"]))

        def cam_sets(cam):
            visible = np.zeros(cap, dtype=np.uint8)
            self._check(lib.r3n_readback_visible_objects(ctx, cam, _ffi.ptr(visible), cap), "readback_visible_objects")
            p = np.zeros(max(total, 1), dtype=np.uint8)
            r = np.zeros(max(total, 1), dtype=np.uint8)
            self._check(lib.r3n_readback_triangle_sets(ctx, cam, _ffi.ptr(p), _ffi.ptr(r), len(p)), "readback_triangle_sets")
            calls = np.zeros((6, 5), dtype=np.uint32)
            self._check(lib.r3n_readback_draw_calls(ctx, cam, _ffi.ptr(calls)), "readback_draw_calls")
            baked = np.zeros((cap, 32), dtype=f32)
            self._check(lib.r3n_readback_baked(ctx, cam, _ffi.ptr(baked), cap), "readback_baked")
            return dict(visible=visible, residual=r, draw_calls=calls, baked=baked, **{"pass": p})

        if cap and self.object_meta:
            for si in range(len(eval_output.shadows)):
                # (multi-GPU: a shadow view this rank does not own was never culled here)
                out["shadows"].append(cam_sets(si) if owned_views is None or si in owned_views else None)
            out.update(cam_sets(_ffi.CAMERA_VIEWPORT))
        vis = np.zeros((height, width) if samples == 1 else (height, width, samples), dtype=np.uint64)
        self._check(lib.r3n_readback_visibility(ctx, _ffi.ptr(vis)), "readback_visibility")
        aw, ah = eval_output.shadow_target_size
        atlas = np.zeros((ah, aw), dtype=f32)
        self._check(lib.r3n_readback_shadow_atlas(ctx, _ffi.ptr(atlas)), "readback_shadow_atlas")
        hdr16 = np.zeros((height, width, 4), dtype=np.uint16)
        self._check(lib.r3n_readback_hdr(ctx, _ffi.ptr(hdr16)), "readback_hdr")
        rgba8 = np.zeros((height, width, 4), dtype=np.uint8)
        rgba_f = np.zeros((height, width, 4), dtype=f32)
        self._check(lib.r3n_readback_output(ctx, _ffi.ptr(rgba8), _ffi.ptr(rgba_f)), "readback_output")
        out.update(vis=vis, atlas=atlas, atlas_size=(aw, ah), hdr16=hdr16, rgba8=rgba8, rgba_f32=rgba_f)
        return out

    def readback_mesh_words(self, byte_offset, n_words):
        out = np.zeros(n_words, dtype=np.uint32)
        self._check(self.lib.r3n_readback_mesh(self.ctx, byte_offset, _ffi.ptr(out), out.nbytes), "r3n_readback_mesh")
        return out

    def set_output_format(self, fmt):
        """TonemappingRoutine's output_format: 0 Rgba8UnormSrgb (default), 1 Bgra8UnormSrgb, 2 Rgba8Unorm, 3 Bgra8Unorm."""
        self._check(self.lib.r3n_set_output_format(self.ctx, int(fmt)), "r3n_set_output_format")
        self.output_format = int(fmt)

    def set_skinning_mode(self, mode):
        """0 = R3N_SKIN_EXACT (vector ALU, default), 1 = R3N_SKIN_MFMA (matrix cores; rigs of at most four joints)."""
        self._check(self.lib.r3n_set_skinning_mode(self.ctx, int(mode)), "r3n_set_skinning_mode")

    def set_shade_mode(self, mode):
        """0 = R3N_SHADE_EXACT (default, bit-identical to the oracle), 1 = R3N_SHADE_FAST (fused / approximate shading arithmetic,
        framebuffer within 1e-3 after tonemap)."""
        self._check(self.lib.r3n_set_shade_mode(self.ctx, int(mode)), "r3n_set_shade_mode")

    def readback_joint_matrices(self):
        """The joint matrices the last skinning pass read (host-provided and GPU-posed), (n, 16) f32."""
        n = sum(len(sk["matrices"]) for sk in self.skeletons)
        out = np.zeros((max(n, 1), 16), dtype=f32)
        if n:
            self._check(self.lib.r3n_readback_joint_matrices(self.ctx, 0, _ffi.ptr(out), n), "r3n_readback_joint_matrices")
        return out[:n]

    def readback_texels(self, per_texture=False):
        """The decoded texels of every texture (levels back to back, array order): RGBA8 textures as (n, 4) u8, textures
        of the float-decoded formats as (n, 4) f32.  per_texture=False concatenates them into one (n, 4) u8 array (a float
        texel is then four rows).  In the library's pool every texture starts on a 4-word boundary
        (r3n_textures_write_encoded); the gaps are dropped here."""
        from . import containers
        self._flush_textures()
        is_float = [containers.is_float_format(int(d[4])) for d in self.tex_descs]
        sizes = [sum(max(1, int(d[1]) >> k) * max(1, int(d[2]) >> k) for k in range(int(d[3]))) * (4 if fl else 1)
                 for d, fl in zip(self.tex_descs, is_float)]
        starts, cur = [], 0
        for n in sizes:
            cur = (cur + 3) & ~3
            starts.append(cur)
            cur += n
        pool = np.zeros(max(cur, 1), dtype=np.uint32)
        if cur:
            self._check(self.lib.r3n_readback_texels(self.ctx, 0, _ffi.ptr(pool), cur), "r3n_readback_texels")
        parts = [pool[s0:s0 + n] for s0, n in zip(starts, sizes)]
        if per_texture:
            return [p.view(np.float32).reshape(-1, 4) if fl else p.view(np.uint8).reshape(-1, 4) for p, fl in zip(parts, is_float)]
        return (np.concatenate(parts) if parts else pool[:0]).view(np.uint8).reshape(-1, 4)

    def readback_hiz(self, width, height):
        n = 0
        k = 0
        m = max(width, height)
        while m:
            n += max(1, width >> k) * max(1, height >> k)
            k += 1
            m >>= 1
        pyr = np.zeros(n, dtype=f32)
        self._check(self.lib.r3n_readback_hiz(self.ctx, _ffi.ptr(pyr), n), "readback_hiz")
        return pyr

    # ------------------------------------------------------------------ timing taps
    def timing_enable(self, on=True):
        self._check(self.lib.r3n_timing_enable(self.ctx, 1 if on else 0), "r3n_timing_enable")

    def set_multi_stream(self, on=True):
        self._check(self.lib.r3n_set_multi_stream(self.ctx, 1 if on else 0), "r3n_set_multi_stream")

    def stage_times(self, reset=True):
        ms = np.zeros(len(_ffi.STAGES), dtype=np.float64)
        n = np.zeros(len(_ffi.STAGES), dtype=np.uint64)
        self._check(self.lib.r3n_stage_times(self.ctx, _ffi.ptr(ms), _ffi.ptr(n), 1 if reset else 0), "r3n_stage_times")
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(_ffi.STAGES)}

    def hbm_copy_rate(self, nbytes=1 << 30, repeats=5):
        """GB/s of a float4 copy of `nbytes` on this device (read + write bytes): the measured HBM roofline denominator."""
        out = ctypes.c_double(0.0)
        self._check(self.lib.r3n_hbm_copy_rate(self.ctx, int(nbytes), int(repeats), ctypes.byref(out)), "r3n_hbm_copy_rate")
        return float(out.value)

    def sync(self):
        self._check(self.lib.r3n_sync(self.ctx), "r3n_sync")


# ---------------------------------------------------------------------------------------------------- graph
class RenderGraph:
    """Fixed-order stand-in for rend3::graph::RenderGraph: nodes are (name, closure) run in declaration order."""

    def __init__(self):
        self.nodes = []

    def add_node(self, name, body):
        self.nodes.append((name, body))

    def execute(self, renderer, eval_output):
        for _name, body in self.nodes:
            body(renderer, eval_output)


class BaseRenderGraphSettings:
    """rend3-routine/src/base.rs:94-98"""

    def __init__(self, ambient_color=(0, 0, 0, 0), clear_color=(0, 0, 0, 0)):
        self.ambient_color = ambient_color
        self.clear_color = clear_color


class BaseRenderGraphInputs:
    """rend3-routine/src/base.rs:72-92 (eval_output, routines, target{resolution, samples})"""

    def __init__(self, eval_output, routines, resolution, samples=1):
        self.eval_output = eval_output
        self.routines = routines
        self.resolution = resolution
        self.samples = samples


class GpuCuller:
    """rend3-routine/src/culling/culler.rs:185-714"""

    def add_object_uniform_upload_to_graph(self, graph, camera_specifier, resolution, samples, name):
        def body(r, ev):
            cam = r.camera if camera_specifier == CameraSpecifier.VIEWPORT else ev.shadows[camera_specifier]["camera"]
            hdr = host.camera_header(cam, None if camera_specifier == CameraSpecifier.VIEWPORT else camera_specifier,
                                     resolution, samples, r.capacity)
            r._check(r.lib.r3n_uniform_bake(r.ctx, camera_specifier, _ffi.ptr(hdr)), "r3n_uniform_bake")

        graph.add_node(name, body)

    def add_culling_to_graph(self, graph, camera_specifier, name):
        graph.add_node(name, lambda r, ev: r._check(r.lib.r3n_cull(r.ctx, camera_specifier), "r3n_cull"))


class ForwardRoutine:
    """rend3-routine/src/forward.rs:135-316: one (routine type, material key) pipeline."""

    def __init__(self, routine_type, material_key):
        self.routine_type = routine_type
        self.material_key = material_key

    def add_forward_to_graph(self, graph, label, camera, culling_source):
        def body(r, ev):
            r._check(r.lib.r3n_forward(r.ctx, camera, self.routine_type, culling_source, self.material_key), "r3n_forward")

        graph.add_node(label, body)


class HiZRoutine:
    """rend3-routine/src/hi_z.rs:18-235"""

    def add_hi_z_to_graph(self, graph):
        graph.add_node("HiZ", lambda r, ev: r._check(r.lib.r3n_hi_z(r.ctx), "r3n_hi_z"))


class PbrRoutine:
    """rend3-routine/src/pbr/routine.rs:17-133"""

    def __init__(self):
        self.opaque_depth = ForwardRoutine(_ffi.PASS_DEPTH, OPAQUE)
        self.cutout_depth = ForwardRoutine(_ffi.PASS_DEPTH, CUTOUT)
        self.opaque_routine = ForwardRoutine(_ffi.PASS_FORWARD, OPAQUE)
        self.cutout_routine = ForwardRoutine(_ffi.PASS_FORWARD, CUTOUT)
        self.blend_routine = ForwardRoutine(_ffi.PASS_FORWARD, BLEND)
        self.hi_z = HiZRoutine()


class TonemappingRoutine:
    """rend3-routine/src/tonemapping.rs:29-148"""

    def add_to_graph(self, graph):
        def body(r, ev):
            r._check(r.lib.r3n_tonemap(r.ctx, None, 0), "r3n_tonemap")

        graph.add_node("Tonemapping", body)


class BaseRenderGraphRoutines:
    def __init__(self, pbr, tonemapping, skybox=None):
        self.pbr, self.tonemapping, self.skybox = pbr, tonemapping, skybox


class BaseRenderGraph:
    """rend3-routine/src/base.rs:103-186: owns the culler; add_to_graph declares the frame's nodes in order."""

    def __init__(self, renderer):
        self.renderer = renderer
        self.gpu_culler = GpuCuller()
        # measured on the bench scene: handing the GPU the viewport pass 1 before the shadow views costs 7 % (1.29 vs
        # 1.20 ms/frame): the shadow views are the longer chain and want the early start
        self.viewport_first = False

    def default_routines(self):
        return BaseRenderGraphRoutines(PbrRoutine(), TonemappingRoutine())

    def add_to_graph(self, graph, inputs, settings, exchange=None):
        """Node order == base.rs:135-185.  `exchange` (multi-GPU only, not in the reference) is called with
        ("shadow" | "pass1" | "pass2", renderer) at the points where ranks must merge their depth keys."""
        ev = inputs.eval_output
        w, h = inputs.resolution
        pbr = inputs.routines.pbr
        VP = CameraSpecifier.VIEWPORT

        # clear_shadow_buffers + create_frame_uniforms (base.rs:139,142)
        def begin(r, _ev):
            fu = host.frame_uniforms(r.camera, settings.ambient_color, (w, h))
            clear = np.asarray(settings.clear_color, dtype=f32)
            aw, ah = ev.shadow_target_size
            r._check(r.lib.r3n_frame_begin(r.ctx, _ffi.ptr(fu), w, h, inputs.samples, _ffi.ptr(clear), aw, ah),
                     "r3n_frame_begin")
            for si, sh in enumerate(ev.shadows):
                r._check(r.lib.r3n_shadow_viewport(r.ctx, si, sh["offset"][0], sh["offset"][1], sh["size"]),
                         "r3n_shadow_viewport")

        graph.add_node("Frame Uniforms", begin)
        # skinning (base.rs:145, skinning.rs:211-226)
        def skin(r, _ev):
            if r.skeletons:
                sk_in, sk_m = r.skinning_buffers()
                poses = r._pose_requests()
                if len(poses):
                    r._check(r.lib.r3n_pose_skeletons(r.ctx, _ffi.ptr(np.ascontiguousarray(poses)), len(poses)), "r3n_pose_skeletons")
                r._check(r.lib.r3n_skinning(r.ctx, _ffi.ptr(sk_in), len(sk_in), _ffi.ptr(sk_m), len(sk_m)), "r3n_skinning")

        graph.add_node("Skinning", skin)
        def shadow_nodes():
            # multi-GPU: shadow views are sharded by view -- a rank renders the views it owns (whole) and receives the others
            sharded = exchange is not None and hasattr(exchange, "owns_shadow_view")
            mine = [si for si in range(len(ev.shadows)) if not sharded or exchange.owns_shadow_view(si)]
            if sharded:
                # an owned view is drawn WHOLE here (every object slot), whatever the viewport's object range is; set every frame
                # next to the ownership test so the two cannot disagree (r3n_render_frame does the same from its view mask)
                def ranges(r, _ev):
                    for si in range(len(ev.shadows)):
                        r.set_camera_object_range(si, *((0, 0xFFFFFFFE) if si in mine else (0xFFFFFFFF, 0xFFFFFFFF)))
                graph.add_node("shadow view ownership", ranges)
            # shadow_object_uniform_upload (base.rs:148)
            for si in mine:
                sh = ev.shadows[si]
                self.gpu_culler.add_object_uniform_upload_to_graph(graph, si, (sh["size"], sh["size"]), 1, f"Shadow Culling S{si}")
            # pbr_shadow_culling (base.rs:150)
            for si in mine:
                self.gpu_culler.add_culling_to_graph(graph, si, f"Shadow Culling S{si}")
            # pbr_shadow_rendering (base.rs:153,366-396)
            for si in mine:
                for routine in (pbr.opaque_depth, pbr.cutout_depth):
                    routine.add_forward_to_graph(graph, f"pbr shadow renderering S{si}", si, _ffi.SOURCE_RESIDUAL)
            if exchange is not None and len(ev.shadows):
                graph.add_node("exchange shadow atlas", lambda r, _ev: exchange("shadow", r, ev=ev, samples=inputs.samples))

        def viewport_pass1_nodes():
            # object_uniform_upload (base.rs:156)
            self.gpu_culler.add_object_uniform_upload_to_graph(graph, VP, (w, h), inputs.samples, "Uniform Bake")
            # pbr_render_opaque_predicted_triangles (base.rs:159)
            for routine in (pbr.opaque_routine, pbr.cutout_routine):
                routine.add_forward_to_graph(graph, "PBR Forward Pass 1", VP, _ffi.SOURCE_PREDICTED)

        # The two groups are independent (the shadow atlas is first read by the resolve), so their relative order
        # only decides what the GPU is handed first.  Reference order: shadows, then the viewport (base.rs:148-159).
        if self.viewport_first:
            viewport_pass1_nodes()
            shadow_nodes()
        else:
            shadow_nodes()
            viewport_pass1_nodes()
        if exchange is not None:
            graph.add_node("exchange pass-1 depth", lambda r, _ev: exchange("pass1", r, ev=ev, samples=inputs.samples))
        # hi_z (base.rs:162)
        pbr.hi_z.add_hi_z_to_graph(graph)
        # pbr_culling (base.rs:169)
        self.gpu_culler.add_culling_to_graph(graph, VP, "Primary Culling")
        # pbr_render_opaque_residual_triangles (base.rs:172)
        for routine in (pbr.opaque_routine, pbr.cutout_routine):
            routine.add_forward_to_graph(graph, "PBR Forward Pass 2", VP, _ffi.SOURCE_RESIDUAL)
        if exchange is not None:
            graph.add_node("exchange pass-2 keys", lambda r, _ev: exchange("pass2", r, ev=ev, samples=inputs.samples))
        # the deferred evaluation of the opaque passes' fragments (this design's stand-in for their fragment shaders)
        graph.add_node("Resolve Opaque", lambda r, _ev: r._check(r.lib.r3n_resolve_opaque(r.ctx), "r3n_resolve_opaque"))
        # skybox (base.rs:175): out of scope.  pbr_forward_rendering_transparent (base.rs:181)
        pbr.blend_routine.add_forward_to_graph(graph, "PBR Forward Transparent", VP, _ffi.SOURCE_RESIDUAL)
        # tonemapping (base.rs:184)
        inputs.routines.tonemapping.add_to_graph(graph)
        graph.add_node("Frame End", lambda r, _ev: r._check(r.lib.r3n_frame_end(r.ctx), "r3n_frame_end"))
