"""
oracle/world.py -- oracle-side world bookkeeping + the frame schedule of BaseRenderGraph::add_to_graph
(rend3-routine/src/base.rs:129-185), driving the C oracle (oracle/r3o.c).

TEST INFRASTRUCTURE ONLY.

Mirrors, at the level the hot path needs:
  Renderer::{add_mesh, add_material, add_object, remove_object, set_object_transform,
             add_directional_light, add_point_light, set_camera_data}   rend3/src/renderer/mod.rs:126-424
  ObjectManager (128-byte records, deferred removal)                    rend3/src/managers/object.rs:236-364
  MeshManager (SoA attribute runs + indices in one u32 buffer)          rend3/src/managers/mesh.rs:123-184
  temporal two-pass culling state                                       SURVEY.md App. B.4
"""
import time

import numpy as np

from . import host
from .lib import get as get_lib

f32 = np.float32
INVALID = 0xFFFFFFFF

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

OPAQUE, CUTOUT, BLEND = 0, 1, 2  # TransparencyType as u64 key, pbr/material.rs:383-392,497-499



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
    """ShaderMaterial::from_material (pbr/material.rs:548-583) behind the 48-byte texture-id prefix
    (managers/material.rs:25-29).  albedo_mode: "none" | "vertex" | "value" | "value_vertex"
    (AlbedoComponent, pbr/material.rs:60-140; Default = None -> flags 0, value (0,0,0,1)), or with
    `albedo_texture` (a texture handle) "texture" | "texture_vertex" | "texture_value" | "texture_vertex_value".
    nearest: SampleType::Nearest (FLAGS_NEAREST); uv_transform0: 3x3 row-major nested list (Mat3, column-major on
    the GPU as three padded vec4 columns)."""
    rec = np.zeros(52, dtype=f32)
    ru = rec.view(np.uint32)
    # uv transforms = identity mat3 as 3 vec4 columns
    for base in (12, 24):
        rec[base + 0] = rec[base + 5] = rec[base + 10] = 1.0
    flags = 0
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
        assert albedo_texture is not None
        ru[0] = int(albedo_texture) + 1  # NonZeroU32 index into the bindless array (managers/material.rs:25-29)
    if nearest:
        flags |= FLAGS_NEAREST
    if uv_transform0 is not None:
        m = np.asarray(uv_transform0, dtype=f32).reshape(3, 3)
        for col in range(3):
            rec[12 + 4 * col: 12 + 4 * col + 3] = m[:, col]
    # NormalTexture::None -> 0 ; AoMRTextures::None -> AOMR_SPLIT ; ClearcoatTextures::None -> CC_GLTF_COMBINED
    flags |= material_texture_slots(ru, normal_texture, normal_mode, normal_y_down, aomr, reflectance_texture,
                                    clearcoat_textures, emissive_texture, anisotropy_texture)
    if unlit:
        flags |= FLAGS_UNLIT
    rec[36:40] = alb
    rec[40:43] = emissive
    rec[43] = roughness
    rec[44] = metallic
    rec[45] = reflectance
    rec[46] = clear_coat
    rec[47] = clear_coat_roughness
    rec[48] = 0.0  # anisotropy
    rec[49] = ao
    rec[50] = 0.0 if cutout is None else cutout
    ru[51] = flags
    return rec


class _Mesh:
    __slots__ = ("attr_off", "first_index", "index_count", "centre", "radius", "vertex_count", "joint_off", "weight_off")


class OracleRenderer:
    def __init__(self, handedness=host.LEFT, aspect_ratio=None, lib=None):
        self.lib = get_lib() if lib is None else lib  # (lib: oracle.lib.OracleLib(tally=True) -- the op-counting build, stage_tally below)
        self.handedness = handedness
        self.aspect_ratio = aspect_ratio
        self.mesh_words = np.zeros(0, dtype=np.uint32)
        self.meshes = []
        self.materials = []  # (record f32[52], key)
        self.capacity = 16  # FreelistDerivedBuffer::STARTING_SIZE (util/freelist/buffer.rs:19)
        self.objects = np.zeros((self.capacity, 32), dtype=np.uint32)
        self.object_meta = {}  # handle -> dict(mesh, mesh sphere)
        self.free_handles = []
        self.pending_free = []
        self.skip_shadow_draw = False
        self.tex_descs = np.zeros((0, 8), dtype=np.uint32)  # r3o_texture_desc rows
        self.tex_pool = np.zeros(1, dtype=np.uint32)
        self.tex_used = 0
        self.deferred_removals = []
        self.next_handle = 0
        self.dir_lights = []
        self.point_lights = []
        self.camera = host.CameraState(host.identity(), ("raw", host.identity()), handedness, aspect_ratio)
        self.cam_state = {}  # camera specifier -> temporal state
        self.frame_index = 0
        self.object_range = None
        self.shadow_views_owned = None  # multi-rank: set of shadow views this rank renders (whole); None = all, by object range
        self.row_band = None            # multi-rank, sort-first: (row_begin, row_end) -- the viewport's passes touch these rows only
        self.skeletons = []  # dict(mesh, out_off[3], matrices)

    # ------------------------------------------------------------------ skeletons (rend3/src/managers/skeleton.rs:67-163)
    def add_skeleton(self, mesh, joint_matrices):
        m = self.meshes[mesh]
        assert m.joint_off != INVALID, "Mesh must have joint indices to be used in a skeleton"
        out_off = [INVALID] * 3
        for a in range(3):  # position, normal, tangent copies private to the skeleton (skeleton.rs:110-113)
            if m.attr_off[a] != INVALID:
                out_off[a] = 4 * len(self.mesh_words)
                self._mesh_append([np.zeros(3 * m.vertex_count, dtype=np.uint32)])
        self.skeletons.append(dict(mesh=mesh, out_off=out_off, matrices=np.ascontiguousarray(joint_matrices, dtype=f32).reshape(-1, 16)))
        return len(self.skeletons) - 1

    def set_skeleton_joint_matrices(self, sk, joint_matrices):
        self.skeletons[sk]["matrices"] = np.ascontiguousarray(joint_matrices, dtype=f32).reshape(-1, 16)

    def skinning_buffers(self):
        """build_gpu_skinning_input_buffers, rend3-routine/src/skinning.rs:54-139"""
        inputs = np.zeros((len(self.skeletons), 10), dtype=np.uint32)
        mats, base = [], 0
        for i, sk in enumerate(self.skeletons):
            m = self.meshes[sk["mesh"]]
            inputs[i] = [m.attr_off[0], m.attr_off[1], m.attr_off[2], m.joint_off, m.weight_off, sk["out_off"][0],
                         sk["out_off"][1], sk["out_off"][2], base, m.vertex_count]
            mats.append(sk["matrices"])
            base += len(sk["matrices"])
        return inputs, (np.ascontiguousarray(np.concatenate(mats)) if mats else np.zeros((0, 16), dtype=f32))

    # ------------------------------------------------------------------ world edits
    def set_output_format(self, fmt):
        self.output_format = int(fmt)

    def add_texture_2d(self, rgba8, srgb=True, mip_count=1, mip_source="uploaded"):
        """Renderer::add_texture_2d (rend3/src/renderer/mod.rs) with Texture{data, format, size, mip_count, mip_source}:
        rgba8 = (H, W, 4) u8 (mip 0, or every mip concatenated row-major when mip_source == "uploaded" and
        mip_count > 1); format Rgba8UnormSrgb | Rgba8Unorm; mip_count: int or "maximum" (MipmapCount::Maximum);
        mip_source "uploaded" | "generated" (MipmapSource).  Returns the texture handle (index)."""
        host_mod = host
        data, w, h, mips = host_mod.prepare_texture(self.lib, rgba8, srgb, mip_count, mip_source)
        return self._append_texels(data, w, h, mips, srgb)

    def _append_texels(self, data, w, h, mips, srgb, pool_float=False):
        """data: pool words -- one per RGBA8 texel, or (pool_float) four f32 bit patterns per texel (r3o.c tex_fetch)."""
        self.tex_used = (self.tex_used + 3) & ~3 if pool_float else self.tex_used
        desc = np.array([[self.tex_used, w, h, mips, 2 if pool_float else (1 if srgb else 0), 0, 0, 0]], dtype=np.uint32)
        if self.tex_used + len(data) > len(self.tex_pool):
            grown = np.zeros(max(2 * len(self.tex_pool), self.tex_used + len(data)), dtype=np.uint32)
            grown[: self.tex_used] = self.tex_pool[: self.tex_used]
            self.tex_pool = grown
        self.tex_pool[self.tex_used: self.tex_used + len(data)] = data
        self.tex_used += len(data)
        self.tex_descs = np.concatenate([self.tex_descs, desc])
        return len(self.tex_descs) - 1

    def add_texture_2d_encoded(self, fmt, width, height, levels, generate_mips=False):
        """Mirror of the product's add_texture_2d_encoded: every level is decoded to RGBA8 by the oracle's decoders
        (oracle/bcn.c); generate_mips expands level 0 and runs the RGBA8 blit chain."""
        c = self.lib.c
        if c.r3o_texture_is_float(fmt) and generate_mips:
            assert len(levels) == 1 and c.r3o_texture_generates_mips_f32(fmt), "chains are generated for R16Float / Rg16Float / Rgba16Float / Rgb10a2Unorm"
            mips = int(max(width, height)).bit_length()
            chain = np.zeros((host.mip_chain_texels(width, height, mips), 4), dtype=np.float32)
            src = np.frombuffer(levels[0], dtype=np.uint8)
            assert len(src) == c.r3o_texture_level_bytes(fmt, width, height), "level byte count"
            assert c.r3o_texture_decode_level_f32(fmt, width, height, src.ctypes.data, chain.ctypes.data) == 0
            assert c.r3o_generate_mips_f32(fmt, width, height, mips, chain.ctypes.data) == 0
            return self._append_texels(chain.reshape(-1).view(np.uint32), width, height, mips, False, pool_float=True)
        if c.r3o_texture_is_float(fmt):
            words = []
            for k, lv in enumerate(levels):
                w, h = max(1, width >> k), max(1, height >> k)
                src = np.frombuffer(lv, dtype=np.uint8)
                assert len(src) == c.r3o_texture_level_bytes(fmt, w, h), "level byte count"
                out = np.zeros((h, w, 4), dtype=np.float32)
                assert c.r3o_texture_decode_level_f32(fmt, w, h, src.ctypes.data, out.ctypes.data) == 0
                words.append(out.reshape(-1).view(np.uint32))
            return self._append_texels(np.concatenate(words), width, height, len(levels), False, pool_float=True)
        decoded = []
        for k, lv in enumerate(levels):
            w, h = max(1, width >> k), max(1, height >> k)
            src = np.frombuffer(lv, dtype=np.uint8)
            assert len(src) == c.r3o_texture_level_bytes(fmt, w, h), "level byte count"
            out = np.zeros((h, w, 4), dtype=np.uint8)
            assert c.r3o_texture_decode_level(fmt, w, h, src.ctypes.data, out.ctypes.data) == 0
            decoded.append(out)
        srgb = fmt in (1, 5, 7, 9, 11, 15)
        if generate_mips:
            assert len(levels) == 1 and fmt < 6
            return self.add_texture_2d(decoded[0], srgb=srgb, mip_count="maximum", mip_source="generated")
        data = np.concatenate([d.reshape(-1, 4).view(np.uint32).reshape(-1) for d in decoded])
        return self._append_texels(data, width, height, len(levels), srgb)

    def _tex_args(self):
        lib = self.lib
        return (lib.ptr(np.ascontiguousarray(self.tex_descs)) if len(self.tex_descs) else None, len(self.tex_descs),
                lib.ptr(self.tex_pool))

    def _mesh_append(self, chunks):
        """mesh_words is a view of a buffer that grows geometrically (thousands of meshes: no quadratic re-copy)."""
        used = len(self.mesh_words)
        need = used + sum(len(c) for c in chunks)
        buf = getattr(self, "_mesh_buf", None)
        if buf is None or need > len(buf):
            grown = np.zeros(max(need, 2 * (len(buf) if buf is not None else 0), 1024), dtype=np.uint32)
            grown[:used] = self.mesh_words
            self._mesh_buf = buf = grown
        for c in chunks:
            buf[used:used + len(c)] = c
            used += len(c)
        self.mesh_words = buf[:used]

    def add_mesh(self, positions, indices=None, normals=None, colors=None, mesh_handedness=host.LEFT, tangents=None,
                 joint_indices=None, joint_weights=None, uv0=None):
        positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
        if indices is None:
            indices = np.arange(len(positions), dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if normals is None:  # MeshBuilder::build, rend3-types/src/lib.rs:501-504
            normals = host.calculate_normals(positions, indices, mesh_handedness == host.LEFT)
        normals = np.ascontiguousarray(normals, dtype=f32).reshape(-1, 3)
        m = _Mesh()
        m.attr_off = [INVALID] * 6
        chunks = []
        cursor = len(self.mesh_words)

        def push(words):
            nonlocal cursor
            start = cursor
            chunks.append(words)
            cursor += len(words)
            return start

        m.attr_off[0] = 4 * push(positions.view(np.uint32).reshape(-1))
        m.attr_off[1] = 4 * push(normals.view(np.uint32).reshape(-1))
        if tangents is not None:
            m.attr_off[2] = 4 * push(np.ascontiguousarray(tangents, dtype=f32).reshape(-1).view(np.uint32))
        if uv0 is not None:  # VERTEX_ATTRIBUTE_TEXTURE_COORDINATES_0: vec2<f32> (rend3-types/src/attribute.rs)
            m.attr_off[3] = 4 * push(np.ascontiguousarray(uv0, dtype=f32).reshape(-1, 2).reshape(-1).view(np.uint32))
        if colors is not None:
            colors = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 4)
            m.attr_off[5] = 4 * push(colors.view(np.uint32).reshape(-1))
        m.vertex_count = len(positions)
        m.joint_off = m.weight_off = INVALID
        if joint_indices is not None:  # [u16; 4] per vertex, 8 bytes (rend3-types/src/attribute.rs:97-135)
            m.joint_off = 4 * push(np.ascontiguousarray(joint_indices, dtype=np.uint16).reshape(-1, 4).view(np.uint32).reshape(-1))
            m.weight_off = 4 * push(np.ascontiguousarray(joint_weights, dtype=f32).reshape(-1, 4).view(np.uint32).reshape(-1))
        m.first_index = push(indices)
        m.index_count = len(indices)
        self._mesh_append(chunks)
        m.centre, m.radius = host.bounding_sphere_from_mesh(positions)
        self.meshes.append(m)
        return len(self.meshes) - 1

    def add_material(self, record, key=OPAQUE):
        self.materials.append((np.asarray(record, dtype=f32), key))
        return len(self.materials) - 1

    def update_material(self, handle, record, key=None):
        """Renderer::update_material (rend3/src/renderer/mod.rs): replaces the material's data; the transparency key of a
        material cannot change through an update in the reference (archetype), so `key` defaults to the existing one."""
        self.materials[handle] = (np.asarray(record, dtype=f32), self.materials[handle][1] if key is None else key)

    def update_directional_light(self, handle, **changes):
        """Renderer::update_directional_light with a DirectionalLightChange (only the given fields change)."""
        self.dir_lights[handle].update(changes)

    def update_point_light(self, handle, **changes):
        self.point_lights[handle].update(changes)

    def _alloc_handle(self):
        if self.free_handles:
            return self.free_handles.pop(0)
        h = self.next_handle
        self.next_handle += 1
        return h

    def _use_index(self, idx):
        cap = self.capacity
        if idx > cap:  # freelist/buffer.rs:48-52 (sic: strictly greater)
            cap = 1 << (int(idx) - 1).bit_length()
        while cap <= idx:  # never index out of bounds (the reference would)
            cap *= 2
        if cap != self.capacity:
            grown = np.zeros((cap, 32), dtype=np.uint32)
            grown[: self.capacity] = self.objects
            self.objects = grown
            self.capacity = cap

    def _write_object(self, h):
        meta = self.object_meta[h]
        mesh = self.meshes[meta["mesh"]]
        rec = np.zeros(32, dtype=np.uint32)
        rf = rec.view(f32)
        rf[0:16] = meta["transform"]
        c, r = host.bounding_sphere_apply_transform(mesh.centre, mesh.radius, meta["transform"])
        rf[16:19] = c
        rf[19] = r
        rec[20] = mesh.first_index
        rec[21] = mesh.index_count
        rec[22] = meta["material"]
        rec[23:29] = mesh.attr_off
        if meta.get("skeleton") is not None:  # object.rs:250-258: skeleton ranges override the mesh's
            for a, off in enumerate(self.skeletons[meta["skeleton"]]["out_off"]):
                if off != INVALID:
                    rec[23 + a] = off
        rec[29] = 1 if meta["enabled"] else 0
        self._use_index(h)
        self.objects[h] = rec

    def add_object(self, mesh, material, transform, skeleton=None):
        h = self._alloc_handle()
        if skeleton is not None:
            mesh = self.skeletons[skeleton]["mesh"]
        self.object_meta[h] = dict(mesh=mesh, material=material, transform=np.asarray(transform, dtype=f32).copy(),
                                   enabled=True, skeleton=skeleton)
        self._write_object(h)
        # object.rs:273: a new object's sorting location is its transformed bounding-sphere centre
        self.object_meta[h]["location"] = self.objects[h].view(f32)[16:19].copy()
        return h

    def add_objects_bulk(self, mesh_ids, material_ids, transforms):
        transforms = np.ascontiguousarray(transforms, dtype=f32).reshape(len(mesh_ids), 16)
        return [self.add_object(int(m), int(k), t) for m, k, t in zip(mesh_ids, material_ids, transforms)]

    def set_object_transform(self, h, transform):
        self.object_meta[h]["transform"] = np.asarray(transform, dtype=f32).copy()
        self._write_object(h)
        # object.rs:313: ... and after a transform update it is the translation (transform_point3a(ZERO))
        self.object_meta[h]["location"] = self.object_meta[h]["transform"][12:15].copy()

    def remove_object(self, h):
        # object.rs:330-342: only disabled now, really removed at the next evaluate
        self.object_meta[h]["enabled"] = False
        self._write_object(h)
        self.deferred_removals.append(h)

    def add_directional_light(self, color=(1, 1, 1), intensity=1.0, direction=(0, -1, 0), distance=100.0,
                              resolution=2048):
        self.dir_lights.append(dict(color=color, intensity=intensity, direction=direction, distance=distance,
                                    resolution=resolution))
        return len(self.dir_lights) - 1

    def add_point_light(self, position, color=(1, 1, 1), intensity=1.0, radius=1.0):
        self.point_lights.append(dict(position=position, color=color, intensity=intensity, radius=radius))
        return len(self.point_lights) - 1

    def set_camera_data(self, view, projection):
        self.camera = host.CameraState(view, projection, self.handedness, self.aspect_ratio)

    # ------------------------------------------------------------------ boundary buffers
    def material_buffers(self):
        n = max(1, len(self.materials))
        recs = np.zeros((n, 52), dtype=f32)
        keys = np.zeros(n, dtype=np.uint8)
        for i, (r, k) in enumerate(self.materials):
            recs[i] = r
            keys[i] = k
        return recs, keys

    def tri_base(self):
        counts = (self.objects[:, 21] // 3) * (self.objects[:, 29] != 0)
        base = np.zeros(self.capacity, dtype=np.uint32)
        base[1:] = np.cumsum(counts[:-1], dtype=np.uint64).astype(np.uint32)
        return base, int(counts.sum())

    # ------------------------------------------------------------------ per-camera cull
    def _cull(self, spec, hdr, baked, hiz, hiz_w, hiz_h):
        lib = self.lib
        cap = self.capacity
        visible = np.zeros(cap, dtype=np.uint8)
        lib.r3o_frustum_cull(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(visible))
        whole_view = self.shadow_views_owned is not None and isinstance(spec, tuple)  # an owned shadow view draws every object
        if getattr(self, "object_owners", None) is not None and not whole_view:  # multi-rank sharding by owner byte (spatial partition)
            owners, my_rank = self.object_owners
            _mats, mat_keys = self.material_buffers()
            is_blend = mat_keys[np.minimum(self.objects[:, 22], len(mat_keys) - 1)] == BLEND
            outside = np.ones(cap, dtype=bool)
            outside[: len(owners)] = np.asarray(owners[:cap]) != my_rank
            visible[outside & ~is_blend] = 0
        elif self.object_range is not None and not whole_view:  # multi-rank sharding: this rank owns object slots [begin, end)
            b, e = self.object_range
            # ... of the opaque / cutout objects; blend objects are culled and drawn by every rank (ordered blending
            # cannot be merged by a MAX reduce)
            _mats, mat_keys = self.material_buffers()
            is_blend = mat_keys[np.minimum(self.objects[:, 22], len(mat_keys) - 1)] == BLEND
            outside = np.ones(cap, dtype=bool)
            outside[b:e] = False
            visible[outside & ~is_blend] = 0
        tri_base, total = self.tri_base()
        pass_bits = np.zeros(max(total, 1), dtype=np.uint8)
        residual = np.zeros(max(total, 1), dtype=np.uint8)
        st = self.cam_state.get(spec)
        prev_base = prev_pass = None
        if st is not None:
            prev_base = np.full(cap, INVALID, dtype=np.uint32)
            n = min(cap, len(st["tri_base_or_invalid"]))
            prev_base[:n] = st["tri_base_or_invalid"][:n]
            prev_pass = st["pass"]
        lib.r3o_cull_triangles(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(self.mesh_words), lib.ptr(baked),
                               lib.ptr(visible), lib.ptr(tri_base), lib.ptr(hiz), hiz_w, hiz_h,
                               lib.ptr(prev_base), lib.ptr(prev_pass), lib.ptr(pass_bits), lib.ptr(residual))
        # batching.rs:226,230: only objects that were batched (frustum-visible) carry history
        base_or_invalid = np.where(visible != 0, tri_base, np.uint32(INVALID)).astype(np.uint32)
        self.cam_state[spec] = dict(tri_base_or_invalid=base_or_invalid, **{"pass": pass_bits})
        return visible, tri_base, pass_bits, residual

    def _list_from_bits(self, bits, tri_base):
        slots = np.flatnonzero(bits).astype(np.uint32)
        obj = (np.searchsorted(tri_base, slots, side="right") - 1).astype(np.uint32)
        # skip empty objects sharing a base: searchsorted(right)-1 lands on the last object with base<=slot
        tri = (slots - tri_base[obj]).astype(np.uint32)
        return np.ascontiguousarray(obj), np.ascontiguousarray(tri)

    # ------------------------------------------------------------------ frame
    def render(self, width, height, samples=1, ambient=(0, 0, 0, 0), clear_color=(0, 0, 0, 0), exchange=None):
        """exchange(what, ndarray) -- multi-rank only: element-wise MAX all-reduce of the shadow atlas ("shadow") and of
        the visibility keys ("pass1", "pass2") at the points DESIGN.md section 6 names."""
        assert samples in (1, 4), "SampleCount::One | SampleCount::Four (rend3-types/src/lib.rs SampleCount)"
        lib = self.lib
        # wall-clock per stage of this frame (BASELINE.md section 3 split), read by bench.py's cpu_baseline leg
        stage_s = self.stage_s = {k: 0.0 for k in ("bake", "cull", "hiz", "shadow_depth", "forward_raster", "shade", "tonemap")}
        tallying = bool(getattr(lib, "tally", False))
        stage_tally = self.stage_tally = {}

        class _Span:
            def __init__(self, name):
                self.name = name

            def __enter__(self):
                self.t0 = time.perf_counter()
                if tallying:
                    lib.tally_reset()

            def __exit__(self, *exc):
                stage_s[self.name] += time.perf_counter() - self.t0
                if tallying:  # the op-counting build (oracle/tally.h): f32 operations per stage of this frame
                    def add(dst, src):
                        for k, v in src.items():
                            if isinstance(v, dict):
                                add(dst.setdefault(k, {}), v)
                            else:
                                dst[k] = dst.get(k, 0) + v
                    add(stage_tally.setdefault(self.name, {}), lib.tally_read())
        # Renderer::evaluate_instructions (renderer/eval.rs): last frame's removals become real
        for h in self.pending_free:
            self.objects[h] = 0
            self.object_meta.pop(h, None)
            self.free_handles.append(h)
        self.pending_free = self.deferred_removals
        self.deferred_removals = []

        cap = self.capacity
        mats, mat_keys = self.material_buffers()
        cam = self.camera
        atlas_size, shadows, dir_buf = host.evaluate_directional_lights(self.dir_lights, cam)
        point_buf = host.point_light_buffer(self.point_lights)
        out = {"shadows": []}

        # 1. clear shadow atlas (clear.rs:4-20)
        atlas = np.zeros((atlas_size[1], atlas_size[0]), dtype=f32)
        # 2. frame uniforms (uniforms.rs)
        fu = host.frame_uniforms(cam, ambient, (width, height), lib)
        # 3. skinning (base.rs:145, skinning.rs:211-226)
        if self.skeletons:
            sk_in, sk_m = self.skinning_buffers()
            lib.r3o_skinning(lib.ptr(self.mesh_words), lib.ptr(sk_in), len(sk_in), lib.ptr(sk_m))
        # 4-6. shadow views: bake, cull, depth draw (base.rs:148-153)
        for si, sh in enumerate(shadows):
            if self.shadow_views_owned is not None and si not in self.shadow_views_owned:
                out["shadows"].append(None)  # rendered by its owner rank; the atlas rectangle arrives through the exchange
                continue
            hdr = host.camera_header(sh["camera"], si, (sh["size"], sh["size"]), 1, cap, lib)
            baked = np.zeros((cap, 32), dtype=f32)
            with _Span("bake"):
                lib.r3o_uniform_bake(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(baked))
            with _Span("cull"):
                visible, tri_base, pass_bits, _ = self._cull(("shadow", si), hdr, baked, None, 0, 0)
                lo, lt = self._list_from_bits(pass_bits, tri_base)
            if self.skip_shadow_draw:  # test probe only: leaves the atlas at its 0.0 clear, so every compare passes
                lo, lt = lo[:0], lt[:0]
            with _Span("shadow_depth"):
                lib.r3o_raster_depth(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(self.mesh_words), lib.ptr(baked),
                                     lib.ptr(mats), lib.ptr(mat_keys), lib.ptr(lo), lib.ptr(lt), len(lo),
                                     lib.ptr(atlas), atlas_size[0], sh["offset"][0], sh["offset"][1], sh["size"], *self._tex_args())
            out["shadows"].append(dict(header=hdr, visible=visible, tri_base=tri_base, **{"pass": pass_bits}))
        if exchange is not None and shadows:
            exchange("shadow", atlas, shadows=shadows)

        # 7. viewport bake
        hdr = host.camera_header(cam, None, (width, height), samples, cap, lib)
        baked = np.zeros((cap, 32), dtype=f32)
        with _Span("bake"):
            lib.r3o_uniform_bake(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(baked))
        vis = np.zeros((height, width) if samples == 1 else (height, width, samples), dtype=np.uint64)
        tri_base_now, _ = self.tri_base()

        def draw(lo, lt, entry_keys=None):
            if len(lo):
                with _Span("forward_raster"):
                    lib.r3o_raster_visibility(lib.ptr(hdr), lib.ptr(self.objects), lib.ptr(self.mesh_words),
                                              lib.ptr(baked), lib.ptr(mats), lib.ptr(mat_keys), lib.ptr(tri_base_now),
                                              lib.ptr(lo), lib.ptr(lt), len(lo), width, height, samples, *self._tex_args(), lib.ptr(vis),
                                              None if entry_keys is None else lib.ptr(entry_keys))
            if self.row_band is not None:  # a rank sharded by rows rasterises its band only: elsewhere the keys stay clear
                vis[: self.row_band[0]] = 0
                vis[self.row_band[1]:] = 0

        # 8. pass 1: last frame's predicted triangles, in LAST frame's draw ranges (forward.rs:224-232: the cached DrawCallSet --
        # material_key_ranges as batch_objects made them then, forward.rs:286): an entry is drawn by the pipeline of the key its
        # material had at that cull (opaque: no discard, cutout: discard), with the material record as it is now
        predicted = self.cam_state.get("predicted_list")
        if predicted is not None:
            lo, lt, lk = predicted
            keep = lo < cap
            draw(np.ascontiguousarray(lo[keep]), np.ascontiguousarray(lt[keep]), np.ascontiguousarray(lk[keep]))
        if exchange is not None and samples != 1:
            exchange("pass1", vis)  # multisampled: the keys (min over samples and max over ranks do not commute)
        # 9. Hi-Z from pass-1 depth (hi_z.rs:161-234)
        nm = lib.r3o_hiz_mip_count(width, height)
        pyr = np.zeros(int(lib.r3o_hiz_mip_offset(width, height, nm)), dtype=f32)
        with _Span("hiz"):
            lib.r3o_vis_to_depth(lib.ptr(vis), width * height, samples, lib.ptr(pyr))
            if exchange is not None and samples == 1:
                exchange("pass1_depth", pyr[: width * height])  # only the depth plane has to be global for the Hi-Z cull
            lib.r3o_hiz_build(lib.ptr(pyr), width, height)
        out["depth_pass1"] = pyr[: width * height].reshape(height, width).copy()
        out["hiz"] = pyr
        # 10. cull (culler.rs:531-659)
        with _Span("cull"):
            visible, tri_base, pass_bits, residual = self._cull("viewport", hdr, baked, pyr, width, height)
            resid_list = self._list_from_bits(residual, tri_base)
            # next frame's predicted triangles: the passing triangles of the objects batched atomic-capable (cull.wgsl:361-363:
            # Sorting::OPAQUE, i.e. opaque + cutout keys; a blend object writes residual entries only, cull.wgsl:372-378), each with
            # the key of the region it was batched under (batching.rs:153,191-204)
            plo, plt = self._list_from_bits(pass_bits, tri_base)
            plk = mat_keys[np.minimum(self.objects[plo, 22], len(mat_keys) - 1)].astype(np.uint8)
            atomic = plk <= CUTOUT
            self.cam_state["predicted_list"] = (np.ascontiguousarray(plo[atomic]), np.ascontiguousarray(plt[atomic]), np.ascontiguousarray(plk[atomic]))
        assert np.array_equal(tri_base, tri_base_now)
        # 11. pass 2: residual triangles
        draw(*resid_list)
        if exchange is not None:
            exchange("pass2", vis)

        # opaque shading of the nearest fragment + 14. tonemap
        hdr16 = np.zeros((height, width, 4), dtype=np.uint16)
        n_dir = int(np.frombuffer(dir_buf[:4], dtype=np.uint32)[0])
        n_pt = int(np.frombuffer(point_buf[:4], dtype=np.uint32)[0])
        dir_arr = np.frombuffer(dir_buf, dtype=np.uint8)[16:].copy()
        pt_arr = np.frombuffer(point_buf, dtype=np.uint8)[16:].copy()
        clear = np.asarray(clear_color, dtype=f32)
        # transparent pass input (base.rs:181): this frame's passing triangles of the blend objects, back to front
        blend_objs = [h for h, m in sorted(self.object_meta.items())
                      if m["enabled"] and self.materials[m["material"]][1] == BLEND and visible[h]]
        order = host.blend_draw_order(cam.location, blend_objs, [self.object_meta[h]["location"] for h in blend_objs])
        bo, bt = [], []
        for h in order:
            nt = int(self.meshes[self.object_meta[h]["mesh"]].index_count // 3)
            tris = np.flatnonzero(pass_bits[int(tri_base[h]): int(tri_base[h]) + nt]).astype(np.uint32)
            bo.append(np.full(len(tris), h, dtype=np.uint32))
            bt.append(tris)
        blend_obj = np.ascontiguousarray(np.concatenate(bo)) if bo else np.zeros(0, dtype=np.uint32)
        blend_tri = np.ascontiguousarray(np.concatenate(bt)) if bt else np.zeros(0, dtype=np.uint32)
        out["blend_list"] = (blend_obj, blend_tri)
        with _Span("shade"):
            lib.r3o_shade(lib.ptr(vis), width, height, samples, lib.ptr(fu), lib.ptr(hdr), lib.ptr(self.objects),
                          lib.ptr(self.mesh_words), lib.ptr(baked), lib.ptr(mats), lib.ptr(tri_base), n_dir,
                          lib.ptr(dir_arr) if n_dir else None, n_pt, lib.ptr(pt_arr) if n_pt else None,
                          lib.ptr(atlas), atlas_size[0], atlas_size[1], lib.ptr(clear), *self._tex_args(),
                          lib.ptr(blend_obj) if len(blend_obj) else None, lib.ptr(blend_tri) if len(blend_tri) else None,
                          len(blend_obj), lib.ptr(hdr16))
        rgba_f = np.zeros((height, width, 4), dtype=f32)
        rgba8 = np.zeros((height, width, 4), dtype=np.uint8)
        with _Span("tonemap"):
            lib.r3o_tonemap_format(lib.ptr(hdr16), width * height, lib.ptr(rgba_f), lib.ptr(rgba8), getattr(self, "output_format", 0))

        out.update(header=hdr, frame_uniforms=fu, baked=baked, visible=visible, tri_base=tri_base,
                   residual=residual, vis=vis, atlas=atlas, atlas_size=atlas_size, hdr16=hdr16, rgba_f32=rgba_f,
                   rgba8=rgba8, dir_buf=dir_buf, point_buf=point_buf, objects=self.objects.copy(),
                   materials=mats, material_keys=mat_keys, mesh=self.mesh_words, shadow_descs=shadows,
                   capacity=cap, **{"pass": pass_bits})
        self.frame_index += 1
        return out
