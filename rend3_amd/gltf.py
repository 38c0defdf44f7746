"""glTF 2.0 / GLB reader and scene instancer (row N1 of SURVEY.md section 8f): container parsing, accessor decoding
into numpy arrays, node hierarchy, skins, materials with their textures (PNG / JPEG through PIL, KTX2 / DDS through
containers.py), KHR_lights_punctual directional lights and animations, fed into the Renderer-shaped API the way
rend3-gltf does (rend3-gltf/src/lib.rs: load_meshes :607-678, load_materials_and_textures :806-943, load_image
:984-1130, load_animations :724-773, instance_loaded_scene :493-562; examples/src/static_gltf/mod.rs:5-41).
Not read: cameras, morph targets; TEXCOORD_1 is decoded but not uploaded (no shader of the path reads it).
"""
import json
import os
import struct

import numpy as np

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class Gltf:
    def __init__(self, path):
        self.path = path
        data = open(path, "rb").read()
        self.buffers = []
        if data[:4] == b"glTF":
            version, length = struct.unpack_from("<II", data, 4)
            assert version == 2, "only glTF 2.0"
            off, chunks = 12, []
            while off < length:
                clen, ctype = struct.unpack_from("<II", data, off)
                chunks.append((ctype, data[off + 8: off + 8 + clen]))
                off += 8 + clen
            assert chunks and chunks[0][0] == 0x4E4F534A, "first GLB chunk must be JSON"
            self.json = json.loads(chunks[0][1].decode("utf-8"))
            glb_bin = next((c for t, c in chunks[1:] if t == 0x004E4942), None)
        else:
            self.json = json.loads(data.decode("utf-8"))
            glb_bin = None
        for b in self.json.get("buffers", []):
            if "uri" not in b:
                assert glb_bin is not None, "buffer without uri needs a GLB BIN chunk"
                self.buffers.append(glb_bin)
            elif b["uri"].startswith("data:"):
                import base64
                self.buffers.append(base64.b64decode(b["uri"].split(",", 1)[1]))
            else:
                self.buffers.append(open(os.path.join(os.path.dirname(path), b["uri"]), "rb").read())

    def _read(self, view_index, byte_offset, dt, width, count):
        bv = self.json["bufferViews"][view_index]
        buf = self.buffers[bv["buffer"]]
        base = bv.get("byteOffset", 0) + byte_offset
        elem = dt.itemsize * width
        stride = bv.get("byteStride", 0) or elem
        if stride == elem:
            return np.frombuffer(buf, dtype=dt, count=count * width, offset=base).reshape(count, width).copy()
        raw = np.frombuffer(buf, dtype=np.uint8, count=(count - 1) * stride + elem, offset=base)
        idx = (np.arange(count)[:, None] * stride + np.arange(elem)[None, :]).reshape(-1)
        return raw[idx].view(dt).reshape(count, width).copy()

    def accessor(self, index, raw=False):
        """Accessor data as (count, width): floats as stored; normalised integers converted to f32 (glTF 2.0 section
        3.6.2.4) unless `raw`; sparse accessors (section 3.6.2.5) applied over the base view (or zeros)."""
        a = self.json["accessors"][index]
        dt = np.dtype(_COMPONENT[a["componentType"]])
        width = _WIDTH[a["type"]]
        count = a["count"]
        if "bufferView" in a:
            arr = self._read(a["bufferView"], a.get("byteOffset", 0), dt, width, count)
        else:
            arr = np.zeros((count, width), dtype=dt)
        if "sparse" in a:
            sp = a["sparse"]
            idt = np.dtype(_COMPONENT[sp["indices"]["componentType"]])
            where = self._read(sp["indices"]["bufferView"], sp["indices"].get("byteOffset", 0), idt, 1, sp["count"]).reshape(-1)
            arr[where.astype(np.int64)] = self._read(sp["values"]["bufferView"], sp["values"].get("byteOffset", 0), dt, width, sp["count"])
        if a.get("normalized") and dt != np.float32 and not raw:
            info = np.iinfo(dt)
            arr = np.maximum(arr.astype(np.float32) / np.float32(info.max), -1.0 if info.min < 0 else 0.0)
        return arr

    def primitive(self, mesh=0, primitive=0):
        """Attribute arrays of one primitive, as the reference's loaders read them (u32 indices, f32 attributes)."""
        p = self.json["meshes"][mesh]["primitives"][primitive]
        assert p.get("mode", 4) == 4, "only triangle lists"
        at = p["attributes"]
        out = {"positions": self.accessor(at["POSITION"]).astype(np.float32)}
        if "NORMAL" in at:
            out["normals"] = self.accessor(at["NORMAL"]).astype(np.float32)
        if "TANGENT" in at:
            out["tangents"] = self.accessor(at["TANGENT"]).astype(np.float32)[:, :3]  # Vec4::truncate
        if "TEXCOORD_0" in at:
            out["uv0"] = self.accessor(at["TEXCOORD_0"]).astype(np.float32)
        if "TEXCOORD_1" in at:
            out["uv1"] = self.accessor(at["TEXCOORD_1"]).astype(np.float32)
        if "COLOR_0" in at:
            # read_colors(0).into_rgba_u8() (gltf crate, mesh/util/colors.rs: f32 -> (clamp(x, 0, 1) * 255) as u8, u16 -> x >> 8,
            # RGB gets alpha 255)
            c = self.accessor(at["COLOR_0"], raw=True)
            if c.dtype == np.float32:
                c8 = (np.clip(c, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
            elif c.dtype == np.uint16:
                c8 = (c >> 8).astype(np.uint8)
            else:
                c8 = c.astype(np.uint8)
            if c8.shape[1] == 3:
                c8 = np.concatenate([c8, np.full((len(c8), 1), 255, dtype=np.uint8)], axis=1)
            out["colors"] = c8
        if "JOINTS_0" in at:
            out["joints"] = self.accessor(at["JOINTS_0"]).astype(np.uint16)
            out["weights"] = self.accessor(at["WEIGHTS_0"]).astype(np.float32)
        if "indices" in p:
            out["indices"] = self.accessor(p["indices"]).astype(np.uint32).reshape(-1)
        else:
            out["indices"] = np.arange(len(out["positions"]), dtype=np.uint32)
        out["material"] = p.get("material")
        return out

    def base_color_factor(self, material):
        if material is None:
            return (1.0, 1.0, 1.0, 1.0)
        pbr = self.json["materials"][material].get("pbrMetallicRoughness", {})
        return tuple(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0]))


# ---------------------------------------------------------------------------------------------------- scene instancing
def _node_local_matrix(node, hm):
    """gltf::scene::Transform::matrix(): the node's `matrix`, or T * R * S from translation / rotation (xyzw) / scale."""
    if "matrix" in node:
        return np.asarray(node["matrix"], dtype=np.float32)
    t = node.get("translation", [0.0, 0.0, 0.0])
    q = np.asarray(node.get("rotation", [0.0, 0.0, 0.0, 1.0]), dtype=np.float32)
    s = node.get("scale", [1.0, 1.0, 1.0])
    x, y, z, w = (np.float32(v) for v in q)
    one, two = np.float32(1.0), np.float32(2.0)
    r = np.zeros(16, dtype=np.float32)
    r[0], r[1], r[2] = one - two * (y * y + z * z), two * (x * y + z * w), two * (x * z - y * w)
    r[4], r[5], r[6] = two * (x * y - z * w), one - two * (x * x + z * z), two * (y * z + x * w)
    r[8], r[9], r[10] = two * (x * z + y * w), two * (y * z - x * w), one - two * (x * x + y * y)
    r[15] = 1.0
    return hm.mat4_mul(hm.mat4_mul(hm.translation(t), r), hm.scale(s))


def _image_bytes(g, image):
    """gltf::image::Source: a buffer view, a data URI or a file next to the document (filesystem_io_func)."""
    if "bufferView" in image:
        bv = g.json["bufferViews"][image["bufferView"]]
        off = bv.get("byteOffset", 0)
        return bytes(g.buffers[bv["buffer"]][off: off + bv["byteLength"]])
    uri = image["uri"]
    if uri.startswith("data:"):
        import base64
        return base64.b64decode(uri.split(",", 1)[1])
    from urllib.parse import unquote
    return open(os.path.join(os.path.dirname(g.path), unquote(uri)), "rb").read()


def load_image(g, r, cache, image_index, srgb):
    """load_image_cached + load_image (rend3-gltf/src/lib.rs:951-1128): KTX2 first, then DDS (containers.py: the
    `ktx2` / `ddsfile` readers and the loader's format maps; stored levels are uploaded as they are, a single-level file
    gets a generated chain only when its format can be rendered to, i.e. is not block-compressed), else the `image`
    crate's decoders + util::convert_dynamic_image (:1157-1175): 8-bit luma (with or without alpha) becomes R8Unorm,
    RGB / RGBA become Rgba8Unorm[Srgb], MipmapCount::Maximum + MipmapSource::Generated.
    Returns (texture handle, components)."""
    key = (image_index, bool(srgb))
    if key not in cache:
        from . import containers
        data = _image_bytes(g, g.json["images"][image_index])
        parsed = containers.parse_ktx2(data, bool(srgb)) or containers.parse_dds(data, bool(srgb))
        if parsed is not None:
            fmt = parsed["format"]
            generate = len(parsed["levels"]) == 1 and containers.generate_mips_allowed(fmt)
            handle = r.add_texture_2d_encoded(fmt, parsed["width"], parsed["height"], parsed["levels"], generate_mips=generate)
            comps = containers.COMPONENTS.get(fmt, 4)  # TextureFormat::describe().components
            cache[key] = (handle, comps)
        else:
            import io
            from PIL import Image
            im = Image.open(io.BytesIO(data))
            if im.mode in ("L", "LA"):
                lum = np.ascontiguousarray(np.array(im.convert("L"), dtype=np.uint8))
                cache[key] = (r.add_texture_2d_encoded(containers.R8, lum.shape[1], lum.shape[0], [lum.tobytes()],
                                                       generate_mips=True), 1)
            else:
                rgba = np.array(im.convert("RGBA"), dtype=np.uint8)
                cache[key] = (r.add_texture_2d(rgba, srgb=bool(srgb), mip_count="maximum", mip_source="generated"), 4)
    return cache[key]


def material_from_gltf(g, index, mk, r=None, image_cache=None, normal_y_down=False):
    """load_materials_and_textures (rend3-gltf/src/lib.rs:806-943): albedo = TextureVertexValue / ValueVertex
    {base_color_factor, srgb: false}; sampler from the base colour texture's magFilter; KHR_texture_transform of the
    base colour texture as uv_transform0; normal texture Bicomponent (2 components, e.g. BC5) / Tricomponent (>= 3) with
    GltfLoadSettings::normal_direction (`normal_y_down`: NormalTextureYDirection::Down, the scene viewer's --normal-y-down);
    AO / metallic-roughness packing Combined (same image) |
    Split (AO with < 3 components) | SwizzledSplit; emissive TextureValue; alpha mode -> transparency;
    KHR_materials_unlit.  load_default_material (:777-800) when the primitive has no material.
    `r` (a renderer with add_texture_2d) is only needed when the material has textures.
    Returns (record, material key)."""
    if index is None:
        return mk(albedo=(1.0, 1.0, 1.0, 1.0), albedo_mode="value", roughness=1.0, metallic=1.0, ao=1.0, clear_coat=1.0,
                  clear_coat_roughness=1.0), 0
    m = g.json["materials"][index]
    pbr = m.get("pbrMetallicRoughness", {})
    cache = image_cache if image_cache is not None else {}

    def tex(info, srgb):
        if info is None:
            return None
        if r is None:
            raise ValueError("textured glTF material: pass the renderer")
        t = g.json["textures"][info["index"]]
        return load_image(g, r, cache, t["source"], srgb)

    albedo_info = pbr.get("baseColorTexture")
    albedo = tex(albedo_info, True)
    occlusion = tex(m.get("occlusionTexture"), False)
    emissive = tex(m.get("emissiveTexture"), True)
    normals = tex(m.get("normalTexture"), False)
    mr = tex(pbr.get("metallicRoughnessTexture"), False)

    nearest = False
    uv_transform = None
    if albedo_info is not None:
        t = g.json["textures"][albedo_info["index"]]
        if "sampler" in t:
            nearest = g.json["samplers"][t["sampler"]].get("magFilter") == 9728  # NEAREST
        tt = albedo_info.get("extensions", {}).get("KHR_texture_transform")
        if tt is not None:  # Mat3::from_scale_angle_translation(scale, rotation, offset)
            sx, sy = tt.get("scale", [1.0, 1.0])
            ox, oy = tt.get("offset", [0.0, 0.0])
            a = np.float32(tt.get("rotation", 0.0))
            c, sn = np.float32(np.cos(a)), np.float32(np.sin(a))
            uv_transform = [[np.float32(sx) * c, -np.float32(sy) * sn, ox], [np.float32(sx) * sn, np.float32(sy) * c, oy], [0.0, 0.0, 1.0]]
    if mr is not None and occlusion is not None and mr[0] == occlusion[0]:
        aomr = ("combined", mr[0])
    elif occlusion is not None and occlusion[1] < 3:
        aomr = ("split", occlusion[0], None if mr is None else mr[0])
    else:
        aomr = ("swizzled_split", None if occlusion is None else occlusion[0], None if mr is None else mr[0])
    mode = m.get("alphaMode", "OPAQUE")
    key = {"OPAQUE": 0, "MASK": 1, "BLEND": 2}[mode]
    rec = mk(albedo=tuple(pbr.get("baseColorFactor", [1.0, 1.0, 1.0, 1.0])),
             albedo_mode="value_vertex" if albedo is None else "texture_vertex_value", vertex_srgb=False,
             albedo_texture=None if albedo is None else albedo[0], nearest=nearest, uv_transform0=uv_transform,
             normal_texture=normals[0] if normals is not None and normals[1] >= 2 else None,
             normal_mode="bicomponent" if normals is not None and normals[1] == 2 else "tricomponent",
             normal_y_down=bool(normal_y_down),
             aomr=aomr, emissive_texture=None if emissive is None else emissive[0],
             roughness=pbr.get("roughnessFactor", 1.0), metallic=pbr.get("metallicFactor", 1.0),
             emissive=tuple(m.get("emissiveFactor", [0.0, 0.0, 0.0])),
             cutout=(m.get("alphaCutoff", 0.5) if mode == "MASK" else None),
             unlit="KHR_materials_unlit" in m.get("extensions", {}))
    return rec, key


def instance_scene(g, r, hm, mk, scale=1.0, enable_directional=True, directional_light_shadow_distance=100.0,
                   directional_light_resolution=2048, normal_y_down=False):
    """load_gltf + instance_loaded_scene (rend3-gltf/src/lib.rs:335-379, 493-562): node transforms in topological order
    under parent_transform = scale(s, s, -s for a left-handed renderer); one object per mesh primitive; a skeleton per
    primitive of a skinned node (joint matrices start as identity, add_mesh_by_index :411-457); winding flipped for
    left-handed renderers (load_meshes :628-634); KHR_lights_punctual directional lights become
    directional lights (GltfLoadSettings::enable_directional / directional_light_* defaults).  Returns dict(objects=[handles], skeletons=[handles],
    inverse_bind_matrices=[per skin], node_transforms)."""
    nodes = g.json.get("nodes", [])
    lh = r.handedness == 0
    parent_of = {}
    for i, n in enumerate(nodes):
        for ch in n.get("children", []):
            parent_of[ch] = i
    order, queue = [], [i for i in range(len(nodes)) if i not in parent_of]
    while queue:
        n = queue.pop(0)
        order.append(n)
        queue.extend(nodes[n].get("children", []))
    s = np.float32(scale)
    root = hm.scale((s, s, -s if lh else s))
    meshes = {}
    materials = {}
    image_cache = {}
    xf = [None] * len(nodes)
    out = dict(objects=[], skeletons=[], lights=[], inverse_bind_matrices=[], node_transforms=xf, topological_order=order,
               nodes=[dict(parent=parent_of.get(i), local_transform=None, objects=[], skin=None, skeletons=[]) for i in range(len(nodes))],
               skins=[dict(joints=list(sk["joints"])) for sk in g.json.get("skins", [])])
    for sk in g.json.get("skins", []):
        nj = len(sk["joints"])
        ibm = g.accessor(sk["inverseBindMatrices"]).astype(np.float32) if "inverseBindMatrices" in sk else np.tile(hm.identity(), (nj, 1))
        out["inverse_bind_matrices"].append(ibm)
    for ni in order:
        node = nodes[ni]
        parent = xf[parent_of[ni]] if ni in parent_of else root
        out["nodes"][ni]["local_transform"] = _node_local_matrix(node, hm)
        xf[ni] = hm.mat4_mul(parent, out["nodes"][ni]["local_transform"])
        li = node.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
        if li is not None and enable_directional:
            # instance_loaded_scene :530-545: directional lights only; direction = transform.transform_vector3(-Z)
            light = g.json["extensions"]["KHR_lights_punctual"]["lights"][li]
            if light.get("type") == "directional":
                m = xf[ni]
                direction = tuple(np.float32(-m[8 + k]) for k in range(3))  # -(column 2): the matrix applied to (0, 0, -1, 0)
                out["lights"].append(r.add_directional_light(color=tuple(light.get("color", [1.0, 1.0, 1.0])), intensity=light.get("intensity", 1.0),
                                                             direction=direction, distance=directional_light_shadow_distance,
                                                             resolution=directional_light_resolution))
        if "mesh" not in node:
            continue
        mi = node["mesh"]
        for pi in range(len(g.json["meshes"][mi]["primitives"])):
            if (mi, pi) not in meshes:
                p = g.primitive(mi, pi)
                idx = p["indices"].reshape(-1, 3)[:, ::-1].reshape(-1) if lh else p["indices"]
                meshes[(mi, pi)] = (r.add_mesh(p["positions"], idx, normals=p.get("normals"), tangents=p.get("tangents"),
                                               joint_indices=p.get("joints"), joint_weights=p.get("weights"),
                                               uv0=p.get("uv0"), colors=p.get("colors"), mesh_handedness=r.handedness), p["material"])
            mesh, mat_index = meshes[(mi, pi)]
            if mat_index not in materials:
                rec, key = material_from_gltf(g, mat_index, mk, r, image_cache, normal_y_down=normal_y_down)
                materials[mat_index] = r.add_material(rec, key)
            if "skin" in node:
                nj = len(out["inverse_bind_matrices"][node["skin"]])
                sk = r.add_skeleton(mesh, np.tile(hm.identity(), (nj, 1)))
                out["skeletons"].append(sk)
                out["nodes"][ni]["skin"] = node["skin"]
                out["nodes"][ni]["skeletons"].append(sk)
                out["objects"].append(r.add_object(None, materials[mat_index], xf[ni], skeleton=sk))
            else:
                out["objects"].append(r.add_object(mesh, materials[mat_index], xf[ni]))
            out["nodes"][ni]["objects"].append(out["objects"][-1])
    for k, sk in enumerate(out["skins"]):
        sk["inverse_bind_matrices"] = out["inverse_bind_matrices"][k]
    return out


def load_animations(g):
    """load_animations (rend3-gltf/src/lib.rs:724-773): per animation {"channels": {node: {"translation" | "rotation" |
    "scale": (times, values)}}, "duration"}.  Keyframe values are taken as stored -- the reference ignores the
    sampler's interpolation mode and always blends linearly (rend3-anim/src/lib.rs:163-175); rotations are read as f32
    (normalised integer encodings converted like the gltf crate's into_f32).  Morph-target weights are skipped (:765).
    duration = the latest key time of any channel (compute_animation_duration :706-722)."""
    out = []
    for anim in g.json.get("animations", []):
        channels = {}
        duration = 0.0
        for ch in anim["channels"]:
            target = ch["target"]
            if "node" not in target or target["path"] == "weights":
                continue
            smp = anim["samplers"][ch["sampler"]]
            times = g.accessor(smp["input"]).astype(np.float32).reshape(-1)
            values = g.accessor(smp["output"]).astype(np.float32)
            channels.setdefault(target["node"], {})[target["path"]] = (times, values)
            if len(times):
                duration = max(duration, float(times.max()))
        out.append(dict(channels=channels, duration=np.float32(duration), name=anim.get("name")))
    return out
