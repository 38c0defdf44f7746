"""
Scene builders shared by the oracle tests and the GPU parity tests.  Every builder takes a
`renderer` exposing the Renderer-shaped API (add_mesh/add_material/add_object/...): both
oracle.world.OracleRenderer and rend3_amd.Renderer implement it, so the same scene runs
through both.  `hm` is the host-math module matching the renderer (oracle.host or rend3_amd.host).

The scenes restate the reference's own tests:
  rend3-test/tests/simple.rs, object.rs, msaa.rs, shadow.rs; rend3-test/src/helpers.rs;
  examples/src/cube/mod.rs.
"""
import math
import os

import numpy as np

f32 = np.float32
LEFT, RIGHT = 0, 1
OPAQUE, CUTOUT, BLEND = 0, 1, 2


def plane_mesh(r):
    """rend3-test/src/helpers.rs:55-74"""
    pos = [(-1, -1, 0), (-1, 1, 0), (1, 1, 0), (1, -1, 0)]
    return r.add_mesh(pos, [0, 2, 1, 0, 3, 2], mesh_handedness=LEFT)


CUBE_POS = [
    (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1),
    (-1, 1, -1), (1, 1, -1), (1, -1, -1), (-1, -1, -1),
    (1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1),
    (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1),
    (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, 1, 1),
    (1, -1, 1), (-1, -1, 1), (-1, -1, -1), (1, -1, -1),
]
CUBE_IDX = [0, 1, 2, 2, 3, 0, 4, 5, 6, 6, 7, 4, 8, 9, 10, 10, 11, 8, 12, 13, 14, 14, 15, 12,
            16, 17, 18, 18, 19, 16, 20, 21, 22, 22, 23, 20]


def cube_mesh(r):
    """rend3-test/src/helpers.rs:77-130 == examples/src/cube/mod.rs:5-52"""
    return r.add_mesh(CUBE_POS, CUBE_IDX, mesh_handedness=LEFT)


def unlit(r, mk, color):
    return r.add_material(mk(albedo=color, albedo_mode="value", unlit=True), OPAQUE)


def lit(r, mk, color):
    return r.add_material(mk(albedo=color, albedo_mode="value", unlit=False), OPAQUE)


from rend3_amd.scenes import Pcg32, box, grid_plane, icosphere  # noqa: E402,F401  (input generators, shared with bench.py)


def random_rotation(rng, hm):
    return hm.mat4_mul(hm.mat4_mul(hm.rotation_y(rng.uniform(0, 2 * math.pi)), hm.rotation_x(rng.uniform(0, 2 * math.pi))),
                       hm.rotation_z(rng.uniform(0, 2 * math.pi)))


def build_random_scene(r, hm, mk, n_objects, seed, extent=(30.0, 8.0, 30.0), n_materials=8, handedness=LEFT,
                       lights=1, shadow_res=256, shadow_distance=60.0, with_cutout=False):
    """Small 'scifi-like' scene (SURVEY section 8d cfg 2 shape, reduced): instanced procedural meshes with random
    transforms inside a box around the origin, camera at the box centre."""
    rng = Pcg32(seed)
    meshes = []
    for sub in (0, 1, 2):
        p, i, n = icosphere(sub)
        if handedness == LEFT:
            i = i.reshape(-1, 3)[:, ::-1].reshape(-1)  # outward faces CW for LH front-face
        meshes.append(r.add_mesh(p, i, normals=n))
    p, i, n = box()
    if handedness == RIGHT:
        i = i.reshape(-1, 3)[:, ::-1].reshape(-1)
    meshes.append(r.add_mesh(p, i, normals=n))
    mats = []
    for k in range(n_materials):
        col = (rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), 1.0)
        key = OPAQUE
        cutout = None
        if with_cutout and k % 4 == 3:
            key, cutout = CUTOUT, 0.5
            col = col[:3] + (0.25 if k % 8 == 3 else 0.75,)
        rec = mk(albedo=col, albedo_mode="value", roughness=rng.uniform(0.2, 0.9),
                 metallic=1.0 if rng.uniform() < 0.2 else 0.0, cutout=cutout)
        mats.append(r.add_material(rec, key))
    handles = []
    for _ in range(n_objects):
        pos = (rng.uniform(-extent[0], extent[0]), rng.uniform(-extent[1], extent[1]), rng.uniform(-extent[2], extent[2]))
        s = math.exp(rng.uniform(math.log(0.25), math.log(4.0)))
        xf = hm.mat4_mul(hm.mat4_mul(hm.translation(pos), random_rotation(rng, hm)), hm.scale((s, s, s)))
        handles.append(r.add_object(meshes[rng.randint(len(meshes))], mats[rng.randint(len(mats))], xf))
    for k in range(lights):
        ang = 2 * math.pi * k / max(lights, 1)
        r.add_directional_light(color=(1, 1, 1), intensity=3.0,
                                direction=(math.cos(ang) - 0.3, -2.0, math.sin(ang) + 0.2),
                                distance=shadow_distance, resolution=shadow_res)
    return handles


def _tangents(normals):
    """Some unit tangent per vertex, perpendicular to the normal (any deterministic choice will do for parity)."""
    n = np.asarray(normals, dtype=np.float32)
    ref = np.where(np.abs(n[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]], dtype=np.float32), np.array([[1.0, 0.0, 0.0]], dtype=np.float32))
    t = np.cross(ref, n).astype(np.float32)
    t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), np.float32(1e-20))
    return t.astype(np.float32)


def build_textured_scene(r, hm, mk, n_objects, seed, extent=(20.0, 6.0, 20.0), handedness=LEFT, lights=1,
                         shadow_res=256, shadow_distance=50.0, encoded=False):
    """Row N2 scene: the instanced meshes of build_random_scene with texture coordinates, four RGBA8 textures
    (sRGB and linear formats, square / odd / 1x1 extents, full generated mip chains and a single-mip one) and ten
    materials covering the albedo-texture variants: linear and nearest samplers, value / vertex multipliers, a
    uv transform, unlit, and cutout materials whose alpha comes from the texture (forward and shadow passes).
    encoded: the same materials over textures in the loader's other formats -- BC7 / BC1 / BC3 / BC5 blocks with stored
    mip chains (random block data: every BC7 mode, punch-through BC1 blocks, both BC3 / BC5 endpoint orders), BGRA8,
    and RG8 / R8 with generated chains (add_texture_2d_encoded).  encoded="float": the float-decoded formats in the same
    slots -- Rgba16Float / Rgba16Unorm / Rg16Float with stored chains, Rgb10a2Unorm, Rgba32Float, Rgb9e5Ufloat, BC5 snorm
    blocks as the normal map and BC6H blocks (the committed unsaturated vectors) as the emissive map."""
    rng = Pcg32(seed)
    nrng = np.random.default_rng(seed)
    meshes = []
    for sub in (0, 1, 2):
        p, i, n = icosphere(sub)
        if handedness == LEFT:
            i = i.reshape(-1, 3)[:, ::-1].reshape(-1)
        uv = (p[:, :2] * np.float32(1.5 + sub) + np.float32(0.5)).astype(np.float32)
        meshes.append(r.add_mesh(p, i, normals=n, uv0=uv, tangents=_tangents(n)))
    p, i, n = box()
    if handedness == RIGHT:
        i = i.reshape(-1, 3)[:, ::-1].reshape(-1)
    uv = (p[:, [0, 2]] * np.float32(0.75) + p[:, [1, 1]] * np.float32(0.25)).astype(np.float32)
    meshes.append(r.add_mesh(p, i, normals=n, uv0=uv, tangents=_tangents(n)))

    noise = nrng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    noise[..., 3] = np.where(nrng.random((64, 64)) < 0.45, 40, 230).astype(np.uint8)
    yy, xx = np.mgrid[0:32, 0:128]
    checker = np.zeros((32, 128, 4), dtype=np.uint8)
    checker[..., 0] = np.where(((xx // 8) + (yy // 8)) % 2 == 0, 230, 30)
    checker[..., 1] = (xx * 2).astype(np.uint8)
    checker[..., 2] = (yy * 8).astype(np.uint8)
    checker[..., 3] = np.where(((xx // 4) % 3) == 0, 60, 255)
    odd = nrng.integers(0, 256, (19, 37, 4), dtype=np.uint8)
    one = np.array([[[200, 120, 40, 255]]], dtype=np.uint8)
    def blocks(fmt_id, block_bytes, w, h, levels, k):
        g = np.random.default_rng(seed * 16 + k)
        lv = [g.integers(0, 256, ((max(1, w >> i) + 3) // 4) * ((max(1, h >> i) + 3) // 4) * block_bytes, dtype=np.uint8).tobytes()
              for i in range(levels)]
        return r.add_texture_2d_encoded(fmt_id, w, h, lv)

    def chain(img, levels):
        """Point-sampled levels of an (H, W, C) array (any values do: both sides decode the same bytes)."""
        h, w = img.shape[:2]
        out = []
        for k in range(levels):
            lh, lw = max(1, h >> k), max(1, w >> k)
            out.append(np.ascontiguousarray(img[(np.arange(lh) * h) // lh][:, (np.arange(lw) * w) // lw]))
        return out

    if encoded == "float":
        nf = noise.astype(np.float32) / np.float32(255.0)
        nf[..., :3] *= np.float32(1.5)  # values above 1 too
        t_noise = r.add_texture_2d_encoded(21, 64, 64, [lv.astype(np.float16).tobytes() for lv in chain(nf, 4)])       # Rgba16Float
        c10 = checker.astype(np.uint32)
        packed = (c10[..., 0] * 4 + 1) | ((c10[..., 1] * 4 + 2) << 10) | ((c10[..., 2] * 4 + 3) << 20) | ((c10[..., 3] // 64) << 30)
        t_check = r.add_texture_2d_encoded(27, 128, 32, [packed.astype(np.uint32).tobytes()])                          # Rgb10a2Unorm
        o16 = odd.astype(np.uint16) * np.uint16(257) ^ np.uint16(0x00A5)
        t_odd = r.add_texture_2d_encoded(25, 37, 19, [lv.tobytes() for lv in chain(o16, 3)])                            # Rgba16Unorm
        t_one = r.add_texture_2d_encoded(24, 1, 1, [np.array([0.6, 0.2, 0.03, 1.0], dtype=np.float32).tobytes()])      # Rgba32Float
        e5 = nrng.integers(0, 1 << 27, (64, 64), dtype=np.uint32) | (nrng.integers(8, 16, (64, 64), dtype=np.uint32) << 27)
        t_flat = r.add_texture_2d_encoded(29, 64, 64, [e5.tobytes()])                                                  # Rgb9e5Ufloat
    elif encoded:
        t_noise = blocks(15, 16, 64, 64, 4, 0)       # Bc7RgbaUnormSrgb, 4 stored levels
        t_check = blocks(6, 8, 128, 32, 1, 1)        # Bc1RgbaUnorm, single level
        t_odd = blocks(11, 16, 37, 19, 3, 2)         # Bc3RgbaUnormSrgb, extent not a multiple of the block
        t_one = r.add_texture_2d(one, srgb=True, mip_count=1, mip_source="uploaded")
        t_flat = r.add_texture_2d_encoded(4, 64, 64, [noise[..., [2, 1, 0, 3]].tobytes()])  # Bgra8Unorm
    else:
        t_noise = r.add_texture_2d(noise, srgb=True, mip_count="maximum", mip_source="generated")
        t_check = r.add_texture_2d(checker, srgb=False, mip_count="maximum", mip_source="generated")
        t_odd = r.add_texture_2d(odd, srgb=True, mip_count="maximum", mip_source="generated")
        t_one = r.add_texture_2d(one, srgb=True, mip_count=1, mip_source="uploaded")
        t_flat = r.add_texture_2d(noise, srgb=False, mip_count=1, mip_source="uploaded")

    c30, s30 = math.cos(0.5), math.sin(0.5)
    mats = [
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_noise, roughness=0.5)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_check, roughness=0.3, nearest=True)),
        r.add_material(mk(albedo_mode="texture_value", albedo_texture=t_odd, albedo=(0.9, 0.7, 0.5, 1.0), roughness=0.7, metallic=1.0)),
        r.add_material(mk(albedo_mode="texture_vertex", albedo_texture=t_check, roughness=0.6)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_one, roughness=0.4)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_flat, unlit=True)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_noise, roughness=0.5,
                          uv_transform0=[[2.0 * c30, -2.0 * s30, 0.25], [2.0 * s30, 2.0 * c30, -0.5], [0.0, 0.0, 1.0]])),
        r.add_material(mk(albedo=(0.4, 0.8, 0.3, 1.0), albedo_mode="value", roughness=0.5)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_noise, roughness=0.5, cutout=0.5), CUTOUT),
        r.add_material(mk(albedo_mode="texture_value", albedo_texture=t_check, albedo=(1.0, 1.0, 1.0, 0.9), roughness=0.5,
                          cutout=0.5, nearest=True), CUTOUT),
    ]
    # normal map (tangent-space bumps), packed AO/roughness/metallic, emissive stripes -- linear formats, full mip chains
    yy, xx = np.mgrid[0:64, 0:64]
    nx = 0.35 * np.sin(xx * (2 * math.pi / 16.0)) + 0.5
    ny = 0.35 * np.cos(yy * (2 * math.pi / 8.0)) + 0.5
    nmap = np.stack([nx * 255, ny * 255, np.full_like(nx, 235.0), ny * 255], axis=2).astype(np.uint8)
    aomr_tex = nrng.integers(40, 256, (32, 32, 4), dtype=np.uint8)
    emis = np.zeros((16, 16, 4), dtype=np.uint8)
    emis[::4, :, 0] = 255
    emis[:, ::4, 2] = 200
    emis[..., 3] = 255
    if encoded == "float":
        t_nmap = blocks(31, 16, 64, 64, 7, 3)        # Bc5RgSnorm, full stored chain
        am = aomr_tex[..., :2].astype(np.float32) / np.float32(255.0)
        t_aomr = r.add_texture_2d_encoded(20, 32, 32, [lv.astype(np.float16).tobytes() for lv in chain(am, 6)])        # Rg16Float
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bcn_float_blocks.npz"))
        t_emis = r.add_texture_2d_encoded(32, 16, 16, [gold["bc6h_uf_data"][:16 * 16].tobytes()])                     # Bc6hRgbUfloat
    elif encoded:
        t_nmap = blocks(13, 16, 64, 64, 7, 3)        # Bc5RgUnorm, full stored chain
        t_aomr = r.add_texture_2d_encoded(3, 32, 32, [np.ascontiguousarray(aomr_tex[..., :2]).tobytes()], generate_mips=True)  # Rg8Unorm
        t_emis = r.add_texture_2d_encoded(2, 16, 16, [np.ascontiguousarray(emis[..., 0]).tobytes()], generate_mips=True)       # R8Unorm
    else:
        t_nmap = r.add_texture_2d(nmap, srgb=False, mip_count="maximum", mip_source="generated")
        t_aomr = r.add_texture_2d(aomr_tex, srgb=False, mip_count="maximum", mip_source="generated")
        t_emis = r.add_texture_2d(emis, srgb=True, mip_count="maximum", mip_source="generated")
    mats += [
        r.add_material(mk(albedo=(0.8, 0.8, 0.8, 1.0), albedo_mode="value", roughness=0.6, normal_texture=t_nmap)),
        r.add_material(mk(albedo_mode="texture", albedo_texture=t_noise, roughness=0.8, metallic=1.0, normal_texture=t_nmap,
                          normal_mode="bicomponent", normal_y_down=True, aomr=("combined", t_aomr))),
        r.add_material(mk(albedo=(0.7, 0.6, 0.5, 1.0), albedo_mode="value", roughness=0.9, metallic=0.8, normal_texture=t_nmap,
                          normal_mode="bicomponent_swizzled", aomr=("swizzled_split", t_aomr, t_aomr))),
        r.add_material(mk(albedo=(0.5, 0.6, 0.9, 1.0), albedo_mode="value", roughness=0.7, metallic=0.5,
                          aomr=("split", None, t_aomr), emissive=(0.5, 0.4, 0.3), emissive_texture=t_emis)),
        r.add_material(mk(albedo=(0.9, 0.9, 0.2, 1.0), albedo_mode="value", roughness=0.9, metallic=1.0,
                          aomr=("bw_split", t_aomr, t_check, t_noise), reflectance_texture=t_aomr,
                          clear_coat=0.8, clear_coat_roughness=0.6, clearcoat_textures=("gltf_combined", t_aomr))),
        r.add_material(mk(albedo=(0.3, 0.9, 0.8, 1.0), albedo_mode="value", roughness=0.5, clear_coat=0.9,
                          clear_coat_roughness=0.7, clearcoat_textures=("gltf_split", t_aomr, t_check), nearest=True)),
        r.add_material(mk(albedo=(0.9, 0.3, 0.8, 1.0), albedo_mode="value", roughness=0.5, clear_coat=0.9,
                          clear_coat_roughness=0.7, clearcoat_textures=("bw_split", None, t_aomr), anisotropy_texture=t_aomr)),
    ]
    handles = []
    for _ in range(n_objects):
        pos = (rng.uniform(-extent[0], extent[0]), rng.uniform(-extent[1], extent[1]), rng.uniform(-extent[2], extent[2]))
        s = math.exp(rng.uniform(math.log(0.3), math.log(2.5)))
        xf = hm.mat4_mul(hm.mat4_mul(hm.translation(pos), random_rotation(rng, hm)), hm.scale((s, s, s)))
        handles.append(r.add_object(meshes[rng.randint(len(meshes))], mats[rng.randint(len(mats))], xf))
    for k in range(lights):
        ang = 2 * math.pi * k / max(lights, 1)
        r.add_directional_light(color=(1, 1, 1), intensity=3.0,
                                direction=(math.cos(ang) - 0.3, -2.0, math.sin(ang) + 0.2),
                                distance=shadow_distance, resolution=shadow_res)
    return handles


def write_textured_gltf(path, containers=False):
    """containers: the base colour image is a DDS file (DX10 header, BC7, 3 stored levels), the normal map a KTX2 file
    (BC5: two components -> the loader's bicomponent normal mode), the occlusion / metallic-roughness image a legacy
    DXT1 DDS and the luminance occlusion image a single-level R8 KTX2 (-> generated chain).

    A small self-contained .gltf (buffers and PNG images as data URIs) exercising the texture side of the glTF
    loader: base colour texture with a NEAREST sampler and KHR_texture_transform, normal map, one image used for both
    occlusion and metallic-roughness (-> Combined packing), emissive texture, a MASK material whose alpha comes from
    the texture, and a luminance-only occlusion image (-> Split packing)."""
    import base64
    import io
    import json

    from PIL import Image

    rng = np.random.default_rng(77)

    def png(arr, mode):
        b = io.BytesIO()
        Image.fromarray(arr, mode).save(b, format="PNG")
        return "data:image/png;base64," + base64.b64encode(b.getvalue()).decode()

    base = rng.integers(0, 256, (32, 32, 4), dtype=np.uint8)
    base[..., 3] = np.where(rng.random((32, 32)) < 0.5, 30, 240).astype(np.uint8)
    yy, xx = np.mgrid[0:16, 0:16]
    nrm = np.stack([128 + 60 * np.sin(xx), 128 + 60 * np.cos(yy), np.full_like(xx, 230)], axis=2).astype(np.uint8)
    orm = rng.integers(60, 256, (16, 16, 3), dtype=np.uint8)
    emi = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)
    lum = rng.integers(100, 256, (8, 8), dtype=np.uint8)
    images = [png(base, "RGBA"), png(nrm, "RGB"), png(orm, "RGB"), png(emi, "RGB"), png(lum, "L")]
    if containers:
        import test_texture_formats as T

        def uri(blob):
            return "data:application/octet-stream;base64," + base64.b64encode(blob).decode()

        def blocks(bb, w, h, n):
            return [rng.integers(0, 256, ((max(1, w >> k) + 3) // 4) * ((max(1, h >> k) + 3) // 4) * bb, dtype=np.uint8).tobytes() for k in range(n)]

        if containers == "float":
            # base colour: BC6H (unsaturated blocks of the committed vectors), 3 stored levels; normal map: BC5 snorm, 5 levels;
            # occlusion / metallic-roughness: Rgb10a2Unorm DDS-less KTX2, one level -> generated chain; emissive: Rgba16Float KTX2,
            # one level -> generated chain; luminance occlusion: R16Float KTX2 with its two levels stored
            gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bcn_float_blocks.npz"))["bc6h_uf_data"].reshape(-1, 16)
            images[0] = uri(T.write_dds(32, 32, [gold[:64].tobytes(), gold[64:80].tobytes(), gold[80:84].tobytes()], dxgi=95))
            images[1] = uri(T.write_ktx2(142, 16, 16, blocks(16, 16, 16, 5)))
            o10 = orm.astype(np.uint32) * 4 + 1
            images[2] = uri(T.write_ktx2(64, 16, 16, [(o10[..., 0] | (o10[..., 1] << 10) | (o10[..., 2] << 20) | (3 << 30)).astype(np.uint32).tobytes()]))
            e16 = np.concatenate([emi.astype(np.float32) / np.float32(64.0), np.ones((8, 8, 1), np.float32)], axis=2).astype(np.float16)
            images[3] = uri(T.write_ktx2(97, 8, 8, [e16.tobytes()]))
            l16 = (lum.astype(np.float32) / np.float32(255.0)).astype(np.float16)
            images[4] = uri(T.write_ktx2(76, 8, 8, [l16.tobytes(), l16[::2, ::2].copy().tobytes()]))
        else:
            images[0] = uri(T.write_dds(32, 32, blocks(16, 32, 32, 3), dxgi=98))
            images[1] = uri(T.write_ktx2(141, 16, 16, blocks(16, 16, 16, 5)))
            images[2] = uri(T.write_dds(16, 16, blocks(8, 16, 16, 1), fourcc=b"DXT1"))
            images[4] = uri(T.write_ktx2(9, 8, 8, [lum.tobytes()]))

    # one quad (two triangles) with normals, tangents, uvs
    pos = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32)
    nor = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (4, 1))
    tan = np.tile(np.array([[1, 0, 0, 1]], dtype=np.float32), (4, 1))
    uv = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=np.float32) * np.float32(1.7)
    idx = np.array([0, 1, 2, 2, 3, 0], dtype=np.uint16)
    blobs = [pos.tobytes(), nor.tobytes(), tan.tobytes(), uv.tobytes(), idx.tobytes()]
    offs, cur = [], 0
    for b in blobs:
        offs.append(cur)
        cur += (len(b) + 3) // 4 * 4
    buf = bytearray(cur)
    for o, b in zip(offs, blobs):
        buf[o: o + len(b)] = b
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(buf), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(buf)).decode()}],
        "bufferViews": [{"buffer": 0, "byteOffset": o, "byteLength": len(b)} for o, b in zip(offs, blobs)],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [-1, -1, 0], "max": [1, 1, 0]},
            {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
            {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC4"},
            {"bufferView": 3, "componentType": 5126, "count": 4, "type": "VEC2"},
            {"bufferView": 4, "componentType": 5123, "count": 6, "type": "SCALAR"},
        ],
        "images": [{"uri": u} for u in images],
        "samplers": [{"magFilter": 9728}, {"magFilter": 9729}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}, {"source": 2, "sampler": 1},
                     {"source": 3}, {"source": 4, "sampler": 1}, {"source": 0, "sampler": 1}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorFactor": [1.0, 0.9, 0.8, 1.0], "baseColorTexture": {"index": 0, "extensions": {
                "KHR_texture_transform": {"offset": [0.1, 0.2], "rotation": 0.3, "scale": [1.5, 0.75]}}},
                "metallicRoughnessTexture": {"index": 2}, "roughnessFactor": 0.9, "metallicFactor": 0.7},
             "normalTexture": {"index": 1}, "occlusionTexture": {"index": 2}, "emissiveTexture": {"index": 3},
             "emissiveFactor": [0.3, 0.2, 0.1]},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 5}, "roughnessFactor": 0.5, "metallicFactor": 0.0},
             "alphaMode": "MASK", "alphaCutoff": 0.4, "occlusionTexture": {"index": 4}},
        ],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TANGENT": 2, "TEXCOORD_0": 3}, "indices": 4, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TANGENT": 2, "TEXCOORD_0": 3}, "indices": 4, "material": 1}]}],
        "nodes": [{"mesh": 0, "translation": [-1.2, 0.0, 0.0], "rotation": [0.0, 0.3826834, 0.0, 0.9238795]},
                  {"mesh": 1, "translation": [1.2, 0.0, 0.5], "scale": [1.0, 1.3, 1.0]}],
        "scenes": [{"nodes": [0, 1]}],
        "scene": 0,
    }
    with open(path, "w") as fh:
        json.dump(doc, fh)
    return path


def add_blend_objects(r, hm, mk, seed, n=12, textured=False):
    """Translucent boxes and spheres (TransparencyType::Blend) scattered in front of the camera of the random scenes:
    overlapping each other and the opaque geometry, different alphas, one unlit, optionally a textured one."""
    rng = Pcg32(seed)
    p, i, nrm = box()
    if getattr(r, "handedness", LEFT) == RIGHT:
        i = i.reshape(-1, 3)[:, ::-1].reshape(-1)
    uv = (p[:, [0, 2]] * np.float32(0.75) + p[:, [1, 1]] * np.float32(0.25)).astype(np.float32)
    mbox = r.add_mesh(p, i, normals=nrm, uv0=uv)
    ps, is_, ns = icosphere(2)
    if getattr(r, "handedness", LEFT) == LEFT:
        is_ = is_.reshape(-1, 3)[:, ::-1].reshape(-1)
    msph = r.add_mesh(ps, is_, normals=ns, uv0=(ps[:, :2] * np.float32(2.0)).astype(np.float32))
    mats = []
    for k in range(5):
        col = (rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.25, 0.8))
        mats.append(r.add_material(mk(albedo=col, albedo_mode="value", roughness=rng.uniform(0.2, 0.8), unlit=(k == 3)), BLEND))
    if textured:
        nrng = np.random.default_rng(seed)
        img = nrng.integers(0, 256, (32, 32, 4), dtype=np.uint8)
        t = r.add_texture_2d(img, srgb=True, mip_count="maximum", mip_source="generated")
        mats.append(r.add_material(mk(albedo_mode="texture_value", albedo_texture=t, albedo=(1.0, 1.0, 1.0, 0.7), roughness=0.5), BLEND))
    handles = []
    for k in range(n):
        pos = (rng.uniform(-6.0, 6.0), rng.uniform(-1.0, 3.0), rng.uniform(3.0, 14.0))
        sc = (rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0.05, 1.0))
        xf = hm.mat4_mul(hm.mat4_mul(hm.translation(pos), random_rotation(rng, hm)), hm.scale(sc))
        handles.append(r.add_object(mbox if k % 3 else msph, mats[rng.randint(len(mats))], xf))
    return handles


def write_animated_gltf(path):
    """A skinned, animated .gltf for row N4 (rend3-anim): a 4 x 1 x 1 bar of 5 vertex rings skinned to a 4-joint chain
    (joint 0 = root, parented to a plain node; one extra joint hangs off joint 1), one clip with rotation channels on
    three joints, a translation channel on one, a scale channel on one, an unanimated joint, keys that start after t = 0
    on one channel, plus an animated plain node carrying a cube
    (node-transform half of pose_animation_frame)."""
    import base64
    import json

    rings = 9
    pos, nor, jnt, wgt, idx = [], [], [], [], []
    for i in range(rings):
        x = 4.0 * i / (rings - 1)
        for (y, z) in ((-0.3, -0.3), (0.3, -0.3), (0.3, 0.3), (-0.3, 0.3)):
            pos.append([x, y, z])
            n = np.array([0.0, y, z]) / np.hypot(y, z)
            nor.append(n.tolist())
            f = x  # joints sit at x = 0, 1, 2, 3
            j0 = min(int(f), 3)
            j1 = min(j0 + 1, 3)
            w1 = f - int(f) if j0 < 3 else 0.0
            jnt.append([j0, j1, 4 if i == rings - 1 else 0, 0])
            wgt.append([1.0 - w1 if i != rings - 1 else 0.75 * (1.0 - w1), w1 if i != rings - 1 else 0.75 * w1, 0.25 if i == rings - 1 else 0.0, 0.0])
    for i in range(rings - 1):
        for k in range(4):
            a, b = 4 * i + k, 4 * i + (k + 1) % 4
            idx += [a, b, a + 4, b, b + 4, a + 4]
    pos, nor = np.array(pos, np.float32), np.array(nor, np.float32)
    jnt, wgt, idx = np.array(jnt, np.uint16), np.array(wgt, np.float32), np.array(idx, np.uint16)
    ibm = np.stack([np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -x, 0, 0, 1], np.float32) for x in (0.0, 1.0, 2.0, 3.0, 1.0)])
    cube_p = np.array([[x, y, z] for x in (-.4, .4) for y in (-.4, .4) for z in (-.4, .4)], np.float32)
    cube_i = np.array([0, 1, 3, 0, 3, 2, 4, 6, 7, 4, 7, 5, 0, 4, 5, 0, 5, 1, 2, 3, 7, 2, 7, 6, 0, 2, 6, 0, 6, 4, 1, 5, 7, 1, 7, 3], np.uint16)

    def quat_z(a):
        return [0.0, 0.0, float(np.sin(a / 2)), float(np.cos(a / 2))]

    t_a = np.array([0.0, 0.5, 1.0, 2.0], np.float32)
    t_b = np.array([0.25, 1.5], np.float32)          # starts after t = 0
    rot1 = np.array([quat_z(0.0), quat_z(0.6), quat_z(-0.4), quat_z(0.9)], np.float32)
    rot2 = np.array([quat_z(0.3), [0.0, 0.0, -float(np.sin(0.5)), -float(np.cos(0.5))]], np.float32)  # second key on the far hemisphere
    rot3 = np.array([[float(np.sin(0.2)), 0.0, 0.0, float(np.cos(0.2))], quat_z(0.0), quat_z(0.0), [0.0, float(np.sin(0.4)), 0.0, float(np.cos(0.4))]], np.float32)
    tra0 = np.array([[0, 0, 0], [0, 0.5, 0], [0.2, 0.5, 0.1], [0, 0, 0]], np.float32)
    sca2 = np.array([[1, 1, 1], [1.5, 0.8, 1.2], [1, 1, 1], [0.7, 1.3, 1.0]], np.float32)
    cube_t = np.array([[0, 2, 0], [1, 2.5, 0], [2, 2, 1], [0, 2, 0]], np.float32)
    blobs = [pos, nor, jnt, wgt, idx, ibm, cube_p, cube_i, t_a, t_b, rot1, rot2, rot3, tra0, sca2, cube_t]
    raw = [np.ascontiguousarray(b).tobytes() for b in blobs]
    offs, cur = [], 0
    for b in raw:
        offs.append(cur)
        cur += (len(b) + 3) // 4 * 4
    buf = bytearray(cur)
    for o, b in zip(offs, raw):
        buf[o:o + len(b)] = b

    def acc(i, ctype, count, typ, **kw):
        return dict(bufferView=i, componentType=ctype, count=count, type=typ, **kw)

    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": len(buf), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(buf)).decode()}],
        "bufferViews": [{"buffer": 0, "byteOffset": o, "byteLength": len(b)} for o, b in zip(offs, raw)],
        "accessors": [
            acc(0, 5126, len(pos), "VEC3", min=pos.min(0).tolist(), max=pos.max(0).tolist()), acc(1, 5126, len(nor), "VEC3"),
            acc(2, 5123, len(jnt), "VEC4"), acc(3, 5126, len(wgt), "VEC4"), acc(4, 5123, len(idx), "SCALAR"), acc(5, 5126, 5, "MAT4"),
            acc(6, 5126, 8, "VEC3", min=[-.4] * 3, max=[.4] * 3), acc(7, 5123, 36, "SCALAR"),
            acc(8, 5126, 4, "SCALAR", min=[0.0], max=[2.0]), acc(9, 5126, 2, "SCALAR", min=[0.25], max=[1.5]),
            acc(10, 5126, 4, "VEC4"), acc(11, 5126, 2, "VEC4"), acc(12, 5126, 4, "VEC4"), acc(13, 5126, 4, "VEC3"), acc(14, 5126, 4, "VEC3"),
            acc(15, 5126, 4, "VEC3"),
        ],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.5, 0.3, 1.0], "roughnessFactor": 0.6, "metallicFactor": 0.1}},
                      {"pbrMetallicRoughness": {"baseColorFactor": [0.3, 0.6, 0.9, 1.0], "roughnessFactor": 0.4, "metallicFactor": 0.0}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "JOINTS_0": 2, "WEIGHTS_0": 3}, "indices": 4, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 6}, "indices": 7, "material": 1}]}],
        # 0: armature root (plain node), 1-4: joint chain, 5: side joint under joint 2 (node), 6: skinned mesh node, 7: cube
        "nodes": [
            {"children": [1], "translation": [-2.0, 0.0, 0.0]},
            {"children": [2], "rotation": quat_z(0.1)},
            {"children": [3, 5], "translation": [1.0, 0.0, 0.0]},
            {"children": [4], "translation": [1.0, 0.0, 0.0], "scale": [1.0, 1.0, 1.0]},
            {"translation": [1.0, 0.0, 0.0]},
            {"translation": [0.0, 0.5, 0.0]},
            {"mesh": 0, "skin": 0},
            {"mesh": 1, "translation": [0.0, 2.0, 0.0]},
        ],
        "skins": [{"joints": [1, 2, 3, 4, 5], "inverseBindMatrices": 5, "skeleton": 1}],
        "animations": [{
            "name": "bend",
            "samplers": [{"input": 8, "output": 10}, {"input": 9, "output": 11}, {"input": 8, "output": 12}, {"input": 8, "output": 13},
                         {"input": 8, "output": 14}, {"input": 8, "output": 15}],
            "channels": [{"sampler": 0, "target": {"node": 2, "path": "rotation"}}, {"sampler": 1, "target": {"node": 3, "path": "rotation"}},
                         {"sampler": 2, "target": {"node": 4, "path": "rotation"}}, {"sampler": 3, "target": {"node": 1, "path": "translation"}},
                         {"sampler": 4, "target": {"node": 3, "path": "scale"}}, {"sampler": 5, "target": {"node": 7, "path": "translation"}}],
        }],
        "scenes": [{"nodes": [0, 6, 7]}],
        "scene": 0,
    }
    with open(path, "w") as fh:
        json.dump(doc, fh)
    return path
