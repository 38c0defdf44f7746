"""KTX2 and DDS containers -> (format, extent, level data), the way rend3-gltf's load_image reads them
(rend3-gltf/src/lib.rs:1013-1105) through the `ktx2` and `ddsfile` crates, with its format maps
(util::map_ktx2_format :1176-1410, map_d3d_format :1419-1490, map_dxgi_format :1493-1610).

The loader decides sRGB-ness from the texture's USE (albedo / emissive -> srgb, everything else linear), not from the
file: every map takes the `srgb` flag and ignores the _SRGB suffix of the stored format.  Formats the maps turn into
TextureFormats a float-sampled texture binding cannot hold or this library has no decoder for (integer, depth, ETC2 /
EAC, ASTC) raise TextureUnsupported; formats the maps reject raise the reference's own error kinds.

Host-side parsing only; decoding happens on the GPU (r3n_textures_write_encoded, csrc/texture_decode.hip).
"""
import struct

# R3N_TEXTURE_* (include/r3n.h)
RGBA8, RGBA8_SRGB, R8, RG8, BGRA8, BGRA8_SRGB = 0, 1, 2, 3, 4, 5
BC1, BC1_SRGB, BC2, BC2_SRGB, BC3, BC3_SRGB, BC4, BC5, BC7, BC7_SRGB = 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
# decoded to float texels (snorm, 16-bit, float, packed float, BC4 / BC5 snorm, BC6H)
R8_SNORM, RG8_SNORM, RGBA8_SNORM, R16F, RG16F, RGBA16F, R32F, RG32F, RGBA32F = 16, 17, 18, 19, 20, 21, 22, 23, 24
RGBA16_UNORM, RGBA16_SNORM, RGB10A2, RG11B10F, RGB9E5, BC4_SNORM, BC5_SNORM, BC6H_UF, BC6H_SF = 25, 26, 27, 28, 29, 30, 31, 32, 33
FORMAT_COUNT = 34
FORMAT_NAMES = {RGBA8: "Rgba8Unorm", RGBA8_SRGB: "Rgba8UnormSrgb", R8: "R8Unorm", RG8: "Rg8Unorm", BGRA8: "Bgra8Unorm",
                BGRA8_SRGB: "Bgra8UnormSrgb", BC1: "Bc1RgbaUnorm", BC1_SRGB: "Bc1RgbaUnormSrgb", BC2: "Bc2RgbaUnorm",
                BC2_SRGB: "Bc2RgbaUnormSrgb", BC3: "Bc3RgbaUnorm", BC3_SRGB: "Bc3RgbaUnormSrgb", BC4: "Bc4RUnorm",
                BC5: "Bc5RgUnorm", BC7: "Bc7RgbaUnorm", BC7_SRGB: "Bc7RgbaUnormSrgb", R8_SNORM: "R8Snorm", RG8_SNORM: "Rg8Snorm",
                RGBA8_SNORM: "Rgba8Snorm", R16F: "R16Float", RG16F: "Rg16Float", RGBA16F: "Rgba16Float", R32F: "R32Float",
                RG32F: "Rg32Float", RGBA32F: "Rgba32Float", RGBA16_UNORM: "Rgba16Unorm", RGBA16_SNORM: "Rgba16Snorm",
                RGB10A2: "Rgb10a2Unorm", RG11B10F: "Rg11b10Float", RGB9E5: "Rgb9e5Ufloat", BC4_SNORM: "Bc4RSnorm",
                BC5_SNORM: "Bc5RgSnorm", BC6H_UF: "Bc6hRgbUfloat", BC6H_SF: "Bc6hRgbFloat"}
BLOCK_BYTES = {BC1: 8, BC1_SRGB: 8, BC4: 8, BC2: 16, BC2_SRGB: 16, BC3: 16, BC3_SRGB: 16, BC5: 16, BC7: 16, BC7_SRGB: 16,
               BC4_SNORM: 8, BC5_SNORM: 16, BC6H_UF: 16, BC6H_SF: 16}
TEXEL_BYTES = {RGBA8: 4, RGBA8_SRGB: 4, BGRA8: 4, BGRA8_SRGB: 4, R8: 1, RG8: 2, R8_SNORM: 1, RG8_SNORM: 2, RGBA8_SNORM: 4, R16F: 2,
               RG16F: 4, RGBA16F: 8, R32F: 4, RG32F: 8, RGBA32F: 16, RGBA16_UNORM: 8, RGBA16_SNORM: 8, RGB10A2: 4, RG11B10F: 4, RGB9E5: 4}


# TextureFormat::describe().components where it is not 4 (the loader picks the normal / AO-M-R packing modes from it)
COMPONENTS = {R8: 1, BC4: 1, R8_SNORM: 1, R16F: 1, R32F: 1, BC4_SNORM: 1, RG8: 2, BC5: 2, RG8_SNORM: 2, RG16F: 2, RG32F: 2, BC5_SNORM: 2,
              RG11B10F: 3, RGB9E5: 3, BC6H_UF: 3, BC6H_SF: 3}


class TextureLoadError(ValueError):
    """GltfLoadError's texture variants (rend3-gltf/src/lib.rs:286-310): kind is the variant name."""

    def __init__(self, kind, detail=""):
        super().__init__(f"{kind}: {detail}" if detail else kind)
        self.kind = kind


class TextureUnsupported(TextureLoadError):
    """The reference maps the format to a TextureFormat the GPU would sample; this library has no path for it."""

    def __init__(self, detail):
        super().__init__("TextureUnsupported", detail)


def level_bytes(fmt, w, h):
    if fmt in BLOCK_BYTES:
        return ((w + 3) // 4) * ((h + 3) // 4) * BLOCK_BYTES[fmt]
    return w * h * TEXEL_BYTES[fmt]


def is_block_format(fmt):
    return fmt in BLOCK_BYTES


def is_float_format(fmt):
    """Decoded into float texels (four f32 per texel in the pool) instead of RGBA8."""
    return R8_SNORM <= fmt < FORMAT_COUNT


def _pick(srgb, linear_fmt, srgb_fmt):
    return srgb_fmt if srgb else linear_fmt


# ---------------------------------------------------------------------------------------------- KTX2
KTX2_MAGIC = bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
# VkFormat values (vulkan_core.h)
_VK_FLOAT = {10: R8_SNORM, 17: RG8_SNORM, 38: RGBA8_SNORM, 64: RGB10A2, 76: R16F, 83: RG16F, 91: RGBA16_UNORM, 92: RGBA16_SNORM,
             97: RGBA16F, 100: R32F, 103: RG32F, 109: RGBA32F, 122: RG11B10F, 123: RGB9E5, 140: BC4_SNORM, 142: BC5_SNORM,
             143: BC6H_UF, 144: BC6H_SF}
_VK_UNSUPPORTED = {  # mapped by the reference to formats a float-sampled binding cannot hold
    13: "R8Uint", 14: "R8Sint", 20: "Rg8Uint", 21: "Rg8Sint", 41: "Rgba8Uint", 42: "Rgba8Sint", 74: "R16Uint", 75: "R16Sint",
    81: "Rg16Uint", 82: "Rg16Sint", 95: "Rgba16Uint", 96: "Rgba16Sint", 98: "R32Uint", 99: "R32Sint", 101: "Rg32Uint",
    102: "Rg32Sint", 107: "Rgba32Uint", 108: "Rgba32Sint", 125: "Depth24Plus", 126: "Depth32Float", 129: "Depth24PlusStencil8"}


def map_ktx2_format(vk, srgb):
    """util::map_ktx2_format.  Returns an R3N_TEXTURE_* id, or None where the reference returns None."""
    if vk in (9, 15):          # R8_UNORM | R8_SRGB
        return None if srgb else R8
    if vk in (16, 22):         # R8G8_UNORM | R8G8_SRGB
        return None if srgb else RG8
    if vk in (37, 43):         # R8G8B8A8_UNORM | _SRGB
        return _pick(srgb, RGBA8, RGBA8_SRGB)
    if vk in (44, 50):         # B8G8R8A8_UNORM | _SRGB
        return _pick(srgb, BGRA8, BGRA8_SRGB)
    if vk in (131, 132, 133, 134):
        return _pick(srgb, BC1, BC1_SRGB)
    if vk in (135, 136):
        return _pick(srgb, BC2, BC2_SRGB)
    if vk in (137, 138):
        return _pick(srgb, BC3, BC3_SRGB)
    if vk == 139:
        return BC4
    if vk == 141:
        return BC5
    if vk in (145, 146):
        return _pick(srgb, BC7, BC7_SRGB)
    if vk in _VK_FLOAT:
        return _VK_FLOAT[vk]
    if vk in _VK_UNSUPPORTED or 147 <= vk <= 184:  # ETC2 / EAC / ASTC blocks
        raise TextureUnsupported(f"KTX2 vkFormat {vk} ({_VK_UNSUPPORTED.get(vk, 'ETC2 / EAC / ASTC')})")
    return None


def parse_ktx2(data, srgb):
    """ktx2::Reader::new + the loader's checks.  None when `data` is not a KTX2 file (the loader then tries DDS)."""
    if len(data) < 80 or bytes(data[:12]) != KTX2_MAGIC:
        return None
    (vk, _type_size, w, h, _depth, layers, faces, levels, scheme) = struct.unpack_from("<9I", data, 12)
    fmt = map_ktx2_format(vk, srgb)
    if fmt is None:
        raise TextureLoadError("TextureBadKxt2Format", f"vkFormat {vk}")
    if levels == 0:
        raise TextureLoadError("TextureZeroLevels")
    if layers >= 2:
        raise TextureLoadError("TextureTooManyLayers")
    if scheme != 0:
        raise TextureUnsupported(f"KTX2 supercompression scheme {scheme}")
    if faces != 1 or h == 0:
        raise TextureUnsupported("KTX2 cube maps / 1D textures")
    index = 80  # header 48 + index 32
    if len(data) < index + 24 * levels:
        return None  # the reader fails, as on any truncated file
    out = []
    for k in range(levels):
        off, length, _unc = struct.unpack_from("<3Q", data, index + 24 * k)
        if off + length > len(data):
            return None
        lw, lh = max(1, w >> k), max(1, h >> k)
        if length != level_bytes(fmt, lw, lh):
            raise TextureLoadError("TextureDecode", f"KTX2 level {k}: {length} bytes, expected {level_bytes(fmt, lw, lh)}")
        out.append(bytes(data[off:off + length]))
    return {"format": fmt, "width": w, "height": h, "levels": out}


# ---------------------------------------------------------------------------------------------- DDS
_DXGI = {  # map_dxgi_format; typeless members of a family behave like its unorm one
    27: ("rgba",), 28: ("rgba",), 29: ("rgba",), 48: ("rg",), 49: ("rg",), 60: ("r",), 61: ("r",),
    70: ("bc1",), 71: ("bc1",), 72: ("bc1",), 73: ("bc2",), 74: ("bc2",), 75: ("bc2",), 76: ("bc3",), 77: ("bc3",), 78: ("bc3",),
    79: ("bc4",), 80: ("bc4",), 82: ("bc5",), 83: ("bc5",), 87: ("bgra",), 90: ("bgra",), 91: ("bgra",),
    97: ("bc7",), 98: ("bc7",), 99: ("bc7",)}
_DXGI_FLOAT = {1: RGBA32F, 2: RGBA32F, 9: RGBA16F, 10: RGBA16F, 15: RG32F, 16: RG32F, 26: RG11B10F, 31: RGBA8_SNORM, 33: RG16F,
               34: RG16F, 39: R32F, 41: R32F, 51: RG8_SNORM, 53: R16F, 54: R16F, 63: R8_SNORM, 67: RGB9E5, 81: BC4_SNORM,
               84: BC5_SNORM, 94: BC6H_UF, 95: BC6H_UF, 96: BC6H_SF}
_DXGI_UNSUPPORTED = {3: "Rgba32Uint", 4: "Rgba32Sint", 12: "Rgba16Uint", 14: "Rgba16Sint", 17: "Rg32Uint", 18: "Rg32Sint",
                     30: "Rgba8Uint", 32: "Rgba8Sint", 36: "Rg16Uint", 38: "Rg16Sint", 40: "Depth32Float", 42: "R32Uint",
                     43: "R32Sint", 44: "Depth24PlusStencil8", 45: "Depth24PlusStencil8", 46: "Depth24Plus", 50: "Rg8Uint",
                     52: "Rg8Sint", 57: "R16Uint", 59: "R16Sint", 62: "R8Uint", 64: "R8Sint"}
_FAMILY = {"rgba": (RGBA8, RGBA8_SRGB), "bgra": (BGRA8, BGRA8_SRGB), "bc1": (BC1, BC1_SRGB), "bc2": (BC2, BC2_SRGB),
           "bc3": (BC3, BC3_SRGB), "bc7": (BC7, BC7_SRGB), "r": (R8, R8), "rg": (RG8, RG8), "bc4": (BC4, BC4), "bc5": (BC5, BC5)}


def map_dxgi_format(dxgi, srgb):
    if dxgi in _DXGI:
        lin, s = _FAMILY[_DXGI[dxgi][0]]
        return _pick(srgb, lin, s)
    if dxgi in _DXGI_FLOAT:
        return _DXGI_FLOAT[dxgi]
    if dxgi in _DXGI_UNSUPPORTED:
        raise TextureUnsupported(f"DXGI format {dxgi} ({_DXGI_UNSUPPORTED[dxgi]})")
    return None


def _d3d_format(flags, fourcc, bits, rm, gm, bm, am):
    """ddsfile::D3DFormat::try_from_pixel_format for the members map_d3d_format accepts.  Returns a name or None."""
    if flags & 0x4:  # DDPF_FOURCC: a four-character code, or a D3DFORMAT number for the float formats
        code = int.from_bytes(fourcc, "little")
        if code in (111, 112, 113, 114, 115, 116):
            return {111: "R16F", 112: "G16R16F", 113: "A16B16G16R16F", 114: "R32F", 115: "G32R32F", 116: "A32B32G32R32F"}[code]
        return {b"DXT1": "DXT1", b"DXT2": "DXT2", b"DXT3": "DXT3", b"DXT4": "DXT4", b"DXT5": "DXT5"}.get(fourcc)
    if flags & 0x40 and bits == 32 and flags & 0x1:  # DDPF_RGB | DDPF_ALPHAPIXELS
        if (rm, gm, bm, am) == (0xFF, 0xFF00, 0xFF0000, 0xFF000000):
            return "A8B8G8R8"
        if (rm, gm, bm, am) == (0xFF0000, 0xFF00, 0xFF, 0xFF000000):
            return "A8R8G8B8"
    if flags & 0x2 and bits == 8 and am == 0xFF:  # DDPF_ALPHA
        return "A8"
    return None


def map_d3d_format(name, srgb):
    if name in ("R16F", "G16R16F", "A16B16G16R16F", "R32F", "G32R32F", "A32B32G32R32F"):
        return {"R16F": R16F, "G16R16F": RG16F, "A16B16G16R16F": RGBA16F, "R32F": R32F, "G32R32F": RG32F, "A32B32G32R32F": RGBA32F}[name]
    fam = {"A8B8G8R8": "rgba", "A8R8G8B8": "bgra", "A8": "r", "DXT1": "bc1", "DXT2": "bc2", "DXT3": "bc2", "DXT4": "bc3",
           "DXT5": "bc3"}.get(name)
    if fam is None:
        return None
    lin, s = _FAMILY[fam]
    return _pick(srgb, lin, s)


def parse_dds(data, srgb):
    """ddsfile::Dds::read + the loader's checks.  None when `data` is not a DDS file (the loader then tries the image
    decoders)."""
    if len(data) < 128 or bytes(data[:4]) != b"DDS ":
        return None
    (size, flags, h, w, _pitch, _depth, mips) = struct.unpack_from("<7I", data, 4)
    if size != 124:
        return None
    (pf_size, pf_flags, fourcc, bits, rm, gm, bm, am) = struct.unpack_from("<II4s5I", data, 76)
    if pf_size != 32:
        return None
    offset = 128
    layers = 1
    if pf_flags & 0x4 and fourcc == b"DX10":
        if len(data) < 148:
            return None
        (dxgi, dim, misc, array_size, _misc2) = struct.unpack_from("<5I", data, 128)
        offset = 148
        fmt = map_dxgi_format(dxgi, srgb)
        if fmt is None:
            raise TextureLoadError("TextureBadDxgiFormat", f"DXGI format {dxgi}")
        if dim != 3 or misc & 0x4:
            raise TextureUnsupported("DDS cube maps / volume textures")
        layers = max(array_size, 1)
    else:
        name = _d3d_format(pf_flags, fourcc, bits, rm, gm, bm, am)
        if name is None:
            # the reference unwraps a None here (a panic); report it as a bad format instead
            raise TextureLoadError("TextureBadD3DFormat", f"pixel format flags {pf_flags:#x} fourcc {fourcc!r}")
        fmt = map_d3d_format(name, srgb)
        if fmt is None:
            raise TextureLoadError("TextureBadD3DFormat", name)
    levels = mips if flags & 0x20000 else 1  # DDSD_MIPMAPCOUNT: header.mip_map_count.unwrap_or(1)
    if levels == 0:
        raise TextureLoadError("TextureZeroLevels")
    out = []
    for k in range(levels):  # get_data(0): the first array layer, its levels back to back
        lw, lh = max(1, w >> k), max(1, h >> k)
        n = level_bytes(fmt, lw, lh)
        if offset + n > len(data):
            raise TextureLoadError("TextureTooManyLayers", "DDS data shorter than its header says")
        out.append(bytes(data[offset:offset + n]))
        offset += n
    del layers
    return {"format": fmt, "width": w, "height": h, "levels": out}


def generate_mips_allowed(fmt):
    """load_image generates a chain for single-level files only when the format is filterable AND a render attachment
    (the blit chain of util/mipmap.rs renders into it) under the default device features: the uncompressed 8-bit unorm
    formats, R16Float / Rg16Float / Rgba16Float and Rgb10a2Unorm; not BCn, snorm, 32-bit float (not filterable), 16-bit norm,
    Rg11b10Float or Rgb9e5Ufloat (not render attachments)."""
    return (not is_block_format(fmt) and not is_float_format(fmt)) or fmt in (R16F, RG16F, RGBA16F, RGB10A2)
