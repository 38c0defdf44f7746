"""Row N2, texture formats: the oracle's block decoders against an independent decoder's output (committed vectors,
tests/golden/bcn_blocks.npz from Pillow 12.2 -- tests/golden/make_bcn_goldens.py), and the KTX2 / DDS container
readers against files written here (and against Pillow's DDS reader where it is installed).  CPU only."""
import io
import os
import struct

import numpy as np
import pytest

from oracle import lib as olib
from rend3_amd import containers as C

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "bcn_blocks.npz"))
CASES = sorted({k.rsplit("_", 1)[0] for k in GOLD.files})


def oracle_decode(fmt, w, h, data):
    c = olib.get().c
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
    assert c.r3o_texture_level_bytes(fmt, w, h) == len(src)
    out = np.zeros((h, w, 4), dtype=np.uint8)
    assert c.r3o_texture_decode_level(fmt, w, h, src.ctypes.data, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("case", CASES)
def test_oracle_block_decoders_match_independent_decoder(case):
    fmt, w, h = (int(v) for v in GOLD[case + "_meta"])
    data = GOLD[case + "_data"].tobytes()
    got = oracle_decode(fmt, w, h, data)
    want = GOLD[case + "_rgba"].copy()
    if case.startswith("bc7"):
        # reserved mode (no mode bit in the first byte): the specification decodes the block to zeros in every channel
        # (Khronos Data Format Specification 1.3, BPTC: "mode 8 ... returns 0"); the independent decoder leaves
        # alpha at 255 there.  The oracle follows the specification.
        blocks = np.frombuffer(data, dtype=np.uint8).reshape(-1, 16)
        reserved = np.nonzero(blocks[:, 0] == 0)[0]
        assert len(reserved) >= 1
        for b in reserved:
            by, bx = divmod(int(b), w // 4)
            assert (got[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] == 0).all()
            want[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = 0
    assert np.array_equal(got, want), f"{case}: {(got != want).any(axis=2).sum()} texels differ"


def test_bc7_crafted_vectors_cover_every_mode_and_partition():
    blocks = GOLD["bc7_1_data"].reshape(-1, 16)
    seen = set()
    for b in blocks:
        if b[0] == 0:
            continue
        mode = (int(b[0]) & -int(b[0])).bit_length() - 1
        v = int.from_bytes(b.tobytes(), "little") >> (mode + 1)
        pb = {0: 4, 1: 6, 2: 6, 3: 6, 4: 3, 5: 2, 6: 0, 7: 6}[mode]
        seen.add((mode, v & ((1 << pb) - 1)))
    for mode, n in ((0, 16), (1, 64), (2, 64), (3, 64), (4, 8), (5, 4), (6, 1), (7, 64)):
        assert all((mode, k) in seen for k in range(n)), mode


def test_partial_blocks_and_uncompressed_expansion():
    rng = np.random.default_rng(5)
    # 6 x 5 texels = 2 x 2 blocks, only the top-left 6 x 5 texels are kept
    data = rng.integers(0, 256, 4 * 16, dtype=np.uint8).tobytes()
    full = oracle_decode(C.BC3, 8, 8, data)
    part = oracle_decode(C.BC3, 6, 5, data)
    assert np.array_equal(part, full[:5, :6])
    r8 = rng.integers(0, 256, 15, dtype=np.uint8)
    out = oracle_decode(C.R8, 5, 3, r8.tobytes())
    assert np.array_equal(out[..., 0].reshape(-1), r8) and (out[..., 1:3] == 0).all() and (out[..., 3] == 255).all()
    rg = rng.integers(0, 256, 30, dtype=np.uint8)
    out = oracle_decode(C.RG8, 5, 3, rg.tobytes())
    assert np.array_equal(out[..., :2].reshape(-1), rg) and (out[..., 2] == 0).all() and (out[..., 3] == 255).all()
    bgra = rng.integers(0, 256, 60, dtype=np.uint8)
    out = oracle_decode(C.BGRA8_SRGB, 5, 3, bgra.tobytes())
    assert np.array_equal(out.reshape(-1, 4), bgra.reshape(-1, 4)[:, [2, 1, 0, 3]])


# ------------------------------------------------------------------------------------------------ containers
def write_ktx2(vk, w, h, levels, layers=0, scheme=0):
    n = len(levels)
    index = 80
    data_off = index + 24 * n
    out = bytearray(C.KTX2_MAGIC + struct.pack("<9I", vk, 1, w, h, 0, layers, 1, n, scheme) + struct.pack("<4I2Q", 0, 0, 0, 0, 0, 0))
    # levels are stored smallest first in the file, the index is by level number (KTX 2.0 specification 3.9)
    offs = {}
    cur = data_off
    for k in reversed(range(n)):
        cur = (cur + 15) & ~15
        offs[k] = cur
        cur += len(levels[k])
    for k in range(n):
        out += struct.pack("<3Q", offs[k], len(levels[k]), len(levels[k]))
    body = bytearray(cur - data_off)
    for k in range(n):
        body[offs[k] - data_off: offs[k] - data_off + len(levels[k])] = levels[k]
    return bytes(out + body)


def write_dds(w, h, levels, fourcc=None, dxgi=None, masks=None):
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | (0x20000 if len(levels) > 1 else 0)
    hdr = struct.pack("<4s7I44x", b"DDS ", 124, flags, h, w, len(levels[0]), 0, len(levels))
    if masks:
        pf = struct.pack("<II4s5I", 32, masks[0], b"\0\0\0\0", masks[1], *masks[2:])
    else:
        pf = struct.pack("<II4s5I", 32, 0x4, fourcc or b"DX10", 0, 0, 0, 0, 0)
    out = hdr + pf + struct.pack("<5I", 0x1000, 0, 0, 0, 0)
    if not fourcc and not masks:
        out += struct.pack("<5I", dxgi, 3, 0, 1, 0)
    return out + b"".join(levels)


def chain(fmt, w, h, n, rng):
    return [rng.integers(0, 256, C.level_bytes(fmt, max(1, w >> k), max(1, h >> k)), dtype=np.uint8).tobytes() for k in range(n)]


def test_ktx2_reader_and_format_map():
    rng = np.random.default_rng(11)
    for vk, srgb, fmt in ((145, True, C.BC7_SRGB), (146, False, C.BC7), (131, False, C.BC1), (134, True, C.BC1_SRGB), (137, False, C.BC3),
                          (139, False, C.BC4), (141, True, C.BC5), (37, True, C.RGBA8_SRGB), (43, False, C.RGBA8), (44, True, C.BGRA8_SRGB),
                          (9, False, C.R8), (16, False, C.RG8), (135, True, C.BC2_SRGB)):
        levels = chain(fmt, 20, 12, 3, rng)
        got = C.parse_ktx2(write_ktx2(vk, 20, 12, levels), srgb)
        assert got["format"] == fmt and (got["width"], got["height"]) == (20, 12) and got["levels"] == levels, vk
    assert C.parse_ktx2(b"not a ktx2 file" * 10, False) is None
    for vk, srgb, kind in ((9, True, "TextureBadKxt2Format"), (16, True, "TextureBadKxt2Format"), (23, False, "TextureBadKxt2Format"),
                           (70, False, "TextureBadKxt2Format"), (95, False, "TextureUnsupported"), (126, False, "TextureUnsupported"),
                           (157, True, "TextureUnsupported"), (99, False, "TextureUnsupported")):
        with pytest.raises(C.TextureLoadError) as e:
            C.parse_ktx2(write_ktx2(vk, 8, 8, [bytes(64)]), srgb)
        assert e.value.kind == kind, (vk, e.value.kind)
    with pytest.raises(C.TextureLoadError) as e:
        C.parse_ktx2(write_ktx2(145, 8, 8, []), False)
    assert e.value.kind == "TextureZeroLevels"
    with pytest.raises(C.TextureLoadError) as e:
        C.parse_ktx2(write_ktx2(145, 8, 8, [bytes(64)], layers=2), False)
    assert e.value.kind == "TextureTooManyLayers"
    with pytest.raises(C.TextureLoadError) as e:
        C.parse_ktx2(write_ktx2(145, 8, 8, [bytes(64)], scheme=2), False)
    assert e.value.kind == "TextureUnsupported"


def test_dds_reader_and_format_maps():
    rng = np.random.default_rng(12)
    for kw, srgb, fmt in ((dict(fourcc=b"DXT1"), True, C.BC1_SRGB), (dict(fourcc=b"DXT3"), False, C.BC2), (dict(fourcc=b"DXT5"), False, C.BC3),
                          (dict(dxgi=98), True, C.BC7_SRGB), (dict(dxgi=99), False, C.BC7), (dict(dxgi=80), False, C.BC4), (dict(dxgi=83), True, C.BC5),
                          (dict(dxgi=71), False, C.BC1), (dict(dxgi=28), True, C.RGBA8_SRGB), (dict(dxgi=87), False, C.BGRA8), (dict(dxgi=61), True, C.R8),
                          (dict(masks=(0x41, 32, 0xFF, 0xFF00, 0xFF0000, 0xFF000000)), True, C.RGBA8_SRGB),
                          (dict(masks=(0x41, 32, 0xFF0000, 0xFF00, 0xFF, 0xFF000000)), False, C.BGRA8)):
        levels = chain(fmt, 16, 8, 4, rng)
        got = C.parse_dds(write_dds(16, 8, levels, **kw), srgb)
        assert got["format"] == fmt and (got["width"], got["height"]) == (16, 8) and got["levels"] == levels, kw
    one = C.parse_dds(write_dds(8, 8, [bytes(64)], dxgi=98), False)
    assert len(one["levels"]) == 1
    assert C.parse_dds(b"DDSx" + bytes(200), False) is None
    for kw, kind in ((dict(dxgi=12), "TextureUnsupported"), (dict(dxgi=88), "TextureBadDxgiFormat"), (dict(dxgi=40), "TextureUnsupported"),
                     (dict(dxgi=24), "TextureBadDxgiFormat"), (dict(dxgi=11), "TextureBadDxgiFormat"), (dict(dxgi=56), "TextureBadDxgiFormat"),
                     (dict(fourcc=b"ATI2"), "TextureBadD3DFormat"), (dict(masks=(0x40, 24, 0xFF0000, 0xFF00, 0xFF, 0)), "TextureBadD3DFormat")):
        with pytest.raises(C.TextureLoadError) as e:
            C.parse_dds(write_dds(8, 8, [bytes(256)], **kw), False)
        assert e.value.kind == kind, (kw, e.value.kind)
    with pytest.raises(C.TextureLoadError) as e:
        C.parse_dds(write_dds(64, 64, [bytes(64)], dxgi=98), False)  # header promises more data than the file holds
    assert e.value.kind == "TextureTooManyLayers"


def test_dds_files_agree_with_independent_reader():
    """The same DDS bytes through this reader + the oracle's decoders, and through Pillow's DdsImagePlugin."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(13)
    for kw, fmt in ((dict(fourcc=b"DXT1"), C.BC1), (dict(fourcc=b"DXT5"), C.BC3), (dict(dxgi=98), C.BC7), (dict(dxgi=83), C.BC5)):
        data = chain(fmt, 24, 16, 1, rng)
        if fmt == C.BC7:  # keep reserved-mode blocks out (see the decoder test)
            a = np.frombuffer(data[0], dtype=np.uint8).copy().reshape(-1, 16)
            a[a[:, 0] == 0, 0] = 1
            data = [a.tobytes()]
        blob = write_dds(24, 16, data, **kw)
        parsed = C.parse_dds(blob, False)
        mine = oracle_decode(parsed["format"], parsed["width"], parsed["height"], parsed["levels"][0])
        im = PIL.open(io.BytesIO(blob))
        im.load()
        assert im.size == (24, 16)
        theirs = np.asarray(im.convert("RGBA"))
        if fmt == C.BC5:
            assert np.array_equal(mine[..., :2], theirs[..., :2])
        else:
            assert np.array_equal(mine, theirs)


# ------------------------------------------------------------------------------------------------ float-decoded formats
FLOAT_GOLD = np.load(os.path.join(HERE, "golden", "bcn_float_blocks.npz"))


def oracle_decode_f32(fmt, w, h, data):
    c = olib.get().c
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
    assert c.r3o_texture_is_float(fmt) == 1
    assert c.r3o_texture_level_bytes(fmt, w, h) == len(src) == C.level_bytes(fmt, w, h)
    out = np.zeros((h, w, 4), dtype=np.float32)
    assert c.r3o_texture_decode_level_f32(fmt, w, h, src.ctypes.data, out.ctypes.data) == 0
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def small_ufloat(v, mant_bits):
    """Unsigned small float (5 exponent bits, bias 15) -> f64, from the format's definition."""
    v = v.astype(np.int64)
    e, m = v >> mant_bits, v & ((1 << mant_bits) - 1)
    normal = (1.0 + m / float(1 << mant_bits)) * np.exp2((e - 15).astype(np.float64))
    denorm = m / float(1 << mant_bits) * 2.0 ** -14
    out = np.where(e == 0, denorm, normal)
    out = np.where((e == 31) & (m == 0), np.inf, out)
    return np.where((e == 31) & (m != 0), np.nan, out)


def numpy_decode_f32(fmt, w, h, data):
    """An independent restatement on numpy: IEEE conversions by numpy's float16, integer ratios evaluated in f64 and rounded
    once to f32 (a correctly rounded f32 division of two exactly representable integers gives the same value)."""
    raw = np.frombuffer(data, dtype=np.uint8)
    n = w * h
    out = np.zeros((n, 4), dtype=np.float64)
    out[:, 3] = 1.0
    if fmt in (C.R8_SNORM, C.RG8_SNORM, C.RGBA8_SNORM):
        ch = {C.R8_SNORM: 1, C.RG8_SNORM: 2, C.RGBA8_SNORM: 4}[fmt]
        out[:, :ch] = np.maximum(raw.view(np.int8).reshape(n, ch).astype(np.float64) / 127.0, -1.0)
    elif fmt in (C.R16F, C.RG16F, C.RGBA16F):
        ch = {C.R16F: 1, C.RG16F: 2, C.RGBA16F: 4}[fmt]
        res = np.zeros((n, 4), dtype=np.float32)
        res[:, 3] = 1.0
        res[:, :ch] = raw.view(np.float16).reshape(n, ch).astype(np.float32)
        return res.reshape(h, w, 4)
    elif fmt in (C.R32F, C.RG32F, C.RGBA32F):
        ch = {C.R32F: 1, C.RG32F: 2, C.RGBA32F: 4}[fmt]
        res = np.zeros((n, 4), dtype=np.float32)
        res[:, 3] = 1.0
        res[:, :ch] = raw.view(np.float32).reshape(n, ch)
        return res.reshape(h, w, 4)
    elif fmt == C.RGBA16_UNORM:
        out[:] = raw.view(np.uint16).reshape(n, 4).astype(np.float64) / 65535.0
    elif fmt == C.RGBA16_SNORM:
        out[:] = np.maximum(raw.view(np.int16).reshape(n, 4).astype(np.float64) / 32767.0, -1.0)
    elif fmt == C.RGB10A2:
        v = raw.view(np.uint32).astype(np.int64)
        out[:, 0], out[:, 1], out[:, 2], out[:, 3] = (v & 1023) / 1023.0, ((v >> 10) & 1023) / 1023.0, ((v >> 20) & 1023) / 1023.0, (v >> 30) / 3.0
    elif fmt == C.RG11B10F:
        v = raw.view(np.uint32)
        out[:, 0], out[:, 1], out[:, 2] = small_ufloat(v & 2047, 6), small_ufloat((v >> 11) & 2047, 6), small_ufloat(v >> 22, 5)
    elif fmt == C.RGB9E5:
        v = raw.view(np.uint32).astype(np.int64)
        sc = np.exp2(((v >> 27) - 24).astype(np.float64))
        out[:, 0], out[:, 1], out[:, 2] = (v & 511) * sc, ((v >> 9) & 511) * sc, ((v >> 18) & 511) * sc
    else:
        raise AssertionError(fmt)
    return out.astype(np.float32).reshape(h, w, 4)


UNCOMPRESSED_FLOAT = [C.R8_SNORM, C.RG8_SNORM, C.RGBA8_SNORM, C.R16F, C.RG16F, C.RGBA16F, C.R32F, C.RG32F, C.RGBA32F, C.RGBA16_UNORM,
                      C.RGBA16_SNORM, C.RGB10A2, C.RG11B10F, C.RGB9E5]


def float_format_level(fmt, w, h, rng):
    """Random bytes, with the special encodings planted: every 8-bit / 16-bit code of the narrow formats, -128 / -32768,
    infinities, NaNs, subnormals, zero exponents."""
    data = rng.integers(0, 256, C.level_bytes(fmt, w, h), dtype=np.uint8)
    if fmt in (C.R8_SNORM, C.RG8_SNORM, C.RGBA8_SNORM):
        data[:256] = np.arange(256, dtype=np.uint8)
    elif fmt in (C.R16F, C.RG16F, C.RGBA16F, C.RGBA16_UNORM, C.RGBA16_SNORM):
        sp = np.array([0, 0x8000, 1, 0x8001, 0x03FF, 0x0400, 0x7BFF, 0x7C00, 0xFC00, 0x7C01, 0x7E00, 0xFFFF, 0x7FFF, 0x8001, 0x3C00, 0xBC00], dtype=np.uint16)
        data[:32] = sp.view(np.uint8)
    elif fmt in (C.RG11B10F, C.RGB9E5, C.RGB10A2):
        sp = np.array([0, 0xFFFFFFFF, 0x7C0 | (0x7C1 << 11) | (0x3E0 << 22), 0x7FF, 1 | (1 << 11) | (1 << 22), 0x3F | (0x40 << 11) | (0x1F << 22),
                       31 << 27, 0xF8000000 | 0x1FF, 0x07FFFFFF, 1 << 27], dtype=np.uint32)
        data[:40] = sp.view(np.uint8)
    return data.tobytes()


@pytest.mark.parametrize("fmt", UNCOMPRESSED_FLOAT)
def test_oracle_float_formats_match_numpy_restatement(fmt):
    rng = np.random.default_rng(100 + fmt)
    w, h = 37, 23
    data = float_format_level(fmt, w, h, rng)
    got, want = oracle_decode_f32(fmt, w, h, data), numpy_decode_f32(fmt, w, h, data)
    # bit patterns: signed zeros, subnormals and (where numpy carries them: binary16 / binary32) NaN payloads included
    if fmt == C.RG11B10F:  # the f64 restatement only says "NaN"
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).any()
        got, want = np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0)
    assert np.array_equal(bits(got), bits(want)), f"{C.FORMAT_NAMES[fmt]}: {(bits(got) != bits(want)).sum()} values differ"


def bc6h_half_levels(signed, variant, w, h, data):
    c = olib.get().c
    src = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8))
    out = np.zeros((h, w, 3), dtype=np.uint16)
    assert c.r3o_bc6h_decode_level_half(signed, variant, w, h, src.ctypes.data, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("name", ["bc6h_uf", "bc6h_sf"])
def test_bc6h_decoder_matches_independent_decoder(name):
    """Every texel of every mode (96 blocks per mode, chosen for unsaturated 8-bit values) under the independent decoder's
    two arithmetic deviations (oracle/bcn.c r3o_bc6h_decode_level_half); the specification variant is what the f32 decode
    returns, differs from it only through those two expressions, and decodes reserved modes to zeros."""
    fmt, w, h = (int(v) for v in FLOAT_GOLD[name + "_meta"])
    data = FLOAT_GOLD[name + "_data"].tobytes()
    signed = 1 if name == "bc6h_sf" else 0
    half = bc6h_half_levels(signed, 1, w, h, data)
    value = half.view(np.float16).astype(np.float32)
    with np.errstate(invalid="ignore"):
        as8 = (np.clip(np.nan_to_num(value, nan=0.0), 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)  # the decoder's 8-bit image
    want = FLOAT_GOLD[name + "_rgb8"]
    assert ((want > 0) & (want < 255)).mean() > 0.6  # the vectors do exercise the value range an 8-bit image resolves
    assert np.array_equal(as8, want), f"{(as8 != want).any(axis=2).sum()} texels differ"
    # the specification's variant, through the public decode
    spec = bc6h_half_levels(signed, 0, w, h, data)
    f32 = oracle_decode_f32(fmt, w, h, data)
    assert np.array_equal(bits(f32[..., :3]), bits(spec.view(np.float16).astype(np.float32)))
    assert (f32[..., 3] == 1.0).all()
    blocks = np.frombuffer(data, dtype=np.uint8).reshape(-1, 16)
    reserved = np.nonzero(np.isin(blocks[:, 0] & 31, [19, 23, 27, 31]))[0]
    assert len(reserved) == 4
    for b in reserved:
        by, bx = divmod(int(b), w // 4)
        assert (f32[4 * by:4 * by + 4, 4 * bx:4 * bx + 4, :3] == 0).all()
    if not signed:
        # rounding term only: the two variants are at most one binary16 step apart, and differ somewhere
        d = spec.astype(np.int32) - half.astype(np.int32)
        assert d.min() >= 0 and d.max() == 1


def test_bc6h_mode_coverage_of_the_vectors():
    blocks = FLOAT_GOLD["bc6h_uf_data"].reshape(-1, 16)
    two = blocks[:, 0] & 3
    five = blocks[:, 0] & 31
    for m in (0, 1):
        assert (two == m).sum() >= 96
    for m in (2, 6, 10, 14, 18, 22, 26, 30, 3, 7, 11, 15):
        assert (five == m).sum() >= 96, m
    # two-region modes: all 32 partitions occur (bits 77..81)
    v = np.array([int.from_bytes(b.tobytes(), "little") for b in blocks], dtype=object)
    parts = {int((x >> 77) & 31) for x, t in zip(v, two) if t < 2}
    assert parts == set(range(32))


def rgtc_signed_numpy(block8):
    """(n, 8) u8 RGTC signed blocks -> the palette index of each texel, endpoints, ordering: the format's definition on numpy."""
    a0, a1 = block8[:, 0].view(np.int8).astype(np.int64), block8[:, 1].view(np.int8).astype(np.int64)
    sel = np.zeros(len(block8), dtype=np.uint64)
    for i in range(6):
        sel |= block8[:, 2 + i].astype(np.uint64) << np.uint64(8 * i)
    k = np.stack([(sel >> np.uint64(3 * i)) & np.uint64(7) for i in range(16)], axis=1).astype(np.int64)
    return a0, a1, k


def test_bc5_snorm_decoder():
    fmt, w, h = (int(v) for v in FLOAT_GOLD["bc5s_meta"])
    data = FLOAT_GOLD["bc5s_data"]
    got = oracle_decode_f32(fmt, w, h, data.tobytes())
    pil = FLOAT_GOLD["bc5s_rgb8"].astype(np.int64) - 128
    blocks = data.reshape(-1, 16)
    for ch in range(2):
        a0, a1, k = rgtc_signed_numpy(np.ascontiguousarray(blocks[:, 8 * ch:8 * ch + 8]))
        f0 = np.maximum(a0.astype(np.float32) / np.float32(127.0), np.float32(-1.0))[:, None]
        f1 = np.maximum(a1.astype(np.float32) / np.float32(127.0), np.float32(-1.0))[:, None]
        kf = k.astype(np.float32)
        six = ((np.float32(8.0) - kf) * f0 + (kf - np.float32(1.0)) * f1) / np.float32(7.0)
        four = ((np.float32(6.0) - kf) * f0 + (kf - np.float32(1.0)) * f1) / np.float32(5.0)
        four = np.where(k == 6, np.float32(-1.0), np.where(k == 7, np.float32(1.0), four))
        want = np.where(k == 0, f0, np.where(k == 1, f1, np.where((a0 > a1)[:, None], six, four))).astype(np.float32)
        # the independent decoder's integer palette (flooring division), same selectors and ordering
        i6 = np.floor(((8 - k) * a0[:, None] + (k - 1) * a1[:, None]) / 7.0)
        i4 = np.floor(((6 - k) * a0[:, None] + (k - 1) * a1[:, None]) / 5.0)
        i4 = np.where(k == 6, -128, np.where(k == 7, 127, i4))
        theirs = np.where(k == 0, a0[:, None], np.where(k == 1, a1[:, None], np.where((a0 > a1)[:, None], i6, i4)))
        img_want = np.zeros((h, w), np.float32)
        img_theirs = np.zeros((h, w), np.int64)
        for b in range(len(blocks)):
            by, bx = divmod(b, w // 4)
            img_want[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = want[b].reshape(4, 4)
            img_theirs[4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = theirs[b].reshape(4, 4)
        assert np.array_equal(img_theirs, pil[..., ch])           # the restatement reads the blocks like the independent decoder
        assert np.array_equal(bits(got[..., ch]), bits(img_want))  # and the oracle is the restatement with the f32 palette
        assert np.abs(got[..., ch] * 127.0 - pil[..., ch]).max() < 2.0
    assert (got[..., 2] == 0).all() and (got[..., 3] == 1).all()
    # BC4 snorm = the red half alone
    red = np.ascontiguousarray(blocks[:, :8]).tobytes()
    r = oracle_decode_f32(C.BC4_SNORM, w, h, red)
    assert np.array_equal(bits(r[..., 0]), bits(got[..., 0])) and (r[..., 1:3] == 0).all() and (r[..., 3] == 1).all()


def test_float_format_container_maps():
    rng = np.random.default_rng(21)
    for vk, fmt in ((10, C.R8_SNORM), (17, C.RG8_SNORM), (38, C.RGBA8_SNORM), (64, C.RGB10A2), (76, C.R16F), (83, C.RG16F), (91, C.RGBA16_UNORM),
                    (92, C.RGBA16_SNORM), (97, C.RGBA16F), (100, C.R32F), (103, C.RG32F), (109, C.RGBA32F), (122, C.RG11B10F), (123, C.RGB9E5),
                    (140, C.BC4_SNORM), (142, C.BC5_SNORM), (143, C.BC6H_UF), (144, C.BC6H_SF)):
        for srgb in (False, True):  # no sRGB variants: the flag is ignored (rend3-gltf/src/lib.rs:1204-1330)
            levels = chain(fmt, 12, 20, 3, rng)
            got = C.parse_ktx2(write_ktx2(vk, 12, 20, levels), srgb)
            assert got["format"] == fmt and got["levels"] == levels, vk
        # chains are generated for the filterable render-target formats only (rend3-gltf/src/lib.rs:1031-1036)
        assert C.is_float_format(fmt) and C.generate_mips_allowed(fmt) == (fmt in (C.R16F, C.RG16F, C.RGBA16F, C.RGB10A2))
    for kw, fmt in ((dict(dxgi=2), C.RGBA32F), (dict(dxgi=10), C.RGBA16F), (dict(dxgi=16), C.RG32F), (dict(dxgi=26), C.RG11B10F),
                    (dict(dxgi=31), C.RGBA8_SNORM), (dict(dxgi=34), C.RG16F), (dict(dxgi=41), C.R32F), (dict(dxgi=51), C.RG8_SNORM),
                    (dict(dxgi=54), C.R16F), (dict(dxgi=63), C.R8_SNORM), (dict(dxgi=67), C.RGB9E5), (dict(dxgi=81), C.BC4_SNORM),
                    (dict(dxgi=84), C.BC5_SNORM), (dict(dxgi=94), C.BC6H_UF), (dict(dxgi=95), C.BC6H_UF), (dict(dxgi=96), C.BC6H_SF),
                    (dict(fourcc=struct.pack("<I", 111)), C.R16F), (dict(fourcc=struct.pack("<I", 113)), C.RGBA16F),
                    (dict(fourcc=struct.pack("<I", 114)), C.R32F), (dict(fourcc=struct.pack("<I", 116)), C.RGBA32F)):
        levels = chain(fmt, 16, 8, 2, rng)
        got = C.parse_dds(write_dds(16, 8, levels, **kw), True)
        assert got["format"] == fmt and got["levels"] == levels, kw
    assert not C.is_float_format(C.BC7) and C.generate_mips_allowed(C.RG8)


def test_bc6h_dds_file_agrees_with_independent_reader():
    """A BC6H DDS file through this reader + the oracle, and through Pillow's plugin, on the committed unsaturated blocks."""
    PIL = pytest.importorskip("PIL.Image")
    fmt, w, h = (int(v) for v in FLOAT_GOLD["bc6h_uf_meta"])
    data = FLOAT_GOLD["bc6h_uf_data"].tobytes()
    blob = write_dds(w, h, [data], dxgi=95)
    parsed = C.parse_dds(blob, False)
    assert parsed["format"] == C.BC6H_UF
    mine = oracle_decode_f32(parsed["format"], w, h, parsed["levels"][0])
    im = PIL.open(io.BytesIO(blob))
    im.load()
    theirs = np.asarray(im).astype(np.float32) / 255.0
    # the specification's rounding term moves a value by at most one binary16 step: below one 8-bit step here
    assert np.abs(np.clip(mine[..., :3], 0, 1) - theirs).max() < 1.5 / 255.0


def test_half_rounding_matches_numpy():
    """f32 -> binary16, round to nearest even (the render-target write of a generated level): every exponent range, ties,
    the subnormal and overflow boundaries, both signs -- against numpy's conversion."""
    c = olib.get().c
    rng = np.random.default_rng(3)
    special = np.array([0, 0x80000000, 0x33000000, 0x33000001, 0x337FFFFF, 0x33800000, 0x38800000, 0x387FFFFF, 0x387FE000, 0x387FF000,
                        0x477FE000, 0x477FEFFF, 0x477FF000, 0x47800000, 0x7F800000, 0xFF800000, 0x3F800000, 0x3F801000, 0x3F803000,
                        0x3F802FFF, 0x3F801001, 0x38000000, 0x37FFFFFF], dtype=np.uint32)
    sub = rng.integers(0x33000000, 0x38800000, 20000, dtype=np.uint64).astype(np.uint32)
    ties = ((rng.integers(0x38800000 >> 13, 0x47800000 >> 13, 20000, dtype=np.uint64) << 13) | 0x1000).astype(np.uint32)
    bits32 = np.concatenate([rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32), special, sub, sub | np.uint32(0x80000000), ties])
    f = bits32.view(np.float32)
    keep = ~np.isnan(f)
    with np.errstate(over="ignore"):
        want = f[keep].astype(np.float16).view(np.uint16)
    got = np.array([c.r3o_float_to_half(float(x)) for x in f[keep]], dtype=np.uint16)
    assert np.array_equal(got, want), f"{(got != want).sum()} differ"


def test_generated_float_chain_is_the_box_average_rounded_to_the_format():
    """Power-of-two extents: the Linear / ClampToEdge blit at texel centres is ((a + b) / 2 + (c + d) / 2) / 2 in the blit's
    operation order; each level is rounded to the format before the next one reads it."""
    c = olib.get().c
    rng = np.random.default_rng(4)
    w, h = 16, 8
    lvl0 = (rng.random((h, w, 4)) * 8.0 - 4.0).astype(np.float16)
    mips = 5
    chain = np.zeros((sum(max(1, w >> k) * max(1, h >> k) for k in range(mips)), 4), dtype=np.float32)
    chain[: w * h] = lvl0.astype(np.float32).reshape(-1, 4)
    assert c.r3o_generate_mips_f32(C.RGBA16F, w, h, mips, chain.ctypes.data) == 0
    src, at = lvl0.astype(np.float32), w * h
    half = np.float32(0.5)
    for k in range(1, mips):
        sh, sw = src.shape[:2]
        dh, dw = max(1, sh // 2), max(1, sw // 2)
        if sh >= 2 and sw >= 2:
            top = src[0::2, 0::2] * half + src[0::2, 1::2] * half
            bot = src[1::2, 0::2] * half + src[1::2, 1::2] * half
            dst = top * half + bot * half
        else:  # one texel high: both rows clamp to it (weights 0.5 / 0.5 of the same value)
            row = src[0:1, 0::2] * half + src[0:1, 1::2] * half
            dst = row * half + row * half
        dst = dst.astype(np.float32).astype(np.float16).astype(np.float32)
        got = chain[at: at + dw * dh].reshape(dh, dw, 4)
        assert np.array_equal(got.view(np.uint32), dst.view(np.uint32)), k
        src, at = dst, at + dw * dh
    # Rgb10a2Unorm: codes survive a 1 x 1 -> chain of length 1, and a generated level holds exact n / 1023 (n / 3) values
    one = np.zeros((5, 4), dtype=np.float32)
    one[:4] = np.array([[0.1, 0.5, 0.9, 1.0], [0.3, 0.25, 0.0, 0.0], [0.7, 0.125, 1.0, 1.0], [0.2, 0.75, 0.5, 0.34]], dtype=np.float32)
    assert c.r3o_generate_mips_f32(C.RGB10A2, 2, 2, 2, one.ctypes.data) == 0
    rgb = one[4, :3].astype(np.float64) * 1023.0
    assert np.abs(rgb - np.rint(rgb)).max() < 1e-3 and abs(one[4, 3] * 3.0 - round(float(one[4, 3]) * 3.0)) < 1e-6
