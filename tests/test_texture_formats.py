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
                           (70, False, "TextureBadKxt2Format"), (143, False, "TextureUnsupported"), (97, False, "TextureUnsupported"),
                           (157, True, "TextureUnsupported"), (140, False, "TextureUnsupported")):
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
    for kw, kind in ((dict(dxgi=95), "TextureUnsupported"), (dict(dxgi=88), "TextureBadDxgiFormat"), (dict(dxgi=10), "TextureUnsupported"),
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
