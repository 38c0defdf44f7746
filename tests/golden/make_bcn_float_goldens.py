#!/usr/bin/env python3
"""Generates tests/golden/bcn_float_blocks.npz: BC6H (unsigned / signed) and BC5 snorm blocks and what an INDEPENDENT
decoder (Pillow 12.2.0 DdsImagePlugin / BcnDecode) makes of them -- 8-bit images: BC6H as clamp(value, 0, 1) * 255
truncated, BC5 snorm as the signed 8-bit palette value + 128.

BC6H: for each of the 14 modes, the 96 blocks (of 8192 random ones) whose decoded values are least saturated at 8 bits --
random endpoints mostly decode above 1.0, where an 8-bit image says nothing -- plus one block of every reserved mode.
BC5 snorm: random blocks plus crafted endpoint pairs (both orderings, equal, -128).

  python tests/golden/make_bcn_float_goldens.py        (writes tests/golden/bcn_float_blocks.npz)
"""
import io
import os
import struct

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
# (mode bits value, mode bit count) in the order of the layout tables (tools/gen_bc6h_tables.py)
BC6H_MODES = [(0, 2), (1, 2), (2, 5), (6, 5), (10, 5), (14, 5), (18, 5), (22, 5), (26, 5), (30, 5), (3, 5), (7, 5), (11, 5), (15, 5)]
BC6H_RESERVED = [19, 23, 27, 31]


def dds(w, h, dxgi, data):
    hdr = struct.pack('<4sIIIIIII44x', b'DDS ', 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, len(data), 0, 1)
    pf = struct.pack('<II4sIIIII', 32, 0x4, b'DX10', 0, 0, 0, 0, 0)
    return hdr + pf + struct.pack('<IIIII', 0x1000, 0, 0, 0, 0) + struct.pack('<IIIII', dxgi, 3, 0, 1, 0) + data


def pillow_rgb(dxgi, w, h, data):
    im = Image.open(io.BytesIO(dds(w, h, dxgi, data)))
    im.load()
    assert im.mode == "RGB"
    return np.asarray(im).copy()


def main():
    rng = np.random.default_rng(0xBC6)
    out = {}
    for name, dxgi, fid in (("bc6h_uf", 95, 32), ("bc6h_sf", 96, 33)):
        keep = []
        for val, bits in BC6H_MODES:
            n = 8192
            blocks = rng.integers(0, 256, (n, 16), dtype=np.uint8)
            blocks[:, 0] = (blocks[:, 0] & (255 ^ ((1 << bits) - 1))) | val
            img = pillow_rgb(dxgi, 256, 4 * (n // 64), blocks.tobytes())
            per_block = img.reshape(n // 64, 4, 64, 4, 3).transpose(0, 2, 1, 3, 4).reshape(n, 48)
            score = ((per_block > 0) & (per_block < 255)).sum(axis=1)
            keep.append(blocks[np.argsort(-score, kind="stable")[:96]])
        res = rng.integers(0, 256, (len(BC6H_RESERVED), 16), dtype=np.uint8)
        res[:, 0] = (res[:, 0] & 0xE0) | np.array(BC6H_RESERVED, dtype=np.uint8)
        blocks = np.concatenate(keep + [res])
        pad = (-len(blocks)) % 8
        blocks = np.concatenate([blocks, np.zeros((pad, 16), np.uint8)])  # zeros = mode 00: a valid block
        w, h = 32, 4 * (len(blocks) // 8)
        out[f"{name}_data"] = blocks.reshape(-1)
        out[f"{name}_rgb8"] = pillow_rgb(dxgi, w, h, blocks.tobytes())
        out[f"{name}_meta"] = np.array([fid, w, h], np.uint32)
        mid = ((out[f"{name}_rgb8"] > 0) & (out[f"{name}_rgb8"] < 255)).mean()
        print(name, len(blocks), "blocks,", f"{mid:.2f} of the 8-bit values unsaturated")
    blocks = [rng.integers(0, 256, (64, 16), dtype=np.uint8)]
    for a0, a1 in ((10, 100), (100, 10), (77, 77), (0x80, 0x7F), (0x7F, 0x80), (0x81, 0x80), (0x80, 0x81), (0xF0, 0x10), (0x10, 0xF0), (0, 0),
                   (0x80, 0x80), (0x81, 0x81), (1, 0xFF), (0xFF, 1), (0x7F, 0x7F), (0x80, 0)):
        b = rng.integers(0, 256, 16, dtype=np.uint8)
        b[0], b[1], b[8], b[9] = a0, a1, a1, a0
        blocks.append(b[None])
    blocks = np.concatenate(blocks)
    w, h = 32, 4 * (len(blocks) // 8)
    out["bc5s_data"] = blocks.reshape(-1)
    out["bc5s_rgb8"] = pillow_rgb(84, w, h, blocks.tobytes())
    out["bc5s_meta"] = np.array([31, w, h], np.uint32)
    np.savez_compressed(os.path.join(HERE, "bcn_float_blocks.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("rgb8")})


if __name__ == "__main__":
    main()
