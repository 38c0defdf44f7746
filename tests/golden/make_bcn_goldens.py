#!/usr/bin/env python3
"""Generates tests/golden/bcn_blocks.npz: block-compressed texture data and what an INDEPENDENT decoder (Pillow's
DdsImagePlugin, Pillow 12.2.0 in the build container) makes of it.  The oracle's decoders (oracle/bcn.c) are pinned
against these vectors on the CPU; the HIP decode kernels are then compared with the oracle.

Per format: 32 x 32 texels (64 blocks) of uniformly random bytes, plus crafted cases -- for BC7 every mode with every
partition / rotation / index-selection value, for BC1 both orderings of the endpoints (punch-through), for BC3-5
both orderings of the alpha endpoints.

  python tests/golden/make_bcn_goldens.py        (writes tests/golden/bcn_blocks.npz)
"""
import io
import os
import struct

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
# name -> (R3N_TEXTURE_* id, FourCC or None, DXGI format or None, block bytes)
FORMATS = {"bc1": (6, b"DXT1", None, 8), "bc2": (8, b"DXT3", None, 16), "bc3": (10, b"DXT5", None, 16),
           "bc4": (12, b"BC4U", None, 8), "bc5": (13, b"BC5U", None, 16), "bc7": (14, None, 98, 16)}


def dds(w, h, fourcc, dxgi, data):
    hdr = struct.pack('<4sIIIIIII44x', b'DDS ', 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, len(data), 0, 1)
    pf = struct.pack('<II4sIIIII', 32, 0x4, fourcc or b'DX10', 0, 0, 0, 0, 0)
    out = hdr + pf + struct.pack('<IIIII', 0x1000, 0, 0, 0, 0)
    if not fourcc:
        out += struct.pack('<IIIII', dxgi, 3, 0, 1, 0)
    return out + data


def pillow_decode(name, w, h, data):
    _id, fourcc, dxgi, _bb = FORMATS[name]
    im = Image.open(io.BytesIO(dds(w, h, fourcc, dxgi, data)))
    im.load()
    a = np.asarray(im)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., 3] = 255
    if im.mode == "L":
        out[..., 0] = a
    elif im.mode == "RGB":   # BC5: Pillow fills blue itself; sampled BC5 textures read (r, g, 0, 1)
        out[..., 0] = a[..., 0]
        out[..., 1] = a[..., 1]
    else:
        out[...] = np.asarray(im.convert("RGBA"))
    return out


def bc7_crafted(rng):
    """One block per (mode, partition / rotation / index-selection) with random payload bits."""
    blocks = []
    modes = {0: (4, 0, 0), 1: (6, 0, 0), 2: (6, 0, 0), 3: (6, 0, 0), 4: (0, 2, 1), 5: (0, 2, 0), 6: (0, 0, 0), 7: (6, 0, 0)}
    for mode, (pb, rb, isb) in modes.items():
        for sel in range(1 << (pb + rb + isb)):
            v = int.from_bytes(rng.integers(0, 256, 16, dtype=np.uint8).tobytes(), "little")
            v &= ~((1 << (mode + 1 + pb + rb + isb)) - 1)
            v |= (1 << mode) | (sel << (mode + 1))
            blocks.append(v.to_bytes(16, "little"))
    blocks.append(bytes(16))  # reserved mode (no mode bit set)
    return b"".join(blocks)


def main():
    rng = np.random.default_rng(0xBC7)
    out = {}
    for name, (fid, _fc, _dx, bb) in FORMATS.items():
        data = rng.integers(0, 256, 64 * bb, dtype=np.uint8).tobytes()
        cases = [(32, 32, data)]
        if name == "bc7":
            crafted = bc7_crafted(rng)
            n = len(crafted) // 16
            pad = (-n) % 8
            crafted += bytes(16) * pad
            cases.append((32, 4 * ((n + pad) // 8), crafted))
        if name == "bc1":
            # equal endpoints and both orderings, every selector value
            blk = []
            for c0, c1 in ((0x1234, 0x1234), (0x0001, 0xFFFE), (0xFFFE, 0x0001), (0x8410, 0x8411)):
                blk.append(struct.pack('<HHI', c0, c1, 0xE4E4E4E4))
            blk += [blk[0]] * 4
            cases.append((32, 4, b"".join(blk)))
        if name in ("bc3", "bc4", "bc5"):
            blk = []
            for a0, a1 in ((10, 200), (200, 10), (77, 77), (0, 255), (255, 0), (1, 2), (254, 253), (128, 127)):
                sel = int.from_bytes(rng.integers(0, 256, 6, dtype=np.uint8).tobytes(), "little")
                alpha = bytes([a0, a1]) + sel.to_bytes(6, "little")
                if name == "bc3":
                    blk.append(alpha + rng.integers(0, 256, 8, dtype=np.uint8).tobytes())
                elif name == "bc4":
                    blk.append(alpha)
                else:
                    blk.append(alpha + alpha[::-1][:2][::-1] + rng.integers(0, 256, 6, dtype=np.uint8).tobytes())
            cases.append((32, 4, b"".join(blk)))
        for k, (w, h, d) in enumerate(cases):
            out[f"{name}_{k}_data"] = np.frombuffer(d, np.uint8)
            out[f"{name}_{k}_rgba"] = pillow_decode(name, w, h, d)
            out[f"{name}_{k}_meta"] = np.array([fid, w, h], np.uint32)
    np.savez_compressed(os.path.join(HERE, "bcn_blocks.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("rgba")})


if __name__ == "__main__":
    main()
