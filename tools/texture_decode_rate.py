#!/usr/bin/env python3
"""GPU box: one 4096 x 4096 level per block format through r3n_textures_write_encoded (csrc/texture_decode.hip), to be
run under `rocprofv3 --kernel-trace --stats`; tools/texture_decode_rate.py --report <kernel_trace.csv> then prints the
decode kernel's HBM rate per format (algorithmic bytes: block bytes in + 64 B of RGBA8 out per block)."""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

FORMATS = [(6, "BC1", 8), (10, "BC3", 16), (12, "BC4", 8), (13, "BC5", 16), (14, "BC7", 16)]
W = H = 4096

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_decode_blocks" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    blocks = (W // 4) * (H // 4)
    for (fid, name, bb), r in zip(FORMATS, rows):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gb = blocks * (bb + 64) / 1e9
        print(f"{name}: {us:8.1f} us  {gb / (us * 1e-6) / 1e3:6.2f} TB/s  ({blocks * 16 / us:.0f} Mtexel/s)")
    sys.exit(0)

import rend3_amd as r3
rng = np.random.default_rng(3)
for fid, name, bb in FORMATS:
    p = r3.Renderer()
    data = rng.integers(0, 256, (W // 4) * (H // 4) * bb, dtype=np.uint8)
    if name == "BC7":
        data.reshape(-1, 16)[:, 0] |= 1 << (np.arange(len(data) // 16) % 8).astype(np.uint8)  # every mode, evenly
    p.add_texture_2d_encoded(fid, W, H, [data.tobytes()])
    p.sync()
    p.close()
