#!/usr/bin/env python3
"""VERDICT r4 "what's weak" #1: 40 % of rend3-test/shadow/cube.png (the lit plane, 29 376 px) differs from the oracle by exactly
one LSB in G -- golden [61, 86, 104], oracle [61, 85, 104].  Which rounding did the golden's adapter use?  This tool evaluates the
lit plane's colour under every rounding the pipeline offers between the f32 shading result and the 8-bit store and prints the
table (profiles/r05_lit_plane_lsb.md).  Pure CPU; the oracle supplies the f32 / f16 values it actually produces.

Pipeline of that pixel (opaque.wgsl:440-468,548-550 -> Rgba16Float target -> blit.wgsl:22-31 into Rgba8UnormSrgb, tonemapping.rs:44):
  L = albedo * (1 - metallic) / pi * intensity * nol        (roughness 0: D = 0, no specular; shadow = 1 on the lit plane)
  h = f16(L)          the HDR target's store
  e = OETF(h)         the sRGB encode of the 8-bit target's store (fixed function on a GPU)
  c = round(e * 255)
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def oetf64(x):
    return 12.92 * x if x <= 0.0031308 else 1.055 * x ** (1.0 / 2.4) - 0.055


def oetf32(x):
    x = np.float32(x)
    if x <= np.float32(0.0031308):
        return np.float32(12.92) * x
    return np.float32(1.055) * np.float32(np.power(x, np.float32(1.0 / 2.4), dtype=np.float32)) - np.float32(0.055)


def f16_round(x, mode):
    """x (float64) to the f16 grid under `mode`: rne, rtz, rtp (toward +inf), rtn."""
    m, e = math.frexp(x)              # x = m * 2^e, 0.5 <= m < 1
    ulp = 2.0 ** (e - 11)             # 11 significant bits
    k = x / ulp
    lo, hi = math.floor(k), math.ceil(k)
    if mode == "rtz" or mode == "rtn":
        q = lo
    elif mode == "rtp":
        q = hi
    else:
        q = lo if (k - lo < 0.5 or (k - lo == 0.5 and lo % 2 == 0)) else hi
    return q * ulp


def main():
    import scenes
    from oracle import host as hm
    from oracle.world import OracleRenderer, material_record as mk
    r = OracleRenderer(hm.LEFT)
    r.add_directional_light(color=(1, 1, 1), intensity=1.0, direction=(-1.0, -1.0, 1.0), distance=5.0, resolution=256)
    m1 = scenes.lit(r, mk, (0.25, 0.5, 0.75, 1.0))
    r.add_object(scenes.plane_mesh(r), m1, hm.rotation_x(-math.pi / 2))
    r.set_camera_data(hm.look_at_lh((0.0, 1.0, -1.0), (0, 0, 0), (0, 1, 0)), ("orthographic", (2.5, 2.5, 5.0)))
    out = r.render(256, 256)
    y, x = 128, 128
    h16 = out["hdr16"][y, x].view(np.float16).astype(np.float64)
    print("# The lit plane's one LSB (VERDICT r4 weak #1) -- `python tools/lit_plane_lsb.py`\n")
    print(f"golden `rend3-test/shadow/plane.png` / `cube.png`, lit plane: [61, 86, 104]; oracle rgba8 at ({x},{y}): {out['rgba8'][y, x][:3].tolist()}; "
          f"oracle Rgba16Float there: {h16[:3].tolist()}\n")
    nol = 1.0 / math.sqrt(3.0)
    print("| channel | analytic L (f64) | e * 255 from f64 | f16 RNE -> code | f16 toward zero -> code | f16 toward +inf -> code | oracle's f16 -> OETF in f32 -> code | code that needs |")
    print("|---|---|---|---|---|---|---|---|")
    for name, alb, gold in (("R", 0.25, 61), ("G", 0.5, 86), ("B", 0.75, 104)):
        L = alb / math.pi * nol
        cells = []
        for mode in ("rne", "rtz", "rtp"):
            hq = f16_round(L, mode)
            cells.append(f"{hq:.7f} -> {oetf64(hq) * 255:.3f} -> {int(oetf64(hq) * 255 + 0.5)}")
        ho = float(h16["RGB".index(name)])
        e32 = float(oetf32(ho))
        # smallest linear value whose code is the golden's
        lo, hi = 0.0, 1.0
        for _ in range(60):
            mid = (lo + hi) / 2
            if oetf64(mid) * 255 + 0.5 >= gold: hi = mid
            else: lo = mid
        cells.append(f"{ho:.7f} -> {e32 * 255:.3f} -> {int(e32 * 255 + 0.5)}")
        print(f"| {name} (golden {gold}) | {L:.7f} | {oetf64(L) * 255:.3f} | " + " | ".join(cells) + f" | L >= {hi:.7f} (= analytic {(hi / L - 1) * 100:+.3f} %, {(hi - L) / 2.0 ** (math.frexp(L)[1] - 11):+.2f} f16 ulp) |")
    print("""
Reading: G is the one channel whose exact code lies near a rounding boundary (85.47), yet NO rounding of the f16 store reaches 86: the
smallest radiance that encodes to 86 is one f16 ulp above the analytic one (1.5 ulps above the value RNE stores), and evaluating the OETF in f32 instead of f64 moves the code
by < 0.001.  The PCF factor is exactly 1 on the lit plane (all twelve comparisons pass: weights sum to 1 within 1 ulp of f32, and 1 ulp of
f32 is 1e-4 f16 ulps).  What is left is the 8-bit target's fixed-function sRGB encode: the APIs under wgpu allow it 0.6 ULP of the 8-bit
code (D3D11.3 functional spec 3.2.3.6 FLOAT -> UNORM_SRGB; Vulkan defers to the same table-based conversions), and the golden's adapter
returned 86 for an exact 85.47-85.49 -- inside that tolerance, outside a correctly rounded OETF.  A restatement can only match it by
adopting one adapter's conversion table, which would move R (60.51 -> 61, already equal) and every other golden by unknown amounts; the
oracle keeps the exact OETF (tonemapping.rs:44: the format does the conversion), the 1-LSB pixels stay pinned by count in
tests/test_oracle_goldens.py::test_shadow_cube / test_shadow_plane.""")


if __name__ == "__main__":
    main()
