#!/usr/bin/env python3
"""Writes the BC6H bit-layout tables (rend3_amd/csrc/bc6h_tables.h for the HIP decoder, oracle/bc6h_tables.h for the
oracle) from the textual block layouts below.

The layouts restate the block format tables of the Khronos Data Format Specification 1.3 (chapter "BPTC", BC6H) /
the D3D11 "BC6H Format" page: per mode, the fields in stream order after the mode bits.  `r0` .. `b3` are the four
endpoints' channels (r0 = endpoint 0 red, the "base"; r1 the second endpoint of region 0; r2 / r3 region 1), `d` the
partition; `f[a:b]` = bits a down to b of field f with the LOWER bit first in the stream; `f[a:b]r` the same bits with
the HIGHER bit first (modes 13 and 14 store the high bits of the base reversed); `f[a]` a single bit.

Each table entry packs (field, lowest bit, bit count, reversed): field | lsb << 4 | (count - 1) << 8 | rev << 12.
Field ids: r0 g0 b0 r1 g1 b1 r2 g2 b2 r3 g3 b3 = 0..11, d = 12.  0xFFFF terminates a mode.

  python tools/gen_bc6h_tables.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["r0", "g0", "b0", "r1", "g1", "b1", "r2", "g2", "b2", "r3", "g3", "b3", "d"]

# mode bits (value, count), endpoint bits, delta bits (r, g, b), transformed, regions, layout
MODES = [
    (0b00, 2, 10, (5, 5, 5), 1, 2,
     "g2[4] b2[4] b3[4] r0[9:0] g0[9:0] b0[9:0] r1[4:0] g3[4] g2[3:0] g1[4:0] b3[0] g3[3:0] b1[4:0] b3[1] b2[3:0] "
     "r2[4:0] b3[2] r3[4:0] b3[3] d[4:0]"),
    (0b01, 2, 7, (6, 6, 6), 1, 2,
     "g2[5] g3[4] g3[5] r0[6:0] b3[0] b3[1] b2[4] g0[6:0] b2[5] b3[2] g2[4] b0[6:0] b3[3] b3[5] b3[4] r1[5:0] g2[3:0] "
     "g1[5:0] g3[3:0] b1[5:0] b2[3:0] r2[5:0] r3[5:0] d[4:0]"),
    (0b00010, 5, 11, (5, 4, 4), 1, 2,
     "r0[9:0] g0[9:0] b0[9:0] r1[4:0] r0[10] g2[3:0] g1[3:0] g0[10] b3[0] g3[3:0] b1[3:0] b0[10] b3[1] b2[3:0] "
     "r2[4:0] b3[2] r3[4:0] b3[3] d[4:0]"),
    (0b00110, 5, 11, (4, 5, 4), 1, 2,
     "r0[9:0] g0[9:0] b0[9:0] r1[3:0] r0[10] g3[4] g2[3:0] g1[4:0] g0[10] g3[3:0] b1[3:0] b0[10] b3[1] b2[3:0] "
     "r2[3:0] b3[0] b3[2] r3[3:0] g2[4] b3[3] d[4:0]"),
    (0b01010, 5, 11, (4, 4, 5), 1, 2,
     "r0[9:0] g0[9:0] b0[9:0] r1[3:0] r0[10] b2[4] g2[3:0] g1[3:0] g0[10] b3[0] g3[3:0] b1[4:0] b0[10] b2[3:0] "
     "r2[3:0] b3[1] b3[2] r3[3:0] b3[4] b3[3] d[4:0]"),
    (0b01110, 5, 9, (5, 5, 5), 1, 2,
     "r0[8:0] b2[4] g0[8:0] g2[4] b0[8:0] b3[4] r1[4:0] g3[4] g2[3:0] g1[4:0] b3[0] g3[3:0] b1[4:0] b3[1] b2[3:0] "
     "r2[4:0] b3[2] r3[4:0] b3[3] d[4:0]"),
    (0b10010, 5, 8, (6, 5, 5), 1, 2,
     "r0[7:0] g3[4] b2[4] g0[7:0] b3[2] g2[4] b0[7:0] b3[3] b3[4] r1[5:0] g2[3:0] g1[4:0] b3[0] g3[3:0] b1[4:0] b3[1] "
     "b2[3:0] r2[5:0] r3[5:0] d[4:0]"),
    (0b10110, 5, 8, (5, 6, 5), 1, 2,
     "r0[7:0] b3[0] b2[4] g0[7:0] g2[5] g2[4] b0[7:0] g3[5] b3[4] r1[4:0] g3[4] g2[3:0] g1[5:0] g3[3:0] b1[4:0] b3[1] "
     "b2[3:0] r2[4:0] b3[2] r3[4:0] b3[3] d[4:0]"),
    (0b11010, 5, 8, (5, 5, 6), 1, 2,
     "r0[7:0] b3[1] b2[4] g0[7:0] b2[5] g2[4] b0[7:0] b3[5] b3[4] r1[4:0] g3[4] g2[3:0] g1[4:0] b3[0] g3[3:0] b1[5:0] "
     "b2[3:0] r2[4:0] b3[2] r3[4:0] b3[3] d[4:0]"),
    (0b11110, 5, 6, (6, 6, 6), 0, 2,
     "r0[5:0] g3[4] b3[0] b3[1] b2[4] g0[5:0] g2[5] b2[5] b3[2] g2[4] b0[5:0] g3[5] b3[3] b3[5] b3[4] r1[5:0] g2[3:0] "
     "g1[5:0] g3[3:0] b1[5:0] b2[3:0] r2[5:0] r3[5:0] d[4:0]"),
    (0b00011, 5, 10, (10, 10, 10), 0, 1, "r0[9:0] g0[9:0] b0[9:0] r1[9:0] g1[9:0] b1[9:0]"),
    (0b00111, 5, 11, (9, 9, 9), 1, 1, "r0[9:0] g0[9:0] b0[9:0] r1[8:0] r0[10] g1[8:0] g0[10] b1[8:0] b0[10]"),
    (0b01011, 5, 12, (8, 8, 8), 1, 1, "r0[9:0] g0[9:0] b0[9:0] r1[7:0] r0[11:10]r g1[7:0] g0[11:10]r b1[7:0] b0[11:10]r"),
    (0b01111, 5, 16, (4, 4, 4), 1, 1, "r0[9:0] g0[9:0] b0[9:0] r1[3:0] r0[15:10]r g1[3:0] g0[15:10]r b1[3:0] b0[15:10]r"),
]


def parse(layout):
    out = []
    for tok in layout.split():
        rev = tok.endswith("r") and tok[-2] == "]"
        if rev:
            tok = tok[:-1]
        name, rng = tok[:-1].split("[")
        hi, lo = (int(x) for x in rng.split(":")) if ":" in rng else (int(rng), int(rng))
        out.append((FIELDS.index(name), lo, hi - lo + 1, 1 if rev else 0))
    return out


def check(mode):
    """Every field gets exactly the bits its width says, the header ends at bit 65 (one region) / 82 (two)."""
    val, mbits, epb, delta, _tr, regions, layout = mode
    seen = {}
    total = mbits
    for f, lo, n, _rev in parse(layout):
        for b in range(lo, lo + n):
            assert (f, b) not in seen, (bin(val), FIELDS[f], b)
            seen[(f, b)] = True
        total += n
    assert total == (65 if regions == 1 else 82), (bin(val), total)
    for c in range(3):
        assert all((c, b) in seen for b in range(epb)), (bin(val), "base", c)
        for e in range(1, 2 * regions):
            got = sorted(b for (f, b) in seen if f == 3 * e + c)
            assert got == list(range(delta[c])), (bin(val), FIELDS[3 * e + c], got)
    if regions == 2:
        assert sorted(b for (f, b) in seen if f == 12) == list(range(5))


def emit(prefix, qual):
    lines = ["// Generated by tools/gen_bc6h_tables.py -- BC6H block layouts (Khronos Data Format Specification 1.3, BPTC).",
             "// entry = field | lsb << 4 | (count - 1) << 8 | reversed << 12; fields r0 g0 b0 r1 g1 b1 r2 g2 b2 r3 g3 b3 d = 0..12",
             "#pragma once", "#include <stdint.h>", f"#define {prefix}_MAX_ENTRIES 28"]
    rows, info = [], []
    for val, mbits, epb, delta, tr, regions, layout in MODES:
        ent = [f | (lo << 4) | ((n - 1) << 8) | (rev << 12) for f, lo, n, rev in parse(layout)]
        assert len(ent) < 28
        ent += [0xFFFF] * (28 - len(ent))
        rows.append("    {" + ", ".join(f"0x{e:04X}" for e in ent) + "},")
        # mode value | mode bits << 5 | endpoint bits << 8 | dr << 13 | dg << 18 | db << 23 | transformed << 28 | (regions - 1) << 29
        info.append(val | (mbits << 5) | (epb << 8) | (delta[0] << 13) | (delta[1] << 18) | (delta[2] << 23) | (tr << 28) | ((regions - 1) << 29))
    lines.append(f"{qual} uint16_t {prefix}_LAYOUT[14][{prefix}_MAX_ENTRIES] = {{")
    lines += rows
    lines.append("};")
    lines.append("// value | mode bits << 5 | endpoint bits << 8 | delta bits r << 13 | g << 18 | b << 23 | transformed << 28 | (regions - 1) << 29")
    lines.append(f"{qual} uint32_t {prefix}_MODE[14] = {{" + ", ".join(f"0x{v:08X}u" for v in info) + "};")
    return "\n".join(lines) + "\n"


def main():
    for m in MODES:
        check(m)
    with open(os.path.join(ROOT, "rend3_amd", "csrc", "bc6h_tables.h"), "w") as f:
        f.write(emit("BC6H", "BC6H_TABLE"))
    with open(os.path.join(ROOT, "oracle", "bc6h_tables.h"), "w") as f:
        f.write(emit("BC6H", "static const"))
    print("ok")


if __name__ == "__main__":
    main()
