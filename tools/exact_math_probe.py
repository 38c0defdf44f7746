#!/usr/bin/env python3
"""GPU: csrc/exact_math.h against the compiler's IEEE expansions over all 2^32 f32 bit patterns (r3n_selftest_exact_math).
Prints, per function, the sign + exponent bins in which the UNGUARDED short sequence differs, and the guarded functions' counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rend3_amd  # noqa: E402

lib = rend3_amd.lib()
hist = np.zeros((3, 512), dtype=np.uint64)
guarded = np.zeros(3, dtype=np.uint64)
rc = lib.r3n_selftest_exact_math(0, hist.ctypes.data, guarded.ctypes.data)
assert rc == 0, rc
for f, name in enumerate(("rcp_core", "sqrt_core", "rsqrt_core")):
    bins = np.nonzero(hist[f])[0]
    pos = [int(b) for b in bins if b < 256]
    print(f"{name}: {int(hist[f].sum())} differing patterns; positive-sign exponent bins with differences: {pos[:20]}{' ...' if len(pos) > 20 else ''}"
          f" (clean positive range: {[b for b in range(256) if hist[f][b] == 0][:1]}..{[b for b in range(255, -1, -1) if hist[f][b] == 0][:1]});"
          f" negative-sign bins: {len(bins) - len(pos)}")
    for b in pos:
        print(f"    exponent {b:3d}: {int(hist[f][b])}")
print("guarded functions (rcp, sqrt, rsqrt) differing patterns:", [int(v) for v in guarded])
