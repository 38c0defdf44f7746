"""GPU: csrc/exact_math.h -- the short correctly-rounded reciprocal / square root / reciprocal square root the shading kernels
evaluate -- against hipcc's IEEE expansions over ALL 2^32 f32 bit patterns (r3n_selftest_exact_math): a proof by exhaustion on
the hardware the library is built for.  The guarded functions must not differ on any pattern; the unguarded cores must be clean
on every positive-normal exponent the guards admit (so the guards, chosen from this very histogram, keep their margin)."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_exact_math_all_bit_patterns():
    import torch
    assert torch.cuda.is_available()
    import rend3_amd
    lib = rend3_amd.lib()
    hist = np.zeros((3, 512), dtype=np.uint64)
    guarded = np.zeros(3, dtype=np.uint64)
    assert lib.r3n_selftest_exact_math(0, hist.ctypes.data, guarded.ctypes.data) == 0
    assert [int(v) for v in guarded] == [0, 0, 0], f"guarded rcp / sqrt / rsqrt differ from the compiler's expansion: {guarded}"
    # guards of exact_math.h (biased exponents): rcp [2, 252), sqrt / rsqrt [24, 254)
    assert hist[0, 2:252].sum() == 0 and hist[1, 24:254].sum() == 0 and hist[2, 24:254].sum() == 0
    # and the histogram is what the guards were chosen from: the cores DO differ outside (subnormal operands / results)
    assert hist[0, 0] > 0 and hist[0, 253] > 0 and hist[1, 1] > 0 and hist[1, 22] > 0 and hist[1, 23] == 0


def test_unorm8_arithmetic_is_the_division_cpu():
    """exact_math::unorm8 restated in numpy (an fma = one rounding of the exact a * b + c; the products of 24-bit values are exact in
    f64): q = c * RN(1/255), q' = fma(fma(-255, q, c), RN(1/255), q) equals RN(c / 255) for every 8-bit c -- while the bare product
    does not (which is why the texel tables used a division)."""
    f = np.float32
    c = np.arange(256, dtype=np.float32)
    ref = (c / f(255.0)).astype(np.float32)
    r = f(1.0) / f(255.0)
    q = (c * r).astype(np.float32)
    fma = lambda a, b, d: (a.astype(np.float64) * np.float64(b) + d.astype(np.float64)).astype(np.float32)  # noqa: E731
    e = fma(q, f(-255.0), c)
    q2 = fma(e, r, q)
    assert (q != ref).sum() > 100 and np.array_equal(q2.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_unorm8_all_inputs_on_the_device():
    import torch
    assert torch.cuda.is_available()
    import rend3_amd
    lib = rend3_amd.lib()
    bad = np.full(1, 999, dtype=np.uint32)
    assert lib.r3n_selftest_unorm8(0, bad.ctypes.data) == 0
    assert int(bad[0]) == 0, f"exact_math::unorm8 differs from c / 255.0f for {int(bad[0])} of the 256 inputs"
