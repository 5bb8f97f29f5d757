"""The operand split of precision='bf16x3' (csrc/rollout.hpp split3), restated in numpy: three bf16 pieces reproduce an fp32 value to
2^-24 relative, and the six kept partial products reproduce an fp32 product to a few 2^-24."""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = x.astype(np.float32)
    h0 = bf16_rne(x)
    r1 = (x - h0).astype(np.float32)
    h1 = bf16_rne(r1)
    r2 = (r1 - h1).astype(np.float32)
    h2 = bf16_rne(r2)
    return h0, h1, h2


def test_three_pieces_carry_24_bits_and_six_products_an_fp32_product():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    b = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    a0, a1, a2 = split3(a)
    b0, b1, b2 = split3(b)
    assert (np.abs(a.astype(np.float64) - (a0.astype(np.float64) + a1 + a2)) <= 2.0 ** -24 * np.abs(a)).all()
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = (a0.astype(np.float64) * b0 + (a0.astype(np.float64) * b1 + a1.astype(np.float64) * b0)
           + (a0.astype(np.float64) * b2 + a1.astype(np.float64) * b1 + a2.astype(np.float64) * b0))
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -22  # dropped terms a1 b2 + a2 b1 + a2 b2 and the piece residuals
    assert np.median(rel) <= 2.0 ** -25
    # every kept partial product is exact in fp32 (8 x 8 significand bits)
    p = (a1.astype(np.float64) * b1.astype(np.float64))
    assert (p.astype(np.float32).astype(np.float64) == p).all()
