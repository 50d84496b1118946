#!/usr/bin/env python
"""Can the F(4,3) kernel's operand transform silu(u) be a transcendental-free polynomial in PACKED FP16 (round-5 review, item 1a)?

    silu(u) = u * sigmoid(u),   sigmoid(u) - 1/2 = odd function, clamped at |u| <= c where it saturates.

Fits the odd polynomial of degree d to sigmoid - 1/2 on [-c, c] (Chebyshev nodes, least squares in the Chebyshev basis = near-minimax),
then evaluates it the way v_pk_fma_f16 would: Horner in u^2, EVERY operation rounded to fp16 (fma = one rounding), and reports the
error of the resulting silu against float64 -- next to the same polynomial evaluated in float32 and to what fp16 storage of the exact
silu costs (the operand's own quantisation, 2^-11).  CPU only.   python scripts/silu_poly_fp16.py
"""
import numpy as np

f16 = np.float16


def fma16(a, b, c):
    """fp16 fused multiply-add: exact product and sum in float64 (fp16 products are exact there), one rounding."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f16)


def fit_odd(c, deg):
    """coefficients q_k of sigmoid(c x) - 1/2 ~= x * sum_k q_k (x^2)^k for x = u / c in [-1, 1], k = 0 .. (deg - 1) / 2, via a Chebyshev fit.
    The NORMALISED variable keeps every Horner intermediate in range (in u itself the high coefficients underflow fp16)."""
    n = 4001
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n)                 # Chebyshev nodes on [-1, 1]
    u = c * x
    y = 1.0 / (1.0 + np.exp(-u)) - 0.5
    cheb = np.polynomial.chebyshev.chebfit(x, y, deg)
    cheb[0::2] = 0.0                                             # odd function
    mono = np.polynomial.chebyshev.cheb2poly(cheb)               # in x = u / c
    return np.array([mono[2 * k + 1] for k in range((deg + 1) // 2)])


def silu_poly(u, q, c, dtype):
    """u * (1/2 + x * P(x^2)) with x = clamp(u / c, -1, 1), Horner, every step in `dtype` (fp16: single-rounded fmas)."""
    if dtype is f16:
        u16 = u.astype(f16)
        uc = np.clip((u16.astype(np.float64) / c).astype(f16), f16(-1), f16(1))
        t = (uc.astype(np.float64) ** 2).astype(f16)
        p = np.full(u.shape, f16(q[-1]), f16)
        for k in range(len(q) - 2, -1, -1):
            p = fma16(p, t, np.full(u.shape, f16(q[k]), f16))
        s = fma16(p, uc, np.full(u.shape, f16(0.5), f16))
        return (u16.astype(np.float64) * s.astype(np.float64)).astype(f16).astype(np.float64)
    uu = u.astype(dtype)
    uc = np.clip(uu / dtype(c), dtype(-1), dtype(1))
    t = uc * uc
    p = np.full(u.shape, dtype(q[-1]), dtype)
    for k in range(len(q) - 2, -1, -1):
        p = p * t + dtype(q[k])
    return (uu * (p * uc + dtype(0.5))).astype(np.float64)


def main():
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.standard_normal(400000) * 1.5, rng.uniform(-12, 12, 100000)])   # GroupNorm outputs (a x + d) and a uniform sweep
    exact = u / (1.0 + np.exp(-u))
    scale = np.sqrt(np.mean(exact ** 2))
    rel = lambda v: float(np.sqrt(np.mean((v - exact) ** 2)) / scale)
    print(f"reference: exact silu stored as fp16: rel L2 {rel(exact.astype(f16).astype(np.float64)):.2e}, max abs {np.abs(exact.astype(f16) - exact).max():.2e}"
          f"   (as bf16: {rel((exact.astype(np.float32).view(np.uint32) + 0x8000 & 0xffff0000).view(np.float32).astype(np.float64)):.2e})")
    print(f"{'clamp c':>8s} {'degree':>6s} {'max |q_k| (in u / c)':>20s} | {'fit error (f64)':>16s} | {'Horner f32: rel L2 / max abs':>30s} | {'Horner fp16: rel L2 / max abs':>30s}")
    for c in (6.0, 8.0, 10.0):
        for deg in (9, 13, 17, 21):
            q = fit_odd(c, deg)
            grow = max(abs(q))
            fit = silu_poly(u, q, c, np.float64)
            s32 = silu_poly(u, q, c, np.float32)
            s16 = silu_poly(u, q, c, f16)
            print(f"{c:8.1f} {deg:6d} {grow:20.1f} | {rel(fit):16.2e} | {rel(s32):14.2e} / {np.abs(s32 - exact).max():9.2e} | {rel(s16):14.2e} / {np.abs(s16 - exact).max():9.2e}")


if __name__ == "__main__":
    main()
