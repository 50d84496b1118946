"""Race hunt for the hand-counted waits of conv_wino44f.hip: random shapes (small ones and chip-filling ones), every case launched three
times -- the results must be bit-identical -- and compared with the direct float32 kernel: relative L2 error (the tests' metric, tolerance 5e-6) and the worst element against the
output's largest element (reported).  Prints the number of bad cases."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
rng = np.random.default_rng(int(os.environ.get("RACE_SEED", 1)))
g = torch.Generator(device="cuda").manual_seed(int(os.environ.get("RACE_SEED", 1)))
f32 = torch.float32
bad = 0
worst_l2 = worst_max = 0.0
N = int(os.environ.get("RACE_CASES", 120))
for it in range(N):
    big = it % 6 == 0
    B = int(rng.integers(4, 9)) if big else int(rng.integers(1, 4))
    H = 16 * int(rng.integers(8, 49)) if big else 16 * int(rng.integers(1, 5))
    W = 16 * int(rng.integers(4, 17)) if big else 16 * int(rng.integers(1, 5))
    C0 = 8 * int(rng.integers(1, 33)); C1 = 8 * int(rng.integers(0, 17)) * int(rng.integers(0, 2))
    S0 = 8 * int(rng.integers(1, 17)) * int(rng.integers(0, 2)); S1 = 8 * int(rng.integers(1, 9)) * int(rng.integers(0, 2)) if S0 else 0
    Cout = int(rng.choice([128, 256]))
    use_aff, use_skip, bias_rows = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.choice([0, 1, B]))
    Cin = C0 + C1
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5
    aff = torch.stack([1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g), 0.3 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous() if use_aff else None
    sc0 = sc1 = w_sc = None
    if S0:
        xs = torch.randn(B, H, W, S0 + S1, device="cuda", generator=g)
        sc0 = xs[..., :S0].contiguous(); sc1 = xs[..., S0:].contiguous() if S1 else None
        w_sc = torch.randn(Cout, S0 + S1, 1, 1, device="cuda", generator=g) / (S0 + S1) ** 0.5
    bias = (torch.randn(bias_rows, Cout, device="cuda", generator=g) if bias_rows > 1 else torch.randn(Cout, device="cuda", generator=g)) if bias_rows else None
    skip = torch.randn(B, H, W, Cout, device="cuda", generator=g) if use_skip else None
    kw = dict(x1=x[..., C0:].contiguous() if C1 else None, affine=aff, bias=bias, skip=skip, scale=0.7 if use_skip else 1.0, sc0=sc0, sc1=sc1, want_stats=True)
    x0 = x[..., :C0].contiguous()
    pw = ops.pack_conv_weight(w, C0=C0, dtype=f32, w_sc=w_sc, S0=S0 if S0 else None, winograd=44)
    pd = ops.pack_conv_weight(w, C0=C0, dtype=f32, w_sc=w_sc, S0=S0 if S0 else None)
    o = [ops.conv2d(x0, pw, Cout, 3, winograd=44, **kw) for _ in range(3)]
    od = ops.conv2d(x0, pd, Cout, 3, **kw)
    torch.cuda.synchronize()
    same = all(torch.equal(o[0][0], o[i][0]) and torch.equal(o[0][1], o[i][1]) for i in (1, 2))
    d = (o[0][0] - od[0]).double()
    e = float(d.norm() / od[0].double().norm())
    em = float(d.abs().max() / od[0].abs().max())
    worst_l2, worst_max = max(worst_l2, e), max(worst_max, em)
    if not same or not e < 5e-6:
        bad += 1
        print("BAD", it, (B, H, W, C0, C1, S0, S1, Cout, use_aff, use_skip, bias_rows), "repeatable" if same else "NOT REPEATABLE", f"err {e:.3e}", flush=True)
print(f"seed {os.environ.get('RACE_SEED', 1)}: {N} cases x 3 launches, {bad} bad; against the direct f32 kernel: worst relative L2 error {worst_l2:.3e}, worst element / largest element {worst_max:.3e}", flush=True)
