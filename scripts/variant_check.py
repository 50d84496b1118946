#!/usr/bin/env python
"""Compare a conv tile-configuration variant (fd_tuning_set conv_variant) against the default one on the same inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
v = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for (B, H, W, C0, C1, Cout, aff, skip, S0) in [(2, 96, 64, 256, 0, 256, 1, 1, 0), (1, 50, 40, 256, 64, 256, 1, 0, 0), (2, 64, 64, 256, 0, 256, 1, 0, 512)]:
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
        sc = torch.randn(B, H, W, S0, device="cuda", generator=g).to(dt) if S0 else None
        w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
        wsc = torch.randn(Cout, S0, 1, 1, device="cuda", generator=g) / S0 ** 0.5 if S0 else None
        pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, w_sc=wsc, S0=S0)
        A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(dt) if skip else None
        outs = []
        for var in (0, v):
            L.check(lib.fd_tuning_set(b"conv_variant", var))
            o, st = ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, skip=sk, scale=0.7, sc0=sc, want_stats=True)
            outs.append((o.float().clone(), st.clone()))
        L.check(lib.fd_tuning_set(b"conv_variant", 0))
        d = (outs[0][0] - outs[1][0]).abs().max().item()
        ds = (outs[0][1].sum(1) - outs[1][1].sum(1)).abs().max().item() / outs[0][1].sum(1).abs().max().item()
        print(f"{str(dt):16s} {H}x{W} C={C0}+{C1} sc={S0}: max|out diff| = {d:.3e}   stats rel diff = {ds:.2e}")
