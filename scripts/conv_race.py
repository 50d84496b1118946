#!/usr/bin/env python
"""Race hunt for fd_conv2d: run every case several times, check bit-determinism and parity vs torch conv (bf16)."""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
import torch.nn.functional as F

g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
for (H, W), Cin, Cout, aff, sc in itertools.product([(64, 64), (192, 32), (768, 64)], [32, 64, 256], [32, 128, 256], [0, 1], [0, 64]):
    B = 2
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(dt)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (9 * Cin) ** 0.5
    w = w.to(dt).float()
    A = None; xin = x.float()
    if aff:
        a = 1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g); d = 0.3 * torch.randn(B, Cin, device="cuda", generator=g)
        A = torch.stack([a, d], -1).contiguous()
        v = xin * a[:, None, None, :] + d[:, None, None, :]
        xin = (v * torch.sigmoid(v)).to(dt).float()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w, padding=1)
    s0 = ws = None
    if sc:
        s0 = torch.randn(B, H, W, sc, device="cuda", generator=g).to(dt)
        ws = (torch.randn(Cout, sc, 1, 1, device="cuda", generator=g) / sc ** 0.5).to(dt).float()
        ref = ref + F.conv2d(s0.float().permute(0, 3, 1, 2), ws)
    pw = ops.pack_conv_weight(w, dtype=dt, w_sc=ws)
    outs = []
    for r in range(4):
        o, st = ops.conv2d(x, pw, Cout, 3, affine=A, sc0=s0, want_stats=True)
        outs.append((o.clone(), st.clone()))
    torch.cuda.synchronize()
    det = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    got = outs[0][0].float().permute(0, 3, 1, 2)
    e = float((got - ref).norm() / ref.norm())
    nbad = [int((outs[0][0] != o[0]).sum()) for o in outs[1:]]
    flag = "OK " if det and e < 6e-3 else "BAD"
    print(f"{flag} {H}x{W} Cin={Cin:3d} Cout={Cout:3d} aff={aff} sc={sc:2d} err={e:.2e} det={det} mismatches={nbad}", flush=True)
