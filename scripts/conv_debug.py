#!/usr/bin/env python
"""Debug matrix for fd_conv2d: which factor breaks parity (run on the GPU box)."""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
from oracle import flowdec_oracle as O

def nhwc(a, dt): return torch.from_numpy(np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))).cuda().to(dt)
def back(t): return np.transpose(t.float().cpu().numpy(), (0, 3, 1, 2))

rng = np.random.default_rng(0)
for prec, Cin, Cout, aff, skip, brows, B in itertools.product(["fp32", "bf16"], [16, 32, 64], [128, 256], [0, 1], [0, 1], [1, 2], [1, 2]):
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    H, W = 16, 32
    q = (lambda a: O.round_bf16(a.astype(np.float32))) if prec == "bf16" else (lambda a: a.astype(np.float32))
    x = q(rng.standard_normal((B, Cin, H, W))); w = q(rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin))
    xin = x; A = None
    if aff:
        a = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32); d = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        A = torch.from_numpy(np.stack([a, d], -1)).cuda()
        xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float32)
        if prec == "bf16": xin = O.round_bf16(xin)
    ref = O.conv2d(xin.astype(np.float64), w.astype(np.float64), None)
    bv = rng.standard_normal((brows, Cout)).astype(np.float32)
    ref = ref + (bv[:, :, None, None] if brows > 1 else bv[0][None, :, None, None])
    sk = None
    if skip:
        s_ = q(rng.standard_normal((B, Cout, H, W))); sk = nhwc(s_, dt); ref = ref + s_
    if brows > 1 and B == 1: continue
    out = ops.conv2d(nhwc(x, dt), ops.pack_conv_weight(torch.from_numpy(w).cuda(), dtype=dt), Cout, 3, affine=A,
                     bias=torch.from_numpy(bv if brows > 1 else bv[0]).cuda(), skip=sk, scale=1.0)
    got = back(out)
    e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    # per-batch / per-channel-block error localisation
    eb = [float(np.linalg.norm(got[b] - ref[b]) / np.linalg.norm(ref[b])) for b in range(B)]
    flag = "OK " if e < (6e-3 if prec == "bf16" else 2e-5) else "BAD"
    print(f"{flag} {prec} Cin={Cin:3d} Cout={Cout} aff={aff} skip={skip} brows={brows} B={B} err={e:.2e} per-b={['%.1e' % v for v in eb]}", flush=True)
