#!/usr/bin/env python
"""Winograd F(2,3) conv kernel vs the direct MFMA kernel: parity against an f64 torch reference on small ragged shapes,
then per-launch timing of both on the layer shapes of FlowDec-75m (interleaved A/B rounds in one process)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops  # noqa: E402

PAR = [  # name, B, H, W, C0, C1, Cout, affine, bias_rows, skip, S0, S1
    ("basic", 1, 16, 16, 32, 0, 128, False, 0, False, 0, 0),
    ("two_ntiles", 2, 32, 16, 64, 0, 256, False, 1, False, 0, 0),
    ("aff_bias_skip", 2, 16, 32, 64, 0, 128, True, 2, True, 0, 0),
    ("concat", 1, 16, 16, 64, 32, 128, True, 1, True, 0, 0),
    ("ragged", 1, 24, 8, 32, 0, 128, True, 1, True, 0, 0),
    ("odd_w", 1, 20, 22, 32, 0, 128, True, 1, True, 0, 0),
    ("odd_w2", 1, 17, 13, 32, 0, 128, False, 1, False, 0, 0),
    ("deepk", 1, 16, 16, 256, 256, 256, True, 1, True, 0, 0),
    ("shortcut", 2, 16, 16, 256, 0, 256, True, 1, False, 64, 0),
    ("shortcut_cat", 1, 32, 16, 128, 0, 128, True, 1, False, 128, 256),
    ("shortcut_small", 1, 16, 16, 32, 0, 128, True, 1, False, 32, 32),
    ("wide", 1, 16, 48, 32, 0, 256, True, 1, True, 0, 0),
    ("multi_tile", 2, 48, 80, 64, 0, 256, True, 2, True, 32, 0),
]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def parity():
    g = torch.Generator(device="cuda").manual_seed(0)
    bad = 0
    for name, B, H, W, C0, C1, Cout, aff, brows, skip, S0, S1 in PAR:
        Cin = C0 + C1
        x = torch.randn(B, H, W, Cin, device="cuda", generator=g).bfloat16()
        w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
        affine = None
        xin = x.double()
        if aff:
            a = 1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g)
            d = 0.3 * torch.randn(B, Cin, device="cuda", generator=g)
            affine = torch.stack([a, d], -1).contiguous()
            xin = F.silu(x.float() * a[:, None, None, :] + d[:, None, None, :]).double()
        ref = F.conv2d(xin.permute(0, 3, 1, 2), w.double(), padding=1)
        sc0 = sc1 = wsc = None
        if S0:
            xs = torch.randn(B, H, W, S0 + S1, device="cuda", generator=g).bfloat16()
            wsc = torch.randn(Cout, S0 + S1, 1, 1, device="cuda", generator=g) / (S0 + S1) ** 0.5
            ref = ref + F.conv2d(xs.double().permute(0, 3, 1, 2), wsc.double())
            sc0 = xs[..., :S0].contiguous()
            sc1 = xs[..., S0:].contiguous() if S1 else None
        bias = None
        if brows:
            bias = torch.randn(brows, Cout, device="cuda", generator=g)
            ref = ref + bias.double()[:, :, None, None] if brows > 1 else ref + bias.double()[0][None, :, None, None]
            bias = bias if brows > 1 else bias[0].contiguous()
        sk = None
        scale = 1.0
        if skip:
            sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16()
            ref = ref + sk.double().permute(0, 3, 1, 2)
            scale = 0.5 ** 0.5
        ref = (ref * scale).permute(0, 2, 3, 1)
        x0 = x[..., :C0].contiguous()
        x1 = x[..., C0:].contiguous() if C1 else None
        res = {}
        for wino in (False, True):
            pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=S0 if S0 else None, winograd=wino)
            out, st = ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=affine, bias=bias, skip=sk, scale=scale, sc0=sc0, sc1=sc1, want_stats=True,
                                 winograd=wino)
            torch.cuda.synchronize()
            s = st.double().sum(1)[:, :Cout]
            rs = torch.stack([ref.sum((1, 2)), (ref ** 2).sum((1, 2))], -1)
            res[wino] = (rel(out, ref), float((s - rs).abs().max() / rs.abs().max()))
        ok = res[True][0] < 6e-3 and res[True][1] < 3e-3
        bad += not ok
        print(f"parity {name:16s} direct err {res[False][0]:.2e} stats {res[False][1]:.2e} | wino err {res[True][0]:.2e} stats {res[True][1]:.2e} {'OK' if ok else 'FAIL'}",
              flush=True)
    return bad


SHAPES = [  # name, H, W, C0, C1, Cout, affine, skip, S
    ("L0 256->256 aff+skip", 768, 256, 256, 0, 256, True, True, 0),
    ("L0 256->256 plain", 768, 256, 256, 0, 256, False, False, 0),
    ("L0 512->256 cat aff", 768, 256, 256, 256, 256, True, False, 0),
    ("L0 320->256 cat aff", 768, 256, 256, 64, 256, True, False, 0),
    ("L0 64->256 aff", 768, 256, 64, 0, 256, True, False, 0),
    ("L0 rb31 tail 256+sc512", 768, 256, 256, 0, 256, True, False, 512),
    ("L1 256->256 aff+skip", 384, 128, 256, 0, 256, True, True, 0),
    ("L1 512->256 cat aff", 384, 128, 256, 256, 256, True, False, 0),
    ("L2 256->256 aff+skip", 192, 64, 256, 0, 256, True, True, 0),
    ("L2 384->256 cat aff", 192, 64, 128, 256, 256, True, False, 0),
    ("L3 128->128 aff+skip", 96, 32, 128, 0, 128, True, True, 0),
]


def timing(B, iters, rounds, only, algos=(False, True)):
    g = torch.Generator(device="cuda").manual_seed(1)
    for i, (name, H, W, C0, C1, Cout, aff, skip, S) in enumerate(SHAPES):
        if only >= 0 and i != only:
            continue
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
        w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
        affine = torch.stack([1 + 0.1 * torch.randn(B, Cin, device="cuda", generator=g), 0.1 * torch.randn(B, Cin, device="cuda", generator=g)],
                             -1).contiguous() if aff else None
        bias = torch.randn(Cout, device="cuda", generator=g)
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if skip else None
        sc0 = sc1 = wsc = None
        if S:
            sc0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).bfloat16()
            sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).bfloat16() if S > 256 else None
            wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
        fl = 2.0 * B * H * W * Cout * (Cin * 9 + S)
        fs = {}
        for wino in algos:
            pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None, winograd=wino)
            fs[wino] = (lambda pw=pw, wino=wino: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=affine, bias=bias, skip=sk, scale=0.7071, sc0=sc0,
                                                            sc1=sc1, want_stats=True, winograd=wino))
        best = {False: 1e9, True: 1e9}
        for f in fs.values():
            f()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for wino, f in fs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                best[wino] = min(best[wino], e0.elapsed_time(e1) / iters)
        if len(algos) < 2:
            a = algos[0]
            print(f"time {i:2d} {name:26s} B={B} {'wino' if a else 'direct'} {best[a]:7.3f} ms {fl / best[a] / 1e9:7.1f} TF", flush=True)
            continue
        print(f"time {i:2d} {name:26s} B={B} direct {best[False]:7.3f} ms {fl / best[False] / 1e9:7.1f} TF | wino {best[True]:7.3f} ms "
              f"{fl / best[True] / 1e9:7.1f} TF(eff)  x{best[False] / best[True]:.3f}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--algo", default="both", choices=["both", "direct", "wino"])
    a = ap.parse_args()
    bad = 0 if a.no_parity else parity()
    if not a.no_timing:
        timing(a.B, a.iters, a.rounds, a.only, {"both": (False, True), "direct": (False,), "wino": (True,)}[a.algo])
    sys.exit(1 if bad else 0)
