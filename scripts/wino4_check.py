#!/usr/bin/env python
"""Winograd F(4,3) conv kernel (conv_wino4.hip, FD_WINOGRAD4) vs the direct MFMA kernel: parity against an f64 torch reference
(outputs and the fused GroupNorm partial sums), determinism, then per-launch timing of both on the layer shapes of FlowDec-75m
(interleaved A/B rounds in one process)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops  # noqa: E402

PAR = [  # name, B, H, W, C0, C1, Cout, affine, bias_rows, skip, S0, S1
    ("basic", 1, 16, 16, 32, 0, 256, False, 0, False, 0, 0),
    ("bias", 2, 32, 16, 64, 0, 256, False, 1, False, 0, 0),
    ("aff_bias_skip", 2, 16, 32, 64, 0, 256, True, 2, True, 0, 0),
    ("concat", 1, 16, 16, 64, 32, 256, True, 1, True, 0, 0),
    ("edges", 1, 48, 32, 32, 0, 256, True, 1, True, 0, 0),
    ("deepk", 1, 16, 16, 256, 256, 256, True, 1, True, 0, 0),
    ("wide", 1, 16, 48, 32, 0, 256, True, 1, True, 0, 0),
    ("multi_tile", 3, 48, 80, 64, 0, 256, True, 3, True, 0, 0),
    ("raw_multi", 2, 32, 64, 96, 0, 256, False, 1, True, 0, 0),
    ("cat320", 1, 32, 32, 256, 64, 256, True, 1, False, 0, 0),
    ("shortcut", 2, 16, 16, 256, 0, 256, True, 1, False, 64, 0),
    ("shortcut_cat", 1, 32, 16, 128, 0, 256, True, 1, False, 128, 256),
]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def make_case(g, B, H, W, C0, C1, Cout, aff, brows, skip, S0, S1):
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
    return dict(x0=x0, x1=x1, w=w, wsc=wsc, C0=C0, S0=S0, affine=affine, bias=bias, skip=sk, scale=scale, sc0=sc0, sc1=sc1, ref=ref, Cout=Cout)


def run(c, algo):
    pw = ops.pack_conv_weight(c["w"], C0=c["C0"], dtype=torch.bfloat16, w_sc=c["wsc"], S0=c["S0"] if c["S0"] else None, winograd=algo)
    return ops.conv2d(c["x0"], pw, c["Cout"], 3, x1=c["x1"], affine=c["affine"], bias=c["bias"], skip=c["skip"], scale=c["scale"], sc0=c["sc0"],
                      sc1=c["sc1"], want_stats=True, winograd=algo)


def parity(names):
    g = torch.Generator(device="cuda").manual_seed(0)
    bad = 0
    for name, B, H, W, C0, C1, Cout, aff, brows, skip, S0, S1 in PAR:
        c = make_case(g, B, H, W, C0, C1, Cout, aff, brows, skip, S0, S1)
        if names and name not in names:
            continue
        try:
            ops.pack_conv_weight(c["w"], C0=C0, dtype=torch.bfloat16, w_sc=c["wsc"], S0=S0 if S0 else None, winograd=4)
        except RuntimeError:
            print(f"parity {name:16s} not supported by FD_WINOGRAD4 (skipped)", flush=True)
            continue
        res = {}
        for algo in (False, 4):
            out, st = run(c, algo)
            torch.cuda.synchronize()
            s = st.double().sum(1)[:, :Cout]
            rs = torch.stack([c["ref"].sum((1, 2)), (c["ref"] ** 2).sum((1, 2))], -1)
            res[algo] = (rel(out, c["ref"]), float((s - rs).abs().max() / rs.abs().max()), out)
        out2, _ = run(c, 4)
        same = bool((out2 == res[4][2]).all())
        # per-position error map helps when something is wrong
        ok = res[4][0] < 4e-3 and res[4][1] < 3e-3 and same
        bad += not ok
        print(f"parity {name:16s} direct err {res[False][0]:.2e} stats {res[False][1]:.2e} | wino4 err {res[4][0]:.2e} stats {res[4][1]:.2e} "
              f"deterministic {same} {'OK' if ok else 'FAIL'}", flush=True)
        if not ok and res[4][0] >= 4e-3:
            e = (res[4][2].double() - c["ref"]).abs()
            print("   err by image row   :", [f"{v:.2f}" for v in e.mean((0, 2, 3)).tolist()[:20]])
            print("   err by column      :", [f"{v:.2f}" for v in e.mean((0, 1, 3)).tolist()[:20]])
            print("   err by cout (/16)  :", [f"{v:.2f}" for v in e.mean((0, 1, 2)).reshape(-1, 16).mean(1).tolist()])
            print("   ref magnitude      :", f"{float(c['ref'].abs().mean()):.2f}")
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
]


def timing(B, iters, rounds, only, algos):
    g = torch.Generator(device="cuda").manual_seed(1)
    for i, (name, H, W, C0, C1, Cout, aff, skip, S) in enumerate(SHAPES):
        if only and i not in only:
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
        for algo in algos:
            try:
                pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None, winograd=algo)
            except RuntimeError:
                continue
            fs[algo] = (lambda pw=pw, algo=algo: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=affine, bias=bias, skip=sk, scale=0.7071, sc0=sc0,
                                                            sc1=sc1, want_stats=True, winograd=algo))
        best = {a: 1e9 for a in fs}
        for f in fs.values():
            f()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for algo, f in fs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    f()
                e1.record()
                torch.cuda.synchronize()
                best[algo] = min(best[algo], e0.elapsed_time(e1) / iters)
        label = {False: "direct", True: "wino2", 4: "wino4"}
        cells = " | ".join(f"{label[a]} {best[a]:7.3f} ms {fl / best[a] / 1e9:7.1f} TF" for a in fs)
        ratio = f"  wino4 x{best[False] / best[4]:.3f}" if False in fs and 4 in fs else ""
        print(f"time {i:2d} {name:26s} B={B} {cells}{ratio}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--cases", default="")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--with-wino2", action="store_true")
    a = ap.parse_args()
    bad = 0 if a.no_parity else parity([c for c in a.cases.split(",") if c])
    if not a.no_timing:
        timing(a.B, a.iters, a.rounds, [int(v) for v in a.only.split(",") if v], (False, True, 4) if a.with_wino2 else (False, 4))
    sys.exit(1 if bad else 0)
