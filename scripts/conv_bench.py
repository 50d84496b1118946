#!/usr/bin/env python
"""Micro-benchmark of the MFMA conv kernel at the layer shapes of FlowDec-75m (per-launch TFLOP/s)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops  # noqa: E402

SHAPES = [  # name, H, W, C0, C1, Cout, k, affine, skip
    ("L0 256->256 3x3 aff+skip", 768, 256, 256, 0, 256, 3, True, True),
    ("L0 256->256 3x3 plain", 768, 256, 256, 0, 256, 3, False, False),
    ("L0 512->256 3x3 cat aff", 768, 256, 256, 256, 256, 3, True, False),
    ("L0 320->256 3x3 cat aff", 768, 256, 256, 64, 256, 3, True, False),
    ("L0 64->256 3x3 aff", 768, 256, 64, 0, 256, 3, True, False),
    ("L1 256->256 3x3 aff+skip", 384, 128, 256, 0, 256, 3, True, True),
    ("L1 512->256 3x3 cat aff", 384, 128, 256, 256, 256, 3, True, False),
    ("L2 256->256 3x3 aff+skip", 192, 64, 256, 0, 256, 3, True, True),
    ("L3 128->128 3x3 aff+skip", 96, 32, 128, 0, 128, 3, True, True),
    ("L0 512->256 1x1 cat", 768, 256, 256, 256, 256, 1, False, False),
    ("L0 256->256 1x1", 768, 256, 256, 0, 256, 1, False, False),
    ("L0 256->4 3x3 head aff", 768, 256, 256, 0, 4, 3, True, True),
]
# ResBlock tail with the shortcut conv folded in: Conv_1(256->256 3x3, aff) + Conv_2(512->256 1x1)
FOLDED = [("L0 rb31 tail 256+sc512", 768, 256, 256, 512), ("L0 rb4 tail 256+sc64", 768, 256, 256, 64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stats", type=int, default=1)
    ap.add_argument("--operands", default="", choices=["", "mixed", "x3"], help="with --dtype fp32: bf16 / split-bf16 MFMA operands")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(0)
    for i, (name, H, W, C0, C1, Cout, k, aff, skip) in enumerate(SHAPES):
        if a.only >= 0 and i != a.only:
            continue
        B = a.B
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
        w = torch.randn(Cout, C0 + C1, k, k, device="cuda", generator=g) / ((C0 + C1) * k * k) ** 0.5
        opnd = {"": False, "mixed": True, "x3": "x3"}[a.operands]
        pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, bf16_operands=opnd)
        affine = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g),
                              0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
        bias = torch.randn(Cout, device="cuda", generator=g)
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(dt) if skip else None
        f = lambda: ops.conv2d(x0, pw, Cout, k, x1=x1, affine=affine, bias=bias, skip=sk, scale=0.7071, want_stats=bool(a.stats) and Cout > 4, bf16_operands=opnd)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * B * H * W * Cout * (C0 + C1) * k * k
        print(f"{i:2d} {name:28s} B={B} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    if a.only < 0:
        folded(a)


def folded(a):
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(1)
    for name, H, W, C, S in FOLDED:
        B = a.B
        x0 = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
        s0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).to(dt)
        s1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).to(dt) if S > 256 else None
        w = torch.randn(C, C, 3, 3, device="cuda", generator=g) / (9 * C) ** 0.5
        ws = torch.randn(C, S, 1, 1, device="cuda", generator=g) / S ** 0.5
        pw = ops.pack_conv_weight(w, dtype=dt, w_sc=ws, S0=min(S, 256))
        affine = torch.stack([1 + 0.1 * torch.randn(B, C, device="cuda", generator=g), 0.1 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
        bias = torch.randn(C, device="cuda", generator=g)
        f = lambda: ops.conv2d(x0, pw, C, 3, affine=affine, bias=bias, scale=0.7071, sc0=s0, sc1=s1, want_stats=True)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * B * H * W * C * (C * 9 + S)
        print(f"   {name:28s} B={B} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
