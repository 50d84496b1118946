#!/usr/bin/env python
"""Micro-benchmark of the fused FIR resampling kernels at the FlowDec-75m level shapes (B = 8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
g = torch.Generator(device="cuda").manual_seed(0)
for variant in (0,):
    for (H, W) in ((768, 256), (384, 128)):
        B, C = 8, 256
        x = torch.randn(B, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
        A = torch.stack([1 + 0.1 * torch.randn(B, C, device="cuda", generator=g), 0.1 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
        for direction in (-1, +1):
            if direction > 0 and (variant > 0 or H == 768):
                continue
            f = lambda: ops.fir_resample(x, direction, affine=A)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            inb = x.numel() * 2; outb = 2 * inb * (4 if direction > 0 else 0.25)
            print(f"variant {variant} dir {direction:+d} {H}x{W}: {ms*1e3:8.1f} us  {(inb + outb) / ms / 1e9:7.2f} TB/s(alg)")
