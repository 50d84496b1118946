#!/usr/bin/env python
"""Per-launch time of fd_conv_in (the vector-FMA input convolution 4 -> 64) at BASELINE cfg 2's shape, next to the MFMA path on the
zero-padded weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops  # noqa: E402

B, H, W, C = 8, 768, 256, 64
g = torch.Generator(device="cuda").manual_seed(0)
in8 = torch.randn(B, H, W, 8, device="cuda", generator=g).bfloat16()
w = torch.randn(C, 4, 3, 3, device="cuda", generator=g) / 6
bias = torch.randn(C, device="cuda", generator=g)
w8 = torch.zeros(C, 8, 3, 3, device="cuda"); w8[:, :4] = w
pw = ops.pack_conv_weight(w8, dtype=torch.bfloat16)
fs = {"conv_in": lambda: ops.conv_in(in8, w, bias), "mfma": lambda: ops.conv2d(in8, pw, C, 3, bias=bias, want_stats=True)}
for name, f in fs.items():
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{name:8s} {best * 1e3:7.1f} us  ({(in8.numel() * 2 + B * H * W * C * 2) / best / 1e6:.0f} GB/s algorithmic)")
