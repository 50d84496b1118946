#!/usr/bin/env python
"""One clip at full resolution (B = 1, 768 x 128): 384 workgroups of 256 output channels = 1.5 waves on 256 CUs.  Does a narrower
workgroup (2 x 384 = 3 full waves of half the work) or the two-workgroups-per-CU tile balance better?  min over 3 x 50 launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ab_conv import SHAPES
from flowdec_amd import ops
g = torch.Generator(device="cuda").manual_seed(1)
B = int(os.environ.get("AB_B", "1")); Wd = int(os.environ.get("AB_W", "128")); Hd = int(os.environ.get("AB_H", "768"))
for name, H, W, C0, C1, Cout, k, aff, skip, S in SHAPES:
    if Cout < 128 or not name.startswith("L0"):
        continue
    W = Wd; H = Hd
    Cin = C0 + C1
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    affine = torch.stack([1 + 0.1 * torch.randn(B, Cin, device="cuda", generator=g), 0.1 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous() if aff else None
    bias = torch.randn(Cout, device="cuda", generator=g)
    skp = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if skip else None
    sc0 = sc1 = wsc = None
    if S:
        sc0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).bfloat16()
        sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).bfloat16() if S > 256 else None
        wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None)
    cells = []
    base = None
    for tile in (0, 128, "duo", 64, "64c", 32):
        try:
            f = lambda: ops.conv2d(x0, pw, Cout, k, x1=x1, affine=affine, bias=bias, skip=skp, scale=0.7071, sc0=sc0, sc1=sc1, want_stats=True, tile_bn=tile)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    f()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 50)
            base = base or best
            cells.append(f"{tile}: {best * 1e3:7.1f} us x{base / best:5.3f}")
        except Exception as e:
            cells.append(f"{tile}: {type(e).__name__}")
    print(f"{name:26s} " + "  ".join(cells), flush=True)
