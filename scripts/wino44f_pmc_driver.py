"""Three launches each of the direct f32, F(4,3) f32 and 2-D F(4x4,3x3) f32 convolution on the affine 256 -> 256 shape (B = 8, 768 x 256): the
workload of scripts/pmc_wino44f.sh (rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, H, W, C, Cout = 8, 768, 256, 256, 256
x = torch.randn(B, H, W, C, device="cuda", generator=g)
w = torch.randn(Cout, C, 3, 3, device="cuda", generator=g) / (9 * C) ** 0.5
A = torch.stack([1 + 0.1 * torch.randn(B, C, device="cuda", generator=g), 0.1 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
for algo in (False, 4, 44):
    pw = ops.pack_conv_weight(w, C0=C, dtype=torch.float32, winograd=algo)
    for _ in range(3):
        ops.conv2d(x, pw, Cout, 3, affine=A, scale=0.7, want_stats=True, winograd=algo)
torch.cuda.synchronize()
