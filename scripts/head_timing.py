"""Launch time of the pyramid-head convolution (GroupNorm + SiLU + conv3x3 C -> 4 + pyramid add, conv_head.hip) at the levels of a cfg 2 step
(8 clips of 2 s), next to its HBM floor (the activation read once at 8 TB/s) and its MFMA floor (32 x 32 x 16 tiles with 4 real couts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B = int(os.environ.get("HEAD_B", 8))
DT = torch.float32 if os.environ.get("HEAD_DT", "bf16") == "fp32" else torch.bfloat16   # fp32: conv_headf.hip (v_mfma_f32_4x4x1)
for (H, W, C) in [(768, 256, 128), (384, 128, 256), (192, 64, 256), (96, 32, 256)]:
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(DT)
    w = torch.randn(4, C, 3, 3, device="cuda", generator=g) / (9 * C) ** 0.5
    A = torch.stack([1 + 0.1 * torch.randn(B, C, device="cuda", generator=g), 0.1 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
    sk = torch.randn(B, H, W, 4, device="cuda", generator=g).to(DT)
    bias = torch.randn(4, device="cuda", generator=g)
    pw = ops.pack_conv_weight(w, C0=C, dtype=DT)
    f = lambda: ops.conv2d(x, pw, 4, 3, affine=A, skip=sk, bias=bias)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = B * H * W * C * x.element_size() / 1e6
    mfma_us = ((B * H * W / 32) * 9 * (C / 16) / 1024 * 32 if DT == torch.bfloat16 else (B * H * W / 64) * 9 * C / 1024 * 8) / 2.0e3   # MFMAs per SIMD x cycles at 2.0 GHz (bf16: 32x32x16 tiles with 4 real couts; fp32: 4x4x1 x 16 blocks)
    print(f"B {B}  {H:4d} x {W:3d} x {C:3d} -> 4: {us:8.1f} us | input {mb:7.1f} MB = {mb / us / 1e3 * 1e3:6.2f} TB/s ({mb / 8e6 * 1e6:6.1f} us at 8 TB/s) | MFMA floor {mfma_us:6.1f} us", flush=True)
