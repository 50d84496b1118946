"""Delta probes of the float32 F(4,3) kernel (conv_wino4f.hip): which input pixel / channel / tap arrives at which output position."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
B, H, W, Cin, Cout = 1, 16, 16, 32, 256
f32 = torch.float32
# 1. channel map: x = 1 in channel ci only, centre tap w[o, c] = c + 1 for every cout -> every output = ci + 1
w = torch.zeros(Cout, Cin, 3, 3, device="cuda")
for c in range(Cin):
    w[:, c, 1, 1] = c + 1
pw = ops.pack_conv_weight(w, C0=Cin, dtype=f32, winograd=4)
pd = ops.pack_conv_weight(w, C0=Cin, dtype=f32)
vals = []
for ci in range(Cin):
    x = torch.zeros(B, H, W, Cin, device="cuda"); x[..., ci] = 1.0
    out = ops.conv2d(x, pw, Cout, 3, winograd=4)
    outd = ops.conv2d(x, pd, Cout, 3)
    vals.append((ci, float(out[0, 8, 8, 0]), float(out[0, 8, 8, 200]), float(outd[0, 8, 8, 0]), float((out - outd).abs().max())))
print("channel probe (ci, wino out[co 0], out[co 200], direct, max|diff|):")
for v in vals:
    print("  ", v)
# 2. cout map: w[o, 0, centre] = o + 1, x = 1 in channel 0 -> out[..., o] = o + 1
w = torch.zeros(Cout, Cin, 3, 3, device="cuda"); w[:, 0, 1, 1] = torch.arange(1, Cout + 1, device="cuda").float()
pw = ops.pack_conv_weight(w, C0=Cin, dtype=f32, winograd=4)
x = torch.zeros(B, H, W, Cin, device="cuda"); x[..., 0] = 1.0
out = ops.conv2d(x, pw, Cout, 3, winograd=4)
print("cout probe: out[8, 8, :16] =", out[0, 8, 8, :16].tolist(), " out[8,8,120:136] =", out[0, 8, 8, 120:136].tolist())
print("cout probe ok:", bool((out[0, 8, 8] - torch.arange(1, Cout + 1, device='cuda')).abs().max() < 1e-3))
# 3. spatial / tap probes (position-coded input)
for (ci, co, dy, dx) in [(0, 0, 1, 1), (0, 0, 0, 1), (0, 0, 2, 1), (0, 0, 1, 0), (0, 0, 1, 2), (5, 37, 1, 1), (20, 200, 0, 0)]:
    x = torch.zeros(B, H, W, Cin, device="cuda")
    rr = torch.arange(H, device="cuda")[:, None].float(); cc = torch.arange(W, device="cuda")[None, :].float()
    x[0, :, :, ci] = (rr + 1) * 32 + (cc + 1)
    w = torch.zeros(Cout, Cin, 3, 3, device="cuda"); w[co, ci, dy, dx] = 1.0
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1)[0, co]
    pw = ops.pack_conv_weight(w, C0=Cin, dtype=f32, winograd=4)
    out = ops.conv2d(x, pw, Cout, 3, winograd=4)
    torch.cuda.synchronize()
    got = out[0, :, :, co]
    other = out[0].abs().sum() - got.abs().sum()
    print(f"--- ci {ci} co {co} tap ({dy},{dx}): max err {float((got - ref).abs().max()):.2f}, energy in other couts {float(other):.1f}")
    if float((got - ref).abs().max()) > 0.5:
        print("ref rows 0..5:"); print(ref[:6].int())
        print("got rows 0..5:"); print(got[:6])
