"""Delta-weight probes of the F(4,3) kernel: which input pixel / channel arrives at which output position."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
B, H, W, Cin, Cout = 1, 16, 16, 32, 256
for (ci, co, dy, dx) in [(0, 0, 1, 1), (0, 0, 0, 1), (0, 0, 2, 1), (0, 0, 1, 0), (0, 0, 1, 2), (9, 37, 1, 1), (20, 200, 0, 0)]:
    x = torch.zeros(B, H, W, Cin, device="cuda")
    rr = torch.arange(H, device="cuda")[:, None].float(); cc = torch.arange(W, device="cuda")[None, :].float()
    x[0, :, :, ci] = (rr + 1) * 32 + (cc + 1)
    w = torch.zeros(Cout, Cin, 3, 3, device="cuda"); w[co, ci, dy, dx] = 1.0
    xb = x.bfloat16()
    ref = torch.nn.functional.conv2d(xb.float().permute(0, 3, 1, 2), w, padding=1)[0, co]
    pw = ops.pack_conv_weight(w, C0=Cin, dtype=torch.bfloat16, winograd=4)
    out = ops.conv2d(xb, pw, Cout, 3, winograd=4).float()
    torch.cuda.synchronize()
    got = out[0, :, :, co]
    other = out[0].abs().sum() - got.abs().sum()
    print(f"--- ci {ci} co {co} tap ({dy},{dx}): max err {float((got - ref).abs().max()):.1f}, energy in other couts {float(other):.1f}")
    if float((got - ref).abs().max()) > 1:
        print("ref rows 0..7 (row*32+col coded):"); print(ref[:8].int())
        print("got rows 0..7:"); print(got[:8].int())
        print("got rows 8..15:"); print(got[8:].int())
