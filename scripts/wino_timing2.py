import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
lib.fd_debug_buffer.argtypes = [C.c_void_p]; lib.fd_debug_buffer.restype = C.c_int
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib.fd_debug_buffer(C.c_void_p(dbg.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
for name, C0, C1, aff, skip, S in [("plain 256", 256, 0, 0, 0, 0), ("cat aff 512", 256, 256, 1, 0, 0), ("aff+skip 256", 256, 0, 1, 1, 0), ("aff 256 + sc512", 256, 0, 1, 0, 512)]:
    B, H, W, Cout = 8, 768, 256, 256
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
    sc0 = sc1 = wsc = None
    if S:
        sc0 = torch.randn(B, H, W, 256, device="cuda", generator=g).to(dt); sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).to(dt)
        wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, w_sc=wsc, S0=256 if S else None, winograd=True)
    A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(dt) if skip else None
    f = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, skip=sk, scale=0.7, sc0=sc0, sc1=sc1, want_stats=True, winograd=True)
    f(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().reshape(8192, 8).double()
    d = d[:8192]
    ms = e0.elapsed_time(e1)
    pro, loop, epi, l3, l1 = [d[:, k].mean().item() for k in range(5)]
    tot = pro + loop + epi
    print(f"{name:16s} {ms:.3f} ms | per workgroup ticks: prologue {pro:7.0f} ({100*pro/tot:4.1f}%)  loop {loop:8.0f} ({100*loop/tot:4.1f}%; 3x3 {l3:.0f}, shortcut {l1:.0f})  epilogue {epi:7.0f} ({100*epi/tot:4.1f}%)")
