import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
lib.fd_debug_buffer.argtypes = [C.c_void_p]; lib.fd_debug_buffer.restype = C.c_int
dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
lib.fd_debug_buffer(C.c_void_p(dbg.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
for name, C0, C1, aff, skip in [("plain 256", 256, 0, 0, 0), ("cat aff 512", 256, 256, 1, 0), ("aff+skip 256", 256, 0, 1, 1)]:
    B, H, W, Cout = 8, 768, 256, 256
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=dt)
    A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(dt) if skip else None
    f = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, skip=sk, scale=0.7, want_stats=True)
    f(); torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().tolist(); nw = max(d[5], 1)
    ms = e0.elapsed_time(e1)
    tot = (d[0] + d[1] + d[2]) / nw
    blocks = B * (H // 16) * (W // 16)
    # wall cycles per block slot: kernel time * clock / (blocks / 256 CUs)
    print(f"{name:14s} {ms:.3f} ms | per wave: prologue {d[0]/nw:8.0f}  loop {d[1]/nw:9.0f}  epilogue {d[2]/nw:8.0f}  sum {tot:9.0f} ticks | "
          f"kernel/rounds = {ms * 1e-3 / (blocks / 256) * 1e6:.2f} us per block slot")
