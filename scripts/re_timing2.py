"""Per-workgroup phase split of the persistent register-epilogue conv configuration (-DFD_TIMING2 build; s_memtime ticks, wave 0):
K loop | register epilogue | statistics barrier + combine, summed over the tiles of a workgroup.
    FLOWDEC_HIP_LIB=flowdec_amd/variants/libflowdec_t2.so python scripts/re_timing2.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
lib.fd_debug_buffer.argtypes = [C.c_void_p]; lib.fd_debug_buffer.restype = C.c_int
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib.fd_debug_buffer(C.c_void_p(dbg.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
for name, C0, C1, aff, S in [("plain 256", 256, 0, 0, 0), ("cat aff 512", 256, 256, 1, 0), ("aff 256", 256, 0, 1, 0), ("aff 256 + sc512", 256, 0, 1, 512)]:
    B, H, W, Cout = 8, 768, 256, 256
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
    sc0 = sc1 = wsc = None
    if S:
        sc0 = torch.randn(B, H, W, 256, device="cuda", generator=g).to(dt); sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).to(dt)
        wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, w_sc=wsc, S0=256 if S else None)
    A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
    f = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, scale=0.7, sc0=sc0, sc1=sc1, want_stats=True, tile_bn='persist')
    for _ in range(3):
        f()
    torch.cuda.synchronize(); dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().reshape(8192, 8).double()
    d = d[d[:, 4] > 0]
    ms = e0.elapsed_time(e1)
    pro, loop, epi, stat, tiles, tot = [d[:, k].mean().item() for k in range(6)]
    tt = d[:, 5]
    xcd = [tt[(torch.arange(len(d)) % 8) == k].mean().item() for k in range(8)] if len(d) >= 8 else []
    print(f"{name:16s} per-workgroup total ticks: min {tt.min().item():.0f}  p10 {tt.quantile(0.1).item():.0f}  mean {tt.mean().item():.0f}  p90 {tt.quantile(0.9).item():.0f}  "
          f"max {tt.max().item():.0f}  (max / mean = {tt.max().item() / tt.mean().item():.3f}); by blockIdx % 8: " + " ".join(f"{v / tt.mean().item():.3f}" for v in xcd))
    kl = d[:, 1] / d[:, 4]
    print(f"{'':16s} per-workgroup K-loop ticks per tile: min {kl.min().item():.0f} mean {kl.mean().item():.0f} max {kl.max().item():.0f}")
    print(f"{name:16s} {ms:.3f} ms | {len(d)} workgroups x {tiles:.1f} tiles; per TILE: K loop {loop / tiles:8.0f} ({100 * loop / tot:4.1f}%)  reg. epilogue {epi / tiles:6.0f} "
          f"({100 * epi / tot:4.1f}%)  stats barrier+combine {stat / tiles:6.0f} ({100 * stat / tot:4.1f}%)  | once: prologue {pro:6.0f} ({100 * pro / tot:4.1f}%), "
          f"total {tot:9.0f} ticks, unaccounted (transitions) {100 * (tot - pro - loop - epi - stat) / tot:4.1f}%")
