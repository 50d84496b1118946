"""Per-workgroup phase split (prologue / K loop / epilogue, s_memtime ticks) of the F(4,3) kernel and the direct kernel on the
full-resolution layer shapes.  Needs a -DFD_TIMING2 build: FLOWDEC_HIP_LIB=flowdec_amd/variants/libflowdec_t2.so (scripts/build_t2.sh)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
lib.fd_debug_buffer.argtypes = [C.c_void_p]; lib.fd_debug_buffer.restype = C.c_int
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib.fd_debug_buffer(C.c_void_p(dbg.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.bfloat16
for name, C0, C1, aff, skip in [("plain 256", 256, 0, 0, 0), ("cat aff 512", 256, 256, 1, 0), ("aff+skip 256", 256, 0, 1, 1), ("aff 64", 64, 0, 1, 0)]:
    B, H, W, Cout = 8, 768, 256, 256
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).to(dt)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).to(dt) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
    A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).to(dt) if skip else None
    for algo in (False, 4):
        pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, winograd=algo)
        f = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, skip=sk, scale=0.7, want_stats=True, winograd=algo)
        f(); torch.cuda.synchronize(); dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().reshape(8192, 8).double()[:6144]
        ms = e0.elapsed_time(e1)
        pro, loop, epi = [d[:, k].mean().item() for k in range(3)]
        tot = pro + loop + epi
        print(f"{name:14s} {'wino4 ' if algo else 'direct'} {ms:.3f} ms | per workgroup ticks: prologue {pro:7.0f} ({100*pro/tot:4.1f}%)  loop {loop:8.0f} ({100*loop/tot:4.1f}%)  "
              f"epilogue {epi:7.0f} ({100*epi/tot:4.1f}%)  total {tot:8.0f}" + (f" | round 0: transform+send {d[:, 3].mean():.0f} recv+combine {d[:, 4].mean():.0f} stage {d[:, 5].mean():.0f} sweep {d[:, 6].mean():.0f}, before {d[:, 7].mean():.0f}" if algo else ""), flush=True)
