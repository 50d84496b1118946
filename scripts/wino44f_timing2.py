"""Per-workgroup phase split (s_memtime ticks) of the float32 Winograd kernels: 2-D F(4x4,3x3) (winograd=44) and F(4,3) (winograd=4), next to
the launch time of the direct f32 kernel.  Needs the -DFD_TIMING2 variant library: FLOWDEC_HIP_LIB=flowdec_amd/variants/libflowdec_t2f.so"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops, _lib as L
lib = L.load()
lib.fd_debug_buffer.argtypes = [C.c_void_p]; lib.fd_debug_buffer.restype = C.c_int
dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
lib.fd_debug_buffer(C.c_void_p(dbg.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.float32
for name, C0, C1, aff, skip, Cout in [("plain 256", 256, 0, 0, 0, 256), ("aff 256", 256, 0, 1, 0, 256), ("cat aff 512", 256, 256, 1, 0, 256),
                                      ("aff+skip 256", 256, 0, 1, 1, 256), ("aff 128", 128, 0, 1, 0, 128), ("aff+skip 128", 128, 0, 1, 1, 128)]:
    B, H, W = 8, 768, 256
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g)
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda", generator=g) / (9 * (C0 + C1)) ** 0.5
    A = torch.stack([1 + 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g), 0.1 * torch.randn(B, C0 + C1, device="cuda", generator=g)], -1).contiguous() if aff else None
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g) if skip else None
    for algo in (False, 4, 44):
        if algo == 4 and Cout != 256:
            continue          # (conv_wino4f.hip: 256-cout workgroups only)
        pw = ops.pack_conv_weight(w, C0=C0, dtype=dt, winograd=algo)
        f = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=A, skip=sk, scale=0.7, want_stats=True, winograd=algo)
        f(); torch.cuda.synchronize(); dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().reshape(8192, 8).double()[:6144]
        ms = e0.elapsed_time(e1)
        pro, loop, epi, mf, pr, pwait, pwork, mf1 = [d[:, k].mean().item() for k in range(8)]
        tag = {False: "direct f32", 4: "F(4,3) f32 ", 44: "F(4x4) f32 "}[algo]
        print(f"{name:14s} {tag} {ms:8.3f} ms | ticks: prologue {pro:8.0f} loop {loop:9.0f} epilogue {epi:8.0f}" + (f" | loop = MFMA phases {mf:9.0f} (wave 4, the SIMD partner of wave 0: {mf1:.0f}) + producer phases {pr:9.0f}; inside the producer steps of wave 0: waiting for the halo {pwait:.0f}, working {pwork:.0f}" if algo == 44 else ""), flush=True)
