#!/usr/bin/env python
"""Per-launch time of ONE image per resolution level (B = 1 x 1 s clip: T_pad = 128) for every workgroup width of the direct
kernel and for the Winograd kernel: the data behind the FD_LOW_LATENCY tile rule (model.hip: Fwd::latency_tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowdec_amd import ops  # noqa: E402

SHAPES = [  # name, H, W, C0, C1, Cout, skip, S
    ("L0 256->256", 768, 128, 256, 0, 256, True, 0),
    ("L0 512->256 cat", 768, 128, 256, 256, 256, False, 0),
    ("L0 64->256", 768, 128, 64, 0, 256, False, 0),
    ("L0 256->256 +sc512", 768, 128, 256, 0, 256, False, 512),
    ("L1 256->256", 384, 64, 256, 0, 256, True, 0),
    ("L1 512->256 cat", 384, 64, 256, 256, 256, False, 0),
    ("L2 256->256", 192, 32, 256, 0, 256, True, 0),
    ("L2 384->256 cat", 192, 32, 128, 256, 256, False, 0),
    ("L3 128->128", 96, 16, 128, 0, 128, True, 0),
    ("L3 384->128 cat", 96, 16, 128, 256, 128, False, 0),
    ("L3 256->128", 96, 16, 256, 0, 128, False, 0),
]


def main(B=1, iters=50, rounds=3, scale_w=1):
    g = torch.Generator(device="cuda").manual_seed(1)
    print(f"B = {B}; per-launch microseconds (min over {rounds} rounds of {iters} back-to-back launches)")
    print(f"{'shape':22s} {'bn256':>8s} {'duo':>8s} {'bn128':>8s} {'bn64':>8s} {'bn64c':>8s} {'bn32':>8s} {'bn32c':>8s} {'wino':>8s}")
    for name, H, W, C0, C1, Cout, skip, S in SHAPES:
        W *= scale_w
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
        w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
        affine = torch.stack([1 + 0.1 * torch.randn(B, Cin, device="cuda", generator=g), 0.1 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous()
        bias = torch.randn(Cout, device="cuda", generator=g)
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if skip else None
        sc0 = sc1 = wsc = None
        if S:
            sc0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).bfloat16()
            sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).bfloat16() if S > 256 else None
            wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
        from flowdec_amd import _lib as L
        lib = L.load()
        fs, outs = {}, {}
        for key in (256, "duo", 128, 64, "64c", 32, "32c", "wino"):
            wino = key == "wino"
            if not wino and not isinstance(key, str) and key > Cout:
                continue
            pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None, winograd=wino)
            out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda")
            st = torch.zeros(B, lib.fd_conv_stats_tiles(H, W), lib.fd_conv_cout_pad(Cout), 2, dtype=torch.float32, device="cuda")
            flags = L.dtype_id(torch.bfloat16) | (L.FD_WINOGRAD if wino else 0) | L.FD_TILE[0 if wino or key == 256 else key]
            outs[key] = (out, st, pw)

            def f(pw=pw, out=out, st=st, flags=flags):   # the bare C call: no allocation, no fill kernel
                L.check(lib.fd_conv2d(L.ptr(x0), C0, L.ptr(x1), C1, L.ptr(affine), L.ptr(sc0), min(S, 256), L.ptr(sc1), max(S - 256, 0), L.ptr(pw),
                                      L.ptr(bias), 1, L.ptr(sk), 0.7071, L.ptr(out), Cout, L.ptr(st), B, H, W, 3, flags, L.stream()))
            fs[key] = f
        ref = None
        best = {k: 1e9 for k in fs}
        graphs = {}
        for k, f in fs.items():
            f()
            torch.cuda.synchronize()
            out, st, _ = outs[k]
            if k != "wino":
                if ref is None:
                    ref = (out.clone(), st.clone())
                else:
                    if not os.environ.get("FD_NO_CHECK"):
                        assert torch.equal(out, ref[0]), (name, k)
                        assert torch.allclose(st, ref[1], rtol=1e-4, atol=1e-3), (name, k)
            gr = torch.cuda.CUDAGraph()   # `iters` launches in one graph: the per-launch time is not bounded by the Python call
            with torch.cuda.graph(gr):
                for _ in range(iters):
                    f()
            graphs[k] = gr
        torch.cuda.synchronize()
        for _ in range(rounds):
            for k, gr in graphs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                gr.replay()
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                best[k] = min(best[k], e0.elapsed_time(e1) / iters * 1e3)
        print(f"{name:22s} " + " ".join(f"{best[k]:8.1f}" if k in best else f"{'-':>8s}" for k in (256, "duo", 128, 64, "64c", 32, "32c", "wino")), flush=True)


if __name__ == "__main__":
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 1, scale_w=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
