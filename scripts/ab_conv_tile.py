#!/usr/bin/env python
"""A/B of tile configurations of ONE library on the shapes of scripts/ab_conv.py: python scripts/ab_conv_tile.py [lib.so ...]
columns: default (staged epilogue) and tile_bn='persist' (register epilogue, continuous tiles) per library; min over 3 x 20 launches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from ab_conv import SHAPES


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from flowdec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    out = {}
    B = 8
    for name, H, W, C0, C1, Cout, k, aff, skip, S in SHAPES:
        if skip or Cout < 128:
            continue
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
        w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
        affine = torch.stack([1 + 0.1 * torch.randn(B, Cin, device="cuda", generator=g), 0.1 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous() if aff else None
        bias = torch.randn(Cout, device="cuda", generator=g)
        sc0 = sc1 = wsc = None
        if S:
            sc0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).bfloat16()
            sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).bfloat16() if S > 256 else None
            wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
        pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None)
        for tile in (0, "persist"):
            f = lambda: ops.conv2d(x0, pw, Cout, k, x1=x1, affine=affine, bias=bias, scale=0.7071, sc0=sc0, sc1=sc1, want_stats=True, tile_bn=tile)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    f()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            out[f"{name}|{tile}"] = best
    print("AB_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(); sys.exit(0)
    libs = sys.argv[1:] or [os.path.join(ROOT, "flowdec_amd", "libflowdec_hip.so")]
    res = {}
    for rnd in range(2):
        for lib in libs:
            o = subprocess.run([sys.executable, __file__, "--worker"], env=dict(os.environ, FLOWDEC_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("AB_RESULT ")]
            if not line:
                print(lib, "failed", o.stderr[-800:]); continue
            for k, v in json.loads(line[0][10:]).items():
                key = (os.path.basename(lib), k)
                res[key] = min(res.get(key, 1e9), v)
    names = sorted({k.split("|")[0] for (_, k) in res})
    for n in names:
        cells = []
        base = None
        for lib in libs:
            for tile in ("0", "persist"):
                v = res.get((os.path.basename(lib), f"{n}|{tile}"))
                if base is None:
                    base = v
                cells.append(f"{os.path.basename(lib)[11:-3]}:{tile[:4]} {v:6.3f}ms x{base / v:5.3f}")
        print(f"{n:26s} " + "  ".join(cells))
