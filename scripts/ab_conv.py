#!/usr/bin/env python
"""A/B of conv kernel builds: python scripts/ab_conv.py name=path.so ... [--rounds R].  Every (round, build) runs in its own
process (FLOWDEC_HIP_LIB), builds interleaved; reports the per-shape minimum over rounds and the ratio to the first build."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # name, H, W, C0, C1, Cout, k, affine, skip, S (folded shortcut channels)
    ("L0 256->256 aff+skip", 768, 256, 256, 0, 256, 3, True, True, 0),
    ("L0 256->256 plain", 768, 256, 256, 0, 256, 3, False, False, 0),
    ("L0 512->256 cat aff", 768, 256, 256, 256, 256, 3, True, False, 0),
    ("L0 320->256 cat aff", 768, 256, 256, 64, 256, 3, True, False, 0),
    ("L0 64->256 aff", 768, 256, 64, 0, 256, 3, True, False, 0),
    ("L0 rb31 tail 256+sc512", 768, 256, 256, 0, 256, 3, True, False, 512),
    ("L1 512->256 cat aff", 384, 128, 256, 256, 256, 3, True, False, 0),
    ("L2 256->256 aff+skip", 192, 64, 256, 0, 256, 3, True, True, 0),
    ("L0 256->4 head aff+skip", 768, 256, 256, 0, 4, 3, True, True, 0),
]


def worker(B, iters, reps):
    import torch
    sys.path.insert(0, ROOT)
    from flowdec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    out = {}
    for name, H, W, C0, C1, Cout, k, aff, skip, S in SHAPES:
        Cin = C0 + C1
        x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
        x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
        w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
        affine = torch.stack([1 + 0.1 * torch.randn(B, Cin, device="cuda", generator=g), 0.1 * torch.randn(B, Cin, device="cuda", generator=g)],
                             -1).contiguous() if aff else None
        bias = torch.randn(Cout, device="cuda", generator=g)
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if skip else None
        sc0 = sc1 = wsc = None
        if S:
            sc0 = torch.randn(B, H, W, min(S, 256), device="cuda", generator=g).bfloat16()
            sc1 = torch.randn(B, H, W, S - 256, device="cuda", generator=g).bfloat16() if S > 256 else None
            wsc = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
        pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=min(S, 256) if S else None)
        f = lambda: ops.conv2d(x0, pw, Cout, k, x1=x1, affine=affine, bias=bias, skip=sk, scale=0.7071, sc0=sc0, sc1=sc1, want_stats=Cout > 4)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        out[name] = best
    print("AB_RESULT " + json.dumps(out), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
        return
    rounds, B, iters, reps = 2, 8, 20, 3
    libs = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        elif a == "--iters":
            iters = int(args.pop(0))
        else:
            n, p = a.split("=", 1)
            libs.append((n, os.path.abspath(p)))
    res = {n: {} for n, _ in libs}
    for r in range(rounds):
        for n, p in libs:
            env = dict(os.environ, FLOWDEC_HIP_LIB=p)
            o = subprocess.run([sys.executable, __file__, "--worker", str(B), str(iters), str(reps)], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("AB_RESULT ")]
            if not line:
                print(f"{n}: worker failed\n{o.stderr[-1500:]}")
                continue
            for k, v in json.loads(line[0][10:]).items():
                res[n][k] = min(res[n].get(k, 1e9), v)
    base = libs[0][0]
    print(f"{'shape':26s} " + " ".join(f"{n:>16s}" for n, _ in libs))
    for name, H, W, C0, C1, Cout, k, *_rest in SHAPES:
        S = _rest[2]
        fl = 2.0 * B * H * W * Cout * ((C0 + C1) * k * k + S)
        cells = []
        for n, _ in libs:
            ms = res[n].get(name)
            cells.append("      failed    " if ms is None else f"{ms:6.3f}ms x{res[base][name] / ms:5.3f}")
        print(f"{name:26s} " + " ".join(cells) + f"   ({fl / res[base][name] / 1e9:6.0f} TF base)")


if __name__ == "__main__":
    main()
