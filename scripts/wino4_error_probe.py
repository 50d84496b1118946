#!/usr/bin/env python
"""Where does the bf16-mode waveform error of `enhance` come from, per convolution algorithm?  On the G17 workload (nf = 64, one
0.5 s clip) a Python Euler loop around `model.forward` follows the FP32 trajectory and evaluates every bf16 algorithm at the SAME
states (per-step forward error), then lets every algorithm run its own trajectory (final spectrogram error)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flowdec_amd  # noqa: E402
from oracle import flowdec_oracle as O  # noqa: E402  (weights generator only)


def model(prec, algo):
    m = flowdec_amd.from_preset("flowdec_75m", precision=prec, nf=64, conv_algo=algo)
    g = np.load(os.path.join(ROOT, "tests/golden/g17_enhance_nf64.npz"))
    sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=int(g["seed"]), nf=64).items()}
    m.load_state_dict(sd, strict=False)
    return m.cuda()


def rel(a, b):
    return float((a - b).abs().double().pow(2).sum().sqrt() / b.abs().double().pow(2).sum().sqrt())


def main():
    g = np.load(os.path.join(ROOT, "tests/golden/g17_enhance_nf64.npz"))
    y = torch.from_numpy(g["y"]).cuda()
    noise = torch.from_numpy(g["noise"]).cuda()
    ref = model("fp32", "direct")
    algos = [a for a in sys.argv[1:]] or ["direct", "winograd_lowres", "auto", "winograd"]
    ms = {a: model("bf16", a) for a in algos}
    Y, _, info = ref._preprocess(y)
    sig = torch.from_numpy(g["sigma_y"]).cuda().reshape(1, 1, -1, 1).float()
    x = Y + sig * noise
    N = 6
    ts = torch.linspace(0, 1, N + 1)
    xs = {a: x.clone() for a in algos}
    errs = {a: [] for a in algos}
    for k in range(N):
        t = ts[k:k + 1].cuda()
        dt = float(ts[k + 1] - ts[k])
        v = ref(x, Y, t)
        line = f"step {k} t={float(t):.3f} |v|={float(v.abs().pow(2).mean().sqrt()):.3f}"
        for a in algos:
            va = ms[a](x, Y, t)
            errs[a].append((va - v).flatten())
            line += f" | {a} fwd {rel(va, v):.3e} (mean err {float((va - v).mean().abs()):.2e})"
            xs[a] = xs[a] + dt * ms[a](xs[a], Y, t)
        if k in (0, 3) and os.environ.get("MAPS"):
            for a in algos:
                e = (ms[a](x, Y, t) - v).abs().pow(2)[0, 0]      # [768, T]
                s2 = v.abs().pow(2).mean()
                print(f"   {a:16s} err by column (x1e3):", " ".join(f"{float(c):.1f}" for c in (e.mean(0) / s2).sqrt() * 1e3))
                print(f"   {a:16s} err by 16-row band  :", " ".join(f"{float(c):.1f}" for c in (e.reshape(48, 16, -1).mean((1, 2)) / s2).sqrt() * 1e3))
                print(f"   {a:16s} err by row mod 16   :", " ".join(f"{float(c):.1f}" for c in (e.reshape(48, 16, -1).mean((0, 2)) / s2).sqrt() * 1e3))
        x = x + dt * v
        print(line + " || own-trajectory: " + " ".join(f"{a} {rel(xs[a], x):.3e}" for a in algos), flush=True)
    for a in algos:
        E = torch.stack(errs[a])
        n = E.abs().pow(2).sum(1).sqrt()
        C = (E.conj() @ E.T).real / (n[:, None] * n[None, :])
        print(f"{a:16s} coherence of the per-step forward errors: cos(e_k, e_k+1) =", " ".join(f"{float(C[k, k + 1]):.2f}" for k in range(N - 1)),
              f"| cos(e_0, e_5) = {float(C[0, 5]):.2f} | |sum e| / sqrt(sum |e|^2) = {float(E.sum(0).abs().pow(2).sum().sqrt() / n.pow(2).sum().sqrt()):.2f}")
    w = ref._postprocess(x, info)
    for a in algos:
        print(f"{a:16s} waveform err vs fp32 {rel(ref._postprocess(xs[a], info), w):.3e}")


if __name__ == "__main__":
    main()
