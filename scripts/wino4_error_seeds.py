#!/usr/bin/env python
"""bf16-mode waveform error of enhance (Euler-6, nf = 64 seeded weights, one 0.5 s clip) against the fp32 mode, per convolution
algorithm, over several (clip, noise) draws and two weight seeds: is an algorithm's error systematically larger, or is the spread the
trajectory's sensitivity to the error direction?"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flowdec_amd  # noqa: E402
from oracle import flowdec_oracle as O  # noqa: E402  (weights generator only)


def model(prec, algo, seed):
    m = flowdec_amd.from_preset("flowdec_75m", precision=prec, nf=64, conv_algo=algo)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=seed, nf=64).items()}, strict=False)
    return m.cuda()


def rel(a, b):
    return float((a - b).double().pow(2).sum().sqrt() / b.double().pow(2).sum().sqrt())


algos = ["direct", "winograd_lowres", "auto", "winograd"]
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
tot = {a: [] for a in algos}
for wseed in (int(np.load(os.path.join(ROOT, "tests/golden/g17_enhance_nf64.npz"))["seed"]), 7):
    ref = model("fp32", "direct", wseed)
    ms = {a: model("bf16", a, wseed) for a in algos}
    for s in range(4):
        g = torch.Generator(device="cuda").manual_seed(100 + s)
        y = 0.1 * torch.randn(1, 1, L, device="cuda", generator=g)
        Tp = 64 * ((1 + L // 384 + 63) // 64)
        nz = torch.randn(1, 1, 768, Tp, dtype=torch.complex64, device="cuda", generator=g)
        w = ref.enhance(y, N=6, solver="euler", noise=nz)
        line = f"weights {wseed} draw {s}:"
        for a in algos:
            e = rel(ms[a].enhance(y, N=6, solver="euler", noise=nz), w)
            tot[a].append(e)
            line += f"  {a} {e:.3e}"
        print(line, flush=True)
print("mean:", "  ".join(f"{a} {np.mean(tot[a]):.3e}" for a in algos))
print("rms :", "  ".join(f"{a} {np.sqrt(np.mean(np.square(tot[a]))):.3e}" for a in algos))
