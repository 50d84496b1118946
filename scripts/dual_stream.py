#!/usr/bin/env python
"""Experiment: process the batch as two half-batches on two streams (two model replicas) so that kernels of different
layers overlap (tails, small kernels, epilogue write bursts)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowdec_amd
from flowdec_amd import _lib as L
dev = torch.device("cuda", 0)
def mk():
    m = flowdec_amd.from_preset("flowdec_75m", precision="bf16")
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for k, v in m.state_dict().items():
        if not k.startswith("backbone."): continue
        if k.endswith(".W"): sd[k] = torch.randn(v.shape, generator=g) * 16.0
        elif v.ndim == 1 and k.endswith("weight"): sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"): sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        else: sd[k] = torch.randn(v.shape, generator=g) / v[0].numel() ** 0.5
    m.load_state_dict(sd, strict=False)
    return m.to(dev)
B, Lw = 8, 96000
gen = torch.Generator(device=dev).manual_seed(0)
y = 0.1 * torch.randn(B, 1, Lw, device=dev, generator=gen)
lib = L.load(); Tp = lib.fd_padded_frames(lib.fd_num_frames(Lw, 384))
noise = torch.randn(B, 1, 768, Tp, dtype=torch.complex64, device=dev, generator=gen)
m0 = mk()
def timeit(fn, n=5, w=2):
    for _ in range(w): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
ms1, ref = timeit(lambda: m0.enhance(y, N=6, solver="euler", noise=noise))
print(f"single stream B=8: {ms1:.1f} ms/step  {B*2/ms1*1e3:.1f}x")
for parts in (2, 4):
    ms_ = [m0] + [mk() for _ in range(parts - 1)]
    sts = [torch.cuda.Stream(dev) for _ in range(parts)]
    hb = B // parts
    def dual():
        cur = torch.cuda.current_stream(dev)
        outs = []
        for i in range(parts):
            sts[i].wait_stream(cur)
            with torch.cuda.stream(sts[i]):
                outs.append(ms_[i].enhance(y[i*hb:(i+1)*hb], N=6, solver="euler", noise=noise[i*hb:(i+1)*hb]))
        for i in range(parts): cur.wait_stream(sts[i])
        return torch.cat(outs)
    ms2, out = timeit(dual)
    print(f"{parts} streams x B={hb}: {ms2:.1f} ms/step  {B*2/ms2*1e3:.1f}x   identical={torch.equal(out, ref)}")
