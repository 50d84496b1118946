#!/usr/bin/env python
"""What a hipGraph capture of enhance() costs and what a replay saves: wall time of the first call of a shape
(runs eagerly), the second (capture + instantiate + launch) and the following ones (replay), against use_graph=False; then a
file-by-file run over 50 distinct clip lengths (every call a first sighting: nothing is captured)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flowdec_amd  # noqa: E402


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    m = flowdec_amd.from_preset("flowdec_75m", precision="bf16").cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    for B, sec in ((1, 1.0), (8, 2.0)):
        L = int(48000 * sec)
        y = 0.1 * torch.randn(B, 1, L, device="cuda", generator=g)
        noise = None
        run = lambda ug: m.enhance(y, N=6, solver="euler", generator=g, use_graph=ug)
        run(False)   # warm-up: workspace, packed weights
        eager = min(timed(lambda: run(False)) for _ in range(3))
        first = timed(lambda: run(True))
        second = timed(lambda: run(True))
        replay = min(timed(lambda: run(True)) for _ in range(3))
        print(f"B={B} x {sec} s: eager {eager:.2f} ms | graph: first sighting (eager) {first:.2f} ms, second (capture + instantiate + launch) {second:.2f} ms, "
              f"replay {replay:.2f} ms -> capture costs {second - replay:.1f} ms, a replay saves {eager - replay:.2f} ms", flush=True)
    lens = [48000 + 977 * i for i in range(50)]
    ys = [0.1 * torch.randn(1, 1, n, device="cuda", generator=g) for n in lens]
    m.enhance(ys[0], N=6, solver="euler", generator=g, use_graph=False)
    t_e = timed(lambda: [m.enhance(v, N=6, solver="euler", generator=g, use_graph=False) for v in ys])
    t_g = timed(lambda: [m.enhance(v, N=6, solver="euler", generator=g, use_graph=True) for v in ys])
    print(f"50 files of distinct lengths (1.00-2.00 s): use_graph=False {t_e:.1f} ms, use_graph=True {t_g:.1f} ms (every call a first sighting: runs eagerly, "
          f"nothing captured)")


if __name__ == "__main__":
    main()
