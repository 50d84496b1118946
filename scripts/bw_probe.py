#!/usr/bin/env python
"""HBM ceilings as PyTorch's own kernels see them: fill (write only), copy (read + write), read-reduce (read only), 1.6 GB tensors."""
import torch
n = 800 * 1024 * 1024
x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = t(lambda: x.fill_(1.0)); print(f"fill  {2*n/ms/1e9:6.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy  {4*n/ms/1e9:6.2f} TB/s read+written")
ms = t(lambda: x.view(torch.int16).max()); print(f"max   {2*n/ms/1e9:6.2f} TB/s read")
