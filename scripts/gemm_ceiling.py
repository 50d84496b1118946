#!/usr/bin/env python
"""Practical MFMA ceiling of this box: hipBLASLt/rocBLAS bf16 and f32 GEMM throughput through torch.matmul (plain library
GEMM, no fusion) at sizes comparable to one full-resolution conv layer (M = 1.57 M pixels, N = 256, K = 2304) and at 8192^3."""
import torch, time
def run(M, N, K, dt, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=dt); b = torch.randn(K, N, device="cuda", dtype=dt)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{str(dt):15s} M={M} N={N} K={K}: {ms:8.3f} ms  {2*M*N*K/ms/1e9:8.1f} TFLOP/s")
for dt in (torch.bfloat16, torch.float32):
    run(8192, 8192, 8192, dt)
    run(1572864, 256, 2304, dt)
    run(1572864, 256, 4608, dt)

# HBM: device-to-device copy of 4 GiB (read + write) and a pure write (fill)
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
for name, fn, bytes_ in (("copy (r+w)", lambda: b.copy_(a), 8 * n), ("fill (w)", lambda: a.fill_(1.0), 4 * n)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"HBM {name}: {bytes_ * 10 / e0.elapsed_time(e1) / 1e9:.2f} TB/s")
