#!/usr/bin/env python
"""bench.py -- FlowDec-75m inference throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (FlowModel.enhance: STFT -> 6 x NCSN++ inside the Euler solver ->
iSTFT) over one batch of synthetic clips that is already resident in HBM.  Workload at N = 1 is
BASELINE.json configs[1]: FlowDec-75m, batch = 8 x 2 s clips @ 48 kHz, 6-step Euler, bf16 operands.
With --gpus N every rank processes its own batch of 8 clips (batch sharding, no data-path collective;
weak scaling) and `value` is the whole-job audio-seconds per wall-second.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r01_conv_traffic.json"
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline():
    """Times the NumPy oracle (a port of the reference's fp32 CPU path) on a bounded sample of the same workload: ONE
    vector-field evaluation of the full-width FlowDec-75m NCSN++ on the shape of BASELINE config 1 (one 1 s clip = 126
    frames padded to 768 x 128), extrapolated to the 6 NFE of the benchmark config (the STFT/iSTFT are < 0.1 %)."""
    import numpy as np
    from oracle import flowdec_oracle as O
    rng = np.random.default_rng(0)
    net = O.NCSNppOracle(O.random_state_dict(seed=64, nf=64), nf=64)
    shape = (1, 1, 768, 128)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    y = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    t0 = time.perf_counter()
    net.forward(x, y, np.array([0.5], np.float32))
    dt = time.perf_counter() - t0
    audio_s = 1.0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": audio_s / (6 * dt), "unit": "audio-seconds/second", "cores": int(threads), "kind": "port",
            "sample": "oracle (NumPy/OpenBLAS fp32 port of the reference CPU path): 1 NFE of full-width NCSN++ on one 1 s clip "
                      f"(BASELINE config 1 shape, 768x128 frames) took {dt:.2f} s; x6 NFE extrapolated to the Euler N=6 config"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=2.0, help="clip length")
    ap.add_argument("--N", type=int, default=6)
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--preset", default="flowdec_75m")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import flowdec_amd
    from flowdec_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # synthetic data + seeded random-init weights of the FlowDec-75m architecture (no checkpoints offline)
    model = flowdec_amd.from_preset(args.preset, precision=args.precision)
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for k, v in model.state_dict().items():
        if not k.startswith("backbone."):
            continue
        if k.endswith(".W"):
            sd[k] = torch.randn(v.shape, generator=g) * 16.0
        elif v.ndim == 1 and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) / fan_in ** 0.5
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)

    B, Lw = args.batch, int(round(args.seconds * 48000))
    gen = torch.Generator(device=dev).manual_seed(rank)
    y = 0.1 * torch.randn(B, 1, Lw, device=dev, generator=gen)
    lib = L.load()
    T = lib.fd_num_frames(Lw, 384); Tp = lib.fd_padded_frames(T)
    noise = torch.randn(B, 1, 768, Tp, dtype=torch.complex64, device=dev, generator=gen)
    nfe = {"euler": args.N, "midpoint": 2 * args.N, "heun2": 2 * args.N, "heun2_eulerlast": 2 * args.N - 1, "dopri5": None}[args.solver]

    def step():
        return model.enhance(y, N=args.N, solver=args.solver, noise=noise, use_graph=not args.no_graph)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if nfe is None:
        nfe = model.last_nfe   # adaptive solver: realised number of vector-field evaluations of the last step
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    audio_seconds = world * B * args.seconds * args.steps
    result = {
        "metric": "48 kHz audio-seconds/sec (RTF) for FlowDec-75m @ 6 ODE steps",
        "value": audio_seconds / elapsed, "unit": "audio-seconds/second", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_nfe": 1e3 * elapsed / args.steps / nfe, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (0.1*randn waveforms, seeded random-init weights of the FlowDec-75m architecture)",
        "config": {"workload": f"{args.preset} enhance(): batch={B} x {args.seconds:g} s clips @48 kHz per GPU, {args.N}-step {args.solver} "
                               f"(NFE {nfe}), T_pad={Tp} frames, inputs resident in HBM", "global_batch": world * B, "nfe": nfe,
                   "parallelism": f"batch-shard x{world}", "hipgraph": not args.no_graph},
    }

    if rank == 0 and world == 1 and not args.no_roofline:
        # dominant kernel = the MFMA implicit-GEMM convolution: time every launch of one extra (eager) step with
        # HIP events on the launch stream; algorithmic FLOPs = 2 * pixels * Cout * Cin * k^2 per launch.
        h = model.backbone.handle()
        L.check(lib.fd_profile_enable(h, 1))
        model.enhance(y, N=args.N, solver=args.solver, noise=noise, use_graph=False)
        torch.cuda.synchronize(dev)
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        L.check(lib.fd_profile_read(h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        L.check(lib.fd_profile_enable(h, 0))
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.precision]
        nl = max(int(n.value), 1)
        result["roofline"] = {"bound": "mfma", "kernel": "conv_mfma_kernel (implicit-GEMM 3x3/1x1)", "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches": int(n.value),
                              "avg_launch_ms": ms.value / nl, "conv_ms_per_step": ms.value, "algorithmic_tflop_per_step": fl.value / 1e12,
                              "algorithmic_tflop_per_launch": fl.value / 1e12 / nl, "algorithmic_bytes_per_launch": by.value / nl,
                              "hbm_GBps_algorithmic": by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0}
        # HBM bytes per launch from the committed PMC passes of this same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # corrected as profiles/summarize_pmc.py documents); PMC counters cannot be collected from inside the timed run.
        default_cfg = (args.preset, args.precision, args.solver, args.N, B, args.seconds) == ("flowdec_75m", "bf16", "euler", 6, 8, 2.0)
        tf = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if default_cfg and os.path.exists(tf):
            with open(tf) as f:
                tr = json.load(f)["conv_mfma_kernel"]
            if tr["launches_per_step"] == int(n.value):
                result["roofline"]["traffic"] = tr["traffic_per_launch"]
                result["roofline"]["traffic_source"] = f"profiles/{TRAFFIC_FILE} (bytes/launch: fetch {tr['fetch_corrected_per_launch']:.4g} + write {tr['write_per_launch']:.4g})"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
