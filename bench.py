#!/usr/bin/env python
"""bench.py -- FlowDec-75m inference throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (FlowModel.enhance: STFT -> 6 x NCSN++ inside the Euler solver -> iSTFT) over one
batch of synthetic clips that is already resident in HBM.  Workload at N = 1 is BASELINE.json configs[1]: FlowDec-75m,
batch = 8 x 2 s clips @ 48 kHz, 6-step Euler, bf16.

Multi-GPU: one process per GPU, the GLOBAL batch (default 8 clips per GPU = weak scaling; `--global-batch G`
fixes the total = strong scaling, e.g. 256 for BASELINE config 4) is sharded by clip with `flowdec_amd.dist.sharded_apply`,
whose all-gather of the output waveforms over RCCL/xGMI is the only collective and sits INSIDE the timed region.  `value` is
the whole-job audio-seconds per wall-second (max over ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N > 1: re-launches itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_FILE = "r02_conv_traffic.json"
REF_CPU_FILE = "r02_reference_cpu_timing.json"
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "mixed": 2500.0, "bf16x3": 2500.0 / 3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


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
    sample = ("oracle (NumPy/OpenBLAS fp32 port of the reference CPU path): 1 NFE of full-width NCSN++ on one 1 s clip "
              f"(BASELINE config 1 shape, 768x128 frames) took {dt:.2f} s; x6 NFE extrapolated to the Euler N=6 config")
    ref = os.path.join(ROOT, "profiles", REF_CPU_FILE)
    if os.path.exists(ref):   # the reference implementation itself, measured in the build container (it cannot travel to this box)
        with open(ref) as f:
            r = json.load(f)
        sample += (f".  Reference implementation (PyTorch CPU, real FlowModel.enhance, config 1, {r['threads']} threads, build container, "
                   f"profiles/{REF_CPU_FILE}): {r['seconds_per_enhance']:.1f} s per enhance = {r['audio_seconds_per_second']:.4f} audio-s/s")
    return {"value": audio_s / (6 * dt), "unit": "audio-seconds/second", "cores": int(threads), "kind": "port", "sample": sample}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch(args):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N local ranks (one per GPU)."""
    if args.backend == "nccl" and not args.share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible -- refusing to fall back to fewer ranks")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=None, help="total clips, sharded over the GPUs (strong scaling; overrides --batch)")
    ap.add_argument("--seconds", type=float, default=2.0, help="clip length")
    ap.add_argument("--N", type=int, default=6)
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "mixed", "bf16x3"])
    ap.add_argument("--conv-algo", default="auto", choices=["direct", "winograd", "winograd_lowres", "auto", "latency"])
    ap.add_argument("--preset", default="flowdec_75m")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --stub-step: launcher / collective test on CPU")
    ap.add_argument("--stub-step", action="store_true", help="replace enhance() by a trivial CPU function (tests of the launcher only)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo: exercises the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0:
        if args.gpus > 1:
            relaunch(args)   # does not return
        world = 1
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    from flowdec_amd.dist import shard_range, sharded_apply
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(0 if args.share_gpu else local_rank)
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world

    gbatch = args.global_batch if args.global_batch is not None else args.batch * world
    Lw = int(round(args.seconds * 48000))
    lo, hi = shard_range(gbatch, rank, world)

    if args.stub_step:
        dev = torch.device("cpu")
        y = 0.1 * torch.randn(gbatch, 1, Lw, generator=torch.Generator().manual_seed(0))
        nfe, Tp, model, noise = args.N, 0, None, None

        def local_fn(yb):
            return yb * 2.0 + 1.0
    else:
        import flowdec_amd
        from flowdec_amd import _lib as L
        dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
        torch.cuda.set_device(dev)
        # synthetic data + seeded random-init weights of the FlowDec-75m architecture (no checkpoints offline)
        model = flowdec_amd.from_preset(args.preset, precision=args.precision, conv_algo=args.conv_algo)
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
        # the same global batch on every rank (seeded), resident in HBM before the timed region; each rank works on its slice
        gen = torch.Generator(device=dev).manual_seed(0)
        y = 0.1 * torch.randn(gbatch, 1, Lw, device=dev, generator=gen)
        lib = L.load()
        T = lib.fd_num_frames(Lw, 384); Tp = lib.fd_padded_frames(T)
        noise = torch.randn(gbatch, 1, 768, Tp, dtype=torch.complex64, device=dev, generator=gen)[lo:hi].contiguous()
        nfe = {"euler": args.N, "midpoint": 2 * args.N, "heun2": 2 * args.N, "heun2_eulerlast": 2 * args.N - 1, "dopri5": None}[args.solver]

        def local_fn(yb):
            return model.enhance(yb, N=args.N, solver=args.solver, noise=noise, use_graph=not args.no_graph)

    t_gather = [0.0]

    def step():
        if dist is None:
            return local_fn(y)
        t0 = time.perf_counter()
        local = local_fn(y[lo:hi])
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        out = sharded_apply(lambda _yb: local, y)      # all-gather of the enhanced waveforms (RCCL over xGMI / gloo)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t_gather[0] += time.perf_counter() - t1
        del t0
        return out

    def sync_all():
        if dist is not None:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t_gather[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    assert out.shape[0] == gbatch and torch.isfinite(out).all()
    if nfe is None:
        nfe = model.last_nfe   # adaptive solver: realised number of vector-field evaluations of the last step
    per_rank_ms = [1e3 * elapsed / args.steps]
    gather_ms = 1e3 * t_gather[0] / args.steps
    if dist is not None:
        tt = torch.tensor([elapsed, t_gather[0]], device=dev, dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank_ms = [1e3 * float(t[0]) / args.steps for t in allt]
        gather_ms = 1e3 * max(float(t[1]) for t in allt) / args.steps
        elapsed = max(float(t[0]) for t in allt)

    audio_seconds = gbatch * args.seconds * args.steps
    result = {
        "metric": "48 kHz audio-seconds/sec (RTF) for FlowDec-75m @ 6 ODE steps",
        "value": audio_seconds / elapsed, "unit": "audio-seconds/second", "n_gpus": world if dist is None else dist.get_world_size(),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_nfe": 1e3 * elapsed / args.steps / nfe,
        "higher_is_better": True, "scaling": "strong" if args.global_batch is not None else "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (0.1*randn waveforms, seeded random-init weights of the FlowDec-75m architecture)",
        "config": {"workload": f"{args.preset} enhance(): global batch={gbatch} x {args.seconds:g} s clips @48 kHz ({hi - lo} per GPU), {args.N}-step "
                               f"{args.solver} (NFE {nfe}), T_pad={Tp} frames, inputs resident in HBM, output waveforms all-gathered",
                   "global_batch": gbatch, "nfe": nfe, "parallelism": f"batch-shard x{world}", "hipgraph": not args.no_graph,
                   "conv_algo": args.conv_algo, "backend": args.backend if world > 1 else None},
        "per_rank_ms_per_step": per_rank_ms, "allgather_ms_per_step": gather_ms if world > 1 else 0.0,
    }
    if args.stub_step:
        result["config"]["workload"] = "STUB step (launcher / collective test, no model)"

    if rank == 0 and world == 1 and not args.no_roofline and not args.stub_step:
        from flowdec_amd import _lib as L
        # dominant kernel = the MFMA convolution: time every launch of one extra (eager) step with HIP events on the launch
        # stream; algorithmic FLOPs = 2 * pixels * Cout * Cin * k^2 per launch (the DIRECT convolution's count,
        # also when the Winograd kernel runs -- `achieved` is then an effective rate).
        h = model.backbone.handle()
        L.check(lib.fd_profile_enable(h, 1))
        model.enhance(y, N=args.N, solver=args.solver, noise=noise, use_graph=False)
        torch.cuda.synchronize(dev)
        ms, n, fl, by = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
        L.check(lib.fd_profile_read(h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        fms, fn_, fby = C.c_double(), C.c_longlong(), C.c_double()
        L.check(lib.fd_profile_read_fir(h, C.byref(fms), C.byref(fn_), C.byref(fby)))
        L.check(lib.fd_profile_enable(h, 0))
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.precision]
        nl = max(int(n.value), 1)
        result["roofline"] = {"bound": "mfma", "kernel": "conv_mfma_kernel / conv_wino_kernel / conv_head_kernel (implicit-GEMM 3x3/1x1)", "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches": int(n.value),
                              "avg_launch_ms": ms.value / nl, "conv_ms_per_step": ms.value, "algorithmic_tflop_per_step": fl.value / 1e12,
                              "algorithmic_tflop_per_launch": fl.value / 1e12 / nl, "algorithmic_bytes_per_launch": by.value / nl,
                              "hbm_GBps_algorithmic": by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0}
        fnl = max(int(fn_.value), 1)
        fach = fby.value / (fms.value * 1e-3) / 1e9 if fms.value > 0 else 0.0
        result["roofline_hbm"] = {"bound": "hbm", "kernel": "fir_up_kernel / fir_down_kernel (FIR x2 resampling, fused GroupNorm+SiLU dual output)",
                                  "achieved": fach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fach / HBM_PEAK_GBPS, "traffic": None,
                                  "launches": int(fn_.value), "avg_launch_ms": fms.value / fnl, "ms_per_step": fms.value,
                                  "algorithmic_bytes_per_launch": fby.value / fnl}
        # HBM bytes per launch from the committed PMC passes of this same workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # corrected as profiles/summarize_pmc.py documents); PMC counters cannot be collected from inside the timed run.
        default_cfg = (args.preset, args.precision, args.solver, args.N, gbatch, args.seconds, args.conv_algo) == \
            ("flowdec_75m", "bf16", "euler", 6, 8, 2.0, "auto")
        tf = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if default_cfg and os.path.exists(tf):
            with open(tf) as f:
                tr = json.load(f)
            for key, name in (("roofline", "conv_mfma_kernel"), ("roofline_hbm", "fir_kernels")):
                t = tr.get(name)
                if t and t["launches_per_step"] == result[key]["launches"]:
                    result[key]["traffic"] = t["traffic_per_launch"]
                    result[key]["traffic_source"] = (f"profiles/{TRAFFIC_FILE} (bytes/launch: fetch {t['fetch_corrected_per_launch']:.4g} + "
                                                     f"write {t['write_per_launch']:.4g})")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub_step:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
