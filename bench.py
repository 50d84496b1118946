#!/usr/bin/env python
"""bench.py -- FlowDec-75m inference throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (FlowModel.enhance: STFT -> 6 x NCSN++ inside the Euler solver -> iSTFT) over one
batch of synthetic clips.  Workload at N = 1 is BASELINE.json configs[1]: FlowDec-75m, batch = 8 x 2 s clips @ 48 kHz, 6-step
Euler, bf16.  `value` is timed with the inputs already resident in HBM (the bench contract: "inputs already resident in HBM when
the timed region starts ... the PCIe-inclusive rate is never `value`"); the same K steps are then timed again from PINNED HOST
waveforms to host waveforms ("H2D of waveform -> D2H of waveform") and reported next to it as `e2e` -- the two differ
by the 2 x 6 MB of PCIe copies (< 1 ms of ~120).  The initial noise of every clip is drawn INSIDE the timed call, as the reference
does (flowdec/model.py:512 `_get_noise`): `sharded_enhance(..., seed=)`, one Philox stream per (step, global clip).

`--config cfg2|cfg3|cfg4|cfg5|cfg3_nfe6|cfg4_nfe6|cfg5_dopri5` selects a BASELINE.json configuration by name (explicit flags given after
it still apply): cfg 4 = FlowDec-75m, 32 x 2 s clips PER GPU, midpoint N = 6 (12 evaluations: the reference counts solver steps,
flowdec/model.py:487), bf16 -- under `--gpus 8` that is the 256-clip batch of config 4; `cfg4_nfe6` = the N = 3 (NFE 6) reading.

Multi-GPU: one process per GPU, the GLOBAL batch (default 8 clips per GPU = weak scaling; `--global-batch G`
fixes the total = strong scaling, e.g. 256 for BASELINE config 4) is sharded by clip with `flowdec_amd.dist.sharded_enhance`
-- the API a user calls -- whose single all_gather_into_tensor of the output waveforms over RCCL/xGMI is the only collective
and sits INSIDE the timed region.  `value` is the whole-job audio-seconds per wall-second (max over ranks).

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

TRAFFIC_FILE = "r06_conv_traffic.json"
REF_CPU_FILE = "r02_reference_cpu_timing.json"
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32 MFMA (the exact-f32 DFT GEMMs of the STFT front / back end)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "mixed": 2500.0, "bf16x3": 2500.0 / 3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def cpu_baseline():
    """The same-box CPU baseline: ALL of BASELINE config 1 (FlowDec-75m, one 1 s clip @ 48 kHz, 6-step Euler, fp32: STFT ->
    6 x full-width NCSN++ -> iSTFT, nothing extrapolated) through oracle/flowdec_oracle_torch.py -- the oracle's restatement of
    the graph and the solver on PyTorch's CPU kernels (oneDNN conv / GroupNorm, pocketfft STFT), i.e. on the library kernels the
    reference's own CPU path runs on (the reference's Python cannot travel to this box).  One untimed 1-NFE warm-up (thread
    pool, oneDNN primitive cache), then one timed enhance."""
    import numpy as np
    import torch
    from oracle import flowdec_oracle as O
    from oracle import flowdec_oracle_torch as OT
    rng = np.random.default_rng(0)
    net = OT.NCSNppTorchCPU(O.random_state_dict(seed=64, nf=64), nf=64)
    y = (0.1 * rng.standard_normal((1, 1, 48000))).astype(np.float32)
    Tp = O.padded_frames(O.num_frames(48000))
    noise = ((rng.standard_normal((1, 1, 768, Tp)) + 1j * rng.standard_normal((1, 1, 768, Tp))) / np.sqrt(2)).astype(np.complex64)
    # thread count: torch's default (= physical cores) is not the fastest on a many-core host (measured on the round-3 box, EPYC 9575F:
    # 128 threads 32.6 s per enhance); one warm 1-NFE probe per candidate, the full run with the fastest
    probe = {}
    for nt in sorted({torch.get_num_threads(), 64, 32, 16, 8} & set(range(1, torch.get_num_threads() + 1)), reverse=True):
        torch.set_num_threads(nt)
        OT.enhance(net, y, noise, 0.66, N=1, solver="euler")          # warm-up: thread pool, oneDNN primitive cache
        t0 = time.perf_counter()
        OT.enhance(net, y, noise, 0.66, N=1, solver="euler")
        probe[nt] = time.perf_counter() - t0
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    out = OT.enhance(net, y, noise, 0.66, N=6, solver="euler")
    dt = time.perf_counter() - t0
    assert np.isfinite(out).all()
    cpu = "?"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "?")
    except OSError:
        pass
    sample = (f"BASELINE config 1 in full (1 clip x 1 s, 6-step Euler = 6 NFE of the full-width NCSN++ + STFT/iSTFT, fp32), oracle on PyTorch CPU "
              f"kernels (oracle/flowdec_oracle_torch.py): {dt:.2f} s per enhance on {threads} threads of {os.cpu_count()} logical cores ({cpu}); "
              f"1-NFE probe per thread count: " + ", ".join(f"{k}: {v:.2f} s" for k, v in sorted(probe.items())))
    ref = os.path.join(ROOT, "profiles", REF_CPU_FILE)
    if os.path.exists(ref):   # the reference implementation itself, measured in the build container (it cannot travel to this box)
        with open(ref) as f:
            r = json.load(f)
        sample += (f".  Reference implementation itself (real FlowModel.enhance, PyTorch CPU, config 1, {r['threads']} threads, build container, "
                   f"profiles/{REF_CPU_FILE}): {r['seconds_per_enhance']:.1f} s per enhance = {r['audio_seconds_per_second']:.4f} audio-s/s")
    return {"value": 1.0 / dt, "unit": "audio-seconds/second", "cores": int(threads), "kind": "port", "sample": sample,
            "seconds_per_enhance": dt, "cpu_model": cpu}


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


# BASELINE.json configurations by name: what `--config` sets (flags given explicitly on the command line win)
CONFIGS = {
    # "6-step midpoint": the reference counts SOLVER STEPS -- enhance(N=6, solver='midpoint') evaluates the network 2 N = 12 times
    # (flowdec/model.py:487) -- so cfg3 / cfg4 are N = 6.  The notebook's "NFE 6" operating point (demo.ipynb cell 3: N = 3) stays
    # available under the explicit names *_nfe6 (cfg3_n6 = the former name of today's cfg3).
    "cfg2": dict(preset="flowdec_75m", batch=8, seconds=2.0, N=6, solver="euler", precision="bf16"),
    "cfg3": dict(preset="flowdec_25s", batch=32, seconds=2.0, N=6, solver="midpoint", precision="bf16"),
    "cfg3_n6": dict(preset="flowdec_25s", batch=32, seconds=2.0, N=6, solver="midpoint", precision="bf16"),
    "cfg3_nfe6": dict(preset="flowdec_25s", batch=32, seconds=2.0, N=3, solver="midpoint", precision="bf16"),
    "cfg4": dict(preset="flowdec_75m", batch=32, seconds=2.0, N=6, solver="midpoint", precision="bf16"),       # x 8 GPUs = 256 clips
    "cfg4_n6": dict(preset="flowdec_75m", batch=32, seconds=2.0, N=6, solver="midpoint", precision="bf16"),
    "cfg4_nfe6": dict(preset="flowdec_75m", batch=32, seconds=2.0, N=3, solver="midpoint", precision="bf16"),
    # "32-step adaptive": cfg5 = the fixed-step reading (32 Euler steps); cfg5_dopri5 = torchdyn's adaptive dopri5 over the 33-point
    # t_span at the solver's default tolerances (realised NFE reported; minutes per step in fp32 -- scripts/bench_cfg5_dopri5.py)
    "cfg5": dict(preset="flowdec_75m", batch=8, seconds=4.0, N=32, solver="euler", precision="fp32"),          # x 8 GPUs = 64 clips
    "cfg5_dopri5": dict(preset="flowdec_75m", batch=8, seconds=4.0, N=32, solver="dopri5", precision="fp32"),
}


def main():
    ap = argparse.ArgumentParser(allow_abbrev=False)   # (an abbreviated flag would slip past the --config override test below)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="a BASELINE.json configuration by name (per-GPU shard; see CONFIGS)")
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
    ap.add_argument("--no-calibration", action="store_true", help="skip the ~0.3 s box calibration (fd_calibrate_mfma) and the clock / power sampling")
    ap.add_argument("--no-e2e", action="store_true", help="skip the second timed loop (pinned host waveforms in -> host waveforms out)")
    ap.add_argument("--no-side-stream", action="store_true", help="FD_NO_SIDE_STREAM: keep the side branches on the launch stream")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --stub-step: launcher / collective test on CPU")
    ap.add_argument("--stub-step", action="store_true", help="replace enhance() by a trivial CPU function (tests of the launcher only)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (with --backend gloo: exercises the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()
    if args.config:
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for k, v in CONFIGS[args.config].items():
            if "--" + k.replace("_", "-") not in given and "--" + k not in given:
                setattr(args, k, v)

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
    from flowdec_amd.dist import shard_range, shard_sizes, sharded_apply, sharded_enhance
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
    stats = {} if world > 1 else None      # per-rank local / gather split (synchronises per step; the N = 1 loop stays asynchronous)

    if args.stub_step:
        dev = torch.device("cpu")
        y = 0.1 * torch.randn(gbatch, 1, Lw, generator=torch.Generator().manual_seed(0))
        y_host = None
        nfe = {"euler": args.N, "midpoint": 2 * args.N, "heun2": 2 * args.N, "heun2_eulerlast": 2 * args.N - 1, "dopri5": args.N}[args.solver]
        Tp, model, noise = 0, None, None

        def step(src, k=0):
            t1 = time.perf_counter()
            out = sharded_apply(lambda yb: yb * 2.0 + 1.0, src)
            if stats is not None:
                stats["gather_s"] = stats.get("gather_s", 0.0) + (time.perf_counter() - t1)
            return out
    else:
        import flowdec_amd
        from flowdec_amd import _lib as L
        dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
        torch.cuda.set_device(dev)
        # synthetic data + seeded random-init weights of the FlowDec-75m architecture (no checkpoints offline)
        model = flowdec_amd.from_preset(args.preset, precision=args.precision, conv_algo=args.conv_algo, side_stream=not args.no_side_stream)
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
        # the same global batch on every rank (seeded), resident in HBM before the timed region (and once more in pinned host memory
        # for the e2e loop); each rank enhances its rows.  The initial noise is a GLOBAL tensor indexed by clip, so the result does
        # not depend on the number of ranks.
        gen = torch.Generator(device=dev).manual_seed(0)
        y = 0.1 * torch.randn(gbatch, 1, Lw, device=dev, generator=gen)
        y_host = y.cpu().pin_memory()
        lib = L.load()
        T = lib.fd_num_frames(Lw, 384); Tp = lib.fd_padded_frames(T)
        nfe = {"euler": args.N, "midpoint": 2 * args.N, "heun2": 2 * args.N, "heun2_eulerlast": 2 * args.N - 1, "dopri5": None}[args.solver]

        def step(src, k=0):   # src: the global batch, on the device (timed `value`) or in pinned host memory (`e2e`); k: step index
            # the initial noise is drawn inside the call (model.py:512): clip i of step k has its own stream (seed 1000 + k, i)
            return sharded_enhance(model, src, N=args.N, solver=args.solver, seed=1000 + k, use_graph=not args.no_graph, stats=stats)

    def sync_all():
        if dist is not None:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def timed(src, steps):
        sync_all()
        if stats is not None:
            stats.clear()
        t0 = time.perf_counter()
        for k in range(steps):
            out = step(src, k)
        sync_all()
        el = time.perf_counter() - t0
        gs = stats.get("gather_s", 0.0) if stats is not None else 0.0
        per_rank = [1e3 * el / steps]
        if dist is not None:
            tt = torch.tensor([el, gs], device=dev, dtype=torch.float64)
            allt = [torch.empty_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            per_rank = [1e3 * float(t[0]) / steps for t in allt]
            gs = max(float(t[1]) for t in allt)
            el = max(float(t[0]) for t in allt)
        return out, el, per_rank, 1e3 * gs / steps

    # Box calibration (round 6): the boxes of the pool differ by +-2.5 % and the chip runs the step at its power limit, so the line
    # carries what THIS box sustains on a fixed matrix-core loop (+ clock / power while it runs) and the clock / power of the timed
    # region itself; `value_per_calibration` = value / mfma_tflops is the figure to compare across boxes and rounds.
    calib = power = None
    probe = dev.type == "cuda" and not args.no_calibration and not args.stub_step
    if probe:
        from flowdec_amd import boxprobe
        try:
            calib = boxprobe.calibrate(dev)
        except Exception as e:   # never lose a bench line to the probe
            calib = {"error": repr(e)}
    for _ in range(args.warmup):
        out = step(y)
    if probe:
        with boxprobe.PowerSampler(dev.index or 0, period_s=0.1) as sampler:
            out, elapsed, per_rank_ms, gather_ms = timed(y, args.steps)
        power = sampler.summary()
    else:
        out, elapsed, per_rank_ms, gather_ms = timed(y, args.steps)
    assert out.shape[0] == gbatch and out.device == y.device and torch.isfinite(out).all()
    if nfe is None:
        nfe = model.last_nfe   # adaptive solver: realised number of vector-field evaluations of the last step

    audio_seconds = gbatch * args.seconds * args.steps
    result = {
        "metric": "48 kHz audio-seconds/sec (RTF) for FlowDec-75m @ 6 ODE steps",
        "value": audio_seconds / elapsed, "unit": "audio-seconds/second", "n_gpus": world if dist is None else dist.get_world_size(),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_nfe": 1e3 * elapsed / args.steps / nfe,
        "higher_is_better": True, "scaling": "strong" if args.global_batch is not None else "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic (0.1*randn waveforms, seeded random-init weights of the FlowDec-75m architecture)",
        "config": {"workload": f"{args.preset} enhance(): global batch={gbatch} x {args.seconds:g} s clips @48 kHz ({hi - lo} per GPU), {args.N}-step "
                               f"{args.solver} (NFE {nfe}), T_pad={Tp} frames, inputs resident in HBM, initial noise drawn inside the timed call, "
                               f"output waveforms all-gathered" + (f" [--config {args.config}]" if args.config else ""),
                   "global_batch": gbatch, "clips_per_rank": shard_sizes(gbatch, world), "nfe": nfe, "parallelism": f"batch-shard x{world}",
                   "hipgraph": not args.no_graph, "conv_algo": args.conv_algo, "backend": args.backend if world > 1 else None},
        "per_rank_ms_per_step": per_rank_ms, "allgather_ms_per_step": gather_ms if world > 1 else 0.0,
    }
    if calib is not None:
        result["box_calibration"] = calib
        result["power"] = power
        if calib.get("mfma_tflops"):
            result["value_per_calibration"] = result["value"] / world / calib["mfma_tflops"]   # per-GPU audio-s/s per calibration TFLOP/s
    if dev.type == "cuda" and not args.stub_step:
        # which devices the ranks really ran on (the first SCALE record must show N DISTINCT GPUs behind the RCCL world)
        pr = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "device": str(dev), "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
              "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
              "mfma_tflops": (calib or {}).get("mfma_tflops")}
        devs = [me]
        if dist is not None:
            devs = [None] * world
            dist.all_gather_object(devs, me)
        result["devices"] = devs
        result["rccl_world"] = dist.get_world_size() if (dist is not None and args.backend == "nccl") else (1 if dist is None else 0)
        result["distinct_devices"] = len({(d["uuid"], d["pci"]) for d in devs})
    if args.stub_step:
        result["config"]["workload"] = "STUB step (launcher / collective test, no model)"
    elif not args.no_e2e:
        # the same K steps from pinned host memory to host memory: H2D of this rank's rows, solve, gather, D2H
        out_h = step(y_host, args.steps - 1)
        assert out_h.device.type == "cpu" and torch.equal(out_h, out.cpu()), "e2e result differs from the HBM-resident one"
        _, el2, _, _ = timed(y_host, args.steps)
        result["e2e"] = {"value": audio_seconds / el2, "unit": "audio-seconds/second", "ms_per_step": 1e3 * el2 / args.steps,
                         "path": f"pinned host float32 [{gbatch}, 1, {Lw}] -> H2D -> enhance -> all-gather -> D2H host float32 (same K steps, same clock)",
                         "pcie_bytes_per_step": 2 * 4 * (hi - lo) * Lw if world == 1 else 4 * ((hi - lo) + gbatch) * Lw}

    if rank == 0 and world == 1 and not args.no_roofline and not args.stub_step:
        from flowdec_amd import _lib as L
        # dominant kernel = the MFMA convolution: time every launch of one extra (eager) step with HIP events on the launch
        # stream; algorithmic FLOPs = 2 * pixels * Cout * Cin * k^2 per launch (the DIRECT convolution's count,
        # also when the Winograd kernel runs -- `achieved` is then an effective rate).
        h = model.backbone.handle()
        L.check(lib.fd_profile_enable(h, 1))
        model.enhance(y, N=args.N, solver=args.solver, generator=gen, use_graph=False)
        torch.cuda.synchronize(dev)
        ms, n, fl, by, flx = C.c_double(), C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
        L.check(lib.fd_profile_read_executed(h, C.byref(flx)))
        L.check(lib.fd_profile_read(h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        fms, fn_, fby = C.c_double(), C.c_longlong(), C.c_double()
        L.check(lib.fd_profile_read_fir(h, C.byref(fms), C.byref(fn_), C.byref(fby)))
        sms, scalls = (C.c_double * 6)(), (C.c_int * 2)()
        L.check(lib.fd_profile_read_stft(h, C.byref(sms), C.byref(scalls)))
        L.check(lib.fd_profile_enable(h, 0))
        achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.precision]
        nl = max(int(n.value), 1)
        result["roofline"] = {"bound": "mfma", "kernel": "conv_wino4_kernel / conv_mfma_kernel / conv_wino_kernel / conv_head_kernel (3x3/1x1 convolutions: Winograd F(4,3) and direct implicit GEMM)", "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches": int(n.value),
                              "avg_launch_ms": ms.value / nl, "conv_ms_per_step": ms.value, "algorithmic_tflop_per_step": fl.value / 1e12,
                              "algorithmic_tflop_per_launch": fl.value / 1e12 / nl, "algorithmic_bytes_per_launch": by.value / nl,
                              # what the matrix cores really executed (Winograd launches execute 1/2 resp. 2/3 of the direct count of their
                              # 3x3 part): `achieved` / `frac` above are EFFECTIVE rates on the direct convolution's count
                              "executed_tflop_per_step": flx.value / 1e12,
                              "executed_TFLOPs": flx.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0,
                              "executed_frac_of_peak": flx.value / (ms.value * 1e-3) / 1e12 / peak if ms.value > 0 else 0.0,
                              "hbm_GBps_algorithmic": by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0}
        fnl = max(int(fn_.value), 1)
        fach = fby.value / (fms.value * 1e-3) / 1e9 if fms.value > 0 else 0.0
        result["roofline_hbm"] = {"bound": "hbm", "kernel": "fir_up_kernel / fir_down_kernel (FIR x2 resampling, fused GroupNorm+SiLU dual output)",
                                  "achieved": fach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": fach / HBM_PEAK_GBPS, "traffic": None,
                                  "launches": int(fn_.value), "avg_launch_ms": fms.value / fnl, "ms_per_step": fms.value,
                                  "algorithmic_bytes_per_launch": fby.value / fnl}
        # STFT / iSTFT (ComplexSTFT + compression and their inverses, once per step each): the two exact-f32 DFT GEMMs against the
        # dense f32 MFMA peak, the four elementwise kernels against HBM.  M = B * T frames, K = 1536 (n_fft 1534 padded):
        #   GEMM flops = 2 * M * K * K each;  frame: read y, write M * K f32;  compress: read M * K, write B * 768 * T_pad c64;
        #   decompress: read B * 768 * T c64, write M * K;  overlap-add: read M * K (each sample of the 4 overlapping frames), write y.
        if scalls[0] >= 1 and scalls[1] >= 1:
            M, K = gbatch * T, 1536
            gflop = 2.0 * M * K * K
            by4 = [4.0 * (gbatch * Lw + M * K), 4.0 * M * K + 8.0 * gbatch * 768 * Tp, 8.0 * gbatch * 768 * T + 4.0 * M * K, 4.0 * (M * K + gbatch * Lw)]
            ms_gemm = sms[1] + sms[4]
            ms_elem = [sms[0], sms[2], sms[3], sms[5]]
            tf = 2 * gflop / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
            gbps = sum(by4) / (sum(ms_elem) * 1e-3) / 1e9 if sum(ms_elem) > 0 else 0.0
            result["roofline_stft"] = {
                "kernel": "sgemm_mfma_kernel (1534-point DFT / inverse DFT as exact-f32 GEMM) + absmax/frame/compress/decompress/overlap_add",
                "dft_gemm": {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TFLOPS,
                             "launches": 2, "avg_launch_ms": ms_gemm / 2, "algorithmic_gflop_per_launch": gflop / 1e9, "M": M, "K": K},
                "elementwise": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "launches": 5,
                                "ms": {"absmax+frame": sms[0], "compress": sms[2], "decompress": sms[3], "overlap_add": sms[5]},
                                "algorithmic_bytes": {"absmax+frame": by4[0], "compress": by4[1], "decompress": by4[2], "overlap_add": by4[3]}},
                "ms_per_step": sum(sms)}
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
