"""Multi-GPU path on the hardware that exists (one MI355X per test box): SURVEY section 8(e).

  * two ranks SHARING the GPU over gloo: `sharded_enhance` == the unsharded `enhance`, bit for bit, at BASELINE config 2's
    size (8 x 2 s, Euler-6, bf16) and with an uneven split (3 clips over 2 ranks), host and device inputs;
  * RCCL itself: backend 'nccl', world size 1, the collective forced on (`always_gather=True`) -- communicator init and
    all_gather_into_tensor execute on the MI355X and return the unsharded result.
The 8-GPU scaling curve is the driver's to measure; what can be wrong in OUR code (shard arithmetic, noise indexing, the
gather layout, device / host plumbing) is covered here and by the gloo world-2 tests of tests/test_host_cpu.py.
"""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_COMMON = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import flowdec_amd
from flowdec_amd.dist import sharded_enhance, clip_noise

def build(nf, precision="bf16"):
    m = flowdec_amd.from_preset("flowdec_75m", precision=precision, nf=nf)
    g = torch.Generator().manual_seed(1234)
    sd = {}
    for k, v in m.state_dict().items():
        if not k.startswith("backbone."):
            continue
        if k.endswith(".W"): sd[k] = torch.randn(v.shape, generator=g) * 16.0
        elif v.ndim == 1 and k.endswith("weight"): sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"): sd[k] = 0.05 * torch.randn(v.shape, generator=g)
        else: sd[k] = torch.randn(v.shape, generator=g) / v[0].numel() ** 0.5
    m.load_state_dict(sd, strict=False)
    return m.cuda()

def data(B, L, seed=0):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    y = 0.1 * torch.randn(B, 1, L, device="cuda", generator=gen)
    Tp = 64 * ((1 + L // 384 + 63) // 64)
    nz = torch.randn(B, 1, 768, Tp, dtype=torch.complex64, device="cuda", generator=gen)
    return y, nz
'''

_GLOO2 = _COMMON + r'''
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
m = build(64)
# BASELINE config 2: 8 clips x 2 s, Euler N = 6, bf16 -- 4 clips per rank
y, nz = data(8, 96000)
ref = m.enhance(y, N=6, solver="euler", noise=nz)
out = sharded_enhance(m, y, N=6, solver="euler", noise=nz)
assert out.shape == ref.shape and out.is_cuda and torch.equal(out, ref), "cfg 2: sharded != unsharded"
out_h = sharded_enhance(m, y.cpu().pin_memory(), N=6, solver="euler", noise=nz)        # host in -> host out
assert out_h.device.type == "cpu" and torch.equal(out_h, ref.cpu()), "cfg 2, host input: sharded != unsharded"
# uneven split (2 + 1 clips), midpoint, per-clip noise streams: clip i draws from (seed, i) wherever it is processed
y3, _ = data(3, 48000, seed=1)
nz3 = torch.stack([clip_noise(99, i, (1, 768, 128), "cuda") for i in range(3)])
ref3 = m.enhance(y3, N=3, solver="midpoint", noise=nz3)
assert torch.equal(sharded_enhance(m, y3, N=3, solver="midpoint", seed=99), ref3), "uneven split / seed= mode"
one = sharded_enhance(m, y3[:1], N=3, solver="midpoint", seed=99)                      # one clip, rank 1 idles
assert torch.equal(one, ref3[:1])
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''

_NCCL1 = _COMMON + r'''
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)       # RCCL communicator on the one GPU
assert dist.get_backend() == "nccl"
m = build(8)
y, nz = data(3, 24000)
ref = m.enhance(y, N=2, solver="euler", noise=nz)
st = {}
out = sharded_enhance(m, y, N=2, solver="euler", noise=nz, always_gather=True, stats=st)   # all_gather_into_tensor over RCCL
assert out.is_cuda and torch.equal(out, ref), "RCCL world-1 gather changed the result"
assert st["gather_s"] > 0
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("rccl ok")
'''


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def test_two_ranks_one_gpu_bit_identical(tmp_path):
    script = tmp_path / "gloo2.py"
    script.write_text(_GLOO2)
    env = _env(WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    assert all("ok" in o for o in outs)


def test_rccl_world1_gather(tmp_path):
    script = tmp_path / "nccl1.py"
    script.write_text(_NCCL1)
    r = subprocess.run([sys.executable, str(script), ROOT], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("flags,clips", [
    (["--global-batch", "8", "--seconds", "2", "--solver", "midpoint", "--N", "3"], [2, 2, 2, 2]),          # cfg 4's split (strong scaling), 4 ranks
    (["--global-batch", "6", "--seconds", "1", "--precision", "fp32", "--N", "2"], [2, 2, 1, 1]),          # cfg 5's flavour, ragged split
])
def test_bench_four_ranks_share_gpu(flags, clips):
    """The driver's multi-GPU launch line (python -m torch.distributed.run ... bench.py --gpus N) with 4 ranks on the ONE GPU of the
    test box (--share-gpu, gloo): rank 0 prints exactly one JSON line, the split is the one INTEGRATION.md section 4 states, the
    value is whole-job audio seconds over the max-over-ranks time."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--share-gpu",
           "--backend", "gloo", "--no-cpu-baseline"] + flags
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["steps"] == 2 and j["scaling"] == "strong"
    assert j["config"]["clips_per_rank"] == clips and j["config"]["global_batch"] == sum(clips)
    assert len(j["per_rank_ms_per_step"]) == 4 and j["ms_per_step"] >= max(j["per_rank_ms_per_step"]) * 0.999
    secs = float(flags[flags.index("--seconds") + 1])
    assert abs(j["value"] - sum(clips) * secs / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
