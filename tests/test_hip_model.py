"""GPU parity tests of the whole-model entry points (fd_ncsnpp_forward / fd_ode_solve / fd_enhance) through the
reference-shaped Python API, against golden vectors produced by the reference and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import flowdec_oracle as O
from test_hip_ops import REPORT, check, report

pytestmark = pytest.mark.gpu

# tolerance of the two arithmetic modes against the reference's fp32 results (relative L2):
#   fp32 mode: exact-f32 MFMA, f32 storage -> differences are summation order / transcendental ulps only
#   bf16 mode: bf16 operands + bf16 activation storage, f32 accumulate and f32 ODE state
# measured floors on the nf=8 golden model: rounding ONLY the conv operands to bf16 in the oracle (everything else f32)
# already gives 1.2e-2 on one forward and 6.4e-2 on the final waveform (|X|^(1/0.3) decompression amplifies 3.3x);
# the HIP bf16 path measures 1.6e-2 / 8-10e-2, the f32 path 2e-6 / 1e-5.
# round 5: the nf = 8 bf16 tolerances are DERIVED like the full-width ones (rounds 1-4: fitted, 3e-2 / 0.13 = measured x 1.3):
# tests/golden/g22_bf16_prediction_nf8.json holds the error of the oracle with bf16-rounded operands and storage on G8 / G9 / the
# FlowDec-25s draw (make_golden_bf16_prediction_nf8.py); tol_fwd8 / tol_wave8 below = 1.3 x (one forward) / 1.6 x (one waveform draw)
# of the prediction for THAT golden and solver.
# precision="bf16x3" (split-bf16 operands) is held to the FP32 mode's tolerances; "mixed" (f32 residual stream, bf16 operands): measured x 1.3
TOL_FWD = {"fp32": 2e-4, "bf16x3": 2e-4, "mixed": 1.1e-2}
TOL_WAVE = {"fp32": 5e-4, "bf16x3": 5e-4, "mixed": 1.3e-1}
# round 4: at FULL width the bf16 tolerances are DERIVED, not fitted: tests/golden/g19_bf16_prediction.json holds the error the
# oracle makes on the same goldens when it rounds every conv operand AND every stored tensor to bf16 in float32 NumPy arithmetic
# (make_golden_bf16_prediction.py: forward 1.008e-2 against G10, waveform 1.77e-2 / Euler-6 against G17).  One forward of the HIP
# bf16 mode must stay within 1.3 x that prediction (summation order, transcendental ulps; the fp16-operand Winograd launches round
# less).  The WAVEFORM of enhance() is a different statistic: the random-weight field amplifies a perturbation by a factor that
# depends on its direction, so on ONE (clip, noise) draw the error of any rounding model -- the oracle's included -- scatters by about
# +-25 % around its mean (scripts/wino4_error_seeds.py, 8 draws x 4 convolution algorithms).  The contract is therefore
#   mean over draws  <=  1.3 x mean prediction   and   every draw  <=  1.6 x its own prediction
# over G17's draw plus the draws of g20_bf16_prediction_draws.npz (test_bf16_error_is_the_predicted_one); single-draw tests use 1.6.
import importlib.util as _ilu
import json as _json
import os as _os
from conftest import GOLDEN as _GOLDEN
BF16_PRED = _json.load(open(_os.path.join(_GOLDEN, "g19_bf16_prediction.json")))
BF16_PRED8 = _json.load(open(_os.path.join(_GOLDEN, "g22_bf16_prediction_nf8.json")))
BF16_MEAN_FACTOR, BF16_DRAW_FACTOR = 1.3, 1.6


def tol_fwd8(prec, key):
    """nf = 8 forward tolerance: bf16 = 1.3 x the oracle-with-bf16-roundings prediction for that golden output."""
    return BF16_MEAN_FACTOR * BF16_PRED8["forward_rel_l2"][key] if prec == "bf16" else TOL_FWD[prec]


def tol_wave8(prec, key):
    """nf = 8 waveform tolerance: bf16 = 1.6 x the prediction for that draw and solver setting."""
    return BF16_DRAW_FACTOR * BF16_PRED8["enhance_rel_l2"][key] if prec == "bf16" else TOL_WAVE[prec]
TOL_FWD_FULL = {"fp32": 2e-4, "bf16x3": 2e-4, "bf16": BF16_MEAN_FACTOR * BF16_PRED["forward_rel_l2"]["bf16_operands_and_storage"], "mixed": 1.1e-2}
TOL_WAVE_FULL = {"fp32": 5e-4, "bf16x3": 5e-4, "mixed": 1.7e-2}


def tol_wave_full(prec, solver_key, table="enhance_rel_l2"):
    """Waveform tolerance at full width.  bf16: 1.6 x the oracle-with-bf16-roundings prediction OF THAT SOLVER SETTING (round 4 took the
    max over both settings, which gave the Euler runs 2.0 x their own prediction)."""
    return BF16_DRAW_FACTOR * BF16_PRED[table][solver_key] if prec == "bf16" else TOL_WAVE_FULL[prec]

_cache = {}


def make_model(nf, seed, precision):
    key = (nf, seed, precision)
    if key not in _cache:
        import flowdec_amd
        m = flowdec_amd.from_preset("flowdec_75m", precision=precision, nf=nf)
        sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=seed, nf=nf).items()}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(not k.startswith("backbone.") for k in missing), (missing, unexpected)
        _cache[key] = m.cuda()
    return _cache[key]


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_ncsnpp_nf8_golden(prec):
    g = load_golden("g8_ncsnpp_nf8.npz")
    m = make_model(8, int(g["seed"]), prec)
    out = m(cu(g["x"]), cu(g["y"]), torch.tensor(0.25, device="cuda"))  # 0-dim t like torchdyn passes it
    assert out.shape == (2, 1, 768, 64) and out.dtype == torch.complex64
    check(f"ncsnpp_nf8[{prec}]", out.cpu().numpy(), g["out_t025"], tol_fwd8(prec, "out_t025"))
    out2 = m.backbone(cu(g["x"]), cu(g["y"]), torch.tensor([0.1, 0.9], device="cuda"))  # per-sample t
    check(f"ncsnpp_nf8_per_sample_t[{prec}]", out2.cpu().numpy(), g["out_t01_09"], tol_fwd8(prec, "out_t01_09"))


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "mixed", "bf16"])
def test_ncsnpp_full_width_golden(prec):
    g = load_golden("g10_ncsnpp_nf64.npz")
    m = make_model(64, int(g["seed"]), prec)
    out = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda"))
    check(f"ncsnpp_nf64[{prec}]", out.cpu().numpy(), g["out"], TOL_FWD_FULL[prec])
    # deterministic: same inputs -> bit-identical output
    out_b = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda"))
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(out_b))


def test_bf16_error_is_the_predicted_one():
    """The headline precision's contract as a derived assertion: the HIP bf16 mode's error against the REFERENCE (G10 forward, G17
    enhance, both at full width; further draws against the float32 oracle) is at most 1.3 x the error of the oracle with bf16-rounded
    operands and storage on the same inputs -- and not implausibly smaller either (a kernel that skipped work would not land inside
    [0.3, 1.3] x the model by accident).  enhance: mean over the draws, and 1.6 x per draw (see the comment at TOL_WAVE_FULL)."""
    g = load_golden("g10_ncsnpp_nf64.npz")
    m = make_model(64, int(g["seed"]), "bf16")
    out = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda"))
    e = rel_err(out.cpu().numpy(), g["out"])
    pred = BF16_PRED["forward_rel_l2"]["bf16_operands_and_storage"]
    report("bf16_forward_vs_prediction", e / pred, BF16_MEAN_FACTOR)
    assert 0.3 * pred < e < BF16_MEAN_FACTOR * pred, (e, pred)
    g17 = load_golden("g17_enhance_nf64.npz")
    errs, preds = [], []
    for solver, N in (("euler", 6), ("midpoint", 3)):
        x = m.enhance(torch.from_numpy(g17["y"]), N=N, solver=solver, noise=torch.from_numpy(g17["noise"]))
        errs.append(rel_err(x.numpy(), g17[f"{solver}_N{N}"]))
        preds.append(BF16_PRED["enhance_rel_l2"][f"{solver}_N{N}"])
    g20 = load_golden("g20_bf16_prediction_draws.npz")
    assert int(g20["weights_seed"]) == int(g17["seed"])
    spec = _ilu.spec_from_file_location("_draws", _os.path.join(_GOLDEN, "make_golden_bf16_prediction_draws.py"))
    gen = _ilu.module_from_spec(spec); spec.loader.exec_module(gen)
    for i, seed in enumerate(g20["seeds"]):
        y, nz = gen.draw(int(seed))
        x = m.enhance(torch.from_numpy(y), N=6, solver="euler", noise=torch.from_numpy(nz))
        errs.append(rel_err(x.numpy(), g20[f"truth{i}"]))
        preds.append(float(g20["predicted_rel_l2"][i]))
    for i, (e, p) in enumerate(zip(errs, preds)):
        report(f"bf16_enhance_vs_prediction[draw {i}]", e / p, BF16_DRAW_FACTOR)
        assert 0.3 * p < e < BF16_DRAW_FACTOR * p, (i, e, p)
    ratio = float(np.mean(errs) / np.mean(preds))
    report("bf16_enhance_vs_prediction[mean over draws]", ratio, BF16_MEAN_FACTOR)
    assert 0.3 < ratio < BF16_MEAN_FACTOR, (errs, preds)


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_cfg1_exact_workload(prec):
    """BASELINE config 1 EXACTLY -- FlowDec-75m, one 1 s clip @ 48 kHz, 6-step Euler -- against the reference's own
    FlowModel.enhance (golden G18, tests/golden/make_golden_nf64_enhance.py --cfg1): fp32 (the config's precision) and the f32-tolerance
    mode at 5e-4, bf16 at its derived tolerance."""
    g = load_golden("g18_enhance_nf64_cfg1.npz")
    assert g["y"].shape == (1, 1, 48000)
    m = make_model(64, int(g["seed"]), prec)
    x = m.enhance(torch.from_numpy(g["y"]), N=6, solver="euler", noise=torch.from_numpy(g["noise"]))
    assert x.shape == (1, 1, 48000)
    check(f"cfg1_enhance_nf64[euler,N=6,{prec}]", x.numpy(), g["euler_N6"], tol_wave_full(prec, "euler_N6"))


def test_ncsnpp_batch_independence():
    """Clips are independent end to end (what the multi-GPU batch split relies on)."""
    g = load_golden("g8_ncsnpp_nf8.npz")
    m = make_model(8, int(g["seed"]), "fp32")
    x, y = cu(g["x"]), cu(g["y"])
    t = torch.tensor([0.25], device="cuda")
    both = m(x, y, t)
    one = m(x[1:2].contiguous(), y[1:2].contiguous(), t)
    assert rel_err(one.cpu().numpy(), both[1:2].cpu().numpy()) < 1e-6


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "mixed", "bf16"])
@pytest.mark.parametrize("solver,N", [("euler", 6), ("midpoint", 3), ("heun2", 3), ("heun2_eulerlast", 3)])
def test_enhance_golden(solver, N, prec):
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), prec)
    y = torch.from_numpy(g["y"])                       # CPU input -> output must come back on the CPU (model.py:524)
    x = m.enhance(y, N=N, solver=solver, noise=torch.from_numpy(g["noise"]))
    assert x.shape == (2, 1, 24000) and x.device.type == "cpu" and x.dtype == torch.float32
    check(f"enhance[{solver},N={N},{prec}]", x.numpy(), g[f"{solver}_N{N}"], tol_wave8(prec, f"{solver}_N{N}"))
    # the same difference in the reference's own evaluation units (eval/metrics.py): SI-SDR of the build's output against
    # the reference's output, and the log-spectral MSE between them (dB^2)
    from flowdec_amd import metrics
    sdr = min(metrics.si_sdr(x[i].numpy(), g[f"{solver}_N{N}"][i]) for i in range(2))
    lsm = max(metrics.logspec_mse(x[i].numpy(), g[f"{solver}_N{N}"][i]) for i in range(2))
    with open(REPORT, "a") as f:
        f.write(f"{f'enhance[{solver},N={N},{prec}] vs reference':60s} SI-SDR={sdr:.1f} dB  logspec-MSE={lsm:.3e} dB^2\n")
    assert sdr > {"fp32": 80.0, "bf16x3": 60.0, "mixed": 15.0, "bf16": 15.0}[prec]


def test_enhance_graph_equals_eager():
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), "bf16")
    y, nz = torch.from_numpy(g["y"]).cuda(), torch.from_numpy(g["noise"])
    a = m.enhance(y, N=3, solver="midpoint", noise=nz, use_graph=False)
    b = m.enhance(y, N=3, solver="midpoint", noise=nz, use_graph=True)    # first sighting of this call: runs eagerly
    c = m.enhance(y, N=3, solver="midpoint", noise=nz, use_graph=True)    # second sighting: capture + launch
    d = m.enhance(y, N=3, solver="midpoint", noise=nz, use_graph=True)    # replay
    assert a.is_cuda and torch.equal(a, b) and torch.equal(b, c) and torch.equal(c, d)


def test_enhance_shapes_info_and_traj():
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), "fp32")
    nz = torch.from_numpy(g["noise"])
    x1 = m.enhance(torch.from_numpy(g["y"][0, 0]), N=2, solver="euler", noise=nz[:1])   # 1-D in -> 1-D out
    assert x1.shape == (24000,)
    check("enhance_1d", x1.numpy(), g["euler_N2_1d"], TOL_WAVE["fp32"])
    xh, info = m.enhance(torch.from_numpy(g["y"]), N=2, solver="euler", noise=nz, return_preprocess_info=True)
    assert set(info) == {"orig_length", "normfac", "undo_pad_fn", "squeeze_dims"}
    assert info["orig_length"] == 24000 and info["squeeze_dims"] == 0 and info["normfac"].shape == (2, 1, 1)
    assert np.allclose(info["normfac"].cpu().numpy().ravel(), np.abs(g["y"]).max(axis=(1, 2)), rtol=1e-6)
    assert info["undo_pad_fn"](torch.zeros(2, 1, 768, 64)).shape[-1] == 63
    traj, waves = m.enhance(torch.from_numpy(g["y"]), N=2, solver="euler", noise=nz, return_traj=True)
    assert traj.shape == (3, 2, 1, 768, 64) and len(waves) == 3
    norms = np.array([float(torch.view_as_real(X).double().pow(2).sum().sqrt()) for X in traj])
    assert np.allclose(norms, g["traj_feat_norms"], rtol=1e-3)
    check("enhance_traj_last", waves[-1].cpu().numpy(), g["traj_last_wave"], TOL_WAVE["fp32"])


def test_enhance_seeded_noise_reproducible():
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), "bf16")
    y = torch.from_numpy(g["y"]).cuda()
    a = m.enhance(y, N=1, generator=torch.Generator(device="cuda").manual_seed(5))
    b = m.enhance(y, N=1, generator=torch.Generator(device="cuda").manual_seed(5))
    c = m.enhance(y, N=1, generator=torch.Generator(device="cuda").manual_seed(6))
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()


def test_errors():
    import flowdec_amd
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), "bf16")
    with pytest.raises(ValueError):
        m.enhance(torch.zeros(1, 1, 24000), solver="rk4")
    with pytest.raises(NotImplementedError):
        m.enhance(torch.zeros(1, 1, 24000), with_grad=True)
    with pytest.raises(RuntimeError):
        m.enhance(torch.zeros(2, 2, 24000))           # two channels
    with pytest.raises(RuntimeError):
        m.enhance(torch.zeros(1, 1, 500), N=1)        # shorter than the reflect padding
    cpu_model = flowdec_amd.from_preset("flowdec_75m", nf=8)
    with pytest.raises(RuntimeError):
        cpu_model.enhance(torch.zeros(1, 1, 24000), N=1)   # no CPU path: must fail loudly


def test_full_size_properties():
    """BASELINE config 2 shape (B=8 x 2 s, Euler N=6 is run by bench.py); here one full-width forward at
    B=2 x 2 s checks size-independent properties: finite output, batch independence, determinism."""
    m = make_model(64, 64, "bf16")
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(2, 1, 768, 256, dtype=torch.complex64, device="cuda", generator=gen)
    y = torch.randn(2, 1, 768, 256, dtype=torch.complex64, device="cuda", generator=gen)
    t = torch.tensor([0.3], device="cuda")
    v = m(x, y, t)
    assert torch.isfinite(torch.view_as_real(v)).all()
    v0 = m(x[:1].contiguous(), y[:1].contiguous(), t)
    e = rel_err(v0.cpu().numpy(), v[:1].cpu().numpy())
    report("full_size_batch_independence", e, 1e-6)
    assert e < 1e-6


def test_enhance_ragged_batch_and_global_sigma_vs_oracle():
    """Odd clip length (not a multiple of the hop, T = 33 -> T_pad = 64), an odd batch of 3, a silent clip (the
    normalisation guard, util/other.py:77-80) and the scalar sigma_y of the *_globsigy presets -- against the oracle."""
    import flowdec_amd
    L, B, N = 12345, 3, 2
    rng = np.random.default_rng(77)
    y = (0.05 * rng.standard_normal((B, 1, L))).astype(np.float32)
    y[1] = 0.0                                            # silence: normfac falls back to 1
    sd = O.random_state_dict(seed=8, nf=8)
    m = flowdec_amd.from_preset("flowdec_75m_globsigy", precision="fp32", nf=8)
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and float(m.sigma_y) == pytest.approx(0.66)
    m = m.cuda()
    Tp = O.padded_frames(O.num_frames(L))
    assert O.num_frames(L) == 33 and Tp == 64
    noise = ((rng.standard_normal((B, 1, 768, Tp)) + 1j * rng.standard_normal((B, 1, 768, Tp))) / np.sqrt(2)).astype(np.complex64)
    out = m.enhance(torch.from_numpy(y), N=N, solver="midpoint", noise=torch.from_numpy(noise))
    assert out.shape == (B, 1, L)
    ref = O.enhance(O.NCSNppOracle(sd, nf=8), y, noise, 0.66, N=N, solver="midpoint")
    check("enhance_ragged_globsigy[fp32]", out.numpy(), ref, TOL_WAVE["fp32"])
    assert torch.isfinite(out).all()
    # each clip alone gives the same waveform (ragged batches can be split freely)
    one = m.enhance(torch.from_numpy(y[2:]), N=N, solver="midpoint", noise=torch.from_numpy(noise[2:]))
    assert torch.equal(one, out[2:])


def test_longest_cli_clip_full_width():
    """The reference driver enhances files of up to 30 s (enhance.py:115): one 30 s clip through the full-width bf16 model
    (T = 3751 frames -> T_pad = 3776; 1.5 GB per activation tensor, 32-bit offsets inside one image) stays finite and
    agrees with the same audio processed as the first clip of a batch of two."""
    m = make_model(64, 64, "bf16")
    L = 30 * 48000
    gen = torch.Generator(device="cuda").manual_seed(3)
    y = 0.1 * torch.randn(2, 1, L, device="cuda", generator=gen)
    from flowdec_amd import _lib as L_
    lib = L_.load()
    Tp = lib.fd_padded_frames(lib.fd_num_frames(L, 384))
    assert Tp == 3776
    nz = torch.randn(2, 1, 768, Tp, dtype=torch.complex64, device="cuda", generator=gen)
    one = m.enhance(y[:1], N=1, solver="euler", noise=nz[:1])
    assert one.shape == (1, 1, L) and torch.isfinite(one).all() and float(one.abs().max()) > 0
    two = m.enhance(y, N=1, solver="euler", noise=nz)
    assert torch.equal(two[:1], one)


def test_longest_cli_clip_f32_storage_modes():
    """The same 30 s clip in the f32-storage modes (one activation tensor = 2.97 GB: byte offsets beyond 2^31 inside an image, which
    the direct kernel addresses as UNSIGNED 32 bits): `bf16x3` and `fp32` agree to the fp32 tolerance over the whole clip AND over its
    last two seconds (the highest addresses), and both agree with the bf16 mode (whose images stay below 2^31) to the bf16 tolerance."""
    L = 30 * 48000
    gen = torch.Generator(device="cuda").manual_seed(3)
    y = 0.1 * torch.randn(1, 1, L, device="cuda", generator=gen)
    nz = torch.randn(1, 1, 768, 3776, dtype=torch.complex64, device="cuda", generator=gen)
    outs = {}
    for prec in ("bf16", "bf16x3", "fp32"):
        m = make_model(64, 64, prec)
        outs[prec] = m.enhance(y, N=1, solver="euler", noise=nz).cpu().numpy()
        assert np.isfinite(outs[prec]).all() and np.abs(outs[prec]).max() > 0
        _cache.pop((64, 64, prec), None) if prec != "bf16" else None      # free the f32 workspaces (tens of GB) before the next mode
        del m
        torch.cuda.empty_cache()
    tail = slice(L - 2 * 48000, L)
    check("longest_clip[bf16x3 vs fp32]", outs["bf16x3"], outs["fp32"], TOL_WAVE_FULL["bf16x3"])
    check("longest_clip_tail[bf16x3 vs fp32]", outs["bf16x3"][..., tail], outs["fp32"][..., tail], TOL_WAVE_FULL["bf16x3"])
    check("longest_clip[bf16 vs fp32]", outs["bf16"], outs["fp32"], tol_wave_full("bf16", "euler_N6"))
    check("longest_clip_tail[bf16 vs fp32]", outs["bf16"][..., tail], outs["fp32"][..., tail], tol_wave_full("bf16", "euler_N6"))


def test_many_clip_lengths_graph_cache():
    """A file-by-file driver sees many clip lengths: more distinct captured graphs than the cache holds (32) must keep
    working, and a length seen before gives the same waveform again."""
    g = load_golden("g9_enhance_nf8.npz")
    m = make_model(8, int(g["seed"]), "bf16")
    gen = torch.Generator(device="cuda").manual_seed(11)
    y = 0.1 * torch.randn(1, 1, 40000, device="cuda", generator=gen)
    nz = torch.randn(1, 1, 768, 128, dtype=torch.complex64, device="cuda", generator=gen)
    first = m.enhance(y[..., :9000], N=1, noise=nz[..., :64])
    for k in range(40):
        L = 9000 + 700 * (k + 1)                                    # T_pad 64 or 128
        Tp = 64 if 1 + L // 384 <= 64 else 128
        out = m.enhance(y[..., :L], N=1, noise=nz[..., :Tp])
        assert out.shape == (1, 1, L) and torch.isfinite(out).all()
    again = m.enhance(y[..., :9000], N=1, noise=nz[..., :64])
    assert torch.equal(first, again)


def test_enhance_dopri5_adaptive_vs_oracle():
    """solver='dopri5' (adaptive Dormand-Prince, torchdyn semantics restated -- unpinned): the HIP driver against the NumPy
    restatement of the same controller (fixture g16, produced by tests/golden/make_golden_dopri5.py from the oracle)."""
    g = load_golden("g9_enhance_nf8.npz")
    ref = load_golden("g16_dopri5_oracle_nf8.npz")
    m = make_model(8, int(g["seed"]), "fp32")
    y, nz = torch.from_numpy(g["y"][:1]), torch.from_numpy(g["noise"][:1])
    out = m.enhance(y, N=2, solver="dopri5", noise=nz, atol=1e-3, rtol=1e-3)
    nfe, nfe_ref = m.last_nfe, int(ref["nfe_tol1e-3"])
    assert out.shape == (1, 1, 24000) and (nfe - 2) % 6 == 0 and abs(nfe - nfe_ref) <= 12, (nfe, nfe_ref)
    check("enhance_dopri5[fp32]", out.numpy(), ref["wave_tol1e-3"], 5e-3)
    # trajectory checkpoints: t_span = [0, 0.5, 1] -> 3 states, the last one is the result
    traj, waves = m.enhance(y, N=2, solver="dopri5", noise=nz, atol=1e-3, rtol=1e-3, return_traj=True)
    assert traj.shape[0] == 3 and torch.equal(waves[-1].cpu(), out)
    assert abs(float(torch.view_as_real(traj[1]).double().pow(2).sum().sqrt()) / float(ref["mid_feat_norm"]) - 1) < 1e-3
    # a tighter tolerance costs more evaluations (the random-weight field is too stiff for the two answers to be compared)
    tight = m.enhance(y, N=2, solver="dopri5", noise=nz, atol=1e-5, rtol=1e-5)
    assert m.last_nfe > nfe and torch.isfinite(tight).all()
    # solver='tsit5' (Tsitouras 5(4), torchdyn's NeuralODE default): the same driver with the other tableau
    out5 = m.enhance(y, N=2, solver="tsit5", noise=nz, atol=1e-3, rtol=1e-3)
    n5, n5_ref = m.last_nfe, int(ref["tsit5_nfe_tol1e-3"])
    assert (n5 - 2) % 6 == 0 and abs(n5 - n5_ref) <= 12, (n5, n5_ref)
    check("enhance_tsit5[fp32]", out5.numpy(), ref["tsit5_wave_tol1e-3"], 5e-3)
    traj5, _ = m.enhance(y, N=2, solver="tsit5", noise=nz, atol=1e-3, rtol=1e-3, return_traj=True)
    assert abs(float(torch.view_as_real(traj5[1]).double().pow(2).sum().sqrt()) / float(ref["tsit5_mid_feat_norm"]) - 1) < 1e-3
    with pytest.raises(ValueError):
        m.enhance(y, N=2, solver="rk4")
