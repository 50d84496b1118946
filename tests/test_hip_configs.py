"""GPU tests of the BASELINE.json configurations at their real sizes (size-independent properties + parity where a golden
vector exists), of the reference API surface that sits beside `enhance()` (feature extractor modules, _preprocess /
_postprocess, normalize_mode, return_preprocess_info of the baselines) and of the Winograd convolution path."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import flowdec_oracle as O
from test_hip_model import BF16_PRED, TOL_FWD, TOL_FWD_FULL, TOL_WAVE, TOL_WAVE_FULL, cu, make_model, tol_wave8, tol_wave_full
from test_hip_ops import DT, check, dev, from_nhwc, nhwc, report

pytestmark = pytest.mark.gpu


def _preset_model(preset, nf, seed, precision, **kw):
    import flowdec_amd
    m = flowdec_amd.from_preset(preset, precision=precision, nf=nf, **kw)
    sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=seed, nf=nf).items()}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    return m.cuda(), {k: v.numpy() for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------------------------------------
# full-width enhance against the REFERENCE (golden G17: FlowModel.enhance, nf = 64, one 0.5 s clip, headline solver settings)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("solver,N", [("euler", 6), ("midpoint", 3)])
def test_enhance_full_width_golden(solver, N, prec):
    g = load_golden("g17_enhance_nf64.npz")
    m = make_model(64, int(g["seed"]), prec)
    x = m.enhance(torch.from_numpy(g["y"]), N=N, solver=solver, noise=torch.from_numpy(g["noise"]))
    assert x.shape == (1, 1, 24000)
    check(f"enhance_nf64[{solver},N={N},{prec}]", x.numpy(), g[f"{solver}_N{N}"], tol_wave_full(prec, f"{solver}_N{N}"))


# ---------------------------------------------------------------------------------------------------------------------------
# cfg 3: FlowDec-25s (per-frequency sigma_y curve of data/flowdec_autoparams_25s.npy), midpoint N = 3 (NFE 6)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_flowdec_25s_midpoint_vs_oracle(prec):
    m, sd = _preset_model("flowdec_25s", 8, 8, prec)
    sig = m.sigma_y.detach().cpu().numpy()
    assert sig.shape == (768, 1) and float(sig.std()) > 0          # a curve, not a scalar
    g12 = load_golden("g12_sigma_y.npz")
    assert np.allclose(sig, g12["25s"], rtol=1e-12)
    rng = np.random.default_rng(25)
    L = 19200
    y = (0.1 * rng.standard_normal((2, 1, L))).astype(np.float32)
    Tp = O.padded_frames(O.num_frames(L))
    noise = ((rng.standard_normal((2, 1, 768, Tp)) + 1j * rng.standard_normal((2, 1, 768, Tp))) / np.sqrt(2)).astype(np.complex64)
    out = m.enhance(torch.from_numpy(y), N=3, solver="midpoint", noise=torch.from_numpy(noise))
    ref = O.enhance(O.NCSNppOracle(sd, nf=8), y, noise, sig, N=3, solver="midpoint")
    check(f"enhance_flowdec_25s[midpoint,N=3,{prec}]", out.numpy(), ref, tol_wave8(prec, "flowdec_25s_midpoint_N3"))


def _size_properties(m, B, seconds, N, solver, tag):
    """finite output; hipGraph replay == eager launches; every clip alone == the same clip inside the batch."""
    Lw = int(seconds * 48000)
    gen = torch.Generator(device="cuda").manual_seed(17)
    y = 0.1 * torch.randn(B, 1, Lw, device="cuda", generator=gen)
    from flowdec_amd import _lib as L_
    lib = L_.load()
    Tp = lib.fd_padded_frames(lib.fd_num_frames(Lw, 384))
    nz = torch.randn(B, 1, 768, Tp, dtype=torch.complex64, device="cuda", generator=gen)
    eager = m.enhance(y, N=N, solver=solver, noise=nz, use_graph=False)
    assert eager.shape == (B, 1, Lw) and torch.isfinite(eager).all() and float(eager.abs().max()) > 0
    m.enhance(y, N=N, solver=solver, noise=nz, use_graph=True)             # first sighting of the shape: eager
    cap = m.enhance(y, N=N, solver=solver, noise=nz, use_graph=True)       # captured
    rep = m.enhance(y, N=N, solver=solver, noise=nz, use_graph=True)       # replayed
    assert torch.equal(eager, cap) and torch.equal(cap, rep), f"{tag}: graph replay differs from eager launches"
    for b in (0, B - 1):
        one = m.enhance(y[b:b + 1], N=N, solver=solver, noise=nz[b:b + 1], use_graph=False)
        e = rel_err(one.cpu().numpy(), eager[b:b + 1].cpu().numpy())
        report(f"{tag}_clip{b}_independence", e, 1e-6)
        assert e < 1e-6


def test_cfg3_flowdec_25s_b32_full_width():
    """BASELINE config 3: FlowDec-25s, batch = 32 x 2 s, midpoint (N = 3 -> NFE 6), bf16, full width."""
    m, _ = _preset_model("flowdec_25s", 64, 64, "bf16")
    _size_properties(m, 32, 2.0, 3, "midpoint", "cfg3_b32_midpoint")


def test_cfg3_flowdec_25s_b32_midpoint_N6_full_width():
    """BASELINE config 3's second reading ("6-step midpoint" = enhance(N=6, 'midpoint') = 12 evaluations, the reference's
    docstring model.py:487 "midpoint has NFE=2*N"): same properties at full size."""
    m, _ = _preset_model("flowdec_25s", 64, 64, "bf16")
    _size_properties(m, 32, 2.0, 6, "midpoint", "cfg3_b32_midpoint_N6")


def test_cfg4_flowdec_75m_b32_per_gpu_shard_full_width():
    """BASELINE config 4 BY NAME: FlowDec-75m, 256 x 2 s clips over 8 GPUs = 32 clips per GPU, midpoint, bf16 -- one GPU's shard at full
    width and full size, in both readings of "6-step midpoint" (N = 3: NFE 6, demo.ipynb cell 3; N = 6: 12 evaluations, model.py:487).
    Same kernel schedule as cfg 3 (the presets differ in the sigma_y curve only: config/flowdec_75m.yaml:18-22 vs flowdec_25s.yaml);
    the clip-level parity of this model / size / solver against the reference is test_cfg4_image_size_vs_reference (G24), the
    N-GPU == 1-GPU identity tests/test_hip_dist.py."""
    m = make_model(64, 64, "bf16")
    _size_properties(m, 32, 2.0, 3, "midpoint", "cfg4_shard_b32_midpoint_N3")
    _size_properties(m, 32, 2.0, 6, "midpoint", "cfg4_shard_b32_midpoint_N6")


def test_cfg2_flowdec_75m_b8_euler6_full_width():
    """BASELINE config 2 exactly: FlowDec-75m, batch = 8 x 2 s, 6-step Euler, bf16 (the bench.py workload)."""
    m = make_model(64, 64, "bf16")
    _size_properties(m, 8, 2.0, 6, "euler", "cfg2_b8_euler6")


def _g21_inputs(g):
    """(y, noise) of golden G21: y is stored, the 1.5 MB of noise are re-drawn in the generator's order and pinned by a checksum."""
    rng = np.random.default_rng(int(g["rng_seed"]))
    y = (0.1 * rng.standard_normal(g["y"].shape)).astype(np.float32)
    assert np.array_equal(y, g["y"])
    shape = (1, 1, 768, O.padded_frames(O.num_frames(y.shape[-1])))
    noise = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)
    assert abs(noise.astype(np.complex128).sum() - complex(g["noise_sum"])) < 1e-6
    assert abs((np.abs(noise.astype(np.complex128)) ** 2).sum() - float(g["noise_abs2"])) < 1e-6 * float(g["noise_abs2"])
    return y, noise


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
def test_cfg2_image_size_vs_reference(prec):
    """The image size bench.py times (768 x 256: one 2 s clip, T_pad = 256 frames), 6-step Euler, against the REFERENCE's own
    FlowModel.enhance (flowdec/model.py:476-528; golden G21 = make_golden_nf64_enhance.py --cfg2clip).  At this size the kernel
    schedule differs from every shorter golden: resolution level 1 (384 x 128 = 192 tiles) runs the F(4,3) kernel (F(2,3) in
    G17 / G18), level 2 (96 tiles) the F(2,3) kernel (direct in G10 / G17).  fp32 / bf16x3: 5e-4; bf16: 1.6 x the prediction of the
    oracle with bf16 roundings ON THIS CLIP (g19_bf16_prediction.json: enhance_rel_l2_cfg2clip), and not below 0.3 x it.  The clip
    inside a batch of 8 (cfg 2's batch) gives the same bits."""
    g = load_golden("g21_enhance_nf64_cfg2clip.npz")
    y, noise = _g21_inputs(g)
    m = make_model(64, int(g["seed"]), prec)
    x = m.enhance(torch.from_numpy(y), N=6, solver="euler", noise=torch.from_numpy(noise))
    assert x.shape == (1, 1, 96000)
    tol = tol_wave_full(prec, "euler_N6", "enhance_rel_l2_cfg2clip")
    e = rel_err(x.numpy(), g["euler_N6"])
    check(f"cfg2clip_enhance_nf64[euler,N=6,{prec}]", x.numpy(), g["euler_N6"], tol)
    if prec == "bf16":
        assert e > 0.3 * BF16_PRED["enhance_rel_l2_cfg2clip"]["euler_N6"], e
        rng = np.random.default_rng(8)
        yb = (0.1 * rng.standard_normal((8, 1, 96000))).astype(np.float32)
        nb = ((rng.standard_normal((8, 1, 768, 256)) + 1j * rng.standard_normal((8, 1, 768, 256))) / np.sqrt(2.0)).astype(np.complex64)
        yb[5], nb[5] = y[0], noise[0]
        xb = m.enhance(torch.from_numpy(yb), N=6, solver="euler", noise=torch.from_numpy(nb))
        assert torch.equal(xb[5], x[0]), "a clip inside cfg 2's batch differs from the same clip alone"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_cfg3_image_size_vs_reference(prec):
    """BASELINE config 3's model and solver at its image size, against the REFERENCE's own FlowModel.enhance (golden G23 =
    make_golden_nf64_enhance.py --cfg3clip): FlowDec-25s -- the per-frequency sigma_y curve of data/flowdec_autoparams_25s.npy
    (flowdec/data/sigma_models.py from_file) -- one 2 s clip, midpoint N = 3 (NFE 6).  fp32: 5e-4; bf16: 1.6 x the prediction of the
    oracle with bf16 roundings on this clip, and not below 0.3 x it."""
    g = load_golden("g23_enhance_nf64_cfg3clip.npz")
    y, noise = _g21_inputs(g)
    m, _ = _preset_model("flowdec_25s", 64, int(g["seed"]), prec)
    assert np.allclose(m.sigma_y.detach().cpu().numpy(), g["sigma_y"], rtol=1e-12)
    x = m.enhance(torch.from_numpy(y), N=3, solver="midpoint", noise=torch.from_numpy(noise))
    assert x.shape == (1, 1, 96000)
    check(f"cfg3clip_enhance_nf64[midpoint,N=3,{prec}]", x.numpy(), g["midpoint_N3"], tol_wave_full(prec, "midpoint_N3", "enhance_rel_l2_cfg3clip"))
    if prec == "bf16":
        assert rel_err(x.numpy(), g["midpoint_N3"]) > 0.3 * BF16_PRED["enhance_rel_l2_cfg3clip"]["midpoint_N3"]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_cfg4_image_size_vs_reference(prec):
    """BASELINE config 4's model, solver and image size (FlowDec-75m, 2 s clips, midpoint N = 3 -- what each of the 8 GPUs runs on its 32
    clips) against the REFERENCE's own FlowModel.enhance on one such clip (golden G24 = make_golden_nf64_enhance.py --cfg4clip)."""
    g = load_golden("g24_enhance_nf64_cfg4clip.npz")
    y, noise = _g21_inputs(g)
    m = make_model(64, int(g["seed"]), prec)
    x = m.enhance(torch.from_numpy(y), N=3, solver="midpoint", noise=torch.from_numpy(noise))
    check(f"cfg4clip_enhance_nf64[midpoint,N=3,{prec}]", x.numpy(), g["midpoint_N3"], tol_wave_full(prec, "midpoint_N3", "enhance_rel_l2_cfg4clip"))
    if prec == "bf16":
        assert rel_err(x.numpy(), g["midpoint_N3"]) > 0.3 * BF16_PRED["enhance_rel_l2_cfg4clip"]["midpoint_N3"]


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_cfg5_clip_vs_reference(prec):
    """BASELINE config 5's precision, clip length and step count -- fp32, 4 s clips (T_pad = 512 frames: a 768 x 512 image), 32 solver steps
    in the fixed-step reading (Euler) -- against the REFERENCE's own FlowModel.enhance on one such clip (golden G25 =
    make_golden_nf64_enhance.py --cfg5clip, ~20 min of CPU there): the exact-f32 mode and the fp32-tolerance mode `bf16x3` at 5e-4.
    (The adaptive reading needs torchdyn's controller, which is not available offline: scripts/pin_third_party.py.)"""
    g = load_golden("g25_enhance_nf64_cfg5clip.npz")
    y, noise = _g21_inputs(g)
    assert y.shape == (1, 1, 192000) and noise.shape[-1] == 512
    m = make_model(64, int(g["seed"]), prec)
    x = m.enhance(torch.from_numpy(y), N=32, solver="euler", noise=torch.from_numpy(noise))
    check(f"cfg5clip_enhance_nf64[euler,N=32,{prec}]", x.numpy(), g["euler_N32"], TOL_WAVE_FULL[prec])


def test_cfg5_fp32_4s_32step_and_adaptive():
    """BASELINE config 5, per-GPU shape: fp32, 8 x 4 s clips, 32 solver steps (fixed-step reading), clip independence.  The adaptive
    reading (dopri5 over the same 33-point t_span) at FULL size takes minutes (494 / 530 evaluations at atol = rtol = 1e-4:
    scripts/bench_cfg5_dopri5.py -> profiles/r03_bench_cfg5_dopri5.json); here it runs on 2 of the clips in the f32-tolerance mode."""
    m = make_model(64, 64, "fp32")
    Lw = 4 * 48000
    gen = torch.Generator(device="cuda").manual_seed(5)
    y = 0.1 * torch.randn(8, 1, Lw, device="cuda", generator=gen)
    nz = torch.randn(8, 1, 768, 512, dtype=torch.complex64, device="cuda", generator=gen)
    out = m.enhance(y, N=32, solver="euler", noise=nz)
    assert out.shape == (8, 1, Lw) and torch.isfinite(out).all() and float(out.abs().max()) > 0
    one = m.enhance(y[3:4], N=32, solver="euler", noise=nz[3:4])
    assert rel_err(one.cpu().numpy(), out[3:4].cpu().numpy()) < 1e-6


def test_cfg5_dopri5_default_tolerances():
    """cfg 5's "32-step adaptive" solver as the reference would run it: `enhance(N=32, solver='dopri5')` with NO tolerance arguments
    (flowdec_amd.model.ADAPTIVE_DEFAULT_TOL = 1e-3, torchdyn's NeuralODE default -- oracle/flowdec_oracle.py (f4)), 4 s clips, in
    `bf16x3` (the mode that makes the config usable); fp32 vs bf16x3 at the same tolerance, where both controllers take the same
    decisions (every step lands on a checkpoint: 2 + 6 x 33 evaluations) and the waveforms must agree at the fp32 tolerance; and the
    tighter candidate 1e-4 (more evaluations)."""
    Lw = 4 * 48000
    gen = torch.Generator(device="cuda").manual_seed(5)
    y = 0.1 * torch.randn(2, 1, Lw, device="cuda", generator=gen)
    nz = torch.randn(2, 1, 768, 512, dtype=torch.complex64, device="cuda", generator=gen)
    mx = make_model(64, 64, "bf16x3")
    import flowdec_amd.model as FM
    assert FM.ADAPTIVE_DEFAULT_TOL == 1e-3
    ad = mx.enhance(y, N=32, solver="dopri5", noise=nz)
    nfe_def = mx.last_nfe
    assert ad.shape == (2, 1, Lw) and torch.isfinite(ad).all() and nfe_def >= 2 + 6 * 32 and (nfe_def - 2) % 6 == 0
    t4 = mx.enhance(y, N=32, solver="dopri5", noise=nz, atol=1e-4, rtol=1e-4)
    nfe = mx.last_nfe
    report("cfg5_dopri5_nfe[bf16x3,1e-4]", float(nfe), 1e9)
    assert torch.isfinite(t4).all() and nfe >= nfe_def and (nfe - 2) % 6 == 0
    a3 = mx.enhance(y[:1], N=32, solver="dopri5", noise=nz[:1], atol=1e-3, rtol=1e-3)
    n3 = mx.last_nfe
    assert torch.equal(a3, mx.enhance(y[:1], N=32, solver="dopri5", noise=nz[:1]))   # the default IS 1e-3
    mf = make_model(64, 64, "fp32")
    b3 = mf.enhance(y[:1], N=32, solver="dopri5", noise=nz[:1], atol=1e-3, rtol=1e-3)
    assert mf.last_nfe == n3 <= nfe, (mf.last_nfe, n3, nfe)
    check("cfg5_dopri5[bf16x3 vs fp32, 1e-3]", a3.cpu().numpy(), b3.cpu().numpy(), 5e-4)


# ---------------------------------------------------------------------------------------------------------------------------
# reference API surface next to enhance()
# ---------------------------------------------------------------------------------------------------------------------------
def test_feature_extractor_modules_golden():
    """ComplexSTFT.forward / invert and CompressAmplitudesAndScale.forward / invert as stand-alone modules
    (feature_extractors.py:86-139) against golden G1 (torch.stft / istft of the reference)."""
    g = load_golden("g1_stft.npz")
    m = make_model(8, 8, "fp32")
    fe = m.feature_extractor
    y = cu(g["y"])
    S = fe.complex_stft(y)
    assert S.shape == (2, 1, 768, 13) and S.dtype == torch.complex64
    check("ComplexSTFT.forward", S.cpu().numpy(), g["stft"], 2e-5)
    Cc = fe.compress(S)
    check("Compress.forward", Cc.cpu().numpy(), g["compressed"], 2e-5)
    check("Compress.invert", fe.compress.invert(cu(g["compressed"])).cpu().numpy(), g["decompressed"], 2e-5)
    check("ComplexSTFT.invert", fe.complex_stft.invert(cu(g["stft"]), orig_length=4800).cpu().numpy(), g["roundtrip"], 2e-5)
    check("FE.forward", fe(y).cpu().numpy(), g["compressed"], 2e-5)
    check("FE.invert_len3000", fe.invert(cu(g["Z"]), orig_length=3000).cpu().numpy(), g["istft_len3000"], 2e-5)


def test_preprocess_postprocess_golden():
    """EnhancementModel._preprocess / _postprocess (model.py:129-190) against golden G3 (normalize_noisy + pad_spec) incl. the
    silent clip, and as a round trip."""
    g = load_golden("g3_pad_norm.npz")
    m = make_model(8, 8, "fp32")
    Y, X, info = m._preprocess(torch.from_numpy(g["y"]))                                     # the reference's 3-tuple (model.py:163)
    assert X is None and set(info) == {"orig_length", "normfac", "undo_pad_fn", "squeeze_dims"} and Y.shape == (2, 1, 768, int(g["padded_T"]))
    assert np.array_equal(info["normfac"].cpu().numpy(), g["normfac"])                       # bit exact, silence guard included
    assert float(torch.view_as_real(Y[..., int(g["orig_T"]):]).abs().max()) == 0.0           # zero padding of the frame axis
    assert info["undo_pad_fn"](Y).shape[-1] == int(g["orig_T"])
    Yo, info_o = O.preprocess(g["y"])                                                         # (G3's `spec` is the un-normalised clip's)
    assert Yo.shape == tuple(Y.shape) and info_o["T"] == int(g["orig_T"])
    check("_preprocess", Y.cpu().numpy(), Yo, 2e-5)
    x = m._postprocess(Y, info)
    assert x.shape == (2, 1, 4800)
    check("_postprocess_roundtrip", x.cpu().numpy(), g["y"], 1e-4)
    Y1, _, info1 = m._preprocess(torch.from_numpy(g["y"][0, 0]))                              # 1-D in -> 1-D out
    assert m._postprocess(Y1, info1).shape == (4800,)
    # clean target x: normalised by y's factor, same features (model.py:152-157, util/other.py:79-81)
    xc = 0.5 * g["y"][:, :, ::-1].copy()
    Y2, X2, info2 = m._preprocess(torch.from_numpy(g["y"]), x=torch.from_numpy(xc))
    assert torch.equal(Y2, Y) and X2.shape == Y.shape
    Xo, _ = O.pad_spec(O.compress(O.stft((xc / g["normfac"]).astype(np.float32)), O.ALPHA, O.BETA))
    check("_preprocess_x", X2.cpu().numpy(), Xo, 2e-5)
    # batch_filter (model.py:185-187): only the selected items come back, at their own level
    xf = m._postprocess(Y, info, batch_filter=[False, True])
    assert xf.shape == (1, 1, 4800) and torch.equal(xf[0], x[1])
    with pytest.raises(NotImplementedError):
        m._postprocess(Y, info, inv_kwargs={"window": None})
    with pytest.raises(NotImplementedError):
        m._preprocess(torch.from_numpy(g["y"]), comp_eps=1e-3)


def test_normalize_mode_none_vs_oracle():
    """normalize_mode = 'none' (model.py:52, util/other.py:70): no level normalisation in front of the STFT."""
    import flowdec_amd
    from flowdec_amd.model import BACKBONE_FINAL_NO_ATTN, NCSNpp, AmplitudeCompressedComplexSTFT, FlowModel
    sd = O.random_state_dict(seed=8, nf=8)
    bb = dict(BACKBONE_FINAL_NO_ATTN); bb["nf"] = 8
    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000, alpha=0.3, beta=0.33)
    m = FlowModel(backbone=NCSNpp(precision="fp32", **bb), feature_extractor=fe, sampling_rate=48000, sigma_y=0.66, normalize_mode="none")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.cuda().eval()
    rng = np.random.default_rng(9)
    L = 12000
    y = (0.3 * rng.standard_normal((1, 1, L))).astype(np.float32)
    Tp = O.padded_frames(O.num_frames(L))
    noise = ((rng.standard_normal((1, 1, 768, Tp)) + 1j * rng.standard_normal((1, 1, 768, Tp))) / np.sqrt(2)).astype(np.complex64)
    out, info = m.enhance(torch.from_numpy(y), N=2, solver="euler", noise=torch.from_numpy(noise), return_preprocess_info=True)
    assert float(info["normfac"].cpu().reshape(-1)[0]) == 1.0
    # the oracle always normalises: feed it the clip pre-scaled so that its max-abs is exactly 1 (normfac = 1) -- identical input
    ys = (y / np.abs(y).max()).astype(np.float32)
    out_s = m.enhance(torch.from_numpy(ys), N=2, solver="euler", noise=torch.from_numpy(noise))
    ref = O.enhance(O.NCSNppOracle(sd, nf=8), ys, noise, 0.66, N=2, solver="euler")
    check("enhance_normalize_none[fp32]", out_s.numpy(), ref, TOL_WAVE["fp32"])
    # and the un-normalised call really differs from the normalised one (the network is not scale-equivariant)
    m2 = flowdec_amd.from_preset("flowdec_75m_globsigy", precision="fp32", nf=8)
    m2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    out_n = m2.cuda().enhance(torch.from_numpy(y), N=2, solver="euler", noise=torch.from_numpy(noise))
    assert rel_err(out.numpy(), out_n.numpy()) > 1e-3


def test_baselines_return_preprocess_info():
    """RegressionModel / ScoreModel.enhance(return_preprocess_info=True) (model.py:575, 654)."""
    import flowdec_amd
    sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=8, nf=8).items()}
    y = 0.1 * torch.randn(2, 1, 9600, generator=torch.Generator().manual_seed(1))
    for preset in ("baseline_regression_75s", "baseline_scoredec_75s"):
        m = flowdec_amd.from_preset(preset, precision="fp32", nf=8)
        m.load_state_dict(sd, strict=False)
        m = m.cuda()
        kw = dict(N=2, generator=torch.Generator(device="cuda").manual_seed(2)) if "score" in preset else {}
        x, info = m.enhance(y, return_preprocess_info=True, **kw)
        assert x.shape == (2, 1, 9600) and torch.isfinite(x).all()
        assert set(info) == {"orig_length", "normfac", "undo_pad_fn", "squeeze_dims"} and info["orig_length"] == 9600
        assert np.allclose(info["normfac"].cpu().numpy().ravel(), y.abs().amax(dim=(1, 2)).numpy(), rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------------
# Winograd F(2,3) convolution path (conv_wino.hip; FD_WINOGRAD)
# ---------------------------------------------------------------------------------------------------------------------------
WINO_CASES = [
    # name, B, H, W, C0, C1, Cout, affine, bias_rows, skip, S0, S1
    ("basic", 1, 16, 16, 32, 0, 128, False, 0, False, 0, 0),
    ("two_ntiles_aff_skip", 2, 32, 16, 64, 0, 256, True, 2, True, 0, 0),
    ("concat", 1, 16, 16, 64, 32, 128, True, 1, True, 0, 0),
    ("ragged_odd_w", 1, 20, 13, 32, 0, 128, True, 1, True, 0, 0),       # H, W not multiples of the tile; odd W (half tile)
    ("deepk", 1, 16, 16, 256, 256, 256, True, 1, True, 0, 0),
    ("shortcut", 2, 16, 16, 256, 0, 256, True, 1, False, 64, 0),         # ResBlock 64 -> 256: Conv_1(h) + Conv_2(x) folded in
    ("shortcut_cat", 1, 32, 16, 128, 0, 128, True, 1, False, 128, 256),
    ("multi_tile", 2, 48, 80, 64, 0, 256, True, 2, True, 32, 0),
]


@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_conv2d_winograd(case):
    """Same contract and same reference as test_hip_ops.test_conv2d (f64 conv of the bf16-rounded operands); the activated
    operand is rounded to fp16 (not bf16) by this kernel, so the reference keeps it unrounded."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, S0, S1 = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    Cin = C0 + C1
    x = bf(rng.standard_normal((B, Cin, H, W)))
    w = bf(rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9))
    aff, xin = None, x
    if use_aff:
        a = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        d = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([a, d], axis=-1))
        xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float32)
    ref = O.conv2d(xin.astype(np.float64), w.astype(np.float64), None)
    sc0 = sc1 = w_sc = None
    if S0:
        xs = bf(rng.standard_normal((B, S0 + S1, H, W)))
        ws = bf(rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1))
        ref = ref + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
        sc0 = nhwc(xs[:, :S0], torch.bfloat16)
        sc1 = nhwc(xs[:, S0:], torch.bfloat16) if S1 else None
        w_sc = dev(ws)
    bias = None
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
        ref = ref + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])
    skip, scale = None, 1.0
    if use_skip:
        sk = bf(rng.standard_normal((B, Cout, H, W)))
        skip = nhwc(sk, torch.bfloat16)
        ref = ref + sk
        scale = float(1 / np.sqrt(2))
    ref = ref * scale
    x0 = nhwc(x[:, :C0], torch.bfloat16)
    x1 = nhwc(x[:, C0:], torch.bfloat16) if C1 else None
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=w_sc, S0=S0 if S0 else None, winograd=True)
    out, stats = ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1, want_stats=True, winograd=True)
    torch.cuda.synchronize()
    # error budget: bf16 rounding of the stored output (2^-9) dominates; operands fp16 (2^-11), Winograd transform in fp16
    check(f"conv2d_winograd[{name}]", from_nhwc(out), ref, 6e-3)
    st = stats.double().sum(dim=1).cpu().numpy()[:, :Cout]
    ref_s = np.stack([ref.sum(axis=(2, 3)), (ref ** 2).sum(axis=(2, 3))], axis=-1)
    e = float(np.abs(st - ref_s).max() / np.abs(ref_s).max())
    report(f"conv2d_winograd_stats[{name}]", e, 3e-3)
    assert e < 3e-3
    # and against the direct MFMA kernel on the same inputs (both round their output to bf16: they agree to ~1 bf16 ulp)
    pd = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=w_sc, S0=S0 if S0 else None)
    outd = ops.conv2d(x0, pd, Cout, 3, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1)
    assert rel_err(from_nhwc(out), from_nhwc(outd)) < 6e-3


def test_conv2d_winograd_rejects_unsupported():
    from flowdec_amd import ops
    w = torch.randn(64, 32, 3, 3, device="cuda")
    with pytest.raises(RuntimeError):
        ops.pack_conv_weight(w, dtype=torch.bfloat16, winograd=True)          # Cout % 128 != 0
    with pytest.raises(RuntimeError):
        ops.pack_conv_weight(torch.randn(128, 32, 1, 1, device="cuda"), dtype=torch.bfloat16, winograd=True)   # 1x1
    with pytest.raises(RuntimeError):
        ops.pack_conv_weight(torch.randn(128, 32, 3, 3, device="cuda"), dtype=torch.float32, winograd=True)    # f32 storage
    with pytest.raises(RuntimeError):   # Cout = 384: padded to 384 by the Winograd packing but to 512 by the statistics layout
        ops.pack_conv_weight(torch.randn(384, 32, 3, 3, device="cuda"), dtype=torch.bfloat16, winograd=True)
    ops.pack_conv_weight(torch.randn(512, 32, 3, 3, device="cuda"), dtype=torch.bfloat16, winograd=True)       # 512 = pad_256(512): fine


# ---------------------------------------------------------------------------------------------------------------------------
# Winograd F(4,3) convolution path (conv_wino4.hip; FD_WINOGRAD4): 256-cout workgroups, whole 16 x 16 tiles
# ---------------------------------------------------------------------------------------------------------------------------
WINO4_CASES = [
    # name, B, H, W, C0, C1, Cout, affine, bias_rows, skip, S0, S1
    ("basic", 1, 16, 16, 32, 0, 256, False, 0, False, 0, 0),
    ("bias_two_tiles", 2, 32, 16, 64, 0, 256, False, 1, False, 0, 0),
    ("aff_bias_skip", 2, 16, 32, 64, 0, 256, True, 2, True, 0, 0),
    ("concat", 1, 16, 16, 64, 32, 256, True, 1, True, 0, 0),
    ("deepk", 1, 16, 16, 256, 256, 256, True, 1, True, 0, 0),
    ("multi_tile", 3, 48, 80, 64, 0, 256, True, 3, True, 0, 0),          # interior + all four image borders, per-clip bias
    ("raw_multi", 2, 32, 64, 96, 0, 256, False, 1, True, 0, 0),           # raw (pre-activated) input, odd chunk-pair count of 3
    ("cat320", 1, 32, 32, 256, 64, 256, True, 1, False, 0, 0),
    ("shortcut", 2, 16, 16, 256, 0, 256, True, 1, False, 64, 0),          # ResBlock 64 -> 256: Conv_1(h) + Conv_2(x) in the epilogue
    ("shortcut_cat", 1, 32, 16, 128, 0, 256, True, 1, False, 128, 256),   # shortcut over a virtual concat
    ("shortcut_320", 1, 16, 48, 64, 0, 256, False, 1, False, 256, 64),
]


@pytest.mark.parametrize("case", WINO4_CASES, ids=[c[0] for c in WINO4_CASES])
def test_conv2d_winograd4(case):
    """Same contract and reference as test_conv2d_winograd.  The 3x3 operands are fp16 (activated input, transformed in packed fp16;
    weights G g in fp16); the folded shortcut is a bf16 x bf16 GEMM on the raw tensors (the direct kernel's numerics)."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, S0, S1 = case
    rng = np.random.default_rng(zlib.crc32(("w4" + name).encode()))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    Cin = C0 + C1
    x = bf(rng.standard_normal((B, Cin, H, W)))
    w = bf(rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9))
    aff, xin = None, x
    if use_aff:
        a = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        d = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([a, d], axis=-1))
        xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float32)
    ref = O.conv2d(xin.astype(np.float64), w.astype(np.float64), None)
    sc0 = sc1 = w_sc = None
    if S0:
        xs = bf(rng.standard_normal((B, S0 + S1, H, W)))
        ws = bf(rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1))
        ref = ref + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
        sc0 = nhwc(xs[:, :S0], torch.bfloat16)
        sc1 = nhwc(xs[:, S0:], torch.bfloat16) if S1 else None
        w_sc = dev(ws)
    bias = None
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
        ref = ref + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])
    skip, scale = None, 1.0
    if use_skip:
        sk = bf(rng.standard_normal((B, Cout, H, W)))
        skip = nhwc(sk, torch.bfloat16)
        ref = ref + sk
        scale = float(1 / np.sqrt(2))
    ref = ref * scale
    x0 = nhwc(x[:, :C0], torch.bfloat16)
    x1 = nhwc(x[:, C0:], torch.bfloat16) if C1 else None
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=w_sc, S0=S0 if S0 else None, winograd=4)
    run = lambda: ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1, want_stats=True, winograd=4)
    out, stats = run()
    torch.cuda.synchronize()
    # measured 1.7-2.1e-3 (the bf16 rounding of the stored output, 2^-9, dominates; direct kernel 1.9-2.3e-3)
    check(f"conv2d_winograd4[{name}]", from_nhwc(out), ref, 4e-3)
    st = stats.double().sum(dim=1).cpu().numpy()[:, :Cout]
    ref_s = np.stack([ref.sum(axis=(2, 3)), (ref ** 2).sum(axis=(2, 3))], axis=-1)
    e = float(np.abs(st - ref_s).max() / np.abs(ref_s).max())
    report(f"conv2d_winograd4_stats[{name}]", e, 1e-3)
    assert e < 1e-3
    # deterministic (no atomics, fixed reduction order) ...
    out2, stats2 = run()
    assert torch.equal(out, out2) and torch.equal(stats, stats2)
    # ... whatever the order the tiles are walked in (FD_TILE_REVERSED) ...
    out3, stats3 = ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1, want_stats=True, winograd=4,
                              reversed_tiles=True)
    assert torch.equal(out, out3) and torch.equal(stats, stats3)
    # ... and close to the direct MFMA kernel on the same inputs (both round their output to bf16)
    pd = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=w_sc, S0=S0 if S0 else None)
    outd = ops.conv2d(x0, pd, Cout, 3, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1)
    assert rel_err(from_nhwc(out), from_nhwc(outd)) < 5e-3


W4F_CASES = WINO4_CASES + [("f32_c16", 1, 16, 32, 16, 0, 256, True, 1, False, 0, 0), ("f32_sc48", 1, 16, 16, 48, 16, 256, True, 1, False, 32, 16)]
# the 2-D kernel: Cout multiples of 128, channel counts multiples of 8, folded shortcut AND residual input together
W44F_CASES = W4F_CASES + [("c128_cat", 2, 32, 16, 24, 8, 128, True, 1, True, 0, 0), ("c128_sc", 1, 16, 32, 128, 0, 128, True, 1, False, 64, 8),
                          ("sc_and_skip", 1, 16, 16, 64, 0, 256, True, 1, True, 64, 0), ("c8", 1, 16, 16, 8, 0, 128, False, 0, False, 0, 0),
                          # three cout blocks (the tile decode's modulo is not a power of two), per-clip bias rows, concat + residual
                          ("c384_b3", 3, 16, 48, 40, 24, 384, True, 3, True, 0, 0)]


@pytest.mark.parametrize("case", [(4, c) for c in W4F_CASES] + [(44, c) for c in W44F_CASES], ids=lambda ac: f"w{ac[0]}-{ac[1][0]}")
def test_conv2d_winograd4_f32(case):
    """The two float32 Winograd kernels of the fp32 mode (round 6): algo 44 = 2-D F(4x4, 3x3) (conv_wino44f.hip: a quarter of the direct f32
    kernel's MFMAs; Cout % 128 == 0, channel counts % 8 == 0, shortcut and residual together) and algo 4 = F(4,3) along W x direct along H
    (conv_wino4f.hip: half; Cout = 256, channel counts % 16 == 0) -- f32 storage, f32 transforms, f32 matrix instructions.  Same operator cases as the fp16-operand kernel plus
    channel counts that are multiples of 16 only; reference = the f64 convolution of the SAME f32 tensors: 5e-6 (measured 2.3e-7 ... 9.0e-7:
    the transforms' row sums of 10 / 8 on f32 roundings; the direct f32 kernel: 3e-7), GroupNorm partial sums 5e-6 (measured <= 2.2e-7),
    bit-deterministic in both tile orders, and within 5e-6 of the direct f32 kernel."""
    from flowdec_amd import ops
    import zlib
    algo, case = case
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, S0, S1 = case
    rng = np.random.default_rng(zlib.crc32(("w4f" + name).encode()))
    Cin = C0 + C1
    f32 = torch.float32
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    aff, xin = None, x.astype(np.float64)
    if use_aff:
        a = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        d = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([a, d], axis=-1))
        u = x.astype(np.float64) * a[:, :, None, None] + d[:, :, None, None]
        xin = u / (1.0 + np.exp(-u))
    ref = O.conv2d(xin, w.astype(np.float64), None)
    sc0 = sc1 = w_sc = None
    if S0:
        xs = rng.standard_normal((B, S0 + S1, H, W)).astype(np.float32)
        ws = (rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1)).astype(np.float32)
        ref = ref + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
        sc0 = nhwc(xs[:, :S0], f32)
        sc1 = nhwc(xs[:, S0:], f32) if S1 else None
        w_sc = dev(ws)
    bias = None
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
        ref = ref + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])
    skip, scale = None, 1.0
    if use_skip:
        sk = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
        skip = nhwc(sk, f32)
        ref = ref + sk
        scale = float(1 / np.sqrt(2))
    ref = ref * scale
    x0 = nhwc(x[:, :C0], f32)
    x1 = nhwc(x[:, C0:], f32) if C1 else None
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=f32, w_sc=w_sc, S0=S0 if S0 else None, winograd=algo)
    kw = dict(x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1, want_stats=True)
    out, stats = ops.conv2d(x0, pw, Cout, 3, winograd=algo, **kw)
    torch.cuda.synchronize()
    assert out.dtype == f32
    check(f"conv2d_winograd{algo}_f32[{name}]", from_nhwc(out), ref, 5e-6)
    st = stats.double().sum(dim=1).cpu().numpy()[:, :Cout]
    ref_s = np.stack([ref.sum(axis=(2, 3)), (ref ** 2).sum(axis=(2, 3))], axis=-1)
    e = float(np.abs(st - ref_s).max() / np.abs(ref_s).max())
    report(f"conv2d_winograd{algo}_f32_stats[{name}]", e, 5e-6)
    assert e < 5e-6
    out2, stats2 = ops.conv2d(x0, pw, Cout, 3, winograd=algo, **kw)
    assert torch.equal(out, out2) and torch.equal(stats, stats2)
    if algo == 4:
        out3, stats3 = ops.conv2d(x0, pw, Cout, 3, winograd=4, reversed_tiles=True, **kw)
        assert torch.equal(out, out3) and torch.equal(stats, stats3)
    pd = ops.pack_conv_weight(dev(w), C0=C0, dtype=f32, w_sc=w_sc, S0=S0 if S0 else None)
    outd, _ = ops.conv2d(x0, pd, Cout, 3, **kw)
    assert rel_err(from_nhwc(out), from_nhwc(outd)) < 5e-6


def test_conv2d_winograd44_f32_random_shapes():
    """Randomised sweep of the 2-D float32 Winograd kernel against the DIRECT float32 kernel on the same tensors (both exact-f32 products:
    5e-6): 48 seeded draws over batch, image size in whole tiles, channel counts (multiples of 8: 1 .. 9 chunks per segment, one or two
    3x3 segments, none / one / two shortcut segments), Cout in {128, 256, 384}, and every combination of affine / per-clip bias / residual.
    The chunk cursors, the padded last requests and the run-off of the weight ring see every segment layout the model does not use."""
    from flowdec_amd import ops
    rng = np.random.default_rng(20260)
    f32 = torch.float32
    worst = 0.0
    for it in range(48):
        B = int(rng.integers(1, 4)); H = 16 * int(rng.integers(1, 4)); W = 16 * int(rng.integers(1, 4))
        C0 = 8 * int(rng.integers(1, 10)); C1 = 8 * int(rng.integers(0, 6)) * int(rng.integers(0, 2))
        S0 = 8 * int(rng.integers(1, 6)) * int(rng.integers(0, 2)); S1 = 8 * int(rng.integers(1, 4)) * int(rng.integers(0, 2)) if S0 else 0
        Cout = int(rng.choice([128, 256, 384]))
        use_aff, use_skip, bias_rows = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.choice([0, 1, B]))
        Cin = C0 + C1
        x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
        w = dev((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
        aff = dev(np.stack([1 + 0.2 * rng.standard_normal((B, Cin)), 0.3 * rng.standard_normal((B, Cin))], axis=-1).astype(np.float32)) if use_aff else None
        sc0 = sc1 = w_sc = None
        if S0:
            xs = torch.from_numpy(rng.standard_normal((B, H, W, S0 + S1)).astype(np.float32)).cuda()
            sc0 = xs[..., :S0].contiguous(); sc1 = xs[..., S0:].contiguous() if S1 else None
            w_sc = dev((rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1)).astype(np.float32))
        bias = dev(rng.standard_normal((bias_rows, Cout)).astype(np.float32) if bias_rows > 1 else rng.standard_normal(Cout).astype(np.float32)) if bias_rows else None
        skip = torch.from_numpy(rng.standard_normal((B, H, W, Cout)).astype(np.float32)).cuda() if use_skip else None
        kw = dict(x1=x[..., C0:].contiguous() if C1 else None, affine=aff, bias=bias, skip=skip, scale=0.7 if use_skip else 1.0, sc0=sc0, sc1=sc1, want_stats=True)
        x0 = x[..., :C0].contiguous()
        outs = []
        for algo in (44, False):
            pw = ops.pack_conv_weight(w, C0=C0, dtype=f32, w_sc=w_sc, S0=S0 if S0 else None, winograd=algo)
            outs.append(ops.conv2d(x0, pw, Cout, 3, winograd=algo, **kw))
        torch.cuda.synchronize()
        e = rel_err(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy())
        es = rel_err(outs[0][1].double().sum(dim=1).cpu().numpy()[:, :Cout], outs[1][1].double().sum(dim=1).cpu().numpy()[:, :Cout])
        assert e < 5e-6 and es < 5e-6, (it, B, H, W, C0, C1, S0, S1, Cout, use_aff, use_skip, bias_rows, e, es)
        worst = max(worst, e)
    report("conv2d_winograd44_f32[48 random shapes vs direct f32]", worst, 5e-6)


def test_fp32_auto_runs_winograd4_f32_and_matches_direct():
    """precision='fp32' with conv_algo='auto' (the default) sends the 3x3 convolutions of every ResBlock to the float32 Winograd kernels
    (2-D F(4x4, 3x3)); 'direct' keeps the direct f32 kernel everywhere.  One full-width forward of each on G10's inputs: both inside the fp32
    tolerance against the reference, and within 2e-5 of each other (different summation orders of exact f32 products)."""
    import flowdec_amd
    g = load_golden("g10_ncsnpp_nf64.npz")
    sd = O.random_state_dict(seed=int(g["seed"]), nf=64)
    outs = {}
    for algo in ("auto", "direct"):
        m = flowdec_amd.from_preset("flowdec_75m", precision="fp32", conv_algo=algo)
        assert m.backbone.conv_algo == algo
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        m = m.cuda()
        outs[algo] = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda")).cpu().numpy()
        check(f"ncsnpp_nf64[fp32,{algo}]", outs[algo], g["out"], TOL_FWD_FULL["fp32"])
        del m
    e = rel_err(outs["auto"], outs["direct"])
    report("fp32_forward[auto (2-D F(4x4,3x3) f32) vs direct]", e, 2e-5)
    assert 0 < e < 2e-5, e            # (0 would mean the F(4,3) kernel did not run: G10's image is 768 x 64 = 192 tiles)
    with pytest.raises(ValueError):
        flowdec_amd.from_preset("flowdec_75m", precision="fp32", conv_algo="winograd")


def test_conv2d_winograd4_rejects_unsupported():
    from flowdec_amd import ops
    bad = [(torch.randn(128, 32, 3, 3), torch.bfloat16),    # Cout != 256
           (torch.randn(256, 32, 1, 1), torch.bfloat16),    # 1x1
           (torch.randn(256, 24, 3, 3), torch.float32),     # f32 storage (conv_wino4f.hip): channels % 16 != 0
           (torch.randn(256, 48, 3, 3), torch.bfloat16)]    # channels % 32 != 0
    for w, dt in bad:
        with pytest.raises(RuntimeError):
            ops.pack_conv_weight(w.cuda(), dtype=dt, winograd=4)
    w = torch.randn(256, 32, 3, 3, device="cuda")
    pw = ops.pack_conv_weight(w, dtype=torch.bfloat16, winograd=4)
    with pytest.raises(RuntimeError):   # partial tiles
        ops.conv2d(torch.zeros(1, 24, 16, 32, device="cuda", dtype=torch.bfloat16), pw, 256, 3, winograd=4)
    ws = torch.randn(256, 64, 1, 1, device="cuda")
    pws = ops.pack_conv_weight(w, dtype=torch.bfloat16, w_sc=ws, winograd=4)
    z = lambda c: torch.zeros(1, 16, 16, c, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):   # residual input AND folded shortcut
        ops.conv2d(z(32), pws, 256, 3, sc0=z(64), skip=z(256), winograd=4)
    with pytest.raises(RuntimeError):   # the reversed order exists for the F(4,3) kernel only
        ops.conv2d(z(32), ops.pack_conv_weight(w, dtype=torch.bfloat16), 256, 3, reversed_tiles=True)


@pytest.mark.parametrize("mag", [1e4, 3e3])
def test_conv2d_winograd4_raw_input_range(mag):
    """Raw (not activated) 3x3 inputs -- FIR-resampled GroupNorm+SiLU outputs in the network -- are converted to fp16 for the packed
    input transform, whose rows sum up to 10 |z|: they saturate at +-6000 (documented in the header).  Parity below, finite above."""
    from flowdec_amd import ops
    rng = np.random.default_rng(int(mag))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    x = bf(mag * rng.uniform(-1, 1, (1, 64, 16, 32)))
    w = bf(rng.standard_normal((256, 64, 3, 3)) / 24)
    ref = O.conv2d(x.astype(np.float64), w.astype(np.float64), None)
    pw = ops.pack_conv_weight(dev(w), dtype=torch.bfloat16, winograd=4)
    out = from_nhwc(ops.conv2d(nhwc(x, torch.bfloat16), pw, 256, 3, winograd=4))
    assert np.isfinite(out).all()
    if mag < 6000:
        check(f"conv2d_winograd4_raw_range[{mag:g}]", out, ref, 4e-3)
    # the ACTIVATED path saturates at the same magnitude (silu(a x + d) with a large GroupNorm gain): finite above it, parity below
    a = np.full((1, 64), 1.0, np.float32)
    d = np.zeros((1, 64), np.float32)
    outa = from_nhwc(ops.conv2d(nhwc(x, torch.bfloat16), pw, 256, 3, affine=dev(np.stack([a, d], axis=-1)), winograd=4))
    assert np.isfinite(outa).all()
    if mag < 6000:
        check(f"conv2d_winograd4_act_range[{mag:g}]", outa, O.conv2d(O.silu(x).astype(np.float64), w.astype(np.float64), None), 4e-3)


@pytest.mark.parametrize("wstd", [1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e2])
@pytest.mark.parametrize("algo", ["winograd4", "winograd"])
def test_conv2d_winograd_weight_range(algo, wstd):
    """The Winograd kernels keep U = G g in fp16; G shrinks a kernel row by up to 24 x (F(4,3)) and fp16 goes subnormal below 6.1e-5.
    The reference zero-initialises Conv_1 and the pyramid heads (init_scale = 0, flowdec/backbones/ncsnpp_utils/layers.py:100), so a
    trained checkpoint may hold small weights exactly there.  The pack kernels scale every cout's rows by a power of two (undone
    exactly in the epilogue): parity with the f64 convolution at the usual 4e-3 / 6e-3 for weight magnitudes from 1e-6 to 1e2 --
    also with magnitudes that DIFFER by cout (each row has its own scale), an all-zero cout, and a folded shortcut of another
    magnitude (it shares the cout's factor)."""
    from flowdec_amd import ops
    w4 = algo == "winograd4"
    tol = 4e-3 if w4 else 6e-3
    rng = np.random.default_rng(int(abs(np.log10(wstd)) * 10) + (4 if w4 else 2))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    B, C, H, W, Co, S = 1, 64, 16, 32, 256, 64
    x = bf(rng.standard_normal((B, C, H, W)))
    a = (1 + 0.2 * rng.standard_normal((B, C))).astype(np.float32)
    d = (0.3 * rng.standard_normal((B, C))).astype(np.float32)
    xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float64)
    aff = dev(np.stack([a, d], axis=-1))
    x0 = nhwc(x, torch.bfloat16)
    kw = dict(winograd=4 if w4 else True)
    # (1) one magnitude for the whole layer
    w = bf(wstd * rng.standard_normal((Co, C, 3, 3)) / np.sqrt(C * 9))
    ref = O.conv2d(xin, w.astype(np.float64), None)
    out = from_nhwc(ops.conv2d(x0, ops.pack_conv_weight(dev(w), dtype=torch.bfloat16, **kw), Co, 3, affine=aff, **kw))
    check(f"conv2d_{algo}_weight_range[{wstd:g}]", out, ref, tol)
    # (2) per-cout magnitudes spread over 8 decades below wstd, cout 7 all zero: every cout on its own
    mags = (10.0 ** rng.uniform(-8, 0, Co)).astype(np.float32)
    mags[7] = 0
    w2 = bf(w * mags[:, None, None, None])
    ref2 = O.conv2d(xin, w2.astype(np.float64), None)
    out2 = from_nhwc(ops.conv2d(x0, ops.pack_conv_weight(dev(w2), dtype=torch.bfloat16, **kw), Co, 3, affine=aff, **kw))
    assert np.isfinite(out2).all() and not out2[:, 7].any()
    worst = max(rel_err(out2[:, c], ref2[:, c]) for c in range(Co) if c != 7 and np.abs(ref2[:, c]).max() > 1e-30)
    report(f"conv2d_{algo}_weight_range_per_cout[{wstd:g}]", worst, 2 * tol)
    assert worst < 2 * tol          # (per channel: 512 outputs each, the bf16 output rounding scatters more than over the tensor)
    # (3) a folded shortcut whose weights are 1e3 x larger than the 3x3 weights of the same cout
    xs = bf(rng.standard_normal((B, S, H, W)))
    ws = bf(1e3 * wstd * rng.standard_normal((Co, S, 1, 1)) / np.sqrt(S))
    ref3 = ref + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
    pw3 = ops.pack_conv_weight(dev(w), dtype=torch.bfloat16, w_sc=dev(ws), S0=S, **kw)
    out3 = from_nhwc(ops.conv2d(x0, pw3, Co, 3, affine=aff, sc0=nhwc(xs, torch.bfloat16), **kw))
    check(f"conv2d_{algo}_weight_range_shortcut[{wstd:g}]", out3, ref3, tol)
    # (4) a near-zero-initialised 3x3 layer (~1e-30) next to O(1) shortcut weights: the cout's power-of-two factor is bounded by the
    # SHORTCUT's magnitude too (wino4_scale_kernel: scaled shortcut weights stay below 2^60), so the shared f32 accumulators stay far
    # from the edge of their range -- F(4,3): also with residual-stream magnitudes of 1e4 (its shortcut GEMM reads bf16; the F(2,3)
    # kernel narrows shortcut inputs to fp16, +-32752: test_conv2d_shortcut_dynamic_range, and `auto` never gives it a shortcut)
    if wstd == 1e-2:
        wt = bf(1e-30 * rng.standard_normal((Co, C, 3, 3)))
        w1 = bf(rng.standard_normal((Co, S, 1, 1)) / np.sqrt(S))
        xs4 = bf((1e4 if w4 else 1.0) * rng.standard_normal((B, S, H, W)))
        ref4 = O.conv2d(xin, wt.astype(np.float64), None) + O.conv2d(xs4.astype(np.float64), w1.astype(np.float64), None)
        pw4 = ops.pack_conv_weight(dev(wt), dtype=torch.bfloat16, w_sc=dev(w1), S0=S, **kw)
        out4 = from_nhwc(ops.conv2d(x0, pw4, Co, 3, affine=aff, sc0=nhwc(xs4, torch.bfloat16), **kw))
        assert np.isfinite(out4).all()
        check(f"conv2d_{algo}_tiny3x3_unit_shortcut", out4, ref4, tol)


def test_small_weight_layers_full_width():
    """One full-width forward with the weights of every ResBlock's Conv_1 (3x3, zero-initialised by the reference: layers.py:100,
    init_scale = 0) and of the pyramid heads scaled by 1e-4: the bf16 mode's kernel choice (`auto`: F(4,3) / F(2,3) / direct by image
    size) against the direct kernel everywhere and against fp32, at the derived one-forward bf16 bound."""
    import flowdec_amd
    g = load_golden("g10_ncsnpp_nf64.npz")
    sd = O.random_state_dict(seed=int(g["seed"]), nf=64)
    small = {k: (v * np.float32(1e-4) if (k.endswith("Conv_1.weight") or (k.endswith(".weight") and v.ndim == 4 and v.shape[0] == 4)) else v) for k, v in sd.items()}
    assert sum(1 for k in sd if not np.array_equal(sd[k], small[k])) >= 20
    outs = {}
    for prec, algo in (("fp32", "auto"), ("bf16", "auto"), ("bf16", "direct")):
        m = flowdec_amd.from_preset("flowdec_75m", precision=prec, conv_algo=algo)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in small.items()}, strict=False)
        m = m.cuda()
        outs[prec, algo] = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda")).cpu().numpy()
        del m
    e_auto, e_direct = rel_err(outs["bf16", "auto"], outs["fp32", "auto"]), rel_err(outs["bf16", "direct"], outs["fp32", "auto"])
    report("small_conv1_weights_forward[bf16 auto vs fp32]", e_auto, TOL_FWD_FULL["bf16"])
    report("small_conv1_weights_forward[bf16 direct vs fp32]", e_direct, TOL_FWD_FULL["bf16"])
    assert e_auto < TOL_FWD_FULL["bf16"] and e_direct < TOL_FWD_FULL["bf16"]
    assert e_auto < 1.5 * e_direct, (e_auto, e_direct)     # the fp16-operand kernels are not the weak link


@pytest.mark.parametrize("mag", [1e5, 5e3, 1e-6])
@pytest.mark.parametrize("case", ["shortcut", "shortcut_cat"])
def test_conv2d_shortcut_dynamic_range(case, mag):
    """The folded 1x1 shortcut reads the UN-NORMALISED residual stream.  Direct kernel (what `conv_algo='auto'` runs whenever a
    shortcut is folded in): bf16 range, parity with the f64 convolution at any magnitude.  Winograd kernel (explicit
    `conv_algo='winograd'` only): raw inputs saturate at +-32752 (half the fp16 range, so that the packed-fp16 input transform
    cannot produce inf): exact parity below that, finite above it; 1e-6 inputs lose bits to fp16 subnormals, which is far below
    the main term."""
    from flowdec_amd import ops
    import zlib
    _, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, S0, S1 = next(c for c in WINO_CASES if c[0] == case)
    rng = np.random.default_rng(zlib.crc32(f"{case}{mag}".encode()))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    x = bf(rng.standard_normal((B, C0, H, W)))
    w = bf(rng.standard_normal((Cout, C0, 3, 3)) / np.sqrt(C0 * 9))
    a = (1 + 0.2 * rng.standard_normal((B, C0))).astype(np.float32)
    d = (0.3 * rng.standard_normal((B, C0))).astype(np.float32)
    xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float32)
    xs = bf(mag * rng.standard_normal((B, S0 + S1, H, W)))
    ws = bf(rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1))
    main = O.conv2d(xin.astype(np.float64), w.astype(np.float64), None)
    ref = main + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
    aff = dev(np.stack([a, d], axis=-1))
    x0 = nhwc(x, torch.bfloat16)
    sc0, sc1 = nhwc(xs[:, :S0], torch.bfloat16), (nhwc(xs[:, S0:], torch.bfloat16) if S1 else None)
    outs = {}
    for algo, wino in (("direct", False), ("winograd", True)) + ((("winograd4", 4),) if Cout == 256 else ()):
        pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=dev(ws), S0=S0, winograd=wino)
        outs[algo] = from_nhwc(ops.conv2d(x0, pw, Cout, 3, affine=aff, sc0=sc0, sc1=sc1, winograd=wino))
        assert np.isfinite(outs[algo]).all(), f"{algo}: non-finite output at |shortcut| ~ {mag:g}"
    if "winograd4" in outs:   # the F(4,3) kernel runs the shortcut as a bf16 GEMM on the raw stream: parity at any magnitude
        check(f"conv2d_shortcut_range[winograd4,{case},{mag:g}]", outs["winograd4"], ref, 6e-3)
    # the direct kernel rounds the activated main operand to bf16 (2^-9 relative to the MAIN term); at mag = 1e5 the main term is a
    # 1e-5 fraction of the output, at 1e-6 the shortcut is
    check(f"conv2d_shortcut_range[direct,{case},{mag:g}]", outs["direct"], ref, 6e-3)
    if np.abs(xs).max() < 32752.0:   # every sample representable: parity
        check(f"conv2d_shortcut_range[winograd,{case},{mag:g}]", outs["winograd"], ref, 6e-3)
    else:                            # saturating, not exact: the documented contract of the explicit Winograd mode
        assert np.abs(outs["winograd"]).max() <= 32752.0 * np.abs(ws).sum(axis=1).max() + np.abs(main).max() + 1.0


def test_auto_keeps_residual_stream_out_of_fp16():
    """`conv_algo='auto'` (the default) never sends a folded-shortcut convolution to the Winograd kernel: with an input 3e5 times
    the usual level the un-normalised residual stream is ~1e5-1e6 at every resolution, far outside fp16 -- `auto` must agree with
    `direct` (they differ only in the shortcut-free convolutions of the low-resolution levels) and with the fp32 mode."""
    import flowdec_amd
    g = load_golden("g10_ncsnpp_nf64.npz")
    sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=int(g["seed"]), nf=64).items()}
    outs = {}
    for prec, algo in (("fp32", "direct"), ("bf16", "direct"), ("bf16", "auto")):
        m = flowdec_amd.from_preset("flowdec_75m", precision=prec, conv_algo=algo)
        m.load_state_dict(sd, strict=False)
        m = m.cuda()
        outs[prec, algo] = m(cu(g["x"]) * 3e5, cu(g["y"]) * 3e5, torch.tensor([0.5], device="cuda")).cpu().numpy()
        assert np.isfinite(outs[prec, algo]).all()
        del m
    check("ncsnpp_nf64_x3e5[bf16,direct]", outs["bf16", "direct"], outs["fp32", "direct"], TOL_FWD_FULL["bf16"])
    check("ncsnpp_nf64_x3e5[bf16,auto]", outs["bf16", "auto"], outs["fp32", "direct"], TOL_FWD_FULL["bf16"])


def test_model_create_rejects_flag_combinations():
    """fd_model_create: at most one convolution-algorithm flag, no FD_TILE_* bits, no unknown bits; FD_NO_SIDE_STREAM is a config
    bit (not an environment variable) and gives bit-identical results."""
    import ctypes as C
    from flowdec_amd import _lib as L
    lib = L.load()

    def create(act):
        cfg = L.FdModelConfig()
        cfg.nf = 8
        for i, c in enumerate((4, 4, 4, 2)):
            cfg.ch_mult[i] = c
        cfg.num_levels, cfg.num_res_blocks, cfg.n_fft, cfg.hop, cfg.alpha, cfg.beta, cfg.act_dtype = 4, 1, 1534, 384, 0.3, 0.33, act
        h = C.c_void_p()
        rc = lib.fd_model_create(C.byref(cfg), C.byref(h))
        if rc == 0:
            lib.fd_model_destroy(h)
        return rc
    assert create(L.FD_BF16 | L.FD_WINOGRAD_AUTO) == 0 and create(L.FD_BF16 | L.FD_NO_SIDE_STREAM) == 0 and create(L.FD_F32) == 0
    for bad in (L.FD_BF16 | L.FD_WINOGRAD | L.FD_LOW_LATENCY, L.FD_BF16 | L.FD_WINOGRAD_AUTO | L.FD_WINOGRAD_LOWRES,
                L.FD_BF16 | L.FD_TILE[64], L.FD_BF16 | L.FD_TILE["32c"] | L.FD_WINOGRAD, L.FD_F32 | L.FD_WINOGRAD, L.FD_BF16 | 0x80000):
        assert create(bad) == -1, hex(bad)
    m1, m2 = make_model(8, 8, "bf16"), None
    import flowdec_amd
    m2 = flowdec_amd.from_preset("flowdec_75m", precision="bf16", nf=8, side_stream=False)
    m2.load_state_dict(m1.state_dict(), strict=False)
    m2 = m2.cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(2, 1, 768, 64, dtype=torch.complex64, device="cuda", generator=gen)
    y = torch.randn(2, 1, 768, 64, dtype=torch.complex64, device="cuda", generator=gen)
    t = torch.tensor([0.4], device="cuda")
    assert torch.equal(m1(x, y, t), m2(x, y, t))


def test_model_calls_from_two_threads():
    """One enqueueing call at a time per fd_model (include/flowdec_hip.h "Threading"): the Python module serialises its callers,
    the C ABI answers a concurrent call with FD_EBUSY -- in neither case is shared state corrupted: every result that comes back
    is bit-identical to the single-threaded one."""
    import ctypes as C
    import threading
    from flowdec_amd import _lib as L
    m = make_model(8, 8, "bf16")
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(1, 1, 768, 64, dtype=torch.complex64, device="cuda", generator=gen)
    y = torch.randn(1, 1, 768, 64, dtype=torch.complex64, device="cuda", generator=gen)
    t = torch.tensor([0.25], device="cuda")
    want = m(x, y, t).clone()
    torch.cuda.synchronize()
    # (a) the module from two threads: all calls succeed (queued by the lock)
    res, errs = [], []

    def via_module():
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(10):
                    o = m(x, y, t); s.synchronize(); res.append(torch.equal(o, want))
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=via_module) for _ in range(2)]
    [a.start() for a in th]; [a.join() for a in th]
    assert not errs and len(res) == 20 and all(res)
    # (b) the raw C entry point from two threads, each with its own output / workspace / stream: a call returns 0 or FD_EBUSY
    lib, h = L.load(), m.backbone.handle()
    need = lib.fd_model_workspace_bytes(h, 1, 64)
    xr, yr = torch.view_as_real(x).contiguous(), torch.view_as_real(y).contiguous()
    rcs, good = [], []

    def via_abi():
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        out = torch.empty_like(x)
        s = torch.cuda.Stream()
        for _ in range(20):
            rc = lib.fd_ncsnpp_forward(h, L.ptr(xr), L.ptr(yr), L.ptr(t), 1, L.ptr(torch.view_as_real(out)), 1, 64, L.ptr(ws), ws.numel(),
                                       C.c_void_p(s.cuda_stream))
            s.synchronize()
            rcs.append(rc)
            if rc == 0:
                good.append(torch.equal(out, want))
    th = [threading.Thread(target=via_abi) for _ in range(2)]
    [a.start() for a in th]; [a.join() for a in th]
    assert set(rcs) <= {0, L.FD_EBUSY} and rcs.count(0) >= 2 and all(good), (rcs.count(0), rcs.count(L.FD_EBUSY))


@pytest.mark.parametrize("algo", ["winograd", "winograd_lowres", "auto", "latency", "direct"])
def test_model_winograd_parity(algo):
    """The whole network with the Winograd kernel on every supported 3x3 convolution (or on the low-resolution levels only):
    full-width forward against the reference golden G10, full enhance against G17 -- at the bf16 mode's tolerances."""
    import flowdec_amd
    g = load_golden("g10_ncsnpp_nf64.npz")
    m = flowdec_amd.from_preset("flowdec_75m", precision="bf16", conv_algo=algo)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=int(g["seed"]), nf=64).items()}, strict=False)
    m = m.cuda()
    out = m(cu(g["x"]), cu(g["y"]), torch.tensor([0.5], device="cuda"))
    check(f"ncsnpp_nf64[bf16,{algo}]", out.cpu().numpy(), g["out"], TOL_FWD_FULL["bf16"])
    g17 = load_golden("g17_enhance_nf64.npz")
    x = m.enhance(torch.from_numpy(g17["y"]), N=6, solver="euler", noise=torch.from_numpy(g17["noise"]))
    check(f"enhance_nf64[euler,N=6,bf16,{algo}]", x.numpy(), g17["euler_N6"], tol_wave_full("bf16", "euler_N6"))


TILE_CASES = [
    # name, B, H, W, C0, C1, Cout, k, affine, skip, S0, S1
    ("plain_256", 1, 32, 16, 64, 0, 256, 3, False, False, 0, 0),
    ("aff_skip_ragged", 2, 24, 20, 96, 0, 128, 3, True, True, 0, 0),
    ("concat_shortcut", 1, 16, 48, 128, 64, 256, 3, True, False, 128, 64),      # 6 shortcut steps
    ("one_chunk", 1, 16, 16, 32, 0, 64, 3, True, False, 32, 0),                  # a single 9-tap chunk + 1 shortcut step
    ("long_shortcut", 1, 16, 16, 64, 0, 128, 3, True, False, 256, 384),          # 20 shortcut steps: chunk-ring falls back
    ("conv1x1", 2, 16, 32, 128, 64, 256, 1, False, False, 0, 0),                 # no 9-tap chunk at all
]


@pytest.mark.parametrize("case", TILE_CASES, ids=[c[0] for c in TILE_CASES])
def test_conv2d_tile_widths(case):
    """fd_conv2d with the FD_TILE_* workgroup widths (incl. the chunk-resident low-latency configurations): the convolution
    result is BIT-IDENTICAL to the default configuration (same K order per output), the statistics agree to f32 rounding."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, k, use_aff, use_skip, S0, S1 = case
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()))
    Cin = C0 + C1
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    aff = torch.stack([1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g), 0.3 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous() \
        if use_aff else None
    bias = torch.randn(B, Cout, device="cuda", generator=g)
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if use_skip else None
    sc0 = sc1 = wsc = None
    if S0:
        sc0 = torch.randn(B, H, W, S0, device="cuda", generator=g).bfloat16()
        sc1 = torch.randn(B, H, W, S1, device="cuda", generator=g).bfloat16() if S1 else None
        wsc = torch.randn(Cout, S0 + S1, 1, 1, device="cuda", generator=g) / (S0 + S1) ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=S0 if S0 else None)
    run = lambda bn: ops.conv2d(x0, pw, Cout, k, x1=x1, affine=aff, bias=bias, skip=sk, scale=0.7071, sc0=sc0, sc1=sc1, want_stats=True, tile_bn=bn)
    ref, ref_st = run(0)
    assert torch.isfinite(ref.float()).all()
    for bn in (128, 64, 32, "64c", "32c", "duo", "persist"):
        out, st = run(bn)
        assert torch.equal(out, ref), (name, bn)
        assert torch.allclose(st[:, :, :Cout], ref_st[:, :, :Cout], rtol=1e-4, atol=1e-3), (name, bn)


RE_CASES = [
    # name, B, H, W, C0, C1, Cout, affine, bias_rows (0 = none, 1, -1 = B), stats, S0, S1, other width
    ("many_tiles_256", 3, 256, 160, 32, 0, 256, True, -1, True, 0, 0, 128),            # 480 tiles on <= 256 workgroups, 3 images
    ("many_tiles_shortcut_128", 5, 96, 176, 64, 32, 128, True, 1, True, 32, 0, 64),    # BN = 128 configuration, 1-tap steps at the end of a tile
    ("many_tiles_plain_nostats", 2, 208, 208, 32, 0, 256, False, 0, False, 0, 0, 128),  # no activation, no bias, no statistics
    ("many_tiles_cat_shortcut_256", 2, 176, 160, 64, 64, 256, True, -1, True, 64, 32, 128),
    ("one_tile_per_image", 9, 16, 16, 64, 0, 256, True, -1, True, 0, 0, 128),           # every tile boundary is an image boundary (when ranges hold > 1 tile)
]


@pytest.mark.parametrize("case", RE_CASES, ids=[c[0] for c in RE_CASES])
def test_conv2d_register_epilogue_continuous_tiles(case):
    """The persistent register-epilogue configuration of the direct kernel (conv_mfma.hip `RE`, FD_TILE_PERSIST: whole tiles, Cout ==
    workgroup width, no residual input -- the shape of every full-resolution convolution of the model): MORE tiles than compute units, so that each
    workgroup walks several tiles (weight stream wrap-around, next tile's halo prefetched in the last chunk, stores left in flight
    across the tile boundary) and crosses image boundaries (second affine / bias table).  The convolution result must be
    BIT-IDENTICAL to the staged-epilogue kernel at another workgroup width, the statistics equal to f32 rounding, and a second
    launch must reproduce the first (no race on the ring / halo hand-over)."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, want_stats, S0, S1, other = case
    g = torch.Generator(device="cuda").manual_seed(zlib.crc32(name.encode()))
    Cin = C0 + C1
    x0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
    x1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5
    aff = torch.stack([1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g), 0.3 * torch.randn(B, Cin, device="cuda", generator=g)], -1).contiguous() \
        if use_aff else None
    bias = None if bias_rows == 0 else torch.randn((Cout,) if bias_rows == 1 else (B, Cout), device="cuda", generator=g)
    sc0 = sc1 = wsc = None
    if S0:
        sc0 = torch.randn(B, H, W, S0, device="cuda", generator=g).bfloat16()
        sc1 = torch.randn(B, H, W, S1, device="cuda", generator=g).bfloat16() if S1 else None
        wsc = torch.randn(Cout, S0 + S1, 1, 1, device="cuda", generator=g) / (S0 + S1) ** 0.5
    pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.bfloat16, w_sc=wsc, S0=S0 if S0 else None)

    def run(bn):
        r = ops.conv2d(x0, pw, Cout, 3, x1=x1, affine=aff, bias=bias, scale=0.7071, sc0=sc0, sc1=sc1, want_stats=want_stats, tile_bn=bn)
        return r if want_stats else (r, None)
    out, st = run("persist")
    ref, ref_st = run(other)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    dflt, _ = run(0)
    assert torch.equal(dflt, ref)
    bad = (out != ref).nonzero()
    assert bad.numel() == 0, (name, "first mismatches (b, h, w, c):", bad[:8].tolist(), "count", int(bad.shape[0]))
    if want_stats:
        assert torch.allclose(st[:, :, :Cout], ref_st[:, :, :Cout], rtol=1e-4, atol=2e-3), (name, float((st[:, :, :Cout] - ref_st[:, :, :Cout]).abs().max()))
    for _ in range(3):
        again, st2 = run("persist")
        assert torch.equal(again, out)
        if want_stats:
            assert torch.equal(st2, st)


def test_mixed_precision_mode():
    """precision='mixed' (FD_F32 | FD_BF16_OPERANDS): f32 activations / residual stream / skip tensors, conv inputs rounded to bf16 at
    the LDS store, bf16 weights, f32 accumulation -- the "bf16 operands, f32 residual stream" mode of VERDICT r1 item 4.  One conv
    against the f64 convolution of the bf16-rounded operands (only accumulation-order error is left), then the full-width network
    against the reference goldens G10 / G17."""
    from flowdec_amd import ops
    import flowdec_amd
    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W, C0, C1, Cout, S = 2, 24, 40, 64, 32, 256, 64
    Cin = C0 + C1
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    a = 1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g)
    d = 0.3 * torch.randn(B, Cin, device="cuda", generator=g)
    xs = torch.randn(B, H, W, S, device="cuda", generator=g)
    ws = (torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5).bfloat16().float()
    bias = torch.randn(Cout, device="cuda", generator=g)
    sk = torch.randn(B, H, W, Cout, device="cuda", generator=g)
    v = x * a[:, None, None, :] + d[:, None, None, :]
    act = (v * torch.sigmoid(v)).bfloat16().double()
    ref = torch.nn.functional.conv2d(act.permute(0, 3, 1, 2), w.double(), padding=1) + \
        torch.nn.functional.conv2d(xs.bfloat16().double().permute(0, 3, 1, 2), ws.double())
    ref = ((ref + bias.double()[None, :, None, None]).permute(0, 2, 3, 1) + sk.double()) * 0.7071
    pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.float32, w_sc=ws, bf16_operands=True)
    out, st = ops.conv2d(x[..., :C0].contiguous(), pw, Cout, 3, x1=x[..., C0:].contiguous(), affine=torch.stack([a, d], -1).contiguous(), bias=bias,
                         skip=sk, scale=0.7071, sc0=xs, want_stats=True, bf16_operands=True)
    assert out.dtype == torch.float32
    err = float((out.double() - ref).norm() / ref.norm())
    report("conv2d[mixed: f32 storage, bf16 operands]", err, 1e-3)     # bf16 rounding flips of silu() near ties + f32 accumulation order
    assert err < 1e-3
    rs = torch.stack([ref.sum((1, 2)), (ref ** 2).sum((1, 2))], -1)
    assert float((st.double().sum(1)[:, :Cout] - rs).abs().max() / rs.abs().max()) < 1e-3

    g10 = load_golden("g10_ncsnpp_nf64.npz")
    m = flowdec_amd.from_preset("flowdec_75m", precision="mixed")
    m.load_state_dict({k: torch.from_numpy(v_) for k, v_ in O.random_state_dict(seed=int(g10["seed"]), nf=64).items()}, strict=False)
    m = m.cuda()
    o = m(cu(g10["x"]), cu(g10["y"]), torch.tensor([0.5], device="cuda"))
    check("ncsnpp_nf64[mixed]", o.cpu().numpy(), g10["out"], TOL_FWD["mixed"])
    g17 = load_golden("g17_enhance_nf64.npz")
    xh = m.enhance(torch.from_numpy(g17["y"]), N=6, solver="euler", noise=torch.from_numpy(g17["noise"]))
    check("enhance_nf64[euler,N=6,mixed]", xh.numpy(), g17["euler_N6"], TOL_WAVE_FULL["mixed"])


def test_bf16x3_precision_mode():
    """precision='bf16x3' (FD_F32 | FD_BF16X3_OPERANDS): f32 storage, every conv operand as hi + lo bf16 pair, three bf16 MFMAs per
    product.  Must meet the FP32 mode's tolerances: one conv against the f64 convolution of the UNROUNDED operands, then the
    full-width network against the reference goldens G10 (one forward) and G17 (enhance) at TOL_FWD / TOL_WAVE_FULL['fp32']."""
    from flowdec_amd import ops
    import flowdec_amd
    g = torch.Generator(device="cuda").manual_seed(12)
    for (B, H, W, C0, C1, Cout, S, k) in [(2, 24, 40, 64, 32, 256, 64, 3), (1, 20, 16, 8, 0, 64, 0, 3), (2, 16, 16, 128, 0, 4, 0, 3), (1, 16, 32, 64, 0, 128, 0, 1)]:
        Cin = C0 + C1
        x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
        w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
        a = 1 + 0.2 * torch.randn(B, Cin, device="cuda", generator=g)
        d = 0.3 * torch.randn(B, Cin, device="cuda", generator=g)
        v = (x * a[:, None, None, :] + d[:, None, None, :]).double()
        ref = torch.nn.functional.conv2d((v * torch.sigmoid(v)).permute(0, 3, 1, 2), w.double(), padding=k // 2)
        xs = ws = None
        if S:
            xs = torch.randn(B, H, W, S, device="cuda", generator=g)
            ws = torch.randn(Cout, S, 1, 1, device="cuda", generator=g) / S ** 0.5
            ref = ref + torch.nn.functional.conv2d(xs.double().permute(0, 3, 1, 2), ws.double())
        bias = torch.randn(Cout, device="cuda", generator=g)
        sk = torch.randn(B, H, W, Cout, device="cuda", generator=g)
        ref = ((ref + bias.double()[None, :, None, None]).permute(0, 2, 3, 1) + sk.double()) * 0.7071
        pw = ops.pack_conv_weight(w, C0=C0, dtype=torch.float32, w_sc=ws, bf16_operands="x3")
        out = ops.conv2d(x[..., :C0].contiguous(), pw, Cout, k, x1=x[..., C0:].contiguous() if C1 else None, affine=torch.stack([a, d], -1).contiguous(),
                         bias=bias, skip=sk, scale=0.7071, sc0=xs, bf16_operands="x3")
        err = float((out.double() - ref).norm() / ref.norm())
        report(f"conv2d[bf16x3,{Cin}->{Cout},k{k}]", err, 5e-5)
        assert out.dtype == torch.float32 and err < 5e-5, err

    g10 = load_golden("g10_ncsnpp_nf64.npz")
    m = flowdec_amd.from_preset("flowdec_75m", precision="bf16x3")
    m.load_state_dict({k_: torch.from_numpy(v_) for k_, v_ in O.random_state_dict(seed=int(g10["seed"]), nf=64).items()}, strict=False)
    m = m.cuda()
    o = m(cu(g10["x"]), cu(g10["y"]), torch.tensor([0.5], device="cuda"))
    check("ncsnpp_nf64[bf16x3]", o.cpu().numpy(), g10["out"], TOL_FWD["fp32"])
    g17 = load_golden("g17_enhance_nf64.npz")
    xh = m.enhance(torch.from_numpy(g17["y"]), N=6, solver="euler", noise=torch.from_numpy(g17["noise"]))
    check("enhance_nf64[euler,N=6,bf16x3]", xh.numpy(), g17["euler_N6"], TOL_WAVE_FULL["fp32"])
    xm = m.enhance(torch.from_numpy(g17["y"]), N=3, solver="midpoint", noise=torch.from_numpy(g17["noise"]))
    check("enhance_nf64[midpoint,N=3,bf16x3]", xm.numpy(), g17["midpoint_N3"], TOL_WAVE_FULL["fp32"])


@pytest.mark.parametrize("Cout,B,H,W", [(64, 2, 32, 48), (8, 1, 16, 16), (16, 2, 48, 16), (32, 1, 16, 32), (64, 1, 768, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_in(Cout, B, H, W, dtype):
    """fd_conv_in (the vector-FMA input convolution 4 -> nf of the bf16 mode, all_modules.3) against the f64 convolution of the same
    stored input: f32 storage 2e-6 (f32 fma chains of 36 terms), bf16 storage = its output rounding; the GroupNorm partial sums are
    those of the unrounded f32 values; channels 4..7 of the packed input are ignored; bit-deterministic; and it agrees with the MFMA
    kernel on the zero-padded 8-channel weights (the path of the other modes)."""
    from flowdec_amd import ops
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"convin{Cout}{H}{W}".encode()))
    x4 = rng.standard_normal((B, 4, H, W)).astype(np.float32)
    if dtype == torch.bfloat16:
        x4 = O.round_bf16(x4)
    w = (rng.standard_normal((Cout, 4, 3, 3)) / 6).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    junk = rng.standard_normal((B, 4, H, W)).astype(np.float32)           # channels 4..7: must not matter
    in8 = nhwc(np.concatenate([x4, junk], 1), dtype)
    ref = O.conv2d(x4.astype(np.float64), w.astype(np.float64), bias.astype(np.float64))
    out, stats = ops.conv_in(in8, dev(w), dev(bias))
    torch.cuda.synchronize()
    check(f"conv_in[{Cout},{H}x{W},{str(dtype)[6:]}]", from_nhwc(out), ref, 2e-6 if dtype == torch.float32 else 3e-3)
    st = stats.double().sum(dim=1).cpu().numpy()
    ref_s = np.stack([ref.sum(axis=(2, 3)), (ref ** 2).sum(axis=(2, 3))], axis=-1)
    e = float(np.abs(st - ref_s).max() / np.abs(ref_s).max())
    report(f"conv_in_stats[{Cout},{H}x{W},{str(dtype)[6:]}]", e, 1e-5)
    assert e < 1e-5
    out2, stats2 = ops.conv_in(in8, dev(w), dev(bias))
    assert torch.equal(out, out2) and torch.equal(stats, stats2)
    # the MFMA path on the zero-padded weights
    w8 = np.zeros((Cout, 8, 3, 3), np.float32); w8[:, :4] = w
    in8z = nhwc(np.concatenate([x4, np.zeros_like(x4)], 1), dtype)
    pw = ops.pack_conv_weight(dev(w8), dtype=dtype)
    outm = ops.conv2d(in8z, pw, Cout, 3, bias=dev(bias))
    assert rel_err(from_nhwc(out), from_nhwc(outm)) < (2e-6 if dtype == torch.float32 else 5e-3)


def _pow2_weight_scale(maxabs, u_exp=9):
    """The pack kernels' per-cout scale (conv_wino4.hip wino4_scale_kernel / conv_wino.hip wino_scale_kernel): 2^k with maxabs 2^k in
    [2^(u_exp - 1), 2^u_exp); 1 for an all-zero row."""
    m = np.asarray(maxabs, np.float32)
    _, e = np.frexp(m)
    k = np.clip(u_exp - e, -100, 100)
    return np.where(m > 0, np.ldexp(np.float32(1), k), np.float32(1)).astype(np.float32)


def _wino4_numerics_model(x, w, a=None, d=None):
    """NumPy model of conv_wino4.hip's ARITHMETIC for a raw (un-activated) bf16 input: fp16 operands (z = fp16(x); V = B^T z in the
    kernel's packed-fp16 operation order, every fma rounded once; U = fp16(G g) from the pack kernel's f32 expressions), exact products,
    wide accumulation over (channel, kernel row), output transform A^T.  Returns the f32 result before bias / residual / scale / output
    rounding.  x: [B, C, H, W] (bf16-representable float32), w: [Co, C, 3, 3] float32."""
    B, C, H, W = x.shape
    f16 = np.float16
    z = np.zeros((B, C, H + 2, W + 6), f16)                      # zero halo: one row / column around (+ room up to column 4 j + 4)
    if a is None:
        z[:, :, 1:H + 1, 1:W + 1] = np.clip(x, -6000.0, 6000.0).astype(f16)
    else:       # GroupNorm + SiLU operand transform in f32: u = fma(x, a, d), silu(u) = u / (1 + exp(-u)); zero padding AFTER the activation
        u = (x.astype(np.float64) * a[:, :, None, None].astype(np.float64) + d[:, :, None, None].astype(np.float64)).astype(np.float32)
        z[:, :, 1:H + 1, 1:W + 1] = np.minimum((u / (np.float32(1) + np.exp(-u))).astype(np.float32).astype(f16), f16(6000.0))
    fma = lambda a, b, c: (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f16)   # one rounding, like v_pk_fma_f16
    d = [z[:, :, :, k:k + W:4] for k in range(6)]                # d_k of tile j = padded column 4 j + k = pixel 4 j - 1 + k
    t1, t2 = fma(d[2], -4.0, d[4]), fma(d[1], -4.0, d[3])
    t3, t4 = (d[4].astype(np.float64) - d[2].astype(np.float64)).astype(f16), (d[3].astype(np.float64) - d[1].astype(np.float64)).astype(f16)
    V = [fma(d[0], 4.0, fma(d[2], -5.0, d[4])),
         (t1.astype(np.float64) + t2.astype(np.float64)).astype(f16), (t1.astype(np.float64) - t2.astype(np.float64)).astype(f16),
         fma(t4, 2.0, t3), fma(t4, -2.0, t3),
         fma(d[1], 4.0, fma(d[3], -5.0, d[5]))]
    g0, g1, g2 = (w[..., k].astype(np.float32) for k in range(3))   # [Co, C, 3 (dy)]
    six, tf, tw = np.float32(6), np.float32(24), np.float32(12)
    U = [np.float32(0.25) * g0, -(g0 + g1 + g2) / six, (-g0 + g1 - g2) / six, g0 / tf + g1 / tw + g2 / six, g0 / tf - g1 / tw + g2 / six, g2]
    # per-cout power-of-two scale before the fp16 rounding, undone exactly in the epilogue
    sc = _pow2_weight_scale(np.max([np.abs(u).max(axis=(1, 2)) for u in U], axis=0))[:, None, None]
    U = [(u * sc).astype(f16).astype(np.float64) / sc.astype(np.float64) for u in U]
    M = []
    for xi in range(6):
        v = V[xi].astype(np.float64)                              # [B, C, H + 2, W / 4]
        acc = np.zeros((B, w.shape[0], H, W // 4))
        for dy in range(3):
            acc += np.einsum("oc,bchj->bohj", U[xi][:, :, dy], v[:, :, dy:dy + H, :])
        M.append(acc)
    y = np.zeros((B, w.shape[0], H, W))
    y[..., 0::4] = M[0] + M[1] + M[2] + M[3] + M[4]
    y[..., 1::4] = M[1] - M[2] + 2 * M[3] - 2 * M[4]
    y[..., 2::4] = M[1] + M[2] + 4 * M[3] + 4 * M[4]
    y[..., 3::4] = M[1] - M[2] + 8 * M[3] - 8 * M[4] + M[5]
    return y.astype(np.float32)


@pytest.mark.parametrize("case", WINO4_CASES, ids=[c[0] for c in WINO4_CASES])
def test_conv2d_winograd4_matches_its_numerics_model(case):
    """The F(4,3) kernel against a NumPy model of its own arithmetic (fp16 operand roundings in the kernel's operation order, exact
    products, wide accumulation): what is left between the two is the f32 summation order of the matrix cores, i.e. a bf16 output that
    differs from the model's in a handful of last-place roundings -- 20x tighter than the parity bound against the f64 convolution, so a
    misplaced tap, a wrong transform coefficient or a lost channel cannot hide behind the bf16 tolerance."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, S0, S1 = case
    rng = np.random.default_rng(zlib.crc32(("w4m" + name).encode()))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    Cin = C0 + C1
    x = bf(rng.standard_normal((B, Cin, H, W)))
    w = bf(rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9))
    aff = av = dv = None
    if use_aff:
        av = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        dv = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([av, dv], axis=-1))
    y = _wino4_numerics_model(x, w, av, dv).astype(np.float32)
    sc0 = sc1 = w_sc = None
    if S0:   # the folded 1x1 shortcut: bf16 x bf16 products on the raw tensors, added to the same planes
        xs = bf(rng.standard_normal((B, S0 + S1, H, W)))
        ws = (rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1)).astype(np.float32)
        y = (y + np.einsum("oc,bchw->bohw", bf(ws)[:, :, 0, 0].astype(np.float64), xs.astype(np.float64))).astype(np.float32)
        sc0 = nhwc(xs[:, :S0], torch.bfloat16)
        sc1 = nhwc(xs[:, S0:], torch.bfloat16) if S1 else None
        w_sc = dev(ws)
    bias = None
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
    skip, scale = None, np.float32(1.0)
    if use_skip:
        sk = bf(rng.standard_normal((B, Cout, H, W)))
        skip = nhwc(sk, torch.bfloat16)
        y = y + sk
        scale = np.float32(1 / np.sqrt(2))
    if bias_rows:
        y = y + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])   # the kernel: m (scale / wscale) + (skip scale + bias scale), fused: <= 2 f32 ulps from this
    model = bf(y * scale)
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, w_sc=w_sc, S0=S0 if S0 else None, winograd=4)
    out = ops.conv2d(nhwc(x[:, :C0], torch.bfloat16), pw, Cout, 3, x1=nhwc(x[:, C0:], torch.bfloat16) if C1 else None, affine=aff, bias=bias,
                     skip=skip, scale=float(scale), sc0=sc0, sc1=sc1, winograd=4)
    got = from_nhwc(out)
    e = rel_err(got, model)
    differing = float(np.mean(got != model))
    report(f"conv2d_winograd4_vs_numerics_model[{name}]", e, 2e-4)
    assert e < 2e-4 and differing < 0.01, (e, differing)


def _wino2_numerics_model(x, w, a=None, d=None):
    """NumPy model of conv_wino.hip's arithmetic (Winograd F(2,3) along W, even W): z = fp16(x) or fp16(silu(a x + d)), V = B^T z with
    one fp16 rounding per entry (d0 - d2, d1 + d2, d2 - d1, d1 - d3), U = fp16 of the pack kernel's f32 expressions, exact products,
    wide accumulation, y0 = M0 + M1 + M2, y1 = M1 - M2 - M3."""
    B, C, H, W = x.shape
    f16 = np.float16
    z = np.zeros((B, C, H + 2, W + 2), f16)
    if a is None:
        z[:, :, 1:H + 1, 1:W + 1] = x.astype(f16)
    else:
        u = (x.astype(np.float64) * a[:, :, None, None].astype(np.float64) + d[:, :, None, None].astype(np.float64)).astype(np.float32)
        z[:, :, 1:H + 1, 1:W + 1] = (u / (np.float32(1) + np.exp(-u))).astype(np.float32).astype(f16)
    dd = [z[:, :, :, k:k + W:2].astype(np.float64) for k in range(4)]     # d_k of tile j = padded column 2 j + k
    V = [(dd[0] - dd[2]).astype(f16), (dd[1] + dd[2]).astype(f16), (dd[2] - dd[1]).astype(f16), (dd[1] - dd[3]).astype(f16)]
    g0, g1, g2 = (w[..., k].astype(np.float32) for k in range(3))
    h = np.float32(0.5)
    U = [g0, h * (g0 + g1 + g2), h * (g0 - g1 + g2), g2]
    sc = _pow2_weight_scale(np.max([np.abs(u).max(axis=(1, 2)) for u in U], axis=0))[:, None, None]
    M = []
    for xi in range(4):
        u64, v = (U[xi] * sc).astype(f16).astype(np.float64) / sc.astype(np.float64), V[xi].astype(np.float64)
        acc = np.zeros((B, w.shape[0], H, W // 2))
        for dy in range(3):
            acc += np.einsum("oc,bchj->bohj", u64[:, :, dy], v[:, :, dy:dy + H, :])
        M.append(acc)
    y = np.zeros((B, w.shape[0], H, W))
    y[..., 0::2] = M[0] + M[1] + M[2]
    y[..., 1::2] = M[1] - M[2] - M[3]
    return y.astype(np.float32)


@pytest.mark.parametrize("case", [c for c in WINO_CASES if not c[10] and c[3] % 2 == 0], ids=[c[0] for c in WINO_CASES if not c[10] and c[3] % 2 == 0])
def test_conv2d_winograd_matches_its_numerics_model(case):
    """The F(2,3) kernel against a NumPy model of its own arithmetic, like test_conv2d_winograd4_matches_its_numerics_model."""
    from flowdec_amd import ops
    import zlib
    name, B, H, W, C0, C1, Cout, use_aff, bias_rows, use_skip, _, _ = case
    rng = np.random.default_rng(zlib.crc32(("w2m" + name).encode()))
    bf = lambda a: O.round_bf16(np.asarray(a, np.float32))
    Cin = C0 + C1
    x = bf(rng.standard_normal((B, Cin, H, W)))
    w = bf(rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9))
    aff = av = dv = None
    if use_aff:
        av = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        dv = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([av, dv], axis=-1))
    y = _wino2_numerics_model(x, w, av, dv)
    bias = None
    skip, scale = None, np.float32(1.0)
    if use_skip:
        sk = bf(rng.standard_normal((B, Cout, H, W)))
        skip = nhwc(sk, torch.bfloat16)
        y = y + sk
        scale = np.float32(1 / np.sqrt(2))
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
        y = y + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])
    model = bf(y * scale)
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=torch.bfloat16, winograd=True)
    out = ops.conv2d(nhwc(x[:, :C0], torch.bfloat16), pw, Cout, 3, x1=nhwc(x[:, C0:], torch.bfloat16) if C1 else None, affine=aff, bias=bias,
                     skip=skip, scale=float(scale), winograd=True)
    got = from_nhwc(out)
    e = rel_err(got, model)
    report(f"conv2d_winograd_vs_numerics_model[{name}]", e, 2e-4)
    assert e < 2e-4 and float(np.mean(got != model)) < 0.01, e
