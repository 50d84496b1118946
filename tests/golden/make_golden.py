#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE implementation (/root/reference) on CPU.

Runs only in the build container (the reference does not exist on the GPU box and none of
its code travels -- only the arrays written here do).  Import recipe: SURVEY.md Appendix B
(stub the CUDA JIT loader, skip flowdec/__init__.py, dummy modules for the training-only
dependencies, and a restated fixed-step torchdyn.NeuralODE -- torchdyn itself is not
installed, so the solver driver is NOT pinned by these vectors).

Weights are NOT stored: they are re-derived from `oracle.flowdec_oracle.random_state_dict`
(seeded NumPy), loaded into the reference modules here and into the oracle / HIP path in the
tests.  Inputs are stored (they are small) so the fixtures are self-contained data.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, manifest.json
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FLOWDEC_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import flowdec_oracle as O  # weight generator + manifest only


# --------------------------------------------------------------------------------------
# import recipe
# --------------------------------------------------------------------------------------
def _install_stubs():
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: types.SimpleNamespace()
    pkg = types.ModuleType("flowdec"); pkg.__path__ = [os.path.join(REF, "flowdec")]
    sys.modules["flowdec"] = pkg
    sys.path.insert(0, REF)

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    class _Permissive(types.ModuleType):
        def __getattr__(self, item):  # any other attribute (wandb.Audio, ...) -> dummy class
            if item.startswith("__"):
                raise AttributeError(item)
            return _Dummy

    def mod(name, **attrs):
        m = _Permissive(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=_Dummy, Callback=_Dummy)
    mod("pytorch_lightning.loggers", WandbLogger=_Dummy, TensorBoardLogger=_Dummy)
    mod("pytorch_lightning.plugins", environments=None)
    mod("pytorch_lightning.plugins.environments", SLURMEnvironment=_Dummy)
    mod("omegaconf", OmegaConf=types.SimpleNamespace(create=lambda x: x), DictConfig=dict, ListConfig=list)
    mod("wandb")
    mod("torchcfm", ConditionalFlowMatcher=_Dummy)
    mod("hydra", utils=None)
    mod("hydra.utils", instantiate=lambda *a, **k: None)
    ta = mod("torchaudio", load=None, save=None)
    ta.transforms = mod("torchaudio.transforms", Resample=_Dummy, Spectrogram=_Dummy, MelSpectrogram=_Dummy)
    ta.functional = mod("torchaudio.functional")
    mod("librosa"); mod("pystoi", stoi=None); mod("pesq", pesq=None); mod("speechmos", dnsmos=None)
    mod("onnxruntime"); mod("torch_pesq", PesqLoss=_Dummy)
    ps = mod("pysepm"); ps.qualityMeasures = mod("pysepm.qualityMeasures", SNRseg=None, fwSNRseg=None)
    mod("pandas") if "pandas" not in sys.modules else None

    # torchdyn stand-in: fixed-step driver restated -- NOT the real package.
    class NeuralODE:
        def __init__(self, vf, solver="euler", **kw):
            self.vf, self.solver = vf, solver

        def trajectory(self, x, t_span):
            t = t_span[0]; dt = t_span[1] - t_span[0]
            traj = [x]
            for i in range(1, len(t_span)):
                s = self.solver
                if s == "euler":
                    x = x + dt * self.vf(t, x)
                elif s == "midpoint":
                    xm = x + 0.5 * dt * self.vf(t, x)
                    x = x + dt * self.vf(t + 0.5 * dt, xm)
                else:  # reference's own DiffEqSolver subclasses (sampling/solvers.py)
                    _, x, _ = s.step(self.vf, x, t, dt)
                traj.append(x)
                t = t + dt
                if i < len(t_span) - 1:
                    dt = t_span[i + 1] - t
            return torch.stack(traj)

    class DiffEqSolver:
        def __init__(self, order=1, **kw):
            self.order = order

    mod("torchdyn"); mod("torchdyn.core", NeuralODE=NeuralODE)
    mod("torchdyn.numerics"); mod("torchdyn.numerics.solvers")
    mod("torchdyn.numerics.solvers.templates", DiffEqSolver=DiffEqSolver)


def to_t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def crandn(rng, shape):
    """complex standard normal like torch.randn_like(complex): var 1/2 per part."""
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    _install_stubs()
    from flowdec.backbones.ncsnpp import NCSNpp
    from flowdec.backbones.ncsnpp_utils import layerspp, up_or_down_sampling
    from flowdec.backbones.ncsnpp_utils.op.upfirdn2d import upfirdn2d_native
    from flowdec.data.feature_extractors import AmplitudeCompressedComplexSTFT
    from flowdec.data import sigma_models
    from flowdec.util.other import pad_spec, normalize_noisy

    rng = np.random.default_rng(1234)
    out = {}

    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000,
                                        alpha=0.3, beta=0.33)

    # ---- G1/G2: STFT, compression, iSTFT --------------------------------------------
    y = (0.1 * rng.standard_normal((2, 1, 4800))).astype(np.float32)
    yt = torch.from_numpy(y)
    S = fe.complex_stft(yt)
    C = fe.compress(S)
    Sinv = fe.compress.invert(C)
    yinv = fe.complex_stft.invert(Sinv, orig_length=4800)
    # iSTFT of an arbitrary (non-consistent) spectrogram, with length shorter / longer than natural
    Z = crandn(rng, (2, 1, 768, 9))
    zi_a = fe.invert(torch.from_numpy(Z), orig_length=3000)
    zi_b = fe.invert(torch.from_numpy(Z), orig_length=8 * 384)
    np.savez_compressed(os.path.join(HERE, "g1_stft.npz"), y=y, window=fe.complex_stft.window.numpy(),
                        stft=S.numpy(), compressed=C.numpy(), decompressed=Sinv.numpy(), roundtrip=yinv.numpy(),
                        Z=Z, istft_len3000=zi_a.numpy(), istft_len3072=zi_b.numpy())

    # ---- G3: pad_spec / normalize_noisy ---------------------------------------------
    yz = y.copy(); yz[1] = 0.0
    yn, _, nf_ = normalize_noisy(torch.from_numpy(yz), mode="noisy")
    P, undo = pad_spec(C, mode="zero")
    np.savez_compressed(os.path.join(HERE, "g3_pad_norm.npz"), y=yz, y_norm=yn.numpy(), normfac=nf_.numpy(),
                        spec=C.numpy(), padded_T=np.int64(P.shape[-1]), orig_T=np.int64(C.shape[-1]))

    # ---- G4: upfirdn2d (native path) ------------------------------------------------
    xa = rng.standard_normal((2, 3, 12, 8)).astype(np.float32)
    xb = rng.standard_normal((1, 4, 768, 16)).astype(np.float32)
    k2 = rng.standard_normal((3, 2)).astype(np.float32)  # asymmetric generic kernel
    g4 = dict(xa=xa, xb=xb, k2=k2)
    for nm, x in (("a", xa), ("b", xb)):
        g4[f"up_{nm}"] = up_or_down_sampling.upsample_2d(torch.from_numpy(x), (1, 3, 3, 1), factor=2).numpy()
        g4[f"down_{nm}"] = up_or_down_sampling.downsample_2d(torch.from_numpy(x), (1, 3, 3, 1), factor=2).numpy()
    g4["generic_a"] = upfirdn2d_native(torch.from_numpy(xa), torch.from_numpy(k2), 2, 3, 1, 2, 1, 2, 0, 1).numpy()
    np.savez_compressed(os.path.join(HERE, "g4_upfirdn2d.npz"), **g4)

    # ---- G5: GroupNorm + SiLU ---------------------------------------------------------
    g5 = {}
    for Cc in (64, 256, 320, 384, 512):
        x = (rng.standard_normal((2, Cc, 8, 8)) * 1.5 + 0.3).astype(np.float32)
        gam = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32); bet = (0.1 * rng.standard_normal(Cc)).astype(np.float32)
        gn = torch.nn.GroupNorm(min(Cc // 4, 32), Cc, eps=1e-6)
        gn.weight.data = torch.from_numpy(gam); gn.bias.data = torch.from_numpy(bet)
        g5[f"x{Cc}"] = x; g5[f"gamma{Cc}"] = gam; g5[f"beta{Cc}"] = bet
        g5[f"out{Cc}"] = torch.nn.functional.silu(gn(torch.from_numpy(x))).numpy()
    np.savez_compressed(os.path.join(HERE, "g5_groupnorm_silu.npz"), **g5)

    # ---- G6: single ResnetBlockBigGANpp, full channel widths, small H x W -------------
    g6 = {}
    act = torch.nn.SiLU()
    for nm, (seed, ci, co, up, down) in O.RESBLOCK_CASES.items():
        blk = layerspp.ResnetBlockBigGANpp(act=act, in_ch=ci, out_ch=co, temb_dim=256, up=up, down=down, dropout=0.0,
                                           fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, init_scale=0.0).eval()
        sd = O.random_resblock_params(seed, ci, co, has_conv2=(ci != co or up or down))
        assert set(sd) == set(blk.state_dict())
        blk.load_state_dict(to_t(sd))
        r2 = np.random.default_rng(seed + 100)
        x = r2.standard_normal((2, ci, 16, 8)).astype(np.float32)
        temb = r2.standard_normal((1, 256)).astype(np.float32)
        o = blk(torch.from_numpy(x), torch.from_numpy(temb)).numpy()
        g6[f"{nm}_x"] = x; g6[f"{nm}_temb"] = temb; g6[f"{nm}_out"] = o
    np.savez_compressed(os.path.join(HERE, "g6_resblock.npz"), **g6)

    # ---- G7/G8: time embedding + full NCSN++ (nf=8) ------------------------------------
    bb_kw = dict(nonlinearity="swish", ch_mult=(4, 4, 4, 2), num_res_blocks=1, attn_resolutions=[], resamp_with_conv=True,
                 conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan",
                 progressive="output_skip", progressive_input="input_skip", progressive_combine="sum", init_scale=0.0,
                 fourier_scale=16, image_size=768, embedding_type="fourier", dropout=0.0, num_channels=4,
                 output_layer_kwargs=dict(kernel_size=1, bias=False, padding="same", padding_mode="zeros"),
                 bottleneck_attn=False)
    net8 = NCSNpp(nf=8, **bb_kw).eval()
    sd8 = O.random_state_dict(seed=8, nf=8)
    net8.load_state_dict(to_t(strip(sd8, "backbone.")))
    tvals = np.array([0.0, 1.0 / 6.0, 0.25, 0.5, 1.0], dtype=np.float32)
    m = net8.all_modules
    temb = m[2](torch.nn.functional.silu(m[1](m[0](torch.from_numpy(tvals)))))
    x8 = crandn(rng, (2, 1, 768, 64)); y8 = crandn(rng, (2, 1, 768, 64))
    o8 = net8(torch.from_numpy(x8), torch.from_numpy(y8), torch.tensor([0.25]))
    o8b = net8(torch.from_numpy(x8), torch.from_numpy(y8), torch.tensor([0.1, 0.9]))  # per-sample t
    np.savez_compressed(os.path.join(HERE, "g8_ncsnpp_nf8.npz"), t=tvals, temb=temb.numpy(), x=x8, y=y8,
                        out_t025=o8.numpy(), out_t01_09=o8b.numpy(), seed=np.int64(8))

    # ---- G10: full-width NCSN++ (nf=64), one forward on [1,1,768,64] -------------------
    net64 = NCSNpp(nf=64, **bb_kw).eval()
    sd64 = O.random_state_dict(seed=64, nf=64)
    net64.load_state_dict(to_t(strip(sd64, "backbone.")))
    x64 = crandn(rng, (1, 1, 768, 64)); y64 = crandn(rng, (1, 1, 768, 64))
    o64 = net64(torch.from_numpy(x64), torch.from_numpy(y64), torch.tensor([0.5]))
    np.savez_compressed(os.path.join(HERE, "g10_ncsnpp_nf64.npz"), x=x64, y=y64, out=o64.numpy(),
                        out_sum=np.float64(o64.numpy().astype(np.complex128).sum().real), seed=np.int64(64))
    manifest = {k: list(v.shape) for k, v in net64.state_dict().items()}

    # ---- G12: sigma_y curves ------------------------------------------------------------
    g12 = {}
    for nm in ("75m", "25s"):
        fn = os.path.join(REF, "data", f"flowdec_autoparams_{nm}.npy")
        g12[nm] = sigma_models.from_file(fn, factor=1, kernel_bandwidth=3).numpy()
        g12[nm + "_raw"] = np.load(fn)
    np.savez_compressed(os.path.join(HERE, "g12_sigma_y.npz"), **g12)

    # ---- G9: the real FlowModel.enhance (nf=8), injected noise ----------------------------
    from flowdec.model import FlowModel
    sig = torch.from_numpy(g12["75m"])
    fm = FlowModel(flow_matcher=None, sigma_x=0.0, sigma_y=sig, backbone=NCSNpp(nf=8, **bb_kw),
                   feature_extractor=fe, sampling_rate=48000, lr=1e-4, full_config={}).eval()
    fm.backbone.load_state_dict(to_t(strip(sd8, "backbone.")))
    for k, v in fm.state_dict().items():
        if not k.startswith("backbone."):
            manifest_key = k
            manifest[manifest_key] = list(v.shape)
    manifest = {("backbone." + k if not (k.startswith("feature_extractor") or k.startswith("sigma_")) else k): v
                for k, v in manifest.items()}
    L = 24000
    y9 = (0.1 * rng.standard_normal((2, 1, L))).astype(np.float32)
    y9[1] *= 3.0
    Tp = O.padded_frames(O.num_frames(L))
    noise = crandn(rng, (2, 1, 768, Tp))
    noise_t = torch.from_numpy(noise)
    fm._get_noise = lambda x, sigma: (sigma * noise_t[:x.shape[0]]).type(x.dtype)  # same arithmetic as model.py:536
    g9 = dict(y=y9, noise=noise, sigma_y=g12["75m"], seed=np.int64(8))
    for solver, N in (("euler", 6), ("midpoint", 3), ("heun2", 3), ("heun2_eulerlast", 3)):
        xh = fm.enhance(torch.from_numpy(y9), N=N, solver=solver)
        g9[f"{solver}_N{N}"] = xh.numpy()
    xh1 = fm.enhance(torch.from_numpy(y9[0, 0]), N=2, solver="euler")  # 1-D input path
    g9["euler_N2_1d"] = xh1.numpy()
    Xs, xs = fm.enhance(torch.from_numpy(y9), N=2, solver="euler", return_traj=True)
    g9["traj_feat_norms"] = np.array([float(X.abs().pow(2).sum().sqrt()) for X in Xs])
    g9["traj_last_wave"] = xs[-1].numpy()
    np.savez_compressed(os.path.join(HERE, "g9_enhance_nf8.npz"), **g9)

    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    for fn in sorted(os.listdir(HERE)):
        print(f"{fn:36s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:9.1f} KiB")


if __name__ == "__main__":
    main()
