#!/usr/bin/env python
"""Golden vectors for the ScoreDec / regression baselines, produced by running the
REFERENCE classes (flowdec.model.ScoreModel / RegressionModel, flowdec.sdes.OUVESDE, flowdec.sampling) on CPU.
Same import recipe and the same seeded nf=8 weights as make_golden.py.  The sampler's Gaussian draws
(torch.randn_like) are replaced by the seeded NumPy stream `oracle.seeded_noises(seed, shape)`, which the tests
regenerate, so only the inputs and the reference outputs are stored.

    python tests/golden/make_golden_score.py      # writes tests/golden/g13_score_nf8.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs + helpers)
from oracle import flowdec_oracle as O  # noqa: E402

CASES = {  # name -> enhance kwargs
    "rd_ald_N3": dict(N=3, predictor="reverse_diffusion", corrector="ald", corrector_steps=1, snr=0.5),
    "rd_none_N4": dict(N=4, predictor="reverse_diffusion", corrector="none"),
    "em_ald2_N2": dict(N=2, predictor="euler_maruyama", corrector="ald", corrector_steps=2, snr=0.33),
    "rd_ald_N3_nodenoise": dict(N=3, predictor="reverse_diffusion", corrector="ald", corrector_steps=1, snr=0.5, denoise=False),
}
NOISE_SEED = 1313
OUT_SCALE = 0.02   # random-init output layer scaled down so that the sampler stays in the codec's amplitude range


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    MG._install_stubs()
    from flowdec.backbones.ncsnpp import NCSNpp
    from flowdec.data.feature_extractors import AmplitudeCompressedComplexSTFT
    from flowdec.model import RegressionModel, ScoreModel
    from flowdec.sdes import OUVESDE
    bb_kw = dict(nonlinearity="swish", ch_mult=(4, 4, 4, 2), num_res_blocks=1, attn_resolutions=[], resamp_with_conv=True,
                 conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan",
                 progressive="output_skip", progressive_input="input_skip", progressive_combine="sum", init_scale=0.0,
                 fourier_scale=16, image_size=768, embedding_type="fourier", dropout=0.0, num_channels=4,
                 output_layer_kwargs=dict(kernel_size=1, bias=False, padding="same", padding_mode="zeros"),
                 bottleneck_attn=False)   # config/model/backbone/ncsnpp_final_no_attn.yaml
    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000, alpha=0.3, beta=0.33)
    sd8 = O.random_state_dict(seed=8, nf=8)
    sd8["backbone.output_layer.weight"] = sd8["backbone.output_layer.weight"] * np.float32(OUT_SCALE)
    common = dict(backbone=NCSNpp(nf=8, **bb_kw), feature_extractor=fe, sampling_rate=48000, lr=1e-4, full_config={})
    sm = ScoreModel(sde=OUVESDE(theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30), t_eps=3e-2, **common).eval()
    sm.backbone.load_state_dict(MG.to_t(MG.strip(sd8, "backbone.")))
    rng = np.random.default_rng(13)
    L = 12000
    y = (0.1 * rng.standard_normal((2, 1, L))).astype(np.float32)
    y[1] *= 2.5
    Tp = O.padded_frames(O.num_frames(L))
    out = dict(y=y, noise_seed=np.int64(NOISE_SEED), weight_seed=np.int64(8), out_scale=np.float64(OUT_SCALE), sde=np.array([1.5, 0.05, 0.5]), t_eps=np.float64(3e-2))
    real_randn_like = torch.randn_like
    for name, kw in CASES.items():
        stream = O.seeded_noises(NOISE_SEED, (2, 1, 768, Tp))
        drawn = [0]

        def fake(x, *a, **k):
            assert tuple(x.shape) == (2, 1, 768, Tp) and x.dtype == torch.complex64
            drawn[0] += 1
            return torch.from_numpy(next(stream))
        torch.randn_like = fake
        try:
            out[name] = sm.enhance(torch.from_numpy(y), **kw).numpy()
        finally:
            torch.randn_like = real_randn_like
        pk = {k: kw[k] for k in ("predictor", "corrector", "corrector_steps") if k in kw}
        assert drawn[0] == O.score_noise_count(kw["N"], **pk), (name, drawn[0])
        print(name, "draws", drawn[0], "rms", float(np.sqrt(np.mean(out[name] ** 2))))
    # sde closed forms at a few t (sdes.py:168-192) for the scalar-coefficient check
    ts = torch.tensor([1.0, 0.5, 0.03, 0.2575], dtype=torch.float32)
    out["std_t"] = ts.numpy(); out["std"] = sm.sde._std(ts).numpy()
    out["diffusion"] = sm.sde.sde(torch.zeros(4, 1, 1, 1), ts, torch.zeros(4, 1, 1, 1))[1].numpy()
    out["timesteps_N30"] = torch.linspace(1, 3e-2, 30).numpy()
    # black-box ODE sampler (sampling/__init__.py:75-146; scipy RK45 on the host), one clip, loose tolerance to keep it short
    stream = O.seeded_noises(NOISE_SEED, (1, 1, 768, Tp))
    torch.randn_like = lambda x, *a, **k: torch.from_numpy(next(stream))
    try:
        sm.eval()
        import flowdec.sampling as S
        nfe_box = {}
        orig = S.get_ode_sampler

        def wrapped(*a, **k):
            sampler = orig(*a, **k)

            def run(*aa, **kk):
                x, nfe = sampler(*aa, **kk)
                nfe_box["nfe"] = nfe
                return x, nfe
            return run
        S.get_ode_sampler = wrapped
        out["ode_rk45"] = sm.enhance(torch.from_numpy(y[:1]), sampler_type="ode", N=30, device="cpu", rtol=1e-3, atol=1e-3).numpy()
        out["ode_rk45_nfe"] = np.int64(nfe_box["nfe"])
        print("ode_rk45 nfe", nfe_box["nfe"], "rms", float(np.sqrt(np.mean(out["ode_rk45"] ** 2))))
    finally:
        torch.randn_like = real_randn_like
        S.get_ode_sampler = orig
    rm = RegressionModel(loss_type="l2", **{**common, "backbone": sm.backbone}).eval()
    out["regression"] = rm.enhance(torch.from_numpy(y)).numpy()
    np.savez_compressed(os.path.join(HERE, "g13_score_nf8.npz"), **out)
    print("g13_score_nf8.npz", os.path.getsize(os.path.join(HERE, "g13_score_nf8.npz")) / 1024, "KiB")


if __name__ == "__main__":
    main()
