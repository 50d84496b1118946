#!/usr/bin/env python
"""G17: the REAL reference `FlowModel.enhance` at full width (nf = 64, FlowDec-75m topology) on one 0.5 s clip, for the
headline solver settings (Euler N=6, midpoint N=3 = NFE 6).  Runs only in the build container (imports /root/reference
through the stub recipe of make_golden.py); only the arrays travel.  Weights are re-derived in the tests from
`oracle.flowdec_oracle.random_state_dict(seed=64, nf=64)`.

    python tests/golden/make_golden_nf64_enhance.py          # ~1-2 min of CPU, writes tests/golden/g17_enhance_nf64.npz
    python tests/golden/make_golden_nf64_enhance.py --cfg1   # G18: BASELINE config 1 EXACTLY -- one 1 s clip, 6-step Euler, fp32 -- ~2 min,
                                                             # writes tests/golden/g18_enhance_nf64_cfg1.npz
    python tests/golden/make_golden_nf64_enhance.py --cfg4clip   # G24: cfg 4's solver at its image size (75m, 2 s, midpoint N = 3)
    python tests/golden/make_golden_nf64_enhance.py --cfg5clip   # G25: cfg 5's clip length and step count (75m, 4 s = T_pad 512, 32-step Euler, fp32; ~20 min)
    python tests/golden/make_golden_nf64_enhance.py --cfg3clip   # G23: like G21 for BASELINE config 3: FlowDec-25s (sigma_y curve of
                                                             # data/flowdec_autoparams_25s.npy), one 2 s clip, midpoint N = 3 (NFE 6)
    python tests/golden/make_golden_nf64_enhance.py --cfg2clip   # G21: ONE clip of BASELINE config 2's shape -- 2 s, T_pad = 256 frames, 6-step
                                                             # Euler, fp32 -- ~4 min; the image size bench.py times (the kernel schedule of
                                                             # a 768 x 256 image differs from every shorter golden).  The noise is
                                                             # re-derived in the test from the stored seed (np.random.default_rng: the same
                                                             # draw order as here), so the file holds y + the waveform only
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from oracle import flowdec_oracle as O  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    MG._install_stubs()
    from flowdec.backbones.ncsnpp import NCSNpp
    from flowdec.data.feature_extractors import AmplitudeCompressedComplexSTFT
    from flowdec.data import sigma_models
    from flowdec.model import FlowModel

    bb_kw = dict(nonlinearity="swish", ch_mult=(4, 4, 4, 2), num_res_blocks=1, attn_resolutions=[], resamp_with_conv=True,
                 conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan",
                 progressive="output_skip", progressive_input="input_skip", progressive_combine="sum", init_scale=0.0,
                 fourier_scale=16, image_size=768, embedding_type="fourier", dropout=0.0, num_channels=4,
                 output_layer_kwargs=dict(kernel_size=1, bias=False, padding="same", padding_mode="zeros"),
                 bottleneck_attn=False)
    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000, alpha=0.3, beta=0.33)
    sig = sigma_models.from_file(os.path.join(MG.REF, "data", "flowdec_autoparams_25s.npy" if "--cfg3clip" in sys.argv else "flowdec_autoparams_75m.npy"),
                                 factor=1, kernel_bandwidth=3)
    fm = FlowModel(flow_matcher=None, sigma_x=0.0, sigma_y=sig, backbone=NCSNpp(nf=64, **bb_kw), feature_extractor=fe,
                   sampling_rate=48000, lr=1e-4, full_config={}).eval()
    sd = O.random_state_dict(seed=64, nf=64)
    fm.backbone.load_state_dict(MG.to_t(MG.strip(sd, "backbone.")))
    cfg1 = "--cfg1" in sys.argv
    cfg3 = "--cfg3clip" in sys.argv     # G23: one clip of BASELINE config 3's shape: FlowDec-25s (per-frequency sigma_y curve), 2 s, midpoint N = 3
    cfg4 = "--cfg4clip" in sys.argv     # G24: one clip of BASELINE config 4's shape: FlowDec-75m, 2 s, midpoint N = 3
    cfg5 = "--cfg5clip" in sys.argv     # G25: one clip of BASELINE config 5's shape: FlowDec-75m, 4 s (T_pad = 512), 32-step Euler (fixed-step reading), fp32
    cfg2 = "--cfg2clip" in sys.argv or cfg3 or cfg4 or cfg5
    rng_seed = 2505 if cfg5 else 2404 if cfg4 else 2303 if cfg3 else 2101 if cfg2 else 1801 if cfg1 else 1764
    rng = np.random.default_rng(rng_seed)
    L = 192000 if cfg5 else 96000 if cfg2 else 48000 if cfg1 else 24000
    y = (0.1 * rng.standard_normal((1, 1, L))).astype(np.float32)
    Tp = O.padded_frames(O.num_frames(L))
    noise = MG.crandn(rng, (1, 1, 768, Tp))
    noise_t = torch.from_numpy(noise)
    fm._get_noise = lambda x, sigma: (sigma * noise_t[:x.shape[0]]).type(x.dtype)  # same arithmetic as model.py:536
    g = dict(y=y, noise=noise, sigma_y=sig.numpy(), seed=np.int64(64))
    if cfg2:   # (1.5 MB of noise stay out of the file; a checksum pins the re-derivation)
        del g["noise"]
        g.update(rng_seed=np.int64(rng_seed), noise_sum=np.complex128(noise.astype(np.complex128).sum()),
                 noise_abs2=np.float64((np.abs(noise.astype(np.complex128)) ** 2).sum()))
    for solver, N in ((("euler", 32),) if cfg5 else (("midpoint", 3),) if (cfg3 or cfg4) else (("euler", 6),) if (cfg1 or cfg2) else (("euler", 6), ("midpoint", 3))):
        t0 = time.time()
        xh = fm.enhance(torch.from_numpy(y), N=N, solver=solver)
        print(f"{solver} N={N}: {time.time() - t0:.1f} s on {torch.get_num_threads()} threads", flush=True)
        g[f"{solver}_N{N}"] = xh.numpy()
    name = "g25_enhance_nf64_cfg5clip.npz" if cfg5 else "g24_enhance_nf64_cfg4clip.npz" if cfg4 else "g23_enhance_nf64_cfg3clip.npz" if cfg3 else "g21_enhance_nf64_cfg2clip.npz" if cfg2 else "g18_enhance_nf64_cfg1.npz" if cfg1 else "g17_enhance_nf64.npz"
    np.savez_compressed(os.path.join(HERE, name), **g)
    print(name, os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB")


if __name__ == "__main__":
    main()
