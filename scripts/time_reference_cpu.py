#!/usr/bin/env python
"""Times the REFERENCE implementation's own PyTorch CPU path on BASELINE config 1 (FlowDec-75m, one 1 s 48 kHz clip, 6-step
Euler, fp32) in the build container and writes profiles/r02_reference_cpu_timing.json.  The reference is imported from
/root/reference through the stub recipe of tests/golden/make_golden.py; it does not exist on the GPU box,
so this number is measured HERE and quoted by bench.py beside the same-box timing of the NumPy port.

    python scripts/time_reference_cpu.py [--repeats 2]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as MG  # noqa: E402
from oracle import flowdec_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=2)
    a = ap.parse_args()
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
                 output_layer_kwargs=dict(kernel_size=1, bias=False, padding="same", padding_mode="zeros"), bottleneck_attn=False)
    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000, alpha=0.3, beta=0.33)
    sig = sigma_models.from_file(os.path.join(MG.REF, "data", "flowdec_autoparams_75m.npy"), factor=1, kernel_bandwidth=3)
    fm = FlowModel(flow_matcher=None, sigma_x=0.0, sigma_y=sig, backbone=NCSNpp(nf=64, **bb_kw), feature_extractor=fe,
                   sampling_rate=48000, lr=1e-4, full_config={}).eval()
    fm.backbone.load_state_dict(MG.to_t(MG.strip(O.random_state_dict(seed=64, nf=64), "backbone.")))
    rng = np.random.default_rng(0)
    y = torch.from_numpy((0.1 * rng.standard_normal((1, 1, 48000))).astype(np.float32))
    times = []
    for r in range(a.repeats + 1):          # first run = warm-up (oneDNN primitive creation)
        t0 = time.perf_counter()
        x = fm.enhance(y, N=6, solver="euler")
        times.append(time.perf_counter() - t0)
        print(f"run {r}: {times[-1]:.2f} s", flush=True)
    assert x.shape == (1, 1, 48000) and torch.isfinite(x).all()
    best = min(times[1:])
    res = {"what": "reference FlowModel.enhance (flowdec/model.py:476-528) on CPU, BASELINE config 1: FlowDec-75m full width, one 1 s 48 kHz clip, "
                   "6-step Euler, fp32, random-init weights",
           "seconds_per_enhance": best, "audio_seconds_per_second": 1.0 / best, "seconds_per_nfe": best / 6, "all_runs_s": times,
           "threads": torch.get_num_threads(), "cpu": platform.processor() or platform.machine(), "torch": torch.__version__,
           "where": "build container (no GPU); the reference cannot travel to the GPU box", "torchdyn": "fixed-step driver restated (not installed)"}
    out = os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
