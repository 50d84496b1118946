#!/usr/bin/env python
"""Golden values of the reference's SI-SDR / SI-SIR / SI-SAR (flowdec/eval/metrics.py SISXR) on seeded signals.
Same import recipe as make_golden.py (dummy modules for the evaluation-only dependencies).

    python tests/golden/make_golden_metrics.py     # writes tests/golden/g14_metrics.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG._install_stubs()
    from flowdec.eval.metrics import SISXR
    rng = np.random.default_rng(14)
    out = {}
    m = SISXR(48000)
    for i, (snr_in, art) in enumerate([(5.0, 0.05), (20.0, 0.3), (-3.0, 0.0)]):
        x = rng.standard_normal(4000).astype(np.float32) * 0.1
        n = rng.standard_normal(4000).astype(np.float32) * 0.1 * 10 ** (-snr_in / 20)
        y = x + n
        x_hat = (0.8 * x + 0.2 * n + art * 0.1 * rng.standard_normal(4000)).astype(np.float32)
        if i == 2:
            y = -y        # exercises the phase-flip guard
        out[f"x{i}"], out[f"y{i}"], out[f"xhat{i}"] = x, y, x_hat
        out[f"sisxr{i}"] = np.array(m(torch.from_numpy(x_hat), torch.from_numpy(x), torch.from_numpy(y)), dtype=np.float64)
        print(i, out[f"sisxr{i}"])
    np.savez_compressed(os.path.join(HERE, "g14_metrics.npz"), **out)


if __name__ == "__main__":
    main()
