#!/usr/bin/env python
"""Throughput of the NDAC codec kernels (csrc/ndac.hip) at DAC's published full-width architecture with the ndac-75 frame rate
(encoder_dim 64, rates 2-4-8-10 = hop 640 at 48 kHz, decoder_dim 1536, 10 x 1024 x 8 codebooks; random weights):
encode (encoder + RVQ), from_codes, decode (matrix-core default and the exact vector path) for B clips of S seconds; audio-seconds per second and algorithmic TFLOP/s (f32 FMA).
    python scripts/ndac_bench.py [--batch 8] [--seconds 2] [--iters 5]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowdec_amd.ndac import DAC  # noqa: E402


def conv_flops(m, L):
    """2 * MACs of the encoder / decoder for one clip of L samples (biases, Snake, RVQ not counted)."""
    enc, T, d = 0, L, m.encoder_dim
    enc += 2 * T * d * 1 * 7
    for s in m.encoder_rates:
        d *= 2
        enc += 3 * 2 * T * (d // 2) * (d // 2) * (7 + 1)
        T //= s
        enc += 2 * T * d * (d // 2) * 2 * s
    enc += 2 * T * m.latent_dim * d * 3
    dec, D = 0, m.decoder_dim
    dec += 2 * T * D * m.latent_dim * 7
    od = D
    for i, s in enumerate(m.decoder_rates):
        idim, od = D // 2 ** i, D // 2 ** (i + 1)
        dec += 2 * T * idim * od * 2 * s          # every input sample meets all 2s taps
        T *= s
        dec += 3 * 2 * T * od * od * (7 + 1)
    dec += 2 * T * od * 7
    return enc, dec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--nq", type=int, default=10)
    a = ap.parse_args()
    torch.manual_seed(0)
    m = DAC(encoder_dim=64, encoder_rates=(2, 4, 8, 10), decoder_dim=1536, decoder_rates=(10, 8, 4, 2), n_codebooks=10, codebook_size=1024, codebook_dim=8,
            sample_rate=48000)
    for k, p in m.named_parameters():          # keep activations O(1): g = 0.8 ||v||
        if k.endswith("weight_g"):
            v = dict(m.named_parameters())[k[:-1] + "v"]
            p.data = 0.8 * v.data.pow(2).sum(dim=tuple(range(1, v.ndim)), keepdim=True).sqrt()
    m = m.cuda()
    L = int(a.seconds * 48000)
    x = m.preprocess(0.3 * torch.randn(a.batch, 1, L, device="cuda"), 48000)
    z, codes, _, _, _ = m.encode(x, n_quantizers=a.nq)
    zq, _, _ = m.quantizer.from_codes(codes)
    y = m.decode(zq)
    assert torch.isfinite(y).all() and y.shape[-1] == x.shape[-1], (y.shape, x.shape)
    fe, fd = conv_flops(m, x.shape[-1])
    res = {"config": f"DAC 64/(2,4,8,10)/1536/(10,8,4,2), {a.nq} x 1024 x 8 codebooks, B = {a.batch} x {a.seconds:g} s @ 48 kHz, f32", "audio_seconds": a.batch * a.seconds}
    y_exact = None
    if m.precision != "exact":
        m.precision = "exact"; y_exact = m.decode(zq); m.precision = "mfma_decoder"
        res["decode_vs_exact_rel_to_peak"] = float((y - y_exact).abs().max() / y_exact.abs().max())

    def decode_exact():
        m.precision = "exact"
        try:
            return m.decode(zq)
        finally:
            m.precision = "mfma_decoder"

    def encode_mfma():
        m.precision = "mfma"
        try:
            return m.encode(x, n_quantizers=a.nq)
        finally:
            m.precision = "mfma_decoder"

    codes_m = encode_mfma()[1]
    res["encode_mfma_code_mismatch_fraction"] = float((codes_m != codes).float().mean())
    for name, fn, fl in (("encode", lambda: m.encode(x, n_quantizers=a.nq), fe), ("encode_mfma", encode_mfma, fe),
                         ("from_codes", lambda: m.quantizer.from_codes(codes), 0), ("decode", lambda: m.decode(zq), fd),
                         ("decode_exact", decode_exact, fd)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        res[name] = {"ms": 1e3 * dt, "audio_seconds_per_second": a.batch * a.seconds / dt, "algorithmic_tflop": a.batch * fl / 1e12,
                     "tflops": a.batch * fl / dt / 1e12}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
