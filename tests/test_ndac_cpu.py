"""NDAC codec, CPU side: the oracle (oracle/ndac_oracle.py, a restatement of descript-audio-codec 1.0.0 -- PARITY UNPINNED: the
package is third party and absent, see the oracle's header) is checked against what CAN be checked offline:

  * its primitives against PyTorch's own nn.functional.conv1d / conv_transpose1d / weight_norm / F.normalize-based search,
  * the whole encoder / decoder against the SAME module tree executed by PyTorch (real nn.Conv1d / nn.ConvTranspose1d modules
    with weight norm, assembled as dac/model/dac.py assembles them) -- an independent execution path, like the package's own,
  * the product's state_dict layout (flowdec_amd.ndac.DAC) against the oracle's checkpoint-form manifest, and DAC.load on a
    container written the way audiotools' BaseModel.save writes it.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import ndac_oracle as N

SMALL = dict(encoder_dim=8, encoder_rates=(2, 4, 5), decoder_dim=64, decoder_rates=(5, 4, 2), n_codebooks=4, codebook_size=32, codebook_dim=8, sample_rate=48000)
# the shape of ndac-75 as far as demo.ipynb fixes it: 48 kHz, 75 Hz frame rate -> hop 640, up to 10 codebooks of 1024 x 8; widths reduced
NDAC75_LIKE = dict(encoder_dim=8, encoder_rates=(2, 4, 8, 10), decoder_dim=96, decoder_rates=(10, 8, 4, 2), n_codebooks=10, codebook_size=1024, codebook_dim=8,
                   sample_rate=48000)


def scaled_sd(cfg, seed, gain):
    sd = N.random_checkpoint_state_dict(seed=seed, **cfg)
    return {k: (v * np.float32(gain)).astype(np.float32) if k.endswith("weight_g") else v for k, v in sd.items()}


def test_oracle_primitives_against_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 50)).astype(np.float32)
    xt = torch.from_numpy(x)
    for K, kw in ((7, dict(stride=1, padding=3, dilation=1)), (7, dict(stride=1, padding=9, dilation=3)), (7, dict(stride=1, padding=27, dilation=9)),
                  (8, dict(stride=4, padding=2, dilation=1)), (10, dict(stride=5, padding=3, dilation=1)), (1, dict(stride=1, padding=0, dilation=1))):
        w = rng.standard_normal((5, 6, K)).astype(np.float32); b = rng.standard_normal(5).astype(np.float32)
        assert rel_err(N.conv1d(x, w, b, **kw), F.conv1d(xt, torch.from_numpy(w), torch.from_numpy(b), **kw).numpy()) < 2e-6, (K, kw)
    for s in (2, 4, 5, 8, 10):
        w = rng.standard_normal((6, 3, 2 * s)).astype(np.float32); b = rng.standard_normal(3).astype(np.float32)
        ref = F.conv_transpose1d(xt, torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=math.ceil(s / 2)).numpy()
        got = N.conv_transpose1d(x, w, b, stride=s, padding=math.ceil(s / 2))
        assert got.shape == ref.shape and rel_err(got, ref) < 2e-6, s
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 5, 7)); ct = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 3, 8, stride=4, padding=2))
    for m in (conv, ct):   # dim = 0: per output channel for Conv1d, per INPUT channel for ConvTranspose1d
        assert np.abs(N.weight_norm_effective(m.weight_g.detach().numpy(), m.weight_v.detach().numpy()) - m.weight.detach().numpy()).max() < 1e-7
    a = (1 + 0.3 * rng.standard_normal(6)).astype(np.float32)
    ref = xt + (torch.from_numpy(a).reshape(1, -1, 1) + 1e-9).reciprocal() * torch.sin(torch.from_numpy(a).reshape(1, -1, 1) * xt).pow(2)
    assert rel_err(N.snake(x, a), ref.numpy()) < 1e-6
    # nearest-neighbour search: the published formula evaluated by torch (decode_latents) gives the same indices on generic data
    e = rng.standard_normal((2000, 8)).astype(np.float32); cb = rng.standard_normal((1024, 8)).astype(np.float32)
    en, cn = F.normalize(torch.from_numpy(e)), F.normalize(torch.from_numpy(cb))
    dist = en.pow(2).sum(1, keepdim=True) - 2 * en @ cn.t() + cn.pow(2).sum(1, keepdim=True).t()
    agree = ((-dist).max(1)[1].numpy() == N.vq_nearest(e, cb)).mean()
    assert agree > 0.999, agree      # (a GEMM's summation order may flip a near-tie; the oracle's own order is DEFINED, see vq_nearest)
    cb2 = cb.copy(); cb2[700] = cb2[13]; cb2[20] = cb2[13]       # exact ties -> the lowest index
    idx = N.vq_nearest(cb[13:14] * 3.0, cb2)
    assert idx[0] == 13


def _torch_forward(mod, x):
    """Executes the product's container modules with PyTorch itself (Snake1d by its formula): what dac's own forward does."""
    from flowdec_amd.ndac import Snake1d
    if isinstance(mod, Snake1d):
        return x + (mod.alpha + 1e-9).reciprocal() * torch.sin(mod.alpha * x).pow(2)
    if isinstance(mod, (torch.nn.Conv1d, torch.nn.ConvTranspose1d, torch.nn.Tanh)):
        return mod(x)
    if isinstance(mod, torch.nn.Sequential):
        for m in mod:
            x = _torch_forward(m, x)
        return x
    inner = mod.block
    is_res_unit = len(inner) == 4 and isinstance(inner[1], torch.nn.Conv1d) and inner[1].kernel_size == (7,) and isinstance(inner[3], torch.nn.Conv1d)
    y = _torch_forward(inner, x)
    return x + y if is_res_unit else y


@pytest.mark.parametrize("cfg,gain", [(SMALL, 0.7), (NDAC75_LIKE, 0.75)], ids=["small", "ndac75_like"])
def test_oracle_stack_against_torch_modules(cfg, gain):
    from flowdec_amd.ndac import DAC
    sd = scaled_sd(cfg, 3, gain)
    m = DAC(**cfg)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing and not unexpected
    o = N.DACOracle(sd, **cfg)
    assert o.hop_length == m.hop_length == int(np.prod(cfg["encoder_rates"]))
    rng = np.random.default_rng(1)
    x = (0.3 * rng.standard_normal((2, 1, 3 * o.hop_length + 17))).astype(np.float32)
    xp = o.preprocess(x)
    assert xp.shape[-1] % o.hop_length == 0 and np.array_equal(m.preprocess(torch.from_numpy(x)).numpy(), xp)
    with torch.no_grad():
        z_t = _torch_forward(m.encoder.block, torch.from_numpy(xp)).numpy()
    z_o = o.encoder(xp)
    assert z_o.shape == (2, m.latent_dim, xp.shape[-1] // o.hop_length) and rel_err(z_o, z_t) < 2e-5
    zq, codes, lat = o.quantize(z_o, n_quantizers=min(4, cfg["n_codebooks"]))
    assert codes.shape == (2, min(4, cfg["n_codebooks"]), z_o.shape[-1]) and codes.min() >= 0 and codes.max() < cfg["codebook_size"]
    zq2, zp, _ = o.from_codes(codes)
    assert rel_err(zq2, zq) < 1e-5 and zp.shape == lat.shape         # forward's straight-through value == the table lookup up to rounding
    with torch.no_grad():
        y_t = _torch_forward(m.decoder.model, torch.from_numpy(zq)).numpy()
    y_o = o.decode(zq)
    assert y_o.shape == y_t.shape and rel_err(y_o, y_t) < 2e-5 and 0.05 < float(np.abs(y_o).mean()) < 0.9     # not saturated, not dead


def test_state_dict_layout_and_load(tmp_path):
    """The product module's state_dict is DAC's: weight_g / weight_v / bias per conv, alpha per Snake1d, codebook.weight per
    quantiser, in the oracle's manifest order; DAC.load reads the {'state_dict', 'metadata': {'kwargs'}} container; the parametrised
    weight-norm spelling of newer torch versions is accepted."""
    from flowdec_amd.ndac import DAC
    m = DAC(**SMALL)
    want = []
    for name, shape, kind in N.param_manifest(**SMALL):
        if kind in ("conv", "convT"):
            base = name[: -len(".weight")]
            want += [(base + ".bias", None)] if False else []
            want.append((base + ".weight_g", (shape[0], 1, 1))); want.append((base + ".weight_v", shape))
        else:
            want.append((name, shape))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(got) == {k for k, _ in want}, sorted(set(got) ^ {k for k, _ in want})[:6]
    assert all(got[k] == tuple(s) for k, s in want)
    sd = scaled_sd(SMALL, 5, 0.7)
    path = tmp_path / "weights.pth"
    torch.save({"state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}, "metadata": {"kwargs": dict(SMALL, encoder_rates=list(SMALL["encoder_rates"]),
                                                                                                          decoder_rates=list(SMALL["decoder_rates"]))}}, path)
    m2 = DAC.load(str(path))
    assert m2.hop_length == 40 and m2.sample_rate == 48000 and m2.n_codebooks == 4
    assert all(torch.equal(m2.state_dict()[k], torch.from_numpy(v)) for k, v in sd.items())
    new_style = {k.replace(".weight_g", ".parametrizations.weight.original0").replace(".weight_v", ".parametrizations.weight.original1"): torch.from_numpy(v)
                 for k, v in sd.items()}
    m3 = DAC(**SMALL); m3.load_state_dict(new_style)
    assert all(torch.equal(m3.state_dict()[k], m2.state_dict()[k]) for k in m2.state_dict())
    eff = m2._effective()
    ref = N.effective_state_dict(sd)
    assert set(eff) == set(ref) and all(np.abs(eff[k].numpy() - ref[k]).max() < 1e-6 for k in ref)
    with pytest.raises(RuntimeError):      # no CPU compute path
        m2.encode(torch.zeros(1, 1, 80))
