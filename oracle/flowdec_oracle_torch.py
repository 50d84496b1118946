"""flowdec_oracle_torch.py -- the oracle's arithmetic on PyTorch CPU kernels (oneDNN / MKL / pocketfft).

TEST INFRASTRUCTURE, like flowdec_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import it.  It exists because the same-box CPU baseline should run at the speed the REFERENCE runs at on a CPU: the
reference is PyTorch (`nn.Conv2d`, `nn.GroupNorm`, `nn.SiLU`, `torch.stft`, upfirdn2d_native; flowdec/backbones/ncsnpp.py,
ncsnpp_utils/layerspp.py, data/feature_extractors.py), so the convolutions / normalisations / FIR resampling / STFT here are
`torch.nn.functional` calls on the host -- the same library kernels the reference's CPU path ends up in -- driven by the
oracle's own restatement of the graph and of the solver (the reference's Python cannot travel to the GPU box).  The NumPy
oracle (im2col-free GEMM per tap through OpenBLAS) is 3x slower per core than that and stays the PARITY checker.

Pinned by tests/test_oracle_golden.py::test_torch_oracle_matches_numpy_oracle_and_golden: forward vs golden G8 (reference
NCSNpp, nf = 8), enhance vs golden G9 (reference FlowModel.enhance), both at the fp32 tolerances.
"""
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import flowdec_oracle as O


def _fir_kernel_1d(dtype=torch.float32) -> torch.Tensor:
    return torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=dtype) / 8.0


def downsample_2d(x: torch.Tensor) -> torch.Tensor:
    """up_or_down_sampling.py:252-282 with k = [1,3,3,1], factor 2 -> upfirdn2d(x, k/64, down=2, pad=(1,1)): a depthwise
    4x4 correlation at stride 2 over the input zero-padded by 1 (the kernel is symmetric, so flipping is a no-op)."""
    C = x.shape[1]
    k1 = _fir_kernel_1d(x.dtype)
    k = torch.outer(k1, k1)[None, None].expand(C, 1, 4, 4).contiguous()
    return F.conv2d(x, k, stride=2, padding=1, groups=C)


def upsample_2d(x: torch.Tensor) -> torch.Tensor:
    """up_or_down_sampling.py:220-249: zero-insertion by 2, pad (2, 1), correlate with k * 4 / 64 -- as a depthwise
    transposed convolution (stride 2, 4x4 kernel, padding 1), which is the same sum."""
    C = x.shape[1]
    k1 = _fir_kernel_1d(x.dtype) * 2.0
    k = torch.outer(k1, k1)[None, None].expand(C, 1, 4, 4).contiguous()
    return F.conv_transpose2d(x, k, stride=2, padding=1, groups=C)


class NCSNppTorchCPU:
    """NCSNpp.forward (ncsnpp.py:254-399) with ResnetBlockBigGANpp (layerspp.py:252-284), same module walk as
    flowdec_oracle.NCSNppOracle, tensors are torch CPU float32."""

    def __init__(self, state_dict: Dict[str, np.ndarray], nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1, num_channels=4, prefix="backbone."):
        self.cfg = dict(nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks, num_channels=num_channels)
        self.specs = O.build_module_specs(**self.cfg)
        self.p = {(k[len(prefix):] if k.startswith(prefix) else k): torch.from_numpy(np.asarray(v, np.float32).copy()) for k, v in state_dict.items()}

    def _w(self, i, name):
        return self.p[f"all_modules.{i}.{name}"]

    def time_embedding(self, t: torch.Tensor) -> torch.Tensor:
        x_proj = t.reshape(-1, 1) * self._w(0, "W")[None, :] * 2 * np.pi
        emb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
        h = F.linear(emb, self._w(1, "weight"), self._w(1, "bias"))
        return F.linear(F.silu(h), self._w(2, "weight"), self._w(2, "bias"))

    def _gn(self, x, i, name):
        C = x.shape[1]
        return F.group_norm(x, O.gn_groups(C), self._w(i, name + ".weight"), self._w(i, name + ".bias"), eps=1e-6)

    def resblock(self, i, s, x, temb):
        ci, co = s["cin"], s["cout"]
        h = F.silu(self._gn(x, i, "GroupNorm_0"))
        if s["up"]:
            h, x = upsample_2d(h), upsample_2d(x)
        elif s["down"]:
            h, x = downsample_2d(h), downsample_2d(x)
        h = F.conv2d(h, self._w(i, "Conv_0.weight"), self._w(i, "Conv_0.bias"), padding=1)
        h = h + F.linear(F.silu(temb), self._w(i, "Dense_0.weight"), self._w(i, "Dense_0.bias"))[:, :, None, None]
        h = F.silu(self._gn(h, i, "GroupNorm_1"))
        h = F.conv2d(h, self._w(i, "Conv_1.weight"), self._w(i, "Conv_1.bias"), padding=1)
        if ci != co or s["up"] or s["down"]:
            x = F.conv2d(x, self._w(i, "Conv_2.weight"), self._w(i, "Conv_2.bias"))
        return (x + h) / np.sqrt(2.0)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, y: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """x, y complex64 [B, 1, F, T]; t [1] or [B] -> complex64 [B, 1, F, T]."""
        h = torch.cat([x.real, x.imag, y.real, y.imag], dim=1).float()
        temb = self.time_embedding(t.float())
        specs, R, nrb = self.specs, len(self.cfg["ch_mult"]), self.cfg["num_res_blocks"]
        m = 3
        pyr_in = h
        hs = [F.conv2d(h, self._w(m, "weight"), self._w(m, "bias"), padding=1)]
        m += 1
        for lvl in range(R):
            for _ in range(nrb):
                h = self.resblock(m, specs[m], hs[-1], temb); m += 1
                hs.append(h)
            if lvl != R - 1:
                h = self.resblock(m, specs[m], hs[-1], temb); m += 1
                pyr_in = downsample_2d(pyr_in)
                h = F.conv2d(pyr_in, self._w(m, "Conv_0.weight"), self._w(m, "Conv_0.bias")) + h
                m += 1
                hs.append(h)
        h = hs[-1]
        h = self.resblock(m, specs[m], h, temb); m += 1
        h = self.resblock(m, specs[m], h, temb); m += 1
        pyramid = None
        for lvl in reversed(range(R)):
            for _ in range(nrb + 1):
                h = self.resblock(m, specs[m], torch.cat([h, hs.pop()], dim=1), temb); m += 1
            ph = F.silu(F.group_norm(h, O.gn_groups(h.shape[1]), self._w(m, "weight"), self._w(m, "bias"), eps=1e-6)); m += 1
            ph = F.conv2d(ph, self._w(m, "weight"), self._w(m, "bias"), padding=1); m += 1
            pyramid = ph if pyramid is None else upsample_2d(pyramid) + ph
            if lvl != 0:
                h = self.resblock(m, specs[m], h, temb); m += 1
        assert not hs and m == len(specs)
        out = F.conv2d(pyramid, self.p["output_layer.weight"])
        return torch.complex(out[:, 0:1], out[:, 1:2])


def _window():
    return torch.from_numpy(O.hann_sym(O.N_FFT, np.float64).astype(np.float32))


@torch.no_grad()
def enhance(net: NCSNppTorchCPU, y: np.ndarray, noise: np.ndarray, sigma_y, N: int = 6, solver: str = "euler", sigma_fac: float = 1.0) -> np.ndarray:
    """FlowModel.enhance (model.py:476-528) end to end on the host: normalize_noisy -> torch.stft (n_fft 1534, hop 384, symmetric
    Hann, center / reflect) -> compression -> pad_spec -> fixed-step solver (oracle's `odeint_fixed` stepping, restated in
    torch) -> decompression -> torch.istft(length) -> * normfac.  y [B, 1, L] f32, noise [B, 1, 768, T_pad] c64."""
    yt = torch.from_numpy(np.asarray(y, np.float32))
    B, _, Lw = yt.shape
    normfac = yt.abs().amax(dim=(1, 2), keepdim=True)
    normfac = torch.where(torch.isclose(normfac, torch.zeros_like(normfac)), torch.ones_like(normfac), normfac)
    w = _window()
    X = torch.stft((yt / normfac).reshape(B, Lw), n_fft=O.N_FFT, hop_length=O.HOP, window=w, center=True, onesided=True, return_complex=True)
    X = (X.abs() ** O.ALPHA * torch.exp(1j * X.angle()) * O.BETA)[:, None]
    T = X.shape[-1]
    Tp = O.padded_frames(T)
    Y = F.pad(X, (0, Tp - T))
    sig = torch.as_tensor(np.asarray(sigma_y, np.float64))
    x = Y + sigma_fac * (sig * torch.from_numpy(noise).to(torch.complex128)).to(torch.complex64)
    ts = torch.from_numpy(O.t_span_linspace(N))
    t = ts[0].clone()
    dt = ts[1] - ts[0]
    f = lambda tt, xx: net.forward(xx, Y, tt.reshape(1))
    for i in range(1, N + 1):
        if solver == "euler":
            x = x + dt * f(t, x)
        elif solver == "midpoint":
            half = 0.5 * dt
            x = x + dt * f(t + half, x + half * f(t, x))
        elif solver == "heun2":
            k1 = f(t, x)
            x = x + 0.5 * dt * (k1 + f(t + dt, x + dt * k1))
        else:
            raise ValueError(solver)
        t = t + dt
        if i < N:
            dt = ts[i + 1] - t
    Xd = x[..., :T] / O.BETA
    Xd = Xd.abs() ** (1.0 / O.ALPHA) * torch.exp(1j * Xd.angle())
    out = torch.istft(Xd[:, 0], n_fft=O.N_FFT, hop_length=O.HOP, window=w, center=True, onesided=True, length=Lw)
    return (out[:, None] * normfac).numpy().astype(np.float32)
