"""flowdec_amd.ndac -- the NDAC codec in front of FlowDec, on libflowdec_hip.so (csrc/ndac.hip).

Mirrors the surface of `dac.DAC` (descript-audio-codec 1.0.0) that /root/reference/demo.ipynb uses (cells 2-3):

    dac_model = DAC.load(os.path.join(CKPT_DIR, f'ndac/{ndac_model}/800k/dac/weights.pth')); dac_model.to('cuda'); dac_model.eval()
    x = dac_model.preprocess(signal.audio_data, signal.sample_rate)
    z, codes, latents, _, _ = dac_model.encode(x, n_quantizers=nq)
    zq, _, _ = dac_model.quantizer.from_codes(codes)
    xhat_ndac = dac_model.decode(zq)

The module tree (`encoder.block.*`, `quantizer.quantizers.*`, `decoder.model.*`) is built from real torch modules so that the
`state_dict` keys / shapes are the package's own (`...weight_g`, `...weight_v`, `...alpha`, `...codebook.weight`) and
`load_state_dict` of a DAC checkpoint works unchanged; the modules are parameter containers only -- every forward runs in the
HIP library (there is no PyTorch / CPU compute path: a codec on the CPU raises).

The arithmetic is third party and absent from /root/reference: restated from the published algorithm, PARITY UNPINNED
(oracle/ndac_oracle.py lists the upstream definitions).
"""
import ctypes as C
import math
import threading
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _wn(module):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the deprecated spelling is the one whose keys (weight_g / weight_v) DAC 1.0.0 checkpoints carry
        return torch.nn.utils.weight_norm(module)


class Snake1d(nn.Module):
    """dac.nn.layers.Snake1d: parameter `alpha` [1, C, 1] (container; applied inside the consuming convolution kernel)."""

    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))


def _res_unit(dim, dilation):
    m = nn.Module()
    m.block = nn.Sequential(Snake1d(dim), _wn(nn.Conv1d(dim, dim, 7, dilation=dilation, padding=3 * dilation)), Snake1d(dim), _wn(nn.Conv1d(dim, dim, 1)))
    return m


def _enc_block(dim, stride):
    m = nn.Module()
    m.block = nn.Sequential(_res_unit(dim // 2, 1), _res_unit(dim // 2, 3), _res_unit(dim // 2, 9), Snake1d(dim // 2),
                            _wn(nn.Conv1d(dim // 2, dim, 2 * stride, stride=stride, padding=math.ceil(stride / 2))))
    return m


def _dec_block(idim, odim, stride):
    m = nn.Module()
    m.block = nn.Sequential(Snake1d(idim), _wn(nn.ConvTranspose1d(idim, odim, 2 * stride, stride=stride, padding=math.ceil(stride / 2))),
                            _res_unit(odim, 1), _res_unit(odim, 3), _res_unit(odim, 9))
    return m


class VectorQuantize(nn.Module):
    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        self.in_proj = _wn(nn.Conv1d(input_dim, codebook_dim, 1))
        self.out_proj = _wn(nn.Conv1d(codebook_dim, input_dim, 1))
        self.codebook = nn.Embedding(codebook_size, codebook_dim)


class ResidualVectorQuantize(nn.Module):
    """dac.nn.quantize.ResidualVectorQuantize: `forward(z, n_quantizers)` and `from_codes(codes)` (eval semantics)."""

    def __init__(self, owner, input_dim, n_codebooks, codebook_size, codebook_dim):
        super().__init__()
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim) for _ in range(n_codebooks)])
        self._owner = [owner]   # (list: not a submodule)

    def forward(self, z, n_quantizers: Optional[int] = None):
        """-> (z_q, codes [B, nq, T] int64, latents [B, nq * codebook_dim, T], commitment_loss, codebook_loss); the two losses are
        training quantities (None here)."""
        dac = self._owner[0]
        L.require_cuda(z)
        z = z.float().contiguous()
        B, D, T = z.shape
        nq = self.n_codebooks if n_quantizers is None else max(1, min(int(n_quantizers), self.n_codebooks))
        with dac._lock, torch.cuda.device(z.device):
            h = dac.handle()
            zq = torch.empty_like(z)
            codes = torch.empty(B, nq, T, dtype=torch.int32, device=z.device)
            lat = torch.empty(B, nq * self.codebook_dim, T, dtype=torch.float32, device=z.device)
            ws = torch.empty(z.numel() * 4, dtype=torch.uint8, device=z.device)
            L.check(L.load().fd_rvq_encode(h, L.ptr(z), B, T, nq, L.ptr(zq), L.ptr(codes), L.ptr(lat), L.ptr(ws), ws.numel(), L.stream()))
        return zq, codes.long(), lat, None, None

    def from_codes(self, codes):
        """-> (z_q [B, D, T], z_p [B, nq * codebook_dim, T], codes)."""
        dac = self._owner[0]
        L.require_cuda(codes)
        B, nq, T = codes.shape
        if nq > self.n_codebooks:
            raise RuntimeError(f"from_codes: {nq} code rows but the model has {self.n_codebooks} codebooks")
        ci = codes.to(torch.int32).contiguous()
        with dac._lock, torch.cuda.device(codes.device):
            h = dac.handle()
            zq = torch.empty(B, dac.latent_dim, T, dtype=torch.float32, device=codes.device)
            L.check(L.load().fd_rvq_from_codes(h, L.ptr(ci), B, nq, T, L.ptr(zq), L.stream()))
        z_p = torch.cat([self.quantizers[i].codebook.weight[codes[:, i]].transpose(1, 2) for i in range(nq)], dim=1)   # decode_code: a table lookup
        return zq, z_p, codes


class DAC(nn.Module):
    """Drop-in for `dac.DAC` on the inference surface demo.ipynb uses.  Constructor keywords = dac.DAC.__init__'s."""

    def __init__(self, encoder_dim: int = 64, encoder_rates: Sequence[int] = (2, 4, 8, 8), latent_dim: Optional[int] = None, decoder_dim: int = 1536,
                 decoder_rates: Sequence[int] = (8, 8, 4, 2), n_codebooks: int = 9, codebook_size: int = 1024, codebook_dim=8, quantizer_dropout: bool = False,
                 sample_rate: int = 44100, precision: str = "mfma_decoder", **ignored):
        super().__init__()
        # "mfma_decoder" (default: decoder on the matrix cores, split-bf16 operands, f32 tolerance; encoder exact: code indices bit-identical
        # to the oracle) | "exact" | "mfma" (encoder too: ~3x faster encode, code indices may differ at near-ties)
        self.precision = precision
        if not isinstance(codebook_dim, int):
            raise NotImplementedError("flowdec_amd.ndac.DAC: per-codebook dimensions (a list) are not supported")
        self.encoder_dim, self.encoder_rates, self.decoder_dim, self.decoder_rates = encoder_dim, tuple(encoder_rates), decoder_dim, tuple(decoder_rates)
        self.sample_rate = sample_rate
        if latent_dim is None:
            latent_dim = encoder_dim * 2 ** len(self.encoder_rates)
        self.latent_dim = latent_dim
        self.hop_length = int(np.prod(self.encoder_rates))
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        enc = nn.Module()
        d = encoder_dim
        blocks = [_wn(nn.Conv1d(1, d, 7, padding=3))]
        for s in self.encoder_rates:
            d *= 2
            blocks.append(_enc_block(d, s))
        blocks += [Snake1d(d), _wn(nn.Conv1d(d, latent_dim, 3, padding=1))]
        enc.block = nn.Sequential(*blocks)
        self.encoder = enc
        self.quantizer = ResidualVectorQuantize(self, latent_dim, n_codebooks, codebook_size, codebook_dim)
        dec = nn.Module()
        layers = [_wn(nn.Conv1d(latent_dim, decoder_dim, 7, padding=3))]
        od = decoder_dim
        for i, s in enumerate(self.decoder_rates):
            layers.append(_dec_block(decoder_dim // 2 ** i, decoder_dim // 2 ** (i + 1), s))
            od = decoder_dim // 2 ** (i + 1)
        layers += [Snake1d(od), _wn(nn.Conv1d(od, 1, 7, padding=3)), nn.Tanh()]
        dec.model = nn.Sequential(*layers)
        self.decoder = dec
        for p in self.parameters():
            p.requires_grad_(False)
        self._handle, self._handle_sig, self._lock = None, None, threading.RLock()
        self.eval()

    # ---- checkpoint container of audiotools' BaseModel.save: {"state_dict": ..., "metadata": {"kwargs": {...}}} ----------------
    @classmethod
    def load(cls, location, *args, **kwargs):
        obj = torch.load(str(location), map_location="cpu", weights_only=False)
        if not isinstance(obj, dict) or "state_dict" not in obj:
            raise RuntimeError("DAC.load: expected a dict with 'state_dict' and 'metadata' (audiotools BaseModel.save format)")
        kw = dict((obj.get("metadata") or {}).get("kwargs") or {})
        kw.update(kwargs)
        model = cls(**kw)
        model.load_state_dict(obj["state_dict"])
        return model

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {}
        for k, v in state_dict.items():   # torch >= 2.1 parametrised weight_norm spells the pair original0 (g) / original1 (v)
            k = k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v")
            sd[k] = v
        self._handle_sig = None
        return super().load_state_dict(sd, strict=strict, **kw)

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- native handle ----------------------------------------------------------------------------------------------------
    def _effective(self):
        """state_dict with weight norm folded: `<m>.weight` = g * v / ||v|| (norm over all dims but 0), float64 inside."""
        sd, out = self.state_dict(), {}
        for k, v in sd.items():
            if k.endswith(".weight_g"):
                base = k[: -len(".weight_g")]
                vv = sd[base + ".weight_v"].double()
                n = vv.pow(2).sum(dim=tuple(range(1, vv.ndim)), keepdim=True).sqrt()
                out[base + ".weight"] = (v.double().reshape(n.shape) * vv / n).float()
            elif not k.endswith(".weight_v"):
                out[k] = v.float()
        return out

    def invalidate(self):
        if self._handle is not None:
            L.load().fd_ndac_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_handle"], st["_handle_sig"] = None, None
        st.pop("_lock", None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._lock = threading.RLock()

    def handle(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd.ndac: the codec must be on the GPU (`dac_model.to('cuda')`); there is no CPU path")
        sig = (dev, tuple(p._version for p in self.parameters()))
        if self._handle is not None and sig == self._handle_sig:
            return self._handle
        self.invalidate()
        lib = L.load()
        cfg = L.FdNdacConfig()
        cfg.encoder_dim, cfg.n_encoder_rates, cfg.latent_dim = self.encoder_dim, len(self.encoder_rates), self.latent_dim
        cfg.decoder_dim, cfg.n_decoder_rates = self.decoder_dim, len(self.decoder_rates)
        for i, s in enumerate(self.encoder_rates):
            cfg.encoder_rates[i] = int(s)
        for i, s in enumerate(self.decoder_rates):
            cfg.decoder_rates[i] = int(s)
        cfg.n_codebooks, cfg.codebook_size, cfg.codebook_dim = self.n_codebooks, self.codebook_size, self.codebook_dim
        with torch.cuda.device(dev):
            h = C.c_void_p()
            L.check(lib.fd_ndac_create(C.byref(cfg), C.byref(h)))
            eff = self._effective()
            for i in range(lib.fd_ndac_num_params(h)):
                name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int * 3)()
                L.check(lib.fd_ndac_param_info(h, i, C.byref(name), C.byref(ndim), C.byref(shape)))
                key = name.value.decode()
                if key not in eff:
                    lib.fd_ndac_destroy(h)
                    raise RuntimeError(f"flowdec_amd.ndac: parameter '{key}' missing from the module")
                t = eff[key].detach().cpu().contiguous()
                if tuple(t.shape) != tuple(shape[:ndim.value]):
                    lib.fd_ndac_destroy(h)
                    raise RuntimeError(f"flowdec_amd.ndac: '{key}' has shape {tuple(t.shape)}, the native model expects {tuple(shape[:ndim.value])}")
                L.check(lib.fd_ndac_set_param(h, name.value, C.c_void_p(t.data_ptr()), t.numel()))
            L.check(lib.fd_ndac_finalize(h, L.stream()))
        self._handle, self._handle_sig = h, sig
        return h

    PRECISIONS = {"exact": 0, "mfma_decoder": 1, "mfma": 3}   # include/flowdec_hip.h FD_NDAC_EXACT / _MFMA_DECODER / | _MFMA_ENCODER

    def _apply_precision(self, h):
        if self.precision not in self.PRECISIONS:
            raise ValueError(f"DAC.precision must be one of {sorted(self.PRECISIONS)} (got {self.precision!r})")
        L.check(L.load().fd_ndac_set_precision(h, self.PRECISIONS[self.precision]))

    # ---- dac.DAC API ------------------------------------------------------------------------------------------------------
    def preprocess(self, audio_data, sample_rate=None):
        """dac.DAC.preprocess: right-pad to a multiple of the hop length."""
        if sample_rate is None:
            sample_rate = self.sample_rate
        assert sample_rate == self.sample_rate, f"expected {self.sample_rate} Hz audio (got {sample_rate})"
        length = audio_data.shape[-1]
        right_pad = math.ceil(length / self.hop_length) * self.hop_length - length
        return nn.functional.pad(audio_data, (0, right_pad))

    @torch.no_grad()
    def encode(self, audio_data, n_quantizers: Optional[int] = None):
        """audio_data [B, 1, L] (L % hop_length == 0) -> (z, codes, latents, commitment_loss=None, codebook_loss=None)."""
        L.require_cuda(audio_data)
        if audio_data.ndim != 3 or audio_data.shape[1] != 1:
            raise RuntimeError(f"encode expects [B, 1, L] audio (got {tuple(audio_data.shape)})")
        B, _, Lw = audio_data.shape
        if Lw % self.hop_length:
            raise RuntimeError(f"encode: length {Lw} is not a multiple of the hop length {self.hop_length}; call preprocess() first")
        nq = self.n_codebooks if n_quantizers is None else max(1, min(int(n_quantizers), self.n_codebooks))
        x = audio_data.float().contiguous()
        lib = L.load()
        with self._lock, torch.cuda.device(x.device):
            h = self.handle()
            self._apply_precision(h)
            T = lib.fd_ndac_latent_frames(h, Lw)
            z = torch.empty(B, self.latent_dim, T, dtype=torch.float32, device=x.device)
            codes = torch.empty(B, nq, T, dtype=torch.int32, device=x.device)
            lat = torch.empty(B, nq * self.codebook_dim, T, dtype=torch.float32, device=x.device)
            ws = torch.empty(lib.fd_ndac_workspace_bytes(h, B, Lw), dtype=torch.uint8, device=x.device)
            L.check(lib.fd_ndac_encode(h, L.ptr(x), B, Lw, nq, L.ptr(z), L.ptr(codes), L.ptr(lat), L.ptr(ws), ws.numel(), L.stream()))
        return z, codes.long(), lat, None, None

    @torch.no_grad()
    def decode(self, z):
        """z [B, latent_dim, T] -> audio [B, 1, L'] in (-1, 1)."""
        L.require_cuda(z)
        if z.ndim != 3 or z.shape[1] != self.latent_dim:
            raise RuntimeError(f"decode expects [B, {self.latent_dim}, T] latents (got {tuple(z.shape)})")
        B, _, T = z.shape
        zz = z.float().contiguous()
        lib = L.load()
        with self._lock, torch.cuda.device(z.device):
            h = self.handle()
            self._apply_precision(h)
            Lo = lib.fd_ndac_decoded_length(h, T)
            out = torch.empty(B, 1, Lo, dtype=torch.float32, device=z.device)
            ws = torch.empty(lib.fd_ndac_workspace_bytes(h, B, max(Lo, T * self.hop_length)), dtype=torch.uint8, device=z.device)
            L.check(lib.fd_ndac_decode(h, L.ptr(zz), B, T, L.ptr(out), L.ptr(ws), ws.numel(), L.stream()))
        return out

    def forward(self, audio_data, sample_rate=None, n_quantizers: Optional[int] = None):
        """dac.DAC.forward (inference part): preprocess -> encode -> decode, trimmed to the input length."""
        length = audio_data.shape[-1]
        x = self.preprocess(audio_data, sample_rate)
        z, codes, latents, _, _ = self.encode(x, n_quantizers)
        return {"audio": self.decode(z)[..., :length], "z": z, "codes": codes, "latents": latents}
