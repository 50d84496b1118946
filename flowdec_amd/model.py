"""Host-side mirror of the reference's Python API for the inference hot path.

Same names, argument meaning, defaults, return conventions and state_dict key layout as

* ``flowdec.model.FlowModel``            (flowdec/model.py:391-536)  -> :class:`FlowModel`
* ``flowdec.backbones.ncsnpp.NCSNpp``    (flowdec/backbones/ncsnpp.py:49-411) -> :class:`NCSNpp`
* ``flowdec.data.feature_extractors.AmplitudeCompressedComplexSTFT`` (:29-59) -> same name here

but every FLOP runs in libflowdec_hip.so (hand-written HIP for gfx950).  The ``nn.Module`` tree only
holds parameters (so ``load_state_dict(ckpt['_pl_ema_state_dict'])`` works unchanged); there is no
Lightning / Hydra / torchdyn dependency and no PyTorch compute fallback.
"""
import ctypes as C
import math
import os
import threading
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


# ------------------------------------------------------------------------------------------------
# parameter containers (names/shapes identical to the reference modules)
# ------------------------------------------------------------------------------------------------
class GaussianFourierProjection(nn.Module):
    """layerspp.py:42-51 (parameter container; W is frozen)."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class ResnetBlockBigGANpp(nn.Module):
    """layerspp.py:222-284 (parameter container)."""

    def __init__(self, in_ch, out_ch=None, temb_dim=None, up=False, down=False):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = nn.GroupNorm(min(in_ch // 4, 32), in_ch, eps=1e-6)
        self.Conv_0 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.Dense_0 = nn.Linear(temb_dim, out_ch)
        self.GroupNorm_1 = nn.GroupNorm(min(out_ch // 4, 32), out_ch, eps=1e-6)
        self.Conv_1 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        if in_ch != out_ch or up or down:
            self.Conv_2 = nn.Conv2d(in_ch, out_ch, 1)
        self.up, self.down, self.in_ch, self.out_ch = up, down, in_ch, out_ch


class Combine(nn.Module):
    """layerspp.py:54-69 (parameter container, method 'sum')."""

    def __init__(self, dim1, dim2):
        super().__init__()
        self.Conv_0 = nn.Conv2d(dim1, dim2, 1)


DEFAULT_OUTPUTLAYER_KWARGS = dict(kernel_size=3, bias=False, padding="same", padding_mode="zeros")
# 3x3 convolution algorithm of the bf16 mode: direct MFMA implicit GEMM, Winograd F(2,3) wherever the shape allows it, or
# Winograd only at the low-resolution levels, or chosen per launch by grid fill ('auto'; include/flowdec_hip.h: FD_WINOGRAD*)
# Tolerances of the adaptive solvers when enhance() gets none: the reference builds `NeuralODE(node_fn, solver=solver, sensitivity=
# 'adjoint')` WITHOUT tolerances (flowdec/model.py:503-515), i.e. it runs on torchdyn's NeuralODE defaults -- atol = rtol = 1e-3 in
# torchdyn 1.0.6 (1e-4 are the adjoint's / ODEProblem's).  torchdyn is not installable offline: scripts/pin_third_party.py checks this
# value against the real package the first time it runs online (oracle/flowdec_oracle.py, section f4).
ADAPTIVE_DEFAULT_TOL = 1e-3

CONV_ALGOS = {"direct": 0, "winograd": L.FD_WINOGRAD, "winograd_lowres": L.FD_WINOGRAD_LOWRES, "auto": L.FD_WINOGRAD_AUTO,
              "latency": L.FD_LOW_LATENCY}   # one short clip on the whole chip (FD_LOW_LATENCY)


class NCSNpp(nn.Module):
    """NCSN++ vector field v(x_t, y, t) -- constructor signature of ncsnpp.py:52-75.

    Supported (= every shipped FlowDec config, config/model/backbone/ncsnpp_final_no_attn.yaml):
    swish, BigGAN blocks, FIR [1,3,3,1], skip_rescale, progressive 'output_skip', progressive_input
    'input_skip' combined by 'sum', Fourier embedding, no attention, 1x1 bias-free output layer.
    Anything else raises NotImplementedError.
    """

    def __init__(self, nonlinearity="swish", nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2,
                 attn_resolutions=(64, 32, 16, 8), resamp_with_conv=True, conditional=True, fir=True,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, resblock_type="biggan", progressive="output_skip",
                 progressive_input="input_skip", progressive_combine="sum", init_scale=0.0, fourier_scale=16,
                 image_size=256, embedding_type="fourier", dropout=0.0, num_channels=4,
                 output_layer_kwargs: dict = DEFAULT_OUTPUTLAYER_KWARGS, bottleneck_attn: bool = True,
                 precision: str = "bf16", conv_algo: str = "auto", side_stream: bool = True):
        super().__init__()
        self.side_stream = bool(side_stream)   # fork the time embedding / pyramid-head chain onto a second stream (FD_NO_SIDE_STREAM when False)
        ch_mult = tuple(ch_mult)
        all_res = [image_size // (2 ** i) for i in range(len(ch_mult))]
        unsupported = []
        if nonlinearity != "swish": unsupported.append("nonlinearity != swish")
        if any(r in tuple(attn_resolutions) for r in all_res) or bottleneck_attn: unsupported.append("attention blocks")
        if not (conditional and fir and skip_rescale and resamp_with_conv): unsupported.append("conditional/fir/skip_rescale/resamp_with_conv must be True")
        if tuple(fir_kernel) != (1, 3, 3, 1): unsupported.append("fir_kernel != [1,3,3,1]")
        if resblock_type.lower() != "biggan": unsupported.append("resblock_type != biggan")
        if progressive.lower() != "output_skip" or progressive_input.lower() != "input_skip" or progressive_combine.lower() != "sum":
            unsupported.append("progressive modes other than output_skip/input_skip/sum")
        if embedding_type.lower() != "fourier": unsupported.append("embedding_type != fourier")
        if dropout != 0.0: unsupported.append("dropout != 0 (inference only)")
        if num_channels != 4: unsupported.append("num_channels != 4")
        if dict(output_layer_kwargs).get("kernel_size", 3) != 1 or dict(output_layer_kwargs).get("bias", False):
            unsupported.append("output layer other than 1x1 without bias")
        if unsupported:
            raise NotImplementedError("flowdec_amd.NCSNpp: unsupported configuration: " + "; ".join(unsupported))
        if precision not in ("bf16", "fp32", "mixed", "bf16x3"):
            raise ValueError("precision must be 'bf16', 'fp32', 'mixed' (f32 activations / residual stream, bf16 MFMA operands) or 'bf16x3' "
                             "(f32 storage, two-term bf16 split operands: f32-mode tolerances on the bf16 matrix cores)")
        if conv_algo not in CONV_ALGOS or (conv_algo not in ("direct", "auto") and precision != "bf16"):
            raise ValueError(f"conv_algo must be one of {sorted(CONV_ALGOS)} (precision != 'bf16': 'direct' or 'auto' only)")
        if precision in ("mixed", "bf16x3"):
            conv_algo = "direct"   # 'auto' = best available: the split / mixed operand modes have the direct kernel only
        # precision='fp32' + 'auto': 2-D Winograd F(4x4, 3x3) in exact float32 (conv_wino44f.hip) for every 3x3 layer with a multiple of 128 output
        # channels on whole 16 x 16 tiles
        self.nf, self.ch_mult, self.num_res_blocks, self.precision, self.conv_algo = nf, ch_mult, num_res_blocks, precision, conv_algo
        self.num_resolutions = len(ch_mult)
        self.output_layer = nn.Conv2d(num_channels, 2, kernel_size=1, bias=False)
        temb_dim = nf * 4
        mods = [GaussianFourierProjection(embedding_size=nf, scale=fourier_scale), nn.Linear(2 * nf, temb_dim),
                nn.Linear(temb_dim, temb_dim), nn.Conv2d(num_channels, nf, 3, padding=1)]
        hs_c, in_ch = [nf], nf
        R = self.num_resolutions
        for lvl in range(R):
            for _ in range(num_res_blocks):
                out_ch = nf * ch_mult[lvl]
                mods.append(ResnetBlockBigGANpp(in_ch, out_ch, temb_dim)); in_ch = out_ch; hs_c.append(in_ch)
            if lvl != R - 1:
                mods.append(ResnetBlockBigGANpp(in_ch, temb_dim=temb_dim, down=True))
                mods.append(Combine(num_channels, in_ch)); hs_c.append(in_ch)
        in_ch = hs_c[-1]
        mods += [ResnetBlockBigGANpp(in_ch, temb_dim=temb_dim), ResnetBlockBigGANpp(in_ch, temb_dim=temb_dim)]
        for lvl in reversed(range(R)):
            for _ in range(num_res_blocks + 1):
                out_ch = nf * ch_mult[lvl]
                mods.append(ResnetBlockBigGANpp(in_ch + hs_c.pop(), out_ch, temb_dim)); in_ch = out_ch
            mods.append(nn.GroupNorm(min(in_ch // 4, 32), in_ch, eps=1e-6))
            mods.append(nn.Conv2d(in_ch, num_channels, 3, padding=1))
            if lvl != 0:
                mods.append(ResnetBlockBigGANpp(in_ch, temb_dim=temb_dim, up=True))
        assert not hs_c
        self.all_modules = nn.ModuleList(mods)
        for p in self.parameters():
            p.requires_grad_(False)
        # ---- native state ----
        self._handle = None
        self._handle_sig = None
        self._ws = {}
        self._sigma_y = None
        self._stft_cfg = dict(n_fft=1534, hop=384, alpha=0.3, beta=0.33)
        self._normalize = True
        self._native_lock = threading.RLock()
        self._last_call_done = None   # event at the end of the previous native call (see _NativeCall)

    # -- native handle management ----------------------------------------------------------------
    def _config_struct(self):
        cfg = L.FdModelConfig()
        cfg.nf = self.nf
        for i, c in enumerate(self.ch_mult):
            cfg.ch_mult[i] = int(c)
        cfg.num_levels = len(self.ch_mult)
        cfg.num_res_blocks = self.num_res_blocks
        cfg.n_fft, cfg.hop = self._stft_cfg["n_fft"], self._stft_cfg["hop"]
        cfg.alpha, cfg.beta = self._stft_cfg["alpha"], self._stft_cfg["beta"]
        cfg.act_dtype = (L.FD_BF16 | CONV_ALGOS[self.conv_algo]) if self.precision == "bf16" else \
            (L.FD_F32 | {"mixed": L.FD_BF16_OPERANDS, "bf16x3": L.FD_BF16X3_OPERANDS, "fp32": CONV_ALGOS[self.conv_algo]}[self.precision])
        if not self.side_stream:
            cfg.act_dtype |= L.FD_NO_SIDE_STREAM
        return cfg

    def invalidate(self):
        """Drop the packed device copy (called after parameters change)."""
        if self._handle is not None:
            L.load().fd_model_destroy(self._handle)
        self._handle, self._ws = None, {}

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    # copy.deepcopy / pickling (EMA copies, torch.save of the module): the native handle, its workspaces and the lock stay behind;
    # the copy repacks lazily on first use
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_handle"], st["_handle_sig"], st["_ws"], st["_last_call_done"] = None, None, {}, None
        st.pop("_native_lock", None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._native_lock = threading.RLock()

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._handle_sig = None  # parameters changed -> repack lazily

    def _sig(self):
        p = next(self.parameters())
        sig_sigma = None if self._sigma_y is None else (self._sigma_y.data_ptr(), self._sigma_y._version)
        return (p.device, self.precision, self.conv_algo, self.side_stream, bool(self._normalize), tuple(sorted(self._stft_cfg.items())), sig_sigma,
                tuple(q._version for q in self.parameters()))

    def handle(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd: the model must be on the GPU (`model.cuda()`); there is no CPU path")
        sig = self._sig()
        if self._handle is not None and sig == self._handle_sig:
            return self._handle
        self.invalidate()
        lib = L.load()
        with torch.cuda.device(dev):
            h = C.c_void_p()
            cfg = self._config_struct()
            L.check(lib.fd_model_create(C.byref(cfg), C.byref(h)))
            sd = {"backbone." + k: v for k, v in self.state_dict().items()}
            n = lib.fd_model_num_params(h)
            for i in range(n):
                name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int * 4)()
                L.check(lib.fd_model_param_info(h, i, C.byref(name), C.byref(ndim), C.byref(shape)))
                key = name.value.decode()
                if key not in sd:
                    lib.fd_model_destroy(h)
                    raise RuntimeError(f"flowdec_amd: parameter {key} missing from the module state")
                t = sd[key].detach().to("cpu", torch.float32).contiguous()
                if list(t.shape) != [shape[j] for j in range(ndim.value)]:
                    lib.fd_model_destroy(h)
                    raise RuntimeError(f"flowdec_amd: parameter {key} has shape {list(t.shape)}")
                L.check(lib.fd_model_set_param(h, name.value, C.c_void_p(t.data_ptr()), t.numel()))
            if self._sigma_y is not None:
                s = self._sigma_y.detach().to("cpu", torch.float64).contiguous().reshape(-1)
                L.check(lib.fd_model_set_sigma_y(h, C.c_void_p(s.data_ptr()), s.numel()))
            L.check(lib.fd_model_set_normalize(h, int(self._normalize)))
            L.check(lib.fd_model_finalize(h, L.stream()))
        self._handle, self._handle_sig = h, sig
        return h

    def workspace(self, key, nbytes, device):
        """One scratch buffer per call kind, grown on demand and reused (a file-by-file driver sees many clip lengths;
        keeping one buffer per (batch, length) would pin every size ever seen)."""
        kind = key[0]
        buf = self._ws.get(kind)
        if buf is None or buf.numel() < nbytes or buf.device != torch.device(device):
            self._ws.pop(kind, None)
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._ws[kind] = buf
        return buf

    # -- reference API -----------------------------------------------------------------------------
    def forward(self, x, y, t):
        with _NativeCall(self):
            return self._forward(x, y, t)

    def _forward(self, x, y, t):
        """x, y: complex64 [B, 1, F, T]; t: [1] or [B]  ->  complex64 [B, 1, F, T] (ncsnpp.py:254-399)."""
        L.require_cuda(x, y, t)
        if x.shape != y.shape or x.ndim != 4 or x.shape[1] != 1:
            raise RuntimeError(f"NCSNpp.forward expects x, y of shape [B, 1, F, T] (got {tuple(x.shape)}, {tuple(y.shape)})")
        h = self.handle()
        lib = L.load()
        B, _, F, T = x.shape
        x = x.to(torch.complex64).contiguous(); y = y.to(torch.complex64).contiguous()
        t = t.to(torch.float32).reshape(-1).contiguous()
        out = torch.empty_like(x)
        need = lib.fd_model_workspace_bytes(h, B, T)
        if need == 0:
            raise RuntimeError("flowdec_hip: " + lib.fd_last_error().decode())
        ws = self.workspace(("fwd", B, T), need, x.device)
        with torch.cuda.device(x.device):   # L.stream() must be the stream of the tensors' device
            L.check(lib.fd_ncsnpp_forward(h, L.ptr(torch.view_as_real(x)), L.ptr(torch.view_as_real(y)), L.ptr(t), t.numel(),
                                          L.ptr(torch.view_as_real(out)), B, T, L.ptr(ws), ws.numel(), L.stream()))
        return out


class _NativeCall:
    """One native call at a time per model -- on the host AND on the device.  The packed model, its workspace, its I/O staging
    buffers and its hipGraph cache are shared state: (i) a lock serialises the enqueueing threads (the C ABI answers a concurrent
    call with FD_EBUSY); (ii) an event recorded at the end of every call makes the NEXT call's stream wait for it, so that two
    callers on different streams cannot overlap on the GPU inside the shared workspace.  The reference's nn.Module can be called
    from several threads / streams; this keeps that working -- calls queue up instead of failing or racing."""

    def __init__(self, backbone):
        self.bb = backbone

    def __enter__(self):
        bb = self.bb
        bb._native_lock.acquire()
        try:
            p = next(bb.parameters())
            self.stream = torch.cuda.current_stream(p.device) if p.is_cuda else None
            if self.stream is not None and bb._last_call_done is not None:
                self.stream.wait_event(bb._last_call_done)
        except BaseException:
            bb._native_lock.release()
            raise
        return self

    def __exit__(self, *exc):
        bb = self.bb
        try:
            if self.stream is not None:
                ev = torch.cuda.Event()
                ev.record(self.stream)
                bb._last_call_done = ev
        finally:
            bb._native_lock.release()
        return False


def _serialized(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with _NativeCall(self if isinstance(self, NCSNpp) else self.backbone):
            return fn(self, *args, **kwargs)
    return wrapper


# ------------------------------------------------------------------------------------------------
# feature extractor mirror
# ------------------------------------------------------------------------------------------------
class ComplexSTFT(nn.Module):
    """feature_extractors.py:62-109 -- torch.stft/istft replaced by the HIP DFT-GEMM kernels."""

    def __init__(self, window_fn, n_fft, sampling_rate, hop_length=None, n_hops=None, learnable_window=False):
        super().__init__()
        assert (hop_length is not None) ^ (n_hops is not None), "Exactly one of {hop_length, n_hops} must be specified!"
        if hop_length is None:
            hop_length = int(math.ceil(n_fft / n_hops))
        if window_fn != "hann" or learnable_window:
            raise NotImplementedError("flowdec_amd.ComplexSTFT: only the fixed symmetric Hann window is implemented")
        self.window = nn.Parameter(torch.signal.windows.hann(n_fft), requires_grad=False)
        self.n_fft, self.hop_length, self.sampling_rate, self.center = n_fft, hop_length, sampling_rate, True

    def forward(self, x):
        """x: [..., L] real (GPU) -> complex64 [..., n_fft/2+1, 1 + L // hop]  (feature_extractors.py:86-96)."""
        from . import ops
        shp = x.shape
        Y, _, T = ops.stft_compress(x.reshape(-1, shp[-1]).float().contiguous(), n_fft=self.n_fft, hop=self.hop_length, alpha=1.0, beta=1.0,
                                    normalize=False)
        return Y[:, 0, :, :T].reshape(*shp[:-1], Y.shape[2], T)

    def invert(self, X, orig_length: Optional[int] = None, **kwargs):
        """torch.istft(..., length=orig_length)  (feature_extractors.py:98-109)."""
        from . import ops
        shp = X.shape
        T = shp[-1]
        if orig_length is None:
            orig_length = self.hop_length * (T - 1)
        Xp = X.reshape(-1, 1, shp[-2], T).to(torch.complex64).contiguous()
        y = ops.decompress_istft(Xp, T, orig_length, None, n_fft=self.n_fft, hop=self.hop_length, alpha=1.0, beta=1.0)
        return y.reshape(*shp[:-2], orig_length)


class CompressAmplitudesAndScale(nn.Module):
    """feature_extractors.py:112-139.  On the hot path the arithmetic is fused into the STFT kernels; forward / invert are
    the stand-alone forms (fd_compress_spec)."""

    def __init__(self, compression_exponent: float, scale_factor: float):
        super().__init__()
        self.compression_exponent, self.scale_factor = compression_exponent, scale_factor

    def _apply_spec(self, x, inverse):
        L.require_cuda(x)
        x = x.to(torch.complex64).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            L.check(L.load().fd_compress_spec(L.ptr(torch.view_as_real(x)), L.ptr(torch.view_as_real(out)), x.numel(),
                                              float(self.compression_exponent), float(self.scale_factor), int(inverse), L.stream()))
        return out

    def forward(self, x):
        return self._apply_spec(x, False)

    def invert(self, x):
        return self._apply_spec(x, True)


class AmplitudeCompressedComplexSTFT(nn.Module):
    """feature_extractors.py:29-59."""

    def __init__(self, window_fn, n_fft, sampling_rate, alpha, beta, hop_length=None, n_hops=None, learnable_window=False):
        super().__init__()
        self.complex_stft = ComplexSTFT(window_fn, n_fft, sampling_rate, hop_length=hop_length, n_hops=n_hops,
                                        learnable_window=learnable_window)
        self.compress = CompressAmplitudesAndScale(alpha, beta)

    def _cfg(self):
        return dict(n_fft=self.complex_stft.n_fft, hop=self.complex_stft.hop_length,
                    alpha=float(self.compress.compression_exponent), beta=float(self.compress.scale_factor))

    def forward(self, x, **kwargs):
        """x: [B, C, T] or [B, T] real (GPU) -> complex STFT, compressed (no normalisation, no padding)."""
        from . import ops
        shp = x.shape
        Y, _, T = ops.stft_compress(x.reshape(-1, shp[-1]).float().contiguous(), normalize=False, **self._cfg())
        return Y[:, 0, :, :T].reshape(*shp[:-1], Y.shape[2], T)

    def invert(self, X, orig_length: Optional[int] = None, **kwargs):
        from . import ops
        shp = X.shape
        T = shp[-1]
        if orig_length is None:
            orig_length = self.complex_stft.hop_length * (T - 1)
        Xp = X.reshape(-1, 1, shp[-2], T).to(torch.complex64).contiguous()
        y = ops.decompress_istft(Xp, T, orig_length, None, **self._cfg())
        return y.reshape(*shp[:-2], orig_length)


# ------------------------------------------------------------------------------------------------
# FlowModel
# ------------------------------------------------------------------------------------------------
def sigma_y_from_file(filename: str, factor: float = 1.0, kernel_bandwidth: Optional[float] = None) -> torch.Tensor:
    """flowdec/data/sigma_models/__init__.py:21-47 (gaussian_filter(mode='nearest') restated in NumPy)."""
    if not os.path.isabs(filename):
        filename = os.path.join(_DATA, os.path.basename(filename))
    curve = np.load(filename).astype(np.float64)
    if kernel_bandwidth is not None:
        radius = int(4.0 * float(kernel_bandwidth) + 0.5)
        k = np.arange(-radius, radius + 1, dtype=np.float64)
        w = np.exp(-0.5 * (k / kernel_bandwidth) ** 2); w /= w.sum()
        xp = np.concatenate([np.full(radius, curve[0]), curve, np.full(radius, curve[-1])])
        curve = np.correlate(xp, w, mode="valid")
    return factor * torch.from_numpy(curve).unsqueeze(-1)


def padded_frames_of(num_samples: int, hop: int = 384) -> int:
    """T_pad of a clip of `num_samples` samples: pad_spec(1 + L // hop) (util/other.py:25-52) -- the bucket key of `enhance_batch`."""
    lib = L.load()
    return int(lib.fd_padded_frames(lib.fd_num_frames(int(num_samples), int(hop))))


def _info_from_ws(lib, h, ws, B, Lw, T, squeeze_dims):
    """preprocess_info of the reference (model.py:161-162); normfac is read where fd_enhance's front end left it."""
    off = lib.fd_enhance_normfac_offset(h, B, Lw)
    normfac = ws[off:off + 4 * B].view(torch.float32).clone().reshape(B, 1, 1)
    return dict(orig_length=Lw, normfac=normfac, undo_pad_fn=(lambda Y_, T=T: Y_[..., :T]), squeeze_dims=squeeze_dims)


def _preprocess(model, y, x=None, comp_eps=None):
    """EnhancementModel._preprocess (model.py:129-163): [L] / [1, L] / [B, 1, L] waveform(s) -> the reference's 3-tuple
    (Y, X, preprocess_info): Y [B, 1, F, T_pad] complex64 with the frame axis zero-padded to a multiple of 64, X the same
    features of the optional clean waveform `x` normalised by y's factor (None when x is None)."""
    from . import ops
    if comp_eps is not None:
        raise NotImplementedError("flowdec_amd: comp_eps (a training-time regulariser, feature_extractors.py:124-125) is not supported")
    if x is not None and x.shape != y.shape:
        raise RuntimeError(f"x and y must have the same shape (got {tuple(x.shape)} vs {tuple(y.shape)})")   # model.py:141
    dev = model.device
    squeeze_dims = 0
    while y.ndim < 3:
        y = y.unsqueeze(0); squeeze_dims += 1
        x = x.unsqueeze(0) if x is not None else x
    if y.ndim != 3 or y.shape[1] != 1:
        raise RuntimeError(f"expected [L], [1, L] or [B, 1, L] waveforms (got {tuple(y.shape)})")
    B, Lw = y.shape[0], y.shape[-1]
    cfg = model.feature_extractor._cfg()
    with torch.cuda.device(dev):
        Y, normfac, T = ops.stft_compress(y.reshape(B, Lw).to(dev, torch.float32).contiguous(), normalize=model.normalize_mode == "noisy", **cfg)
        X = None
        if x is not None:   # x / normfac(y) (util/other.py:79-81), then the same feature extractor and padding, without a second normalisation
            xn = x.reshape(B, Lw).to(dev, torch.float32) / normfac.reshape(B, 1)
            X, _, _ = ops.stft_compress(xn.contiguous(), normalize=False, **cfg)
    info = dict(orig_length=Lw, normfac=normfac.reshape(B, 1, 1), undo_pad_fn=(lambda Y_, T=T: Y_[..., :T]), squeeze_dims=squeeze_dims)
    return Y, X, info


def _postprocess(model, X_hat, preprocess_info, inv_kwargs=None, batch_filter=None):
    """EnhancementModel._postprocess (model.py:165-190): undo padding, invert the features, [select batch items,] restore the
    level and the rank.  `inv_kwargs` is forwarded to the feature extractor's invert like the reference does; the native
    inverse has no optional arguments, so a non-empty dict is rejected instead of being silently ignored."""
    from . import ops
    I = preprocess_info
    assert {"orig_length", "normfac", "undo_pad_fn", "squeeze_dims"} <= I.keys()   # model.py:180
    if inv_kwargs:
        raise NotImplementedError(f"flowdec_amd: feature_extractor.invert takes no extra arguments (got {sorted(inv_kwargs)})")
    T = I["undo_pad_fn"](X_hat).shape[-1]   # frames before pad_spec
    Lw = I["orig_length"]
    B = X_hat.shape[0]
    Xp = X_hat.to(torch.complex64).contiguous()             # the kernel reads the first T frames of the (padded) frame axis
    normfac = I["normfac"]   # [B, 1, 1] tensor ('noisy') or the float 1.0 of normalize_mode='none' (util/other.py:70)
    normfac = normfac.reshape(-1).float() if torch.is_tensor(normfac) else torch.full((1,), float(normfac))
    normfac = normfac.expand(B) if normfac.numel() == 1 else normfac
    with torch.cuda.device(Xp.device):
        x_hat = ops.decompress_istft(Xp, T, Lw, normfac.to(Xp.device).contiguous(), **model.feature_extractor._cfg())
    x_hat = x_hat.reshape(B, 1, Lw)
    if batch_filter is not None:   # model.py:185-187 (x * normfac commutes with selecting batch items)
        x_hat = x_hat[torch.as_tensor(batch_filter, device=x_hat.device)]
    for _ in range(I["squeeze_dims"]):
        x_hat = x_hat.squeeze(0)
    return x_hat


class FlowModel(nn.Module):
    """Drop-in for flowdec.model.FlowModel on the inference path (model.py:391-536).

    Extensions over the reference signature: ``enhance(..., noise=None, generator=None)`` to inject /
    seed the initial Gaussian noise (the reference draws it from the global device RNG, model.py:512),
    and ``use_graph`` (hipGraph replay of the whole solve).  ``with_grad=True`` is not supported.
    """
    strict_loading = False

    def __init__(self, backbone: NCSNpp, feature_extractor: AmplitudeCompressedComplexSTFT, sampling_rate: int,
                 sigma_x=0.0, sigma_y=0.66, flow_matcher=None, lr: float = 1e-4, normalize_mode: str = "noisy", **kwargs):
        super().__init__()
        assert normalize_mode in ("noisy", "none")
        self.sampling_rate, self.normalize_mode, self.lr, self.flow_matcher = sampling_rate, normalize_mode, lr, flow_matcher
        self.backbone, self.feature_extractor = backbone, feature_extractor
        self.sigma_x = nn.Parameter(sigma_x if isinstance(sigma_x, torch.Tensor) else torch.tensor(float(sigma_x)), requires_grad=False)
        self.sigma_y = nn.Parameter(sigma_y if isinstance(sigma_y, torch.Tensor) else torch.tensor(float(sigma_y)), requires_grad=False)
        self._io = {}
        self._side_stream = None

    # -- helpers ------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.sigma_y.device

    def load_state_dict(self, state_dict, strict: bool = False, **kw):
        # the reference sets strict_loading = False (model.py:397)
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _sync_native(self):
        self.backbone._sigma_y = self.sigma_y
        self.backbone._stft_cfg = self.feature_extractor._cfg()
        self.backbone._normalize = self.normalize_mode == "noisy"
        return self.backbone.handle()

    # EnhancementModel._preprocess / _postprocess (model.py:129-190) as stand-alone calls
    def _preprocess(self, y, x=None, comp_eps=None):
        return _preprocess(self, y, x, comp_eps)

    def _postprocess(self, X_hat, preprocess_info, inv_kwargs=None, batch_filter=None):
        return _postprocess(self, X_hat, preprocess_info, inv_kwargs, batch_filter)

    def _io_buffers(self, B, Lw, Tp, F, dev):
        key = (B, Lw, str(dev))
        io = self._io.get(key)
        if io is None:
            io = dict(y=torch.empty(B, Lw, dtype=torch.float32, device=dev),
                      noise=torch.empty(B, 1, F, Tp, dtype=torch.complex64, device=dev),
                      out=torch.empty(B, Lw, dtype=torch.float32, device=dev))
            self._io = {key: io}  # keep one shape resident
        return io

    def forward(self, xt, y, t):
        if t.ndim == 0:
            t = t.unsqueeze(0)  # model.py:471-472
        self._sync_native()
        return self.backbone(xt, y, t)

    def _get_noise_tensor(self, shape, dev, noise, generator):
        if noise is not None:
            return noise.to(dev, torch.complex64).reshape(shape)
        return torch.randn(shape, dtype=torch.complex64, device=dev, generator=generator)

    @torch.no_grad()
    @_serialized
    def enhance(self, y, return_preprocess_info: bool = False, N: int = 50, solver: str = "euler", with_grad: bool = False,
                sigma_fac: float = 1.0, return_traj: bool = False, noise=None, generator=None, use_graph: bool = True, **kwargs):
        """Enhances a coded/noisy waveform y (model.py:476-528).  y: [L], [1, L] or [B, 1, L]."""
        if with_grad:
            raise NotImplementedError("flowdec_amd.FlowModel.enhance: with_grad=True (backprop through the solver) is out of scope")
        adaptive = solver in L.ADAPTIVE_SOLVERS
        if not adaptive and solver not in L.SOLVERS:
            raise ValueError(f"unknown solver {solver!r}; supported: {sorted(L.SOLVERS) + sorted(L.ADAPTIVE_SOLVERS)}")
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd: move the model to the GPU first (`model.cuda()`)")
        orig_device = y.device
        squeeze_dims = 0
        y3 = y
        while y3.ndim < 3:  # model.py:146-149
            y3 = y3.unsqueeze(0); squeeze_dims += 1
        if y3.ndim != 3 or y3.shape[1] != 1:
            raise RuntimeError(f"enhance expects [L], [1, L] or [B, 1, L] waveforms (got {tuple(y.shape)})")
        lib = L.load()
        h = self._sync_native()
        cfg = self.feature_extractor._cfg()
        B, Lw = y3.shape[0], y3.shape[-1]
        F = cfg["n_fft"] // 2 + 1
        T = lib.fd_num_frames(Lw, cfg["hop"]); Tp = lib.fd_padded_frames(T)
        with torch.cuda.device(dev):
            io = self._io_buffers(B, Lw, Tp, F, dev)
            io["y"].copy_(y3.reshape(B, Lw))
            io["noise"].copy_(self._get_noise_tensor((B, 1, F, Tp), dev, noise, generator))
            # stream capture is illegal on the legacy default stream: run the solve on a side stream that is
            # ordered after / before the caller's current stream
            cur = torch.cuda.current_stream(dev)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(dev)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if adaptive:
                    res = self._enhance_adaptive(lib, h, cfg, io, B, Lw, F, T, Tp, N, sigma_fac, return_traj, squeeze_dims, dev,
                                                 float(kwargs.get("atol", ADAPTIVE_DEFAULT_TOL)), float(kwargs.get("rtol", ADAPTIVE_DEFAULT_TOL)), L.ADAPTIVE_SOLVERS[solver])
                else:
                    res = self._enhance_native(lib, h, cfg, io, B, Lw, F, T, Tp, N, solver, sigma_fac, return_traj,
                                               return_preprocess_info, squeeze_dims, use_graph, dev)
            cur.wait_stream(side)
        if return_traj:
            res[0].record_stream(cur)
            for w in res[1]:
                w.record_stream(cur)
            return res
        x_hat, info = res
        x_hat.record_stream(cur)
        for _ in range(squeeze_dims):
            x_hat = x_hat.squeeze(0)
        x_hat = x_hat.to(orig_device)
        return (x_hat, info) if return_preprocess_info else x_hat

    @torch.no_grad()
    @_serialized
    def enhance_batch(self, clips, N: int = 50, solver: str = "euler", sigma_fac: float = 1.0, noise=None, generator=None,
                      use_graph: bool = True):
        """`[self.enhance(c, N=N, solver=solver) for c in clips]` as ONE native call (fd_enhance_ragged) for clips of DIFFERENT
        lengths whose spectrograms pad to the same T_pad -- what the reference's driver does file by file (enhance.py:96-137).

        clips: sequence of waveforms [L_i], [1, L_i] or [1, 1, L_i].  Returns a list of tensors, each with the shape and on the
        device of its input.  Every clip's result is BIT-IDENTICAL to `self.enhance(clip)` with the same noise: per-clip
        normalisation, reflect padding, frame count, zero padding of the frame axis, iSTFT length (model.py:129-190); the network
        itself never mixes batch items.  Initial noise: `noise` = one [1, 1, F, T_pad] complex tensor per clip, or `generator` =
        one torch.Generator (drawn clip by clip, i.e. the stream a one-by-one loop would consume) or a list of one per clip."""
        if solver not in L.SOLVERS:
            raise ValueError(f"enhance_batch: fixed-step solvers only ({sorted(L.SOLVERS)}), got {solver!r}")
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd: move the model to the GPU first (`model.cuda()`)")
        clips = list(clips)
        if not clips:
            return []
        lib = L.load()
        h = self._sync_native()
        cfg = self.feature_extractor._cfg()
        hop, F = cfg["hop"], cfg["n_fft"] // 2 + 1
        flat = []
        for i, c in enumerate(clips):
            if c.ndim > 3 or any(d != 1 for d in c.shape[:-1]):
                raise RuntimeError(f"enhance_batch: clip {i} must be [L], [1, L] or [1, 1, L] (got {tuple(c.shape)})")
            flat.append(c.reshape(-1))
        lens = [int(c.numel()) for c in flat]
        Tps = {lib.fd_padded_frames(lib.fd_num_frames(l, hop)) for l in lens}
        if len(Tps) != 1:
            raise RuntimeError(f"enhance_batch: the clips pad to different frame counts {sorted(Tps)}; one call takes one T_pad bucket "
                               f"(bucket with flowdec_amd.model.padded_frames_of)")
        Tp = Tps.pop()
        B = len(flat)
        # the row length is the bucket's LARGEST possible clip (1 + Lrow // hop == T_pad): one workspace / one hipGraph per (B, T_pad)
        Lrow = hop * Tp - 1
        from . import ops
        ops.check_ragged_lengths(lens, Lrow, cfg["n_fft"], hop)
        gens = generator if isinstance(generator, (list, tuple)) else [generator] * B
        if noise is not None and len(noise) != B or len(gens) != B:
            raise RuntimeError("enhance_batch: one noise tensor / generator per clip")
        with torch.cuda.device(dev):
            io = self._io_buffers(B, Lrow, Tp, F, dev)
            if "lens" not in io:
                io["lens"] = torch.empty(B, dtype=torch.int32, device=dev)
            io["y"].zero_()
            for b, c in enumerate(flat):
                io["y"][b, :lens[b]].copy_(c)
                io["noise"][b:b + 1].copy_(self._get_noise_tensor((1, 1, F, Tp), dev, None if noise is None else noise[b], gens[b]))
            io["lens"].copy_(torch.tensor(lens, dtype=torch.int32))
            cur = torch.cuda.current_stream(dev)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(dev)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                need = lib.fd_enhance_workspace_bytes(h, B, Lrow)
                if need == 0:
                    raise RuntimeError("flowdec_hip: " + lib.fd_last_error().decode())
                ws = self.backbone.workspace(("enh", B, Lrow), need, dev)
                L.check(lib.fd_enhance_ragged(h, L.ptr(io["y"]), L.ptr(io["lens"]), L.ptr(torch.view_as_real(io["noise"])), float(sigma_fac), int(N),
                                              L.SOLVERS[solver], L.ptr(io["out"]), B, Lrow, L.ptr(ws), ws.numel(), int(use_graph), L.stream()))
                outs = [io["out"][b, :lens[b]].clone() for b in range(B)]
            cur.wait_stream(side)
        res = []
        for b, c in enumerate(clips):
            outs[b].record_stream(cur)
            res.append(outs[b].reshape(c.shape).to(c.device))
        return res

    def _enhance_adaptive(self, lib, h, cfg, io, B, Lw, F, T, Tp, N, sigma_fac, return_traj, squeeze_dims, dev, atol, rtol, method=0):
        """solver='dopri5' / 'tsit5': adaptive 5(4) pair over t_span = linspace(0, 1, N+1) (torchdyn semantics restated, unpinned);
        host-driven (one read-back per attempted step), so no hipGraph.  The realised NFE is left in `self.last_nfe`."""
        from . import ops
        Y, normfac, _ = ops.stft_compress(io["y"], normalize=self.normalize_mode == "noisy", **cfg)
        traj = torch.empty(N + 1, B, 1, F, Tp, dtype=torch.complex64, device=dev) if return_traj else None
        X = torch.empty_like(Y)
        need = lib.fd_ode_adaptive_workspace_bytes(h, B, Tp)
        ws = self.backbone.workspace(("ode", B, Tp), need, dev)
        nfe = C.c_int(0)
        L.check(lib.fd_ode_solve_adaptive_method(h, L.ptr(torch.view_as_real(Y)), L.ptr(torch.view_as_real(io["noise"])), float(sigma_fac), int(N), int(method),
                                          atol, rtol, L.ptr(torch.view_as_real(X)), L.ptr(torch.view_as_real(traj)) if return_traj else None,
                                          C.byref(nfe), B, Tp, L.ptr(ws), ws.numel(), L.stream()))
        self.last_nfe = int(nfe.value)
        if return_traj:
            x_hats = []
            for i in range(N + 1):
                xh = ops.decompress_istft(traj[i], T, Lw, normfac, **cfg).reshape(B, 1, Lw)
                for _ in range(squeeze_dims):
                    xh = xh.squeeze(0)
                x_hats.append(xh)
            return traj, x_hats
        x_hat = ops.decompress_istft(X, T, Lw, normfac, **cfg).reshape(B, 1, Lw)
        info = dict(orig_length=Lw, normfac=normfac.reshape(B, 1, 1), undo_pad_fn=(lambda Y_, T=T: Y_[..., :T]), squeeze_dims=squeeze_dims)
        return x_hat, info

    def _enhance_native(self, lib, h, cfg, io, B, Lw, F, T, Tp, N, solver, sigma_fac, return_traj, return_preprocess_info,
                        squeeze_dims, use_graph, dev):
        if return_traj:   # every solver state is needed: front end, solver and back end as separate native calls
            from . import ops
            Y, normfac, _ = ops.stft_compress(io["y"], normalize=self.normalize_mode == "noisy", **cfg)
            traj = torch.empty(N + 1, B, 1, F, Tp, dtype=torch.complex64, device=dev)
            X = torch.empty_like(Y)
            need = lib.fd_model_workspace_bytes(h, B, Tp)
            ws = self.backbone.workspace(("ode", B, Tp), need, dev)
            L.check(lib.fd_ode_solve(h, L.ptr(torch.view_as_real(Y)), L.ptr(torch.view_as_real(io["noise"])), float(sigma_fac), int(N),
                                     L.SOLVERS[solver], L.ptr(torch.view_as_real(X)), L.ptr(torch.view_as_real(traj)), B, Tp,
                                     L.ptr(ws), ws.numel(), 0, L.stream()))
            x_hats = []
            for i in range(N + 1):
                xh = ops.decompress_istft(traj[i], T, Lw, normfac, **cfg).reshape(B, 1, Lw)
                for _ in range(squeeze_dims):
                    xh = xh.squeeze(0)
                x_hats.append(xh)
            return traj, x_hats
        need = lib.fd_enhance_workspace_bytes(h, B, Lw)
        if need == 0:
            raise RuntimeError("flowdec_hip: " + lib.fd_last_error().decode())
        ws = self.backbone.workspace(("enh", B, Lw), need, dev)
        L.check(lib.fd_enhance(h, L.ptr(io["y"]), L.ptr(torch.view_as_real(io["noise"])), float(sigma_fac), int(N),
                               L.SOLVERS[solver], L.ptr(io["out"]), B, Lw, L.ptr(ws), ws.numel(), int(use_graph), L.stream()))
        x_hat = io["out"].reshape(B, 1, Lw).clone()
        info = None
        if return_preprocess_info:
            info = _info_from_ws(lib, h, ws, B, Lw, T, squeeze_dims)
        return x_hat, info


# ------------------------------------------------------------------------------------------------
# ScoreDec / regression baselines on the same backbone
# ------------------------------------------------------------------------------------------------
class OUVESDE:
    """flowdec.sdes.OUVESDE (sdes.py:132-206): parameters + the closed forms the sampler needs (float32 like the reference)."""

    def __init__(self, theta, sigma_min, sigma_max, N=1000, **ignored_kwargs):
        self.theta, self.sigma_min, self.sigma_max, self.N = float(theta), float(sigma_min), float(sigma_max), int(N)
        self.logsig = float(np.log(self.sigma_max / self.sigma_min))

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N)

    @property
    def T(self):
        return 1

    def _std(self, t: torch.Tensor) -> torch.Tensor:                       # sdes.py:181-192
        th, ls = self.theta, self.logsig
        return torch.sqrt(self.sigma_min ** 2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + ls) * t) - 1) * ls / (th + ls))

    def _mean(self, x0, t, y):                                            # sdes.py:176-179
        e = torch.exp(-self.theta * t)[:, None, None, None]
        return e * x0 + (1 - e) * y

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)


class _WaveModel(nn.Module):
    """Shared waveform plumbing of the baselines (EnhancementModel._preprocess/_postprocess, model.py:129-190)."""
    strict_loading = False

    def __init__(self, backbone: NCSNpp, feature_extractor: AmplitudeCompressedComplexSTFT, sampling_rate: int = 48000,
                 lr: float = 1e-4, normalize_mode: str = "noisy", **kwargs):
        super().__init__()
        assert normalize_mode in ("noisy", "none")
        self.sampling_rate, self.normalize_mode, self.lr = sampling_rate, normalize_mode, lr
        self.backbone, self.feature_extractor = backbone, feature_extractor
        self._io, self._side_stream = {}, None

    def _preprocess(self, y, x=None, comp_eps=None):
        return _preprocess(self, y, x, comp_eps)

    def _postprocess(self, X_hat, preprocess_info, inv_kwargs=None, batch_filter=None):
        return _postprocess(self, X_hat, preprocess_info, inv_kwargs, batch_filter)

    @property
    def device(self):
        return next(self.backbone.parameters()).device

    def load_state_dict(self, state_dict, strict: bool = False, **kw):
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _sync_native(self):
        self.backbone._sigma_y = None
        self.backbone._stft_cfg = self.feature_extractor._cfg()
        self.backbone._normalize = self.normalize_mode == "noisy"
        return self.backbone.handle()

    @_serialized
    def _wave_call(self, y, n_draws, noise, generator, launch, return_preprocess_info=False):
        """launch(lib, h, y_dev [B, L], noise_dev [n_draws, B, 1, F, Tp] | None, out [B, L], ws) on a capture-safe side stream."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd: move the model to the GPU first (`model.cuda()`)")
        orig_device, squeeze_dims, y3 = y.device, 0, y
        while y3.ndim < 3:
            y3 = y3.unsqueeze(0); squeeze_dims += 1
        if y3.ndim != 3 or y3.shape[1] != 1:
            raise RuntimeError(f"enhance expects [L], [1, L] or [B, 1, L] waveforms (got {tuple(y.shape)})")
        lib = L.load()
        h = self._sync_native()
        cfg = self.feature_extractor._cfg()
        B, Lw = y3.shape[0], y3.shape[-1]
        F = cfg["n_fft"] // 2 + 1
        Tp = lib.fd_padded_frames(lib.fd_num_frames(Lw, cfg["hop"]))
        with torch.cuda.device(dev):
            key = (B, Lw, n_draws, str(dev))
            io = self._io.get(key)
            if io is None:
                io = dict(y=torch.empty(B, Lw, dtype=torch.float32, device=dev), out=torch.empty(B, Lw, dtype=torch.float32, device=dev),
                          noise=torch.empty(n_draws, B, 1, F, Tp, dtype=torch.complex64, device=dev) if n_draws else None)
                self._io = {key: io}
            io["y"].copy_(y3.reshape(B, Lw))
            if n_draws:
                if noise is not None:
                    io["noise"].copy_(noise.to(dev, torch.complex64).reshape(io["noise"].shape))
                else:
                    io["noise"].copy_(torch.randn(io["noise"].shape, dtype=torch.complex64, device=dev, generator=generator))
            need = lib.fd_enhance_workspace_bytes(h, B, Lw)
            if need == 0:
                raise RuntimeError("flowdec_hip: " + lib.fd_last_error().decode())
            ws = self.backbone.workspace(("enh", B, Lw), need, dev)
            cur = torch.cuda.current_stream(dev)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(dev)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                launch(lib, h, io, B, Lw, ws)
                x_hat = io["out"].reshape(B, 1, Lw).clone()
                info = _info_from_ws(lib, h, ws, B, Lw, lib.fd_num_frames(Lw, cfg["hop"]), squeeze_dims) if return_preprocess_info else None
            cur.wait_stream(side)
        x_hat.record_stream(cur)
        for _ in range(squeeze_dims):
            x_hat = x_hat.squeeze(0)
        x_hat = x_hat.to(orig_device)
        return (x_hat, info) if return_preprocess_info else x_hat


class ScoreModel(_WaveModel):
    """Drop-in for flowdec.model.ScoreModel on the inference path (model.py:581-690): `forward` = score estimate,
    `enhance` = predictor-corrector sampling (sampler_type='pc').  Extensions: noise= / generator= / use_graph=."""

    def __init__(self, sde: OUVESDE, t_eps: float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.sde, self.t_eps = sde, t_eps

    def sde_std(self, t_batch):
        return self.sde._std(t_batch)

    def forward(self, xt, y, t_batch):
        """-backbone(xt, y, t) / std(t)  (model.py:613-628)."""
        if t_batch.ndim == 0:
            t_batch = t_batch.unsqueeze(0)
        self._sync_native()
        std = self.sde_std(t_batch.float())
        return -self.backbone(xt, y, t_batch) / std.reshape(-1, 1, 1, 1)

    def num_draws(self, N=None, predictor="reverse_diffusion", corrector="ald", corrector_steps=1) -> int:
        N = self.sde.N if N is None else N
        return 1 + N * ((corrector_steps if corrector == "ald" else 0) + (1 if predictor != "none" else 0))

    @torch.no_grad()
    def enhance(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=30, corrector_steps=1, snr=0.5,
                return_preprocess_info=False, denoise=True, noise=None, generator=None, use_graph: bool = True, **kwargs):
        if sampler_type == "ode":
            return self._enhance_ode(y, N=N, denoise=denoise, noise=noise, generator=generator, return_preprocess_info=return_preprocess_info, **kwargs)
        if sampler_type != "pc":
            raise ValueError(f"{sampler_type} is not a valid sampler type!")
        if predictor not in L.PREDICTORS:
            raise ValueError(f"unknown predictor {predictor!r}; supported: {sorted(L.PREDICTORS)}")
        if corrector not in L.CORRECTORS:
            raise ValueError(f"unknown corrector {corrector!r}; supported: {sorted(L.CORRECTORS)}")
        N = self.sde.N if N is None else int(N)
        cfg = L.FdScoreConfig(self.sde.theta, self.sde.sigma_min, self.sde.sigma_max, float(kwargs.get("eps", self.t_eps)), float(snr), N,
                              L.PREDICTORS[predictor], L.CORRECTORS[corrector], int(corrector_steps), int(bool(denoise)))
        n_draws = self.num_draws(N, predictor, corrector, corrector_steps)

        def launch(lib, h, io, B, Lw, ws):
            assert lib.fd_score_num_draws(C.byref(cfg)) == n_draws
            L.check(lib.fd_score_enhance(h, L.ptr(io["y"]), L.ptr(torch.view_as_real(io["noise"])), C.byref(cfg), L.ptr(io["out"]), B, Lw,
                                         L.ptr(ws), ws.numel(), int(use_graph), L.stream()))
        return self._wave_call(y, n_draws, noise, generator, launch, return_preprocess_info)


    @_serialized
    def _enhance_ode(self, y, N=None, denoise=True, noise=None, generator=None, rtol=1e-5, atol=1e-5, method="RK45", eps=None,
                     return_nfe: bool = False, return_preprocess_info: bool = False, **ignored):
        """sampler_type='ode' (sampling/__init__.py:75-146): the probability-flow ODE integrated by scipy.integrate.solve_ivp
        on the host, exactly like the reference; every drift evaluation is one fd_score_eval (backbone + fused update) on
        the GPU, the STFT / iSTFT run in libflowdec_hip.so.  Host-driven and slow by construction (the state crosses PCIe
        twice per evaluation) -- it exists for API parity with ScoreModel.enhance."""
        from scipy import integrate
        from . import ops
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("flowdec_amd: move the model to the GPU first (`model.cuda()`)")
        orig_device, squeeze_dims, y3 = y.device, 0, y
        while y3.ndim < 3:
            y3 = y3.unsqueeze(0); squeeze_dims += 1
        if y3.ndim != 3 or y3.shape[1] != 1:
            raise RuntimeError(f"enhance expects [L], [1, L] or [B, 1, L] waveforms (got {tuple(y.shape)})")
        N = self.sde.N if N is None else int(N)
        t_eps = float(self.t_eps if eps is None else eps)
        lib = L.load()
        h = self._sync_native()
        cfg = self.feature_extractor._cfg()
        B, Lw = y3.shape[0], y3.shape[-1]
        sc = L.FdScoreConfig(self.sde.theta, self.sde.sigma_min, self.sde.sigma_max, t_eps, 0.0, N, 0, 1, 0, int(bool(denoise)))
        with torch.cuda.device(dev):
            Y, normfac, T = ops.stft_compress(y3.reshape(B, Lw).to(dev, torch.float32), normalize=self.normalize_mode == "noisy", **cfg)
            Tp = Y.shape[-1]
            if noise is None:
                noise = torch.randn(Y.shape, dtype=torch.complex64, device=dev, generator=generator)
            std1 = float(self.sde._std(torch.ones(1))[0])
            # prior sample x_T = Y + std(1) * z (sdes.py:197-202), formed on the host where scipy keeps the state anyway
            x0 = (Y.cpu().numpy() + noise.to("cpu", torch.complex64).reshape(Y.shape).numpy() * np.float32(std1)).astype(np.complex64)
            ws = self.backbone.workspace(("fwd", B, Tp), lib.fd_model_workspace_bytes(h, B, Tp), dev)
            xin, out = torch.empty_like(Y), torch.empty_like(Y)

            def evaluate(x_host, t, mode):
                xin.copy_(torch.from_numpy(np.ascontiguousarray(x_host.reshape(Y.shape).astype(np.complex64))))
                L.check(lib.fd_score_eval(h, L.ptr(torch.view_as_real(xin)), L.ptr(torch.view_as_real(Y)), float(np.float32(t)), C.byref(sc), mode,
                                          L.ptr(torch.view_as_real(out)), B, Tp, L.ptr(ws), ws.numel(), L.stream()))
                return out.cpu().numpy().reshape(-1)

            sol = integrate.solve_ivp(lambda t, xf: evaluate(xf, t, 0), (1.0, t_eps), x0.reshape(-1), rtol=rtol, atol=atol, method=method)
            x = sol.y[:, -1]
            if denoise:
                x = evaluate(x, t_eps, 2)
            X = torch.from_numpy(np.ascontiguousarray(x.reshape(Y.shape).astype(np.complex64))).to(dev)
            x_hat = ops.decompress_istft(X, T, Lw, normfac, **cfg).reshape(B, 1, Lw)
        for _ in range(squeeze_dims):
            x_hat = x_hat.squeeze(0)
        x_hat = x_hat.to(orig_device)
        if return_preprocess_info:
            info = dict(orig_length=Lw, normfac=normfac.reshape(B, 1, 1), undo_pad_fn=(lambda Y_, T=T: Y_[..., :T]), squeeze_dims=squeeze_dims)
            return x_hat, info
        return (x_hat, int(sol.nfev)) if return_nfe else x_hat


class RegressionModel(_WaveModel):
    """Drop-in for flowdec.model.RegressionModel.enhance (model.py:539-578): one backbone call with x_t = Y, t = 0."""

    def __init__(self, *args, loss_type: str = "l2", **kwargs):
        super().__init__(*args, **kwargs)
        self.loss_type = loss_type

    def forward(self, xt, y, t):
        if t.ndim == 0:
            t = t.unsqueeze(0)
        self._sync_native()
        return self.backbone(xt, y, t)

    @torch.no_grad()
    def enhance(self, y, return_preprocess_info=False, use_graph: bool = True, **kwargs):
        def launch(lib, h, io, B, Lw, ws):
            L.check(lib.fd_regression_enhance(h, L.ptr(io["y"]), L.ptr(io["out"]), B, Lw, L.ptr(ws), ws.numel(), int(use_graph), L.stream()))
        return self._wave_call(y, 0, None, None, launch, return_preprocess_info)


# ------------------------------------------------------------------------------------------------
# presets (replace hydra.compose('flowdec_75m' | 'flowdec_25s'), config/flowdec_75m.yaml etc.)
# ------------------------------------------------------------------------------------------------
BACKBONE_FINAL_NO_ATTN = dict(image_size=768, nonlinearity="swish", nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1,
                              attn_resolutions=(), bottleneck_attn=False, resamp_with_conv=True, conditional=True, fir=True,
                              fir_kernel=(1, 3, 3, 1), skip_rescale=True, resblock_type="biggan", progressive="output_skip",
                              progressive_input="input_skip", progressive_combine="sum", init_scale=0.0, embedding_type="fourier",
                              fourier_scale=16, dropout=0.0, num_channels=4,
                              output_layer_kwargs=dict(kernel_size=1, bias=False, padding="same", padding_mode="zeros"))

PRESETS = {
    "flowdec_75m": dict(sigma_file="flowdec_autoparams_75m.npy"),
    "flowdec_25s": dict(sigma_file="flowdec_autoparams_25s.npy"),
    "flowdec_75m_globsigy": dict(sigma_file=None),
    "flowdec_25s_globsigy": dict(sigma_file=None),
    # baselines (config/baseline_scoredec_75s.yaml -> model/score_model_final.yaml + sde/ouve_final.yaml; baseline_regression_75s.yaml)
    "baseline_scoredec_75s": dict(kind="score", sde=dict(theta=1.5, sigma_min=0.05, sigma_max=0.82, N=30), t_eps=3e-2),
    "baseline_regression_75s": dict(kind="regression"),
}


def from_preset(name: str = "flowdec_75m", precision: str = "bf16", **backbone_overrides):
    """Instantiate the model a reference user gets from `instantiate(compose(config_name=name)['model'])`."""
    if name not in PRESETS:
        raise KeyError(f"unknown preset {name!r}; available: {sorted(PRESETS)}")
    bb = dict(BACKBONE_FINAL_NO_ATTN); bb.update(backbone_overrides)
    backbone = NCSNpp(precision=precision, **bb)
    fe = AmplitudeCompressedComplexSTFT(window_fn="hann", n_fft=1534, n_hops=4, sampling_rate=48000, alpha=0.3, beta=0.33)
    kind = PRESETS[name].get("kind", "flow")
    if kind == "score":
        return ScoreModel(sde=OUVESDE(**PRESETS[name]["sde"]), t_eps=PRESETS[name]["t_eps"], backbone=backbone, feature_extractor=fe,
                          sampling_rate=48000).eval()
    if kind == "regression":
        return RegressionModel(backbone=backbone, feature_extractor=fe, sampling_rate=48000).eval()
    sf = PRESETS[name]["sigma_file"]
    sigma_y = sigma_y_from_file(sf, factor=1, kernel_bandwidth=3) if sf else 0.66
    return FlowModel(backbone=backbone, feature_extractor=fe, sampling_rate=48000, sigma_x=0.0, sigma_y=sigma_y).eval()
