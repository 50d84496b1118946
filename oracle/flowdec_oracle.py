"""CPU oracle for the FlowDec inference hot path -- TEST INFRASTRUCTURE ONLY.

This is a from-scratch NumPy restatement of the arithmetic performed by
facebookresearch/FlowDec's ``FlowModel.enhance`` (STFT front-end -> NCSN++ vector
field inside a fixed-step ODE loop -> iSTFT back-end).  It exists to CHECK the HIP
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``flowdec_amd`` never does.

Parity status: PINNED against golden vectors generated in the build container by
importing the reference itself (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).  The one part
that is NOT pinned by reference code is the fixed-step solver driver: it lives in
the un-vendored third-party ``torchdyn==1.0.6`` (reference requirements.txt:53;
call site flowdec/model.py:511-514).  Its published algorithm is restated in
``odeint_fixed`` below -- "parity unpinned" for that function only.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Layout convention everywhere: NCHW with H = frequency (768)
and W = time frames, exactly as the reference.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# Configuration (values: config/flowdec_75m.yaml + its defaults chain;
# config/model/backbone/ncsnpp_final_no_attn.yaml; .../compressed_complex_stft_final.yaml)
# --------------------------------------------------------------------------------------
N_FFT = 1534
N_HOPS = 4
HOP = int(math.ceil(N_FFT / N_HOPS))  # feature_extractors.py:69-71 -> 384
N_FREQ = N_FFT // 2 + 1  # 768
ALPHA = 0.3
BETA = 0.33
FIR_KERNEL = (1, 3, 3, 1)

DEFAULT_BACKBONE = dict(nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1, fourier_scale=16,
                        num_channels=4)


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 -> bfloat16 (round-to-nearest-even) and return as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounding = ((u >> 16) & 1) + np.uint32(0x7FFF)
    r = ((u + rounding) & np.uint32(0xFFFF0000)).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out


def silu(x: np.ndarray) -> np.ndarray:
    """nn.SiLU (layers.py:48-49)."""
    return x / (1.0 + np.exp(-x))


def hann_sym(n: int, dtype=np.float64) -> np.ndarray:
    """torch.signal.windows.hann(n) default sym=True (feature_extractors.py:73-75):
    w[k] = 0.5 - 0.5 cos(2 pi k / (n-1)), so w[0] = w[n-1] = 0."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))).astype(dtype)


# --------------------------------------------------------------------------------------
# (a1) normalize_noisy -- flowdec/util/other.py:55-82
# --------------------------------------------------------------------------------------
def normalize_noisy(y: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """y: [B, C, L].  normfac = max|y| over dims 1.. (keepdim); isclose(.,0) -> 1."""
    normfac = np.abs(y).max(axis=tuple(range(1, y.ndim)), keepdims=True)
    # torch.isclose(normfac, 0): |normfac| <= atol(1e-8) + rtol*0
    normfac = np.where(np.abs(normfac) <= 1e-8, np.ones_like(normfac), normfac)
    return y / normfac, normfac


# --------------------------------------------------------------------------------------
# (a2) ComplexSTFT.forward -- flowdec/data/feature_extractors.py:86-96 (torch.stft)
# --------------------------------------------------------------------------------------
def num_frames(L: int) -> int:
    return 1 + L // HOP


def reflect_index(i: np.ndarray, L: int) -> np.ndarray:
    """Index map of torch 'reflect' padding (no edge repeat) for padded index i - pad."""
    i = np.where(i < 0, -i, i)
    i = np.where(i >= L, 2 * (L - 1) - i, i)
    return i


def frame_signal(y: np.ndarray) -> np.ndarray:
    """y [N, L] -> frames [N, T, N_FFT]; frame j covers padded samples [HOP*j, HOP*j+N_FFT)
    of the signal reflect-padded by N_FFT//2 on both sides (center=True)."""
    N, L = y.shape
    T = num_frames(L)
    pad = N_FFT // 2
    idx = (np.arange(T)[:, None] * HOP + np.arange(N_FFT)[None, :]) - pad  # [T, N_FFT]
    idx = reflect_index(idx, L)
    return y[:, idx]


def stft(y: np.ndarray, dtype=np.float64) -> np.ndarray:
    """y: [B, C, L] real -> [B, C, 768, T] complex.  Computed in `dtype` precision and
    returned as complex64 (reference is fp32/complex64)."""
    B, C, L = y.shape
    fr = frame_signal(y.reshape(B * C, L).astype(dtype)) * hann_sym(N_FFT, dtype)[None, None, :]
    X = np.fft.rfft(fr, n=N_FFT, axis=-1)  # [N, T, 768]
    X = np.transpose(X, (0, 2, 1)).reshape(B, C, N_FREQ, -1)
    return X.astype(np.complex64)


# --------------------------------------------------------------------------------------
# (a3) CompressAmplitudesAndScale -- feature_extractors.py:118-139
# --------------------------------------------------------------------------------------
def compress(X: np.ndarray, alpha: float = ALPHA, beta: float = BETA) -> np.ndarray:
    """beta * |X|^alpha * exp(j angle(X))   (:118-128, alpha != 1 branch)."""
    if alpha != 1:
        X = np.abs(X) ** alpha * np.exp(1j * np.angle(X))
    return (X * beta).astype(np.complex64)


def decompress(X: np.ndarray, alpha: float = ALPHA, beta: float = BETA) -> np.ndarray:
    """|X/beta|^(1/alpha) * exp(j angle(X/beta))   (:130-139)."""
    X = X / beta
    if alpha != 1:
        X = np.abs(X) ** (1.0 / alpha) * np.exp(1j * np.angle(X))
    return X.astype(np.complex64)


# --------------------------------------------------------------------------------------
# (a18) ComplexSTFT.invert -- feature_extractors.py:98-109 (torch.istft)
# --------------------------------------------------------------------------------------
def istft(X: np.ndarray, length: int, dtype=np.float64) -> np.ndarray:
    """X: [B, C, 768, T] complex -> [B, C, length].  irfft (imag of DC/Nyquist ignored,
    1/n normalisation) -> * window -> overlap-add -> / overlap-added window^2 ->
    drop N_FFT//2 leading samples -> trim / zero-pad to `length`."""
    B, C, F, T = X.shape
    w = hann_sym(N_FFT, dtype)
    Xf = np.transpose(X.reshape(B * C, F, T), (0, 2, 1)).astype(np.complex128 if dtype == np.float64 else np.complex64)
    fr = np.fft.irfft(Xf, n=N_FFT, axis=-1).astype(dtype) * w[None, None, :]  # [N, T, N_FFT]
    total = N_FFT + HOP * (T - 1)
    out = np.zeros((B * C, total), dtype=dtype)
    env = np.zeros((total,), dtype=dtype)
    w2 = w * w
    for t in range(T):
        out[:, t * HOP:t * HOP + N_FFT] += fr[:, t]
        env[t * HOP:t * HOP + N_FFT] += w2
    start = N_FFT // 2
    end = start + length
    seg = out[:, start:min(end, total)] / env[start:min(end, total)][None, :]
    if seg.shape[1] < length:
        seg = np.concatenate([seg, np.zeros((B * C, length - seg.shape[1]), dtype=dtype)], axis=1)
    return seg.reshape(B, C, length).astype(np.float32)


# --------------------------------------------------------------------------------------
# (a4) pad_spec(mode='zero') -- flowdec/util/other.py:25-52
# --------------------------------------------------------------------------------------
def padded_frames(T: int) -> int:
    return T if T % 64 == 0 else T + (64 - T % 64)


def pad_spec(Y: np.ndarray) -> Tuple[np.ndarray, int]:
    T = Y.shape[-1]
    Tp = padded_frames(T)
    if Tp != T:
        Y = np.concatenate([Y, np.zeros(Y.shape[:-1] + (Tp - T,), dtype=Y.dtype)], axis=-1)
    return Y, T


# --------------------------------------------------------------------------------------
# (a19) sigma_models.from_file -- flowdec/data/sigma_models/__init__.py:21-47
# --------------------------------------------------------------------------------------
def gaussian_filter1d_nearest(x: np.ndarray, sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage.gaussian_filter(x, sigma, mode='nearest') for 1-D x (float64)."""
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 * (k / sigma) ** 2)
    w /= w.sum()
    xp = np.concatenate([np.full(radius, x[0]), x, np.full(radius, x[-1])])
    return np.correlate(xp, w, mode="valid")


def sigma_y_curve(curve: np.ndarray, factor: float = 1.0, kernel_bandwidth: Optional[float] = 3.0) -> np.ndarray:
    c = np.asarray(curve, dtype=np.float64)
    if kernel_bandwidth is not None:
        c = gaussian_filter1d_nearest(c, kernel_bandwidth)
    return factor * c[:, None]  # (768, 1) float64


# --------------------------------------------------------------------------------------
# NCSN++ building blocks
# --------------------------------------------------------------------------------------
def conv2d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], operand_round=None) -> np.ndarray:
    """nn.Conv2d stride 1, 'same' zero padding (layers.py:110-134).  x [B,Ci,H,W],
    w [Co,Ci,kh,kw].  im2col + matmul.  operand_round='bf16' rounds both operands to
    bfloat16 first (emulates the HIP bf16-MFMA path; accumulation stays wide)."""
    B, Ci, H, W = x.shape
    Co, Ci2, kh, kw = w.shape
    assert Ci == Ci2
    if operand_round == "bf16":
        x = round_bf16(x.astype(np.float32)).astype(x.dtype)
        w = round_bf16(w.astype(np.float32)).astype(w.dtype)
    ph, pw = kh // 2, kw // 2
    if kh == 1 and kw == 1:
        out = np.einsum("oc,bchw->bohw", w[:, :, 0, 0], x, optimize=True)
    else:
        xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
        out = np.zeros((B, Co, H, W), dtype=np.result_type(x.dtype, w.dtype))
        # one GEMM per tap on the shifted window (no im2col matrix: 9x less memory traffic than gathering all taps first)
        wt = np.ascontiguousarray(w.transpose(2, 3, 0, 1))          # [kh][kw][Co][Ci]
        for b_ in range(B):
            acc = np.zeros((Co, H * W), dtype=out.dtype)
            for dy in range(kh):
                for dx in range(kw):
                    acc += wt[dy, dx] @ np.ascontiguousarray(xp[b_, :, dy:dy + H, dx:dx + W]).reshape(Ci, -1)
            out[b_] = acc.reshape(Co, H, W)
    if b is not None:
        out = out + b[None, :, None, None]
    return out


def group_norm(x: np.ndarray, G: int, gamma: np.ndarray, beta: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """nn.GroupNorm(G, C, eps=1e-6) (layerspp.py:229,241): biased variance over
    (C/G, H, W) per (b, g)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, G, -1)
    mean = xg.mean(axis=2, keepdims=True, dtype=np.float64)
    var = ((xg - mean) ** 2).mean(axis=2, keepdims=True, dtype=np.float64)
    xn = ((xg - mean) / np.sqrt(var + eps)).astype(x.dtype).reshape(B, C, H, W)
    return xn * gamma[None, :, None, None] + beta[None, :, None, None]


def gn_groups(C: int) -> int:
    return min(C // 4, 32)


def linear(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    return x @ w.T + b


def setup_fir_kernel(k: Sequence[float]) -> np.ndarray:
    """_setup_kernel (up_or_down_sampling.py:206-213)."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upfirdn2d(x: np.ndarray, kernel: np.ndarray, up: int = 1, down: int = 1, pad: Tuple[int, int] = (0, 0)) -> np.ndarray:
    """upfirdn2d_native (op/upfirdn2d.py:183-224): zero-insert upsample by `up`, pad
    (pad[0] before, pad[1] after, on both axes; negative = crop), correlate with the
    FLIPPED kernel ('valid'), keep every `down`-th sample.  x [B,C,H,W]."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    u = np.zeros((B, C, H * up, W * up), dtype=x.dtype)
    u[:, :, ::up, ::up] = x
    u = np.pad(u, ((0, 0), (0, 0), (max(p0, 0), max(p1, 0)), (max(p0, 0), max(p1, 0))))
    u = u[:, :, max(-p0, 0):u.shape[2] - max(-p1, 0), max(-p0, 0):u.shape[3] - max(-p1, 0)]
    kf = kernel[::-1, ::-1].astype(x.dtype)
    oh = u.shape[2] - kh + 1
    ow = u.shape[3] - kw + 1
    out = np.zeros((B, C, oh, ow), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            out += kf[dy, dx] * u[:, :, dy:dy + oh, dx:dx + ow]
    return out[:, :, ::down, ::down]


def upsample_2d(x: np.ndarray, k=FIR_KERNEL, factor: int = 2) -> np.ndarray:
    """up_or_down_sampling.py:220-249."""
    kk = setup_fir_kernel(k) * (factor ** 2)
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x: np.ndarray, k=FIR_KERNEL, factor: int = 2) -> np.ndarray:
    """up_or_down_sampling.py:252-282."""
    kk = setup_fir_kernel(k)
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, down=factor, pad=((p + 1) // 2, p // 2))


def fir_down2_polyphase(x: np.ndarray) -> np.ndarray:
    """Closed form of downsample_2d for k=[1,3,3,1]: per axis
    out[n] = (x[2n-1] + 3x[2n] + 3x[2n+1] + x[2n+2]) / 8, zeros outside."""
    def ax(v, axis):
        v = np.moveaxis(v, axis, -1)
        n = v.shape[-1]
        vp = np.pad(v, [(0, 0)] * (v.ndim - 1) + [(1, 1)])
        o = (vp[..., 0:n:2] + 3 * vp[..., 1:n + 1:2] + 3 * vp[..., 2:n + 2:2] + vp[..., 3:n + 3:2]) / 8
        return np.moveaxis(o, -1, axis)
    return ax(ax(x, 2), 3)


def fir_up2_polyphase(x: np.ndarray) -> np.ndarray:
    """Closed form of upsample_2d for k=[1,3,3,1]: per axis out[2i] = (x[i-1] + 3x[i])/4,
    out[2i+1] = (3x[i] + x[i+1])/4, zeros outside."""
    def ax(v, axis):
        v = np.moveaxis(v, axis, -1)
        n = v.shape[-1]
        vp = np.pad(v, [(0, 0)] * (v.ndim - 1) + [(1, 1)])
        o = np.empty(v.shape[:-1] + (2 * n,), dtype=v.dtype)
        o[..., 0::2] = (vp[..., 0:n] + 3 * vp[..., 1:n + 1]) / 4
        o[..., 1::2] = (3 * vp[..., 1:n + 1] + vp[..., 2:n + 2]) / 4
        return np.moveaxis(o, -1, axis)
    return ax(ax(x, 2), 3)


# --------------------------------------------------------------------------------------
# NCSN++ module list (ncsnpp.py:102-251) for progressive='output_skip',
# progressive_input='input_skip', combine 'sum', biggan blocks, no attention.
# --------------------------------------------------------------------------------------
def build_module_specs(nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1, num_channels=4) -> List[dict]:
    """Returns one dict per entry of `all_modules` in execution order (ncsnpp.py:102-251)."""
    specs: List[dict] = []
    specs.append(dict(kind="gfp"))
    specs.append(dict(kind="linear", cin=2 * nf, cout=4 * nf))
    specs.append(dict(kind="linear", cin=4 * nf, cout=4 * nf))
    specs.append(dict(kind="conv3", cin=num_channels, cout=nf))
    hs_c = [nf]
    in_ch = nf
    R = len(ch_mult)
    for lvl in range(R):
        for _ in range(num_res_blocks):
            out_ch = nf * ch_mult[lvl]
            specs.append(dict(kind="rb", cin=in_ch, cout=out_ch, up=False, down=False))
            in_ch = out_ch
            hs_c.append(in_ch)
        if lvl != R - 1:
            specs.append(dict(kind="rb", cin=in_ch, cout=in_ch, up=False, down=True))
            specs.append(dict(kind="combine", cin=num_channels, cout=in_ch))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    specs.append(dict(kind="rb", cin=in_ch, cout=in_ch, up=False, down=False))
    specs.append(dict(kind="rb", cin=in_ch, cout=in_ch, up=False, down=False))
    for lvl in reversed(range(R)):
        for _ in range(num_res_blocks + 1):
            out_ch = nf * ch_mult[lvl]
            specs.append(dict(kind="rb", cin=in_ch + hs_c.pop(), cout=out_ch, up=False, down=False))
            in_ch = out_ch
        specs.append(dict(kind="gn", c=in_ch))
        specs.append(dict(kind="conv3", cin=in_ch, cout=num_channels))
        if lvl != 0:
            specs.append(dict(kind="rb", cin=in_ch, cout=in_ch, up=True, down=False))
    assert not hs_c
    return specs


def state_dict_manifest(nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1, num_channels=4,
                        temb_dim=None, prefix="backbone.") -> Dict[str, Tuple[int, ...]]:
    """Key -> shape of the reference backbone state_dict."""
    temb_dim = temb_dim or 4 * nf
    m: Dict[str, Tuple[int, ...]] = {}
    m[prefix + "output_layer.weight"] = (2, num_channels, 1, 1)
    for i, s in enumerate(build_module_specs(nf, ch_mult, num_res_blocks, num_channels)):
        p = f"{prefix}all_modules.{i}."
        k = s["kind"]
        if k == "gfp":
            m[p + "W"] = (nf,)
        elif k == "linear":
            m[p + "weight"] = (s["cout"], s["cin"]); m[p + "bias"] = (s["cout"],)
        elif k == "conv3":
            m[p + "weight"] = (s["cout"], s["cin"], 3, 3); m[p + "bias"] = (s["cout"],)
        elif k == "gn":
            m[p + "weight"] = (s["c"],); m[p + "bias"] = (s["c"],)
        elif k == "combine":
            m[p + "Conv_0.weight"] = (s["cout"], s["cin"], 1, 1); m[p + "Conv_0.bias"] = (s["cout"],)
        elif k == "rb":
            ci, co = s["cin"], s["cout"]
            m[p + "GroupNorm_0.weight"] = (ci,); m[p + "GroupNorm_0.bias"] = (ci,)
            m[p + "Conv_0.weight"] = (co, ci, 3, 3); m[p + "Conv_0.bias"] = (co,)
            m[p + "Dense_0.weight"] = (co, temb_dim); m[p + "Dense_0.bias"] = (co,)
            m[p + "GroupNorm_1.weight"] = (co,); m[p + "GroupNorm_1.bias"] = (co,)
            m[p + "Conv_1.weight"] = (co, co, 3, 3); m[p + "Conv_1.bias"] = (co,)
            if ci != co or s["up"] or s["down"]:
                m[p + "Conv_2.weight"] = (co, ci, 1, 1); m[p + "Conv_2.bias"] = (co,)
    return m


def random_state_dict(seed: int = 0, std: float = 0.05, **cfg) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (the reference's default init is degenerate: init_scale 0
    -> ~1e-10 weights in Conv_1 / pyramid heads).  Conv/linear
    weights ~ N(0, gain/fan_in) so activations stay O(1); GN affine ~ 1 + N(0, .1);
    biases ~ N(0, std)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for k, shp in state_dict_manifest(**cfg).items():
        if k.endswith(".W"):
            sd[k] = (rng.standard_normal(shp) * 16.0).astype(np.float32)
        elif "GroupNorm" in k and k.endswith("weight") or (len(shp) == 1 and k.endswith("weight")):
            sd[k] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif k.endswith("bias"):
            sd[k] = (std * rng.standard_normal(shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = (rng.standard_normal(shp) * math.sqrt(1.0 / fan_in)).astype(np.float32)
    return sd


def random_resblock_params(seed: int, ci: int, co: int, has_conv2: bool, temb_dim: int = 256) -> Dict[str, np.ndarray]:
    """Seeded parameters of ONE ResnetBlockBigGANpp (keys as in its state_dict)."""
    rng = np.random.default_rng(seed)
    shapes = {"GroupNorm_0.weight": (ci,), "GroupNorm_0.bias": (ci,), "Conv_0.weight": (co, ci, 3, 3),
              "Conv_0.bias": (co,), "Dense_0.weight": (co, temb_dim), "Dense_0.bias": (co,),
              "GroupNorm_1.weight": (co,), "GroupNorm_1.bias": (co,), "Conv_1.weight": (co, co, 3, 3),
              "Conv_1.bias": (co,)}
    if has_conv2:
        shapes["Conv_2.weight"] = (co, ci, 1, 1); shapes["Conv_2.bias"] = (co,)
    sd = {}
    for k, shp in shapes.items():
        if "GroupNorm" in k and k.endswith("weight"):
            sd[k] = (1 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif k.endswith("bias"):
            sd[k] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        else:
            sd[k] = (rng.standard_normal(shp) / np.sqrt(np.prod(shp[1:]))).astype(np.float32)
    return sd


RESBLOCK_CASES = dict(  # name -> (seed, cin, cout, up, down)
    plain=(11, 256, 256, False, False), widen=(12, 64, 256, False, False), cat=(13, 320, 256, False, False),
    up=(14, 128, 128, True, False), down=(15, 256, 256, False, True))


class NCSNppOracle:
    """NumPy restatement of NCSNpp.forward (flowdec/backbones/ncsnpp.py:254-399) with
    ResnetBlockBigGANpp (layerspp.py:252-284), Combine 'sum' (:54-69),
    GaussianFourierProjection (:42-51)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], nf=64, ch_mult=(4, 4, 4, 2), num_res_blocks=1,
                 num_channels=4, prefix="backbone.", dtype=np.float32, operand_round=None, storage_round=None):
        """operand_round='bf16': every convolution rounds its two operands to bf16 (f32 accumulation).  storage_round='bf16': every
        tensor the HIP bf16 mode keeps in memory between kernels is rounded to bf16 where that mode stores it (conv / FIR / Combine
        outputs, block outputs, the pyramids) -- together the two model the bf16 mode's arithmetic; tests/test_hip_model.py derives its
        tolerance from this prediction (tests/golden/make_golden_bf16_prediction.py)."""
        self.cfg = dict(nf=nf, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks, num_channels=num_channels)
        self.specs = build_module_specs(**self.cfg)
        self.dtype = dtype
        self.operand_round = operand_round
        self._st = (lambda a: round_bf16(np.asarray(a, np.float32)).astype(dtype)) if storage_round == "bf16" else (lambda a: a)
        self.p = {k[len(prefix):] if k.startswith(prefix) else k: np.asarray(v).astype(dtype)
                  for k, v in state_dict.items()}
        self.taps: Dict[str, np.ndarray] = {}

    # -- pieces -------------------------------------------------------------------------
    def _w(self, i: int, name: str) -> np.ndarray:
        return self.p[f"all_modules.{i}.{name}"]

    def time_embedding(self, t: np.ndarray) -> np.ndarray:
        """ncsnpp.py:263-274.  t [Bt] -> temb [Bt, 4nf]."""
        t = np.asarray(t, dtype=self.dtype).reshape(-1)
        W = self._w(0, "W")
        x_proj = t[:, None] * W[None, :] * self.dtype(2) * self.dtype(np.pi)
        emb = np.concatenate([np.sin(x_proj), np.cos(x_proj)], axis=-1)
        h = linear(emb, self._w(1, "weight"), self._w(1, "bias"))
        return linear(silu(h), self._w(2, "weight"), self._w(2, "bias"))

    def resblock(self, i: int, s: dict, x: np.ndarray, temb: np.ndarray) -> np.ndarray:
        """layerspp.py:252-284."""
        ci, co = s["cin"], s["cout"]
        h = silu(group_norm(x, gn_groups(ci), self._w(i, "GroupNorm_0.weight"), self._w(i, "GroupNorm_0.bias")))
        st = self._st
        if s["up"]:
            h = st(upsample_2d(h)); x = st(upsample_2d(x))
        elif s["down"]:
            h = st(downsample_2d(h)); x = st(downsample_2d(x))
        h = conv2d(h, self._w(i, "Conv_0.weight"), self._w(i, "Conv_0.bias"), self.operand_round)
        tb = linear(silu(temb), self._w(i, "Dense_0.weight"), self._w(i, "Dense_0.bias"))  # [Bt, co]
        h = st(h + tb[:, :, None, None])
        h = silu(group_norm(h, gn_groups(co), self._w(i, "GroupNorm_1.weight"), self._w(i, "GroupNorm_1.bias")))
        h = conv2d(h, self._w(i, "Conv_1.weight"), self._w(i, "Conv_1.bias"), self.operand_round)
        if ci != co or s["up"] or s["down"]:
            x = conv2d(x, self._w(i, "Conv_2.weight"), self._w(i, "Conv_2.bias"), self.operand_round)
        return st(((x + h) / np.sqrt(2.0)).astype(self.dtype))

    # -- forward ------------------------------------------------------------------------
    def forward(self, x: np.ndarray, y: np.ndarray, t: np.ndarray, tap: bool = False) -> np.ndarray:
        """x, y: [B,1,F,T] complex; t: [1] or [B].  Returns [B,1,F,T] complex64."""
        dt = self.dtype
        h = np.concatenate([x.real, x.imag, y.real, y.imag], axis=1).astype(dt)  # :401-404
        temb = self.time_embedding(t)
        specs = self.specs
        R = len(self.cfg["ch_mult"])
        nrb = self.cfg["num_res_blocks"]
        m = 3
        input_pyramid = h
        st = self._st
        hs = [st(conv2d(h, self._w(m, "weight"), self._w(m, "bias"), self.operand_round))]
        m += 1
        taps = {}
        for lvl in range(R):
            for _ in range(nrb):
                h = self.resblock(m, specs[m], hs[-1], temb); m += 1
                hs.append(h)
            if lvl != R - 1:
                h = self.resblock(m, specs[m], hs[-1], temb); m += 1
                input_pyramid = st(downsample_2d(input_pyramid))
                h = st(conv2d(input_pyramid, self._w(m, "Conv_0.weight"), self._w(m, "Conv_0.bias"), self.operand_round) + h)
                m += 1
                hs.append(h)
        h = hs[-1]
        h = self.resblock(m, specs[m], h, temb); m += 1
        h = self.resblock(m, specs[m], h, temb); m += 1
        if tap:
            taps["mid"] = h
        pyramid = None
        for lvl in reversed(range(R)):
            for _ in range(nrb + 1):
                h = self.resblock(m, specs[m], np.concatenate([h, hs.pop()], axis=1), temb); m += 1
            c = specs[m]["c"]
            ph = silu(group_norm(h, gn_groups(c), self._w(m, "weight"), self._w(m, "bias"))); m += 1
            ph = conv2d(ph, self._w(m, "weight"), self._w(m, "bias"), self.operand_round); m += 1
            pyramid = st(ph if pyramid is None else st(upsample_2d(pyramid)) + ph)
            if lvl != 0:
                h = self.resblock(m, specs[m], h, temb); m += 1
        assert not hs and m == len(specs)
        if tap:
            taps["pyramid"] = pyramid
            self.taps = taps
        out = conv2d(pyramid, self.p["output_layer.weight"], None, None)  # 1x1 4->2, no bias, kept fp32
        return (out[:, 0:1] + 1j * out[:, 1:2]).astype(np.complex64)  # :407-411


# --------------------------------------------------------------------------------------
# (a6) fixed-step ODE driver -- torchdyn 1.0.6 `_fixed_odeint` semantics restated
# (third-party, not in /root/reference: PARITY UNPINNED); solvers:
# euler / midpoint = textbook; heun2 / heun2_eulerlast = flowdec/sampling/solvers.py:15-57
# --------------------------------------------------------------------------------------
def t_span_linspace(N: int) -> np.ndarray:
    """torch.linspace(0, 1, N+1) in float32 (model.py:513).  ATen computes step in float32
    and then start + step*i for the first half, end - step*(N-i) for the second half, each
    with a fused multiply-add (single rounding) -- emulated here via an exact float64 product."""
    steps = N + 1
    step = np.float64(np.float32(1.0) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        out[i] = np.float32(step * i) if i < half else np.float32(1.0 - step * (steps - 1 - i))
    return out


def solver_nfe(solver: str, N: int) -> int:
    if solver == "euler":
        return N
    if solver in ("midpoint", "heun2"):
        return 2 * N
    if solver == "heun2_eulerlast":
        return 2 * N - 1
    raise ValueError(solver)


def odeint_fixed(f: Callable[[np.float32, np.ndarray], np.ndarray], x: np.ndarray, t_span: np.ndarray,
                 solver: str = "euler", return_traj: bool = False):
    """t = t_span[0]; dt = t_span[1]-t_span[0]; repeat: x = step(f,x,t,dt); t = t+dt;
    dt = t_span[i+1] - t   (all float32)."""
    t_span = np.asarray(t_span, dtype=np.float32)
    t = t_span[0]
    dt = np.float32(t_span[1] - t)
    traj = [x]
    for i in range(1, len(t_span)):
        cdt = np.complex64(dt)
        if solver == "euler":
            x = x + cdt * f(t, x)
        elif solver == "midpoint":
            half = np.float32(0.5) * dt
            xm = x + np.complex64(half) * f(t, x)
            x = x + cdt * f(np.float32(t + half), xm)
        elif solver == "heun2":
            k1 = f(t, x)
            k2 = f(np.float32(t + dt), x + cdt * k1)
            x = x + np.complex64(dt * np.float32(0.5)) * (k1 + k2)
        elif solver == "heun2_eulerlast":
            k1 = f(t, x)
            xp = x + cdt * k1
            # torch.isclose(t+dt, 1): |a-1| <= 1e-8 + 1e-5*1
            if abs(np.float32(t + dt) - np.float32(1.0)) <= 1e-8 + 1e-5:
                x = xp
            else:
                k2 = f(np.float32(t + dt), xp)
                x = x + np.complex64(dt * np.float32(0.5)) * (k1 + k2)
        else:
            raise ValueError(f"unknown solver {solver}")
        x = x.astype(np.complex64)
        traj.append(x)
        t = np.float32(t + dt)
        if i < len(t_span) - 1:
            dt = np.float32(t_span[i + 1] - t)
    return traj if return_traj else x


# --------------------------------------------------------------------------------------
# (a20) FlowModel.enhance -- flowdec/model.py:476-528 (+ _preprocess :129-163,
# _postprocess :165-190, _get_noise :530-536)
# --------------------------------------------------------------------------------------
def preprocess(y: np.ndarray, alpha=ALPHA, beta=BETA):
    """y [B,1,L] float32 -> (Y [B,1,768,T_pad] c64, info)."""
    yn, normfac = normalize_noisy(y.astype(np.float32))
    Y = compress(stft(yn), alpha, beta)
    Y, T = pad_spec(Y)
    return Y, dict(orig_length=y.shape[-1], normfac=normfac, T=T)


def postprocess(X: np.ndarray, info: dict, alpha=ALPHA, beta=BETA) -> np.ndarray:
    X = X[..., :info["T"]]
    x = istft(decompress(X, alpha, beta), info["orig_length"])
    return (x * info["normfac"]).astype(np.float32)


def initial_state(Y: np.ndarray, sigma_y, noise: np.ndarray, sigma_fac: float = 1.0) -> np.ndarray:
    """Y + sigma_fac * (sigma_y * noise).type(complex64)  (model.py:512, :530-536).
    sigma_y: python float or (768,1) float64; noise: complex64 standard normal."""
    sig = np.asarray(sigma_y, dtype=np.float64)
    n = (sig * noise.astype(np.complex128)).astype(np.complex64)
    return (Y + np.complex64(sigma_fac) * n).astype(np.complex64)


def enhance(net: NCSNppOracle, y: np.ndarray, noise: np.ndarray, sigma_y, N: int = 50, solver: str = "euler",
            sigma_fac: float = 1.0, alpha=ALPHA, beta=BETA, return_traj: bool = False):
    """y [B,1,L]; noise [B,1,768,T_pad] complex64 (the reference draws it from the device
    RNG -- parity tests inject it)."""
    Y, info = preprocess(y, alpha, beta)
    x0 = initial_state(Y, sigma_y, noise, sigma_fac)
    f = lambda t, X: net.forward(X, Y, np.asarray([t], dtype=np.float32))
    res = odeint_fixed(f, x0, t_span_linspace(N), solver, return_traj=return_traj)
    if return_traj:
        return res, [postprocess(X, info, alpha, beta) for X in res]
    return postprocess(res, info, alpha, beta)


# --------------------------------------------------------------------------------------
# (f3) ScoreDec baseline: OUVE SDE + predictor-corrector sampler, and the regression baseline
# (flowdec/sdes.py:132-206, sampling/__init__.py:32-72, sampling/predictors.py:48-71,
#  sampling/correctors.py:42-66, model.py:566-578 (RegressionModel.enhance), :613-657
#  (ScoreModel.forward / enhance)).  Pinned by tests/golden/g13_score_nf8.npz, produced by
#  running the reference classes with torch.randn_like replaced by a seeded NumPy stream.
# --------------------------------------------------------------------------------------
def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """torch.linspace(start, end, steps) in float32: step = (end-start)/(steps-1) in float32, first half
    start + step*i, second half end - step*(steps-1-i), each one fused multiply-add."""
    s, e = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([s], dtype=np.float32)
    step = np.float64(np.float32((e - s) / np.float32(steps - 1)))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        out[i] = np.float32(np.float64(s) + step * i) if i < half else np.float32(np.float64(e) - step * (steps - 1 - i))
    return out


class OUVE:
    """sdes.py:132-206.  Scalar (per-call) arithmetic is float32 like the reference's [B] tensors."""

    def __init__(self, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30):
        self.theta, self.sigma_min, self.sigma_max, self.N = theta, sigma_min, sigma_max, N
        self.logsig = np.log(sigma_max / sigma_min)

    def std(self, t) -> np.float32:                                        # sdes.py:181-192
        t = np.float32(t)
        th, ls = self.theta, self.logsig
        num = (np.float32(self.sigma_min ** 2) * np.exp(np.float32(-2 * th) * t)
               * (np.exp(np.float32(2 * (th + ls)) * t) - np.float32(1)) * np.float32(ls))
        return np.float32(np.sqrt(np.float32(num / np.float32(th + ls))))

    def diffusion(self, t) -> np.float32:                                  # sdes.py:168-172
        sigma = np.float32(self.sigma_min) * np.float32(np.float32(self.sigma_max / self.sigma_min) ** np.float32(t))
        return np.float32(sigma * np.float32(np.sqrt(2 * self.logsig)))


def score_pc_sample(net: NCSNppOracle, Y: np.ndarray, noises, sde: OUVE, N: int, predictor="reverse_diffusion", corrector="ald",
                    corrector_steps=1, snr=0.5, t_eps=3e-2, denoise=True):
    """sampling/__init__.py:57-70.  `noises` = iterator of complex standard-normal arrays of Y's shape, consumed in the
    reference's draw order: prior, then per step [corrector noise x n_steps], predictor noise.  -> (X, nfe)"""
    noises = iter(noises)
    c64 = np.complex64
    score = lambda x, t: (-net.forward(x, Y, np.asarray([t], dtype=np.float32)) / sde.std(t)).astype(c64)   # model.py:613-628
    x = (Y + next(noises) * sde.std(1.0)).astype(c64)                                                         # sdes.py:197-202
    x_mean = x
    ts = linspace_f32(1.0, t_eps, N)
    nfe = 0
    for i in range(N):
        t = ts[i]
        if corrector == "ald":                                                                                # correctors.py:52-66
            std = sde.std(t)
            for _ in range(corrector_steps):
                grad = score(x, t); nfe += 1
                z = next(noises)
                step = np.float32(np.float32(np.float32(snr) * std) ** 2 * np.float32(2))
                x_mean = (x + step * grad).astype(c64)
                x = (x_mean + z * np.float32(np.sqrt(np.float32(step * np.float32(2))))).astype(c64)
        elif corrector != "none":
            raise ValueError(corrector)
        if predictor == "reverse_diffusion":                                                                  # predictors.py:61-71, sdes.py:62-77,111-116
            dt = 1.0 / N
            f = (np.float32(sde.theta) * (Y - x) * np.float32(dt)).astype(c64)
            G = np.float32(sde.diffusion(t) * np.float32(np.sqrt(np.float32(dt))))
            rev_f = (f - np.float32(G ** 2) * score(x, t)).astype(c64); nfe += 1
            z = next(noises)
            x_mean = (x - rev_f).astype(c64)
            x = (x_mean + G * z).astype(c64)
        elif predictor == "euler_maruyama":                                                                   # predictors.py:48-58, sdes.py:93-109
            dt = -1.0 / N
            z = next(noises)
            g = sde.diffusion(t)
            drift = (np.float32(sde.theta) * (Y - x) - np.float32(g ** 2) * score(x, t)).astype(c64); nfe += 1
            x_mean = (x + drift * np.float32(dt)).astype(c64)
            x = (x_mean + np.float32(g * np.float32(np.sqrt(-dt))) * z).astype(c64)
        elif predictor != "none":
            raise ValueError(predictor)
    if denoise == "both":
        return (x_mean, x), nfe
    return (x_mean if denoise else x), nfe


def score_noise_count(N: int, predictor="reverse_diffusion", corrector="ald", corrector_steps=1) -> int:
    return 1 + N * ((corrector_steps if corrector == "ald" else 0) + (1 if predictor != "none" else 0))


def score_enhance(net: NCSNppOracle, y: np.ndarray, noises, sde: OUVE, N=30, alpha=ALPHA, beta=BETA, **kw) -> np.ndarray:
    Y, info = preprocess(y, alpha, beta)
    X, _ = score_pc_sample(net, Y, noises, sde, N, **kw)
    return postprocess(X, info, alpha, beta)


def regression_enhance(net: NCSNppOracle, y: np.ndarray, alpha=ALPHA, beta=BETA) -> np.ndarray:
    """model.py:566-578: X_hat = backbone(Y, Y, t = 0)."""
    Y, info = preprocess(y, alpha, beta)
    return postprocess(net.forward(Y, Y, np.zeros(1, dtype=np.float32)), info, alpha, beta)


def seeded_noises(seed: int, shape):
    """The shared noise stream of the score-sampler fixtures: complex standard normal (var 1/2 per part)."""
    rng = np.random.default_rng(seed)
    while True:
        yield ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)


def score_ode_sample(net: NCSNppOracle, Y: np.ndarray, prior_noise: np.ndarray, sde: OUVE, N: int, t_eps=3e-2, rtol=1e-5, atol=1e-5,
                     method="RK45", denoise=True):
    """sampling/__init__.py:75-146: probability-flow ODE integrated by scipy.integrate.solve_ivp on the flattened complex64
    state from t = 1 to t_eps, then one noise-free reverse-diffusion predictor step (dt = 1/N).  -> (X, nfe)"""
    from scipy import integrate
    c64 = np.complex64
    score = lambda x, t: (-net.forward(x, Y, np.asarray([t], dtype=np.float32)) / sde.std(t)).astype(c64)
    x0 = (Y + prior_noise * sde.std(1.0)).astype(c64)

    def ode_func(t, xf):
        x = xf.reshape(Y.shape).astype(c64)
        t32 = np.float32(t)
        g = sde.diffusion(t32)
        drift = (np.float32(sde.theta) * (Y - x) - np.float32(g ** 2) * score(x, t32) * np.float32(0.5)).astype(c64)
        return drift.reshape(-1)

    sol = integrate.solve_ivp(ode_func, (1.0, t_eps), x0.reshape(-1), rtol=rtol, atol=atol, method=method)
    x = sol.y[:, -1].reshape(Y.shape).astype(c64)
    if denoise:                                                             # predictors.py:61-71 at t = eps, no noise
        t = np.float32(t_eps)
        dt = 1.0 / N
        f = (np.float32(sde.theta) * (Y - x) * np.float32(dt)).astype(c64)
        G = np.float32(sde.diffusion(t) * np.float32(np.sqrt(np.float32(dt))))
        x = (x - (f - np.float32(G ** 2) * score(x, t))).astype(c64)
    return x, int(sol.nfev)


def score_ode_enhance(net: NCSNppOracle, y: np.ndarray, prior_noise: np.ndarray, sde: OUVE, N=30, alpha=ALPHA, beta=BETA, **kw):
    Y, info = preprocess(y, alpha, beta)
    X, nfe = score_ode_sample(net, Y, prior_noise, sde, N, **kw)
    return postprocess(X, info, alpha, beta), nfe


# --------------------------------------------------------------------------------------
# (f4) adaptive Dormand-Prince 5(4) driver -- torchdyn 1.0.6 `odeint(..., solver='dopri5')` semantics RESTATED
# (third-party: requirements.txt:53 `torchdyn==1.0.6`, not in /root/reference and not installable offline: PARITY UNPINNED).
# Call site: flowdec/model.py:511-514 `NeuralODE(node_fn, solver=solver, sensitivity="adjoint").trajectory(x0, t_span)`.
# Every constant below, with the place in the torchdyn 1.0.6 source tree it restates, so that a reader WITH the package can
# check it line by line (paths relative to site-packages/torchdyn/):
#   tableau c, A, b5, b4         numerics/solvers/_constants.py `construct_dopri5` (and `construct_tsit5` for solver='tsit5', below) -- the classical Dormand-Prince 5(4) pair, 7 stages,
#                                FSAL (stage 7 is evaluated at the 5th-order solution and becomes k1 of the next step)
#   order = 5                    numerics/solvers/ode.py `DormandPrince45.__init__` (super().__init__(order=5, stepping_class='adaptive'))
#   safety = 0.9, min_factor = 0.2, max_factor = 10        same class: `self.safety, self.min_factor, self.max_factor`
#   error norm                   numerics/utils.py `hairer_norm`: tensor.abs().pow(2).mean().sqrt() over ALL elements of the state
#                                (for a complex state: mean of |z|^2 over complex elements)
#   scaled error                 numerics/odeint.py `_adaptive_odeint`: x_err / (atol + rtol * max(|x|, |x_new|)), accept iff norm <= 1
#   step update                  numerics/utils.py `adapt_step`: ratio == 0 -> dt * max_factor; ratio < 1 -> min_factor := 1;
#                                dt * min(max_factor, max(safety / ratio^(1/order), min_factor))            [exponent 1/5, not 1/(order+1)]
#   initial step                 numerics/utils.py `init_step` (Hairer II.4): scale = atol + |x0| rtol; d0 = ||x0/scale||, d1 = ||f0/scale||;
#                                h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 d0/d1; one extra evaluation f(t0 + h0, x0 + h0 f0);
#                                d2 = ||(f1 - f0)/scale|| / h0; h1 = max(1e-6, 1e-3 h0) if d1, d2 <= 1e-15 else (0.01 / max(d1, d2))^(1/(order+1));
#                                dt = min(100 h0, h1)                                                       [exponent 1/6 here]
#   checkpoints                  `_adaptive_odeint` without interpolator: when t + dt would pass t_span[i] the step is shortened to land
#                                on it exactly ("save old dt, raise checkpoint flag"), and dt_old - dt is restored before `adapt_step`
#   tolerances                   `NeuralODE.__init__` (core/neuralde.py) forwards its own atol / rtol.  SURVEY section 8(c) records them as
#                                1e-4; the builder's and the round-3 advisor's recollection of 1.0.6 is atol = rtol = 1e-3 for NeuralODE
#                                (1e-4 being `ODEProblem`'s and the adjoint's).  Neither can be checked offline: `enhance(..., solver=
#                                'dopri5', atol=, rtol=)` takes them as arguments, default flowdec_amd.model.ADAPTIVE_DEFAULT_TOL = 1e-3 since
#                                round 4 (1e-4 before); scripts/pin_third_party.py compares it with the installed package;
#                                profiles/r03_bench_cfg5_dopri5*.json has both settings.
#   t, dt                        float32 tensors (t_span = torch.linspace(0, 1, N + 1), model.py:513); the state keeps its dtype
# What IS checked offline: the tableau / controller integrate test problems to tolerance and agree with scipy's RK45
# (tests/test_oracle_golden.py::test_dopri5_driver_against_scipy); the HIP driver follows this restatement (fixture g16).
# --------------------------------------------------------------------------------------
DOPRI5_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
DOPRI5_A = ((), (1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
            (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
DOPRI5_B5 = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
DOPRI5_B4 = (5179 / 57600, 0.0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40)
DOPRI5_E = tuple(b5 - b4 for b5, b4 in zip(DOPRI5_B5, DOPRI5_B4))


def hairer_norm(a: np.ndarray) -> float:
    return float(np.sqrt(np.mean(np.abs(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)) ** 2)))


def dopri5_adapt_step(dt, ratio, safety=0.9, min_factor=0.2, max_factor=10.0, order=5):
    if ratio == 0:
        return np.float32(dt * max_factor)
    if ratio < 1:
        min_factor = 1.0
    return np.float32(dt * min(max_factor, max(safety / ratio ** (1.0 / order), min_factor)))


# Tsitouras 5(4) ('tsit5', torchdyn's NeuralODE DEFAULT solver; numerics/solvers/_constants.py `construct_tsit5`, numerics/solvers/ode.py
# `Tsitouras45`: order 5, the same safety / min_factor / max_factor, FSAL).  Coefficients as published (Ch. Tsitouras, "Runge-Kutta pairs of
# order 5(4) satisfying only the first column simplifying assumption", 2011) -- typed from memory and VERIFIED: every order condition up
# to order 5 holds to 1e-16 and the pair converges with order 5.4 / its estimator with order 5 (tests/test_oracle_golden.py).
TSIT5_C = (0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0)
TSIT5_A = ((), (0.161,), (-0.008480655492356989, 0.335480655492357), (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
           (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
           (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383),
           (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774))
TSIT5_E = (-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552, -0.45808210592918697,
           0.015151515151515152)
ADAPTIVE_TABLEAUS = {"dopri5": (DOPRI5_C, DOPRI5_A, DOPRI5_E), "tsit5": (TSIT5_C, TSIT5_A, TSIT5_E)}


def odeint_adaptive(f: Callable[[np.float32, np.ndarray], np.ndarray], x: np.ndarray, t_span: np.ndarray, method: str = "dopri5", atol=1e-4,
                    rtol=1e-4, return_traj: bool = False, max_steps: int = 100000):
    """-> (x(T) or [x(t_span[i])], nfe).  State dtype is kept (complex64 / float32 / float64); t, dt are float32.  7-stage FSAL pairs of
    order 5(4): 'dopri5' | 'tsit5' (the last row of A are the 5th-order weights; E the error weights)."""
    C_, A_, E_ = ADAPTIVE_TABLEAUS[method]
    dt_ = x.dtype
    t, T = np.float32(t_span[0]), np.float32(t_span[-1])
    t_eval = [np.float32(v) for v in t_span[1:]]
    k1 = f(t, x).astype(dt_); nfe = 1
    # Hairer's initial step
    scale = atol + np.abs(x) * rtol
    d0, d1 = hairer_norm(x / scale), hairer_norm(k1 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f_new = f(np.float32(t + np.float32(h0)), (x + np.float32(h0) * k1).astype(dt_)).astype(dt_); nfe += 1
    d2 = hairer_norm((f_new - k1) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 6.0)
    dt = np.float32(min(100 * h0, h1))
    traj = [x]
    ckpt, steps = 0, 0
    while t < T:
        steps += 1
        if steps > max_steps:
            raise RuntimeError("odeint_adaptive: step limit")
        if np.float32(t + dt) > T:
            dt = np.float32(T - t)
        flag = False
        if ckpt < len(t_eval) and np.float32(t + dt) > t_eval[ckpt]:
            dt_old, flag = dt, True
            dt = np.float32(t_eval[ckpt] - t)
        ks = [k1]
        for s in range(1, 7):
            xs = x
            acc = np.zeros_like(x)
            for j, a in enumerate(A_[s]):
                if a != 0.0:
                    acc = acc + np.float32(a) * ks[j]
            xs = (x + dt * acc).astype(dt_)
            if s == 6:
                x_new = xs
            ks.append(f(np.float32(t + np.float32(C_[s]) * dt), xs).astype(dt_)); nfe += 1
        err = np.zeros_like(x)
        for j, e in enumerate(E_):
            if e != 0.0:
                err = err + np.float32(e) * ks[j]
        err = (dt * err).astype(dt_)
        ratio = hairer_norm(err / (atol + rtol * np.maximum(np.abs(x), np.abs(x_new))))
        if ratio <= 1:
            t = np.float32(t + dt)
            k1, x = ks[6], x_new
            if ckpt < len(t_eval) and abs(float(t) - float(t_eval[ckpt])) <= 1e-7:
                t = t_eval[ckpt]
                traj.append(x); ckpt += 1
        if flag:
            dt = np.float32(dt_old - dt)
        dt = dopri5_adapt_step(dt, ratio)
    return (traj if return_traj else x), nfe


def odeint_dopri5(f, x, t_span, atol=1e-4, rtol=1e-4, return_traj: bool = False, max_steps: int = 100000):
    return odeint_adaptive(f, x, t_span, "dopri5", atol, rtol, return_traj, max_steps)
