"""Thin torch-tensor wrappers over the op-level C ABI (include/flowdec_hip.h).

Tensors are NHWC ([B, H, W, C]) float32 or bfloat16 on the GPU unless stated otherwise; every
wrapper launches on torch's current stream and allocates only its own output.
"""
import math

import torch

from . import _lib as L


def _nhwc(x):
    L.require_cuda(x)
    if x.ndim != 4 or not x.is_contiguous():
        raise RuntimeError("expected a contiguous NHWC tensor [B, H, W, C]")
    return x.shape


def to_nhwc(x_nchw, dtype=None):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(dtype or x_nchw.dtype)


def to_nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def channel_sums(x):
    """Stand-alone statistics pass -> partial sums [B, tiles, C, 2] float32."""
    B, H, W, Cc = _nhwc(x)
    tiles = L.load().fd_channel_sums_tiles(H, W)
    out = torch.empty(B, tiles, Cc, 2, dtype=torch.float32, device=x.device)
    L.check(L.load().fd_channel_sums(L.ptr(x), L.ptr(out), B, H, W, Cc, L.dtype_id(x.dtype), L.stream()))
    return out


def gn_finalize(part0, C0, part1, C1, gamma, beta, groups, hw, eps=1e-6):
    """part: [B, tiles, stride, 2] partial sums (stride >= C) of one or two tensors -> affine [B, C0+C1, 2]."""
    B = part0.shape[0]
    out = torch.empty(B, C0 + C1, 2, dtype=torch.float32, device=part0.device)
    t1, s1 = (0, 0) if part1 is None else (part1.shape[1], part1.shape[2])
    L.check(L.load().fd_gn_finalize(L.ptr(part0), part0.shape[1], part0.shape[2], C0, L.ptr(part1), t1, s1, C1, L.ptr(gamma),
                                    L.ptr(beta), L.ptr(out), B, groups, hw, eps, L.stream()))
    return out


def gn_affine(x0, x1, gamma, beta, eps=1e-6):
    """GroupNorm(min(C//4, 32), C) statistics of the virtual concat [x0 | x1] as per-(b, c) affine pairs."""
    C0, C1 = x0.shape[3], (0 if x1 is None else x1.shape[3])
    s0 = channel_sums(x0)
    s1 = None if x1 is None else channel_sums(x1)
    return gn_finalize(s0, C0, s1, C1, gamma, beta, min((C0 + C1) // 4, 32), x0.shape[1] * x0.shape[2], eps)


def fir_resample(x, direction, affine=None, want_raw=True):
    """direction +1 / -1; returns (raw, act) -- act is None without `affine`."""
    B, H, W, Cc = _nhwc(x)
    oh, ow = (2 * H, 2 * W) if direction > 0 else (H // 2, W // 2)
    raw = torch.empty(B, oh, ow, Cc, dtype=x.dtype, device=x.device) if want_raw else None
    act = torch.empty(B, oh, ow, Cc, dtype=x.dtype, device=x.device) if affine is not None else None
    L.check(L.load().fd_fir_resample(L.ptr(x), L.ptr(affine), L.ptr(raw), L.ptr(act), B, H, W, Cc, direction,
                                     L.dtype_id(x.dtype), L.stream()))
    return raw, act


def conv_in(in8, w, bias):
    """The input convolution conv3x3(4, Cout) on the packed input [B, H, W, 8]; returns (out [B, H, W, Cout], stats partials
    [B, tiles, Cout, 2])."""
    B, H, W, c8 = _nhwc(in8)
    assert c8 == 8 and w.shape[1:] == (4, 3, 3) and w.dtype == torch.float32
    Cout = w.shape[0]
    out = torch.empty(B, H, W, Cout, dtype=in8.dtype, device=in8.device)
    stats = torch.zeros(B, (H // 16) * (W // 16), Cout, 2, dtype=torch.float32, device=in8.device)
    L.check(L.load().fd_conv_in(L.ptr(in8), L.ptr(w.contiguous()), L.ptr(bias), L.ptr(out), L.ptr(stats), B, H, W, Cout, L.dtype_id(in8.dtype),
                                L.stream()))
    return out, stats


_WINO = {False: 0, 0: 0, True: L.FD_WINOGRAD, 4: L.FD_WINOGRAD4, 44: L.FD_WINOGRAD44}   # (True == 1: the F(2,3) kernel; 44: 2-D F(4x4,3x3), float32 only)


def pack_conv_weight(w, C0=None, dtype=torch.bfloat16, w_sc=None, S0=None, winograd=False, bf16_operands=False):
    """w: [Cout, Cin, k, k] float32 (GPU); C0 = channels of the first concat segment (default: all).
    w_sc: optional 1x1 shortcut weight [Cout, S, 1, 1] folded behind the main K loop (S0 = first segment).
    winograd: True = pack for the F(2,3) kernel (FD_WINOGRAD), 4 = for the F(4,3) kernel (FD_WINOGRAD4; bf16 or float32 storage), 44 = for the
    2-D F(4x4, 3x3) kernel (FD_WINOGRAD44; float32 storage only); pass the same value to conv2d.
    bf16_operands (with dtype=float32): True = FD_BF16_OPERANDS (f32 activations, bf16 weights / MFMA operands), "x3" =
    FD_BF16X3_OPERANDS (two-term bf16 split, three MFMAs per product); pass the same value to conv2d."""
    L.require_cuda(w, w_sc)
    w = w.contiguous().float()
    Cout, Cin, k, _ = w.shape
    C0 = Cin if C0 is None else C0
    S = 0 if w_sc is None else w_sc.shape[1]
    S0 = S if S0 is None else S0
    if w_sc is not None:
        w_sc = w_sc.contiguous().float()
    dt = L.dtype_id(dtype) | _WINO[winograd] | {False: 0, True: L.FD_BF16_OPERANDS, "x3": L.FD_BF16X3_OPERANDS}[bf16_operands]
    nbytes = L.load().fd_conv_packed_bytes(Cout, C0, Cin - C0, k, S0, S - S0, dt)
    if nbytes <= 0:
        raise RuntimeError("flowdec_hip: this convolution shape has no %s packing" % ("Winograd" if winograd else "MFMA"))
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    L.check(L.load().fd_conv_pack_weights(L.ptr(w), L.ptr(w_sc), L.ptr(packed), Cout, C0, Cin - C0, k, S0, S - S0, dt, L.stream()))
    return packed


def conv2d(x0, packed_w, Cout, ksize, x1=None, affine=None, bias=None, skip=None, scale=1.0, sc0=None, sc1=None, want_stats=False,
           winograd=False, tile_bn=0, bf16_operands=False, reversed_tiles=False):
    """Returns out, or (out, stats partials [B, tiles, CoutPad, 2]) with want_stats."""
    B, H, W, C0 = _nhwc(x0)
    lib = L.load()
    C1 = 0 if x1 is None else x1.shape[3]
    S0 = 0 if sc0 is None else sc0.shape[3]
    S1 = 0 if sc1 is None else sc1.shape[3]
    out = torch.empty(B, H, W, Cout, dtype=x0.dtype, device=x0.device)
    stats = torch.zeros(B, lib.fd_conv_stats_tiles(H, W), lib.fd_conv_cout_pad(Cout), 2, dtype=torch.float32, device=x0.device) if want_stats else None
    rows = 0 if bias is None else (1 if bias.ndim == 1 else bias.shape[0])
    L.check(lib.fd_conv2d(L.ptr(x0), C0, L.ptr(x1), C1, L.ptr(affine), L.ptr(sc0), S0, L.ptr(sc1), S1, L.ptr(packed_w), L.ptr(bias), rows,
                          L.ptr(skip), float(scale), L.ptr(out), Cout, L.ptr(stats), B, H, W, ksize,
                          L.dtype_id(x0.dtype) | _WINO[winograd] | (L.FD_TILE_REVERSED if reversed_tiles else 0) | L.FD_TILE[tile_bn] | {False: 0, True: L.FD_BF16_OPERANDS, "x3": L.FD_BF16X3_OPERANDS}[bf16_operands],
                          L.stream()))
    return (out, stats) if want_stats else out


def time_embedding(t, gfp_w, w1, b1, w2, b2):
    nt, nf = t.numel(), gfp_w.numel()
    out = torch.empty(nt, 4 * nf, dtype=torch.float32, device=t.device)
    L.check(L.load().fd_time_embedding(L.ptr(t), nt, L.ptr(gfp_w), nf, L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(out),
                                       L.stream()))
    return out


def temb_bias(temb, dense_w, dense_b, conv_bias):
    nt, D = temb.shape
    Cout = dense_w.shape[0]
    out = torch.empty(nt, Cout, dtype=torch.float32, device=temb.device)
    L.check(L.load().fd_temb_bias(L.ptr(temb), nt, D, L.ptr(dense_w), L.ptr(dense_b), L.ptr(conv_bias), Cout, L.ptr(out), L.stream()))
    return out


class _StftPlan:
    """Caller-owned fd_stft_plan (device-resident DFT matrices of one (n_fft, hop)); freed with the object."""

    def __init__(self, n_fft, hop):
        import ctypes as C
        self.handle = C.c_void_p()
        L.check(L.load().fd_stft_plan_create(int(n_fft), int(hop), C.byref(self.handle)))

    def __del__(self):
        try:
            L.load().fd_stft_plan_destroy(self.handle)
        except Exception:
            pass


_PLANS = {}


def stft_plan(n_fft, hop, device):
    """One plan per (device, n_fft, hop), created on first use (allocates and uploads: call it outside graph capture)."""
    key = (torch.device(device).index, int(n_fft), int(hop))
    if key not in _PLANS:
        with torch.cuda.device(device):
            _PLANS[key] = _StftPlan(n_fft, hop)
    return _PLANS[key].handle


def check_ragged_lengths(lengths, Ls, n_fft, hop):
    """A ragged batch is one T_pad bucket: every clip longer than the reflect padding, none longer than the row, and all of them
    padding to the frame count of the row length (util/other.py:25-52).  -> (host list, T_pad).  The native kernels trust this."""
    lib = L.load()
    ls = [int(v) for v in (lengths.tolist() if torch.is_tensor(lengths) else lengths)]
    Tp = lib.fd_padded_frames(lib.fd_num_frames(Ls, hop))
    for b, l in enumerate(ls):
        if not (n_fft // 2 < l <= Ls):
            raise RuntimeError(f"ragged batch: clip {b} has {l} samples, need {n_fft // 2} < length <= row length {Ls}")
        if lib.fd_padded_frames(lib.fd_num_frames(l, hop)) != Tp:
            raise RuntimeError(f"ragged batch: clip {b} ({l} samples) pads to {lib.fd_padded_frames(lib.fd_num_frames(l, hop))} frames, the "
                               f"row length {Ls} to {Tp}: one call takes ONE T_pad bucket")
    return ls, Tp


def stft_compress(y, n_fft=1534, hop=384, alpha=0.3, beta=0.33, normalize=True, lengths=None):
    """y [B, L] float32 -> (Y complex64 [B, 1, F, T_pad], normfac [B], T).  `lengths` (B ints): a ragged batch -- clip b is the first
    lengths[b] samples of its row and is transformed exactly as it would be alone (fd_stft_compress_ragged); T is then the frame
    count of the row length."""
    L.require_cuda(y)
    lib = L.load()
    B, Ls = y.shape
    T = lib.fd_num_frames(Ls, hop)
    Tp = lib.fd_padded_frames(T)
    Y = torch.empty(B, 1, n_fft // 2 + 1, Tp, dtype=torch.complex64, device=y.device)
    nf = torch.empty(B, dtype=torch.float32, device=y.device)
    nws = lib.fd_stft_workspace_bytes(B, Ls, n_fft, hop)
    ws = torch.empty(nws, dtype=torch.uint8, device=y.device)
    if lengths is not None:
        ls, _ = check_ragged_lengths(lengths, Ls, n_fft, hop)
        lens = torch.tensor(ls, dtype=torch.int32, device=y.device)
        L.check(lib.fd_stft_compress_ragged(stft_plan(n_fft, hop, y.device), L.ptr(y), L.ptr(lens), B, Ls, alpha, beta, int(normalize), L.ptr(nf),
                                            L.ptr(Y), Tp, L.ptr(ws), nws, L.stream()))
        lens.record_stream(torch.cuda.current_stream(y.device))
        return Y, nf, T
    L.check(lib.fd_stft_compress(stft_plan(n_fft, hop, y.device), L.ptr(y), B, Ls, alpha, beta, int(normalize), L.ptr(nf), L.ptr(Y), Tp, L.ptr(ws),
                                 nws, L.stream()))
    return Y, nf, T


def decompress_istft(X, T, length, normfac=None, n_fft=1534, hop=384, alpha=0.3, beta=0.33, lengths=None):
    """X complex64 [B, 1, F, T_pad] -> y [B, length] float32.  `lengths` (B ints): ragged batch -- clip b is synthesised from its own
    1 + lengths[b] // hop frames into the first lengths[b] samples of its row (the rest zero); T must be 1 + length // hop."""
    L.require_cuda(X)
    lib = L.load()
    B, Tp = X.shape[0], X.shape[-1]
    y = torch.empty(B, length, dtype=torch.float32, device=X.device)
    nws = lib.fd_stft_workspace_bytes(B, max(length, hop * T), n_fft, hop)
    ws = torch.empty(nws, dtype=torch.uint8, device=X.device)
    if lengths is not None:
        ls, _ = check_ragged_lengths(lengths, length, n_fft, hop)
        lens = torch.tensor(ls, dtype=torch.int32, device=X.device)
        L.check(lib.fd_decompress_istft_ragged(stft_plan(n_fft, hop, X.device), L.ptr(X), L.ptr(lens), B, T, Tp, alpha, beta, L.ptr(normfac), L.ptr(y),
                                               length, L.ptr(ws), nws, L.stream()))
        lens.record_stream(torch.cuda.current_stream(X.device))
        return y
    L.check(lib.fd_decompress_istft(stft_plan(n_fft, hop, X.device), L.ptr(X), B, T, Tp, alpha, beta, L.ptr(normfac), L.ptr(y), length, L.ptr(ws),
                                    nws, L.stream()))
    return y


def upfirdn2d_raw(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """x: [major, in_h, in_w, minor] (the reference binding's view, op/upfirdn2d.py:123)."""
    L.require_cuda(x, kernel)
    lib = L.load()
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    oh = lib.fd_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    ow = lib.fd_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    if oh <= 0 or ow <= 0:
        raise RuntimeError("upfirdn2d: empty output")
    out = torch.empty(major, oh, ow, minor, dtype=x.dtype, device=x.device)
    L.check(lib.fd_upfirdn2d(L.ptr(x.contiguous()), L.ptr(kernel.contiguous().float()), L.ptr(out), major, in_h, in_w, minor, kh, kw,
                             up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, L.dtype_id(x.dtype), L.stream()))
    return out


def fused_bias_act(x, bias, act=3, alpha=0.2, scale=math.sqrt(2.0)):
    """scale * lrelu(x + bias[c]) for NCHW x (fused_act.py:110-121)."""
    L.require_cuda(x)
    x = x.contiguous().float()
    out = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    size_b = 0 if bias is None else bias.numel()
    L.check(L.load().fd_fused_bias_act(L.ptr(x), L.ptr(None if bias is None else bias.contiguous().float()), L.ptr(out), x.numel(),
                                       step_b, size_b, act, alpha, scale, L.stream()))
    return out


def gn_silu_apply(x, affine):
    """silu(a * x + d) with per-(b, c) affine pairs [B, C, 2] on an NHWC tensor (stand-alone; fused on the hot path)."""
    B, H, W, Cc = _nhwc(x)
    L.require_cuda(affine)
    out = torch.empty_like(x)
    L.check(L.load().fd_gn_silu_apply(L.ptr(x), L.ptr(affine.contiguous()), L.ptr(out), B, H * W, Cc, L.dtype_id(x.dtype), L.stream()))
    return out


def resblock(x0, x1, params, temb, up=False, down=False):
    """One ResnetBlockBigGANpp (layerspp.py:252-284) through fd_resblock.  x0 / x1: NHWC (virtual concat); params: the
    module's state_dict entries (GroupNorm_0/1, Conv_0/1[/2], Dense_0) as float32 GPU tensors; temb [nt, temb_dim]."""
    import ctypes as C
    B, H, W, C0 = _nhwc(x0)
    C1 = 0 if x1 is None else x1.shape[3]
    Cout = params["Conv_0.weight"].shape[0]
    has_c2 = "Conv_2.weight" in params
    dt = x0.dtype
    w0 = pack_conv_weight(params["Conv_0.weight"], C0=C0, dtype=dt)
    w1 = pack_conv_weight(params["Conv_1.weight"], C0=Cout, dtype=dt, w_sc=params.get("Conv_2.weight"), S0=C0 if has_c2 else None)
    bias0 = temb_bias(temb.float().contiguous(), params["Dense_0.weight"], params["Dense_0.bias"], params["Conv_0.bias"])
    bias1 = (params["Conv_1.bias"] + (params["Conv_2.bias"] if has_c2 else 0)).float().contiguous()
    keep = [t.float().contiguous() for t in (params["GroupNorm_0.weight"], params["GroupNorm_0.bias"], params["GroupNorm_1.weight"], params["GroupNorm_1.bias"])]
    d = L.FdResblockDesc(C0, C1, Cout, int(up), int(down), int(has_c2), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                         w0.data_ptr(), bias0.data_ptr(), bias0.shape[0], w1.data_ptr(), bias1.data_ptr())
    lib = L.load()
    OH, OW = (2 * H, 2 * W) if up else ((H // 2, W // 2) if down else (H, W))
    out = torch.empty(B, OH, OW, Cout, dtype=dt, device=x0.device)
    need = lib.fd_resblock_workspace_bytes(C.byref(d), B, H, W, L.dtype_id(dt))
    if need == 0:
        raise RuntimeError("flowdec_hip: " + lib.fd_last_error().decode())
    ws = torch.empty(need, dtype=torch.uint8, device=x0.device)
    L.check(lib.fd_resblock(C.byref(d), L.ptr(x0), L.ptr(x1), L.ptr(out), B, H, W, L.dtype_id(dt), L.ptr(ws), need, L.stream()))
    return out
