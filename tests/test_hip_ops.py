"""GPU parity tests of the op-level C ABI (include/flowdec_hip.h) against the oracle and the golden
vectors.  Every test goes through libflowdec_hip.so via ctypes; nothing here runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err
from oracle import flowdec_oracle as O

pytestmark = pytest.mark.gpu

REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def report(name, err, tol):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(f"{name:60s} err={err:.3e} tol={tol:.1e} {'OK' if err < tol else 'FAIL'}\n")


def check(name, got, ref, tol):
    e = rel_err(got, ref)
    report(name, e, tol)
    assert e < tol, f"{name}: rel err {e:.3e} >= {tol:.1e}"


@pytest.fixture(scope="module")
def ops():
    from flowdec_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def nhwc(a_nchw, dtype):
    return dev(np.transpose(a_nchw, (0, 2, 3, 1)), dtype)


def from_nhwc(t):
    return np.transpose(t.float().cpu().numpy(), (0, 3, 1, 2))


DT = {"bf16": torch.bfloat16, "fp32": torch.float32}


def bf16r(a):
    return O.round_bf16(np.asarray(a, np.float32))


# ---------------------------------------------------------------------------------------------------------
# conv (MFMA implicit GEMM)
# ---------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name, B, H, W, C0, C1, Cout, k, affine, bias_rows, skip, S0, S1 (folded 1x1 shortcut conv operand)
    ("3x3_basic", 1, 16, 16, 32, 0, 128, 3, False, 0, False, 0, 0),
    ("3x3_two_ntiles", 2, 32, 16, 64, 0, 256, 3, False, 1, False, 0, 0),
    ("3x3_affine_bias_skip", 2, 16, 32, 64, 0, 128, 3, True, 2, True, 0, 0),
    ("3x3_concat", 1, 16, 16, 64, 32, 128, 3, True, 1, True, 0, 0),
    ("3x3_concat_pad", 2, 16, 16, 32, 16, 128, 3, True, 1, False, 0, 0),     # second segment padded to a chunk
    ("3x3_partial_tiles", 1, 24, 8, 32, 0, 128, 3, True, 1, True, 0, 0),     # W = 8 < tile, H not multiple of 16
    ("3x3_head_cout4", 2, 16, 16, 128, 0, 4, 3, True, 1, True, 0, 0),
    ("3x3_head_odd_chunks", 1, 24, 20, 96, 0, 4, 3, True, 1, True, 0, 0),     # 3 chunks: generic BN = 32 configuration
    ("3x3_head_concat_plain", 2, 20, 36, 64, 64, 4, 3, False, 1, False, 0, 0),  # dedicated head kernel, two segments, ragged tiles
    ("3x3_head_256", 2, 32, 16, 256, 0, 4, 3, True, 2, True, 0, 0),
    ("3x3_small_c8", 1, 16, 16, 8, 0, 8, 3, True, 1, False, 0, 0),
    ("3x3_cout16", 1, 32, 16, 16, 8, 16, 3, True, 1, True, 0, 0),
    ("1x1_basic", 2, 16, 16, 64, 0, 128, 1, False, 1, False, 0, 0),
    ("1x1_concat", 1, 16, 16, 128, 64, 256, 1, False, 1, False, 0, 0),
    ("1x1_small", 2, 8, 8, 8, 0, 32, 1, False, 1, False, 0, 0),
    ("3x3_deepk", 1, 16, 16, 256, 256, 256, 3, True, 1, True, 0, 0),
    ("3x3_shortcut", 2, 16, 16, 256, 0, 256, 3, True, 1, False, 64, 0),      # ResBlock 64->256: Conv_1(h) + Conv_2(x)
    ("3x3_shortcut_cat", 1, 32, 16, 128, 0, 128, 3, True, 1, False, 128, 256),  # up-path block: shortcut over cat(h, skip)
    ("3x3_shortcut_small", 1, 16, 16, 32, 0, 32, 3, True, 1, False, 32, 16),
    ("3x3_wide_image", 1, 16, 48, 32, 0, 256, 3, True, 1, True, 0, 0),
]


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(ops, case, prec):
    name, B, H, W, C0, C1, Cout, k, use_aff, bias_rows, use_skip, S0, S1 = case
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    dt = DT[prec]
    q = bf16r if prec == "bf16" else (lambda a: np.asarray(a, np.float32))
    Cin = C0 + C1
    x = q(rng.standard_normal((B, Cin, H, W)))
    w = q(rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k))
    aff = None
    xin = x
    if use_aff:
        a = (1 + 0.2 * rng.standard_normal((B, Cin))).astype(np.float32)
        d = (0.3 * rng.standard_normal((B, Cin))).astype(np.float32)
        aff = dev(np.stack([a, d], axis=-1))
        xin = O.silu(x * a[:, :, None, None] + d[:, :, None, None]).astype(np.float32)
        if prec == "bf16":
            xin = bf16r(xin)  # the kernel rounds the activated operand to bf16 before the MFMA
    bias = None
    ref = O.conv2d(xin.astype(np.float64), w.astype(np.float64), None)
    sc0 = sc1 = w_sc = None
    if S0:
        xs = q(rng.standard_normal((B, S0 + S1, H, W)))
        ws = q(rng.standard_normal((Cout, S0 + S1, 1, 1)) / np.sqrt(S0 + S1))
        ref = ref + O.conv2d(xs.astype(np.float64), ws.astype(np.float64), None)
        sc0 = nhwc(xs[:, :S0], dt)
        sc1 = nhwc(xs[:, S0:], dt) if S1 else None
        w_sc = dev(ws)
    if bias_rows:
        bv = rng.standard_normal((bias_rows, Cout)).astype(np.float32)
        bias = dev(bv if bias_rows > 1 else bv[0])
        ref = ref + (bv[:, :, None, None] if bias_rows > 1 else bv[0][None, :, None, None])
    skip = None
    scale = 1.0
    if use_skip:
        sk = q(rng.standard_normal((B, Cout, H, W)))
        skip = nhwc(sk, dt)
        ref = ref + sk
        scale = float(1 / np.sqrt(2))
    ref = ref * scale
    x0 = nhwc(x[:, :C0], dt)
    x1 = nhwc(x[:, C0:], dt) if C1 else None
    pw = ops.pack_conv_weight(dev(w), C0=C0, dtype=dt, w_sc=w_sc, S0=S0 if S0 else None)
    out, stats = ops.conv2d(x0, pw, Cout, k, x1=x1, affine=aff, bias=bias, skip=skip, scale=scale, sc0=sc0, sc1=sc1, want_stats=True)
    torch.cuda.synchronize()
    got = from_nhwc(out)
    # bf16: output rounding 2^-9 relative + transcendental differences of the fused SiLU; f32: accumulation order
    tol = 6e-3 if prec == "bf16" else 2e-5
    check(f"conv2d[{name},{prec}]", got, ref, tol)
    if prec == "bf16":
        # `ref` IS the kernel's numerics model (operand rounded to bf16 after the activation, exact products, wide accumulation): rounded
        # to bf16 like the stored output it must agree with the kernel up to a handful of last-place roundings (f32 summation order,
        # transcendental ulps of the fused SiLU) -- 15x tighter than the bound above
        check(f"conv2d_vs_numerics_model[{name}]", got, bf16r(ref.astype(np.float32)), 4e-4)
    # fused GroupNorm partial sums of the (un-rounded) output: reduce over tiles, compare with the reference sums
    st = stats.double().sum(dim=1).cpu().numpy()[:, :Cout]            # [B, Cout, 2]
    ref_s = np.stack([ref.sum(axis=(2, 3)), (ref ** 2).sum(axis=(2, 3))], axis=-1)
    tol_s = 3e-3 if prec == "bf16" else 2e-5
    e = float(np.abs(st - ref_s).max() / np.abs(ref_s).max())
    report(f"conv2d_stats[{name},{prec}]", e, tol_s)
    assert e < tol_s
    assert float(stats[:, :, Cout:].abs().max()) == 0.0 if stats.shape[2] > Cout else True


def test_conv2d_stats_feed_groupnorm(ops):
    """conv output partials -> fd_gn_finalize == GroupNorm statistics of the stored tensor."""
    rng = np.random.default_rng(11)
    B, H, W, Ci, Co = 2, 32, 16, 32, 256
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, 3, 3)) / np.sqrt(9 * Ci)).astype(np.float32)
    gam = (1 + 0.1 * rng.standard_normal(Co)).astype(np.float32); bet = (0.1 * rng.standard_normal(Co)).astype(np.float32)
    out, stats = ops.conv2d(nhwc(x, torch.float32), ops.pack_conv_weight(dev(w), dtype=torch.float32), Co, 3, want_stats=True)
    aff = ops.gn_finalize(stats, Co, None, 0, dev(gam), dev(bet), 32, H * W).cpu().numpy()
    o = from_nhwc(out)
    got = o * aff[:, :, 0][:, :, None, None] + aff[:, :, 1][:, :, None, None]
    ref = O.group_norm(o.astype(np.float64), 32, gam.astype(np.float64), bet.astype(np.float64))
    check("conv_stats_to_groupnorm", got, ref, 5e-6)


def test_conv2d_matches_torch_layout(ops):
    """Transpose-detecting check: asymmetric weights, output must equal F.conv2d exactly up to rounding."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 32, 16, 16)).astype(np.float32)
    w = rng.standard_normal((128, 32, 3, 3)).astype(np.float32) * 0.05
    ref = torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), padding=1).numpy()
    out = ops.conv2d(nhwc(x, torch.float32), ops.pack_conv_weight(dev(w), dtype=torch.float32), 128, 3)
    check("conv2d_vs_torch_f32", from_nhwc(out), ref, 2e-5)


def test_conv2d_errors(ops):
    x = torch.zeros(1, 16, 16, 12, device="cuda")  # 12 channels: not a multiple of 8
    pw = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv2d(x, pw, 128, 3)
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 16, 16, 32, device="cuda"), pw, 128, 5)


# ---------------------------------------------------------------------------------------------------------
# GroupNorm statistics
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("C0,C1", [(64, 0), (256, 0), (256, 64), (128, 256), (8, 0), (32, 16)])
def test_groupnorm_affine(ops, C0, C1, prec):
    rng = np.random.default_rng(C0 * 7 + C1)
    B, H, W = 2, 24, 16
    C = C0 + C1
    q = bf16r if prec == "bf16" else (lambda a: np.asarray(a, np.float32))
    x = q(rng.standard_normal((B, C, H, W)) * 1.5 + 0.7)
    gam = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(C)).astype(np.float32)
    x0 = nhwc(x[:, :C0], DT[prec]); x1 = nhwc(x[:, C0:], DT[prec]) if C1 else None
    aff = ops.gn_affine(x0, x1, dev(gam), dev(bet)).cpu().numpy()
    got = x * aff[:, :, 0][:, :, None, None] + aff[:, :, 1][:, :, None, None]
    ref = O.group_norm(x.astype(np.float64), O.gn_groups(C), gam.astype(np.float64), bet.astype(np.float64))
    check(f"groupnorm[{C0}+{C1},{prec}]", got, ref, 2e-6)


def test_groupnorm_golden(ops):
    g = load_golden("g5_groupnorm_silu.npz")
    for C in (64, 256):
        x = g[f"x{C}"]
        aff = ops.gn_affine(nhwc(x, torch.float32), None, dev(g[f"gamma{C}"]), dev(g[f"beta{C}"])).cpu().numpy()
        got = O.silu(x * aff[:, :, 0][:, :, None, None] + aff[:, :, 1][:, :, None, None])
        check(f"groupnorm_golden[{C}]", got, g[f"out{C}"], 5e-6)


# ---------------------------------------------------------------------------------------------------------
# FIR resampling / upfirdn2d
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("C", [4, 8, 64])
@pytest.mark.parametrize("direction", [1, -1])
def test_fir_resample(ops, direction, C, prec):
    rng = np.random.default_rng(C + direction)
    B, H, W = 2, 12, 8
    q = bf16r if prec == "bf16" else (lambda a: np.asarray(a, np.float32))
    x = q(rng.standard_normal((B, C, H, W)))
    a = (1 + 0.2 * rng.standard_normal((B, C))).astype(np.float32)
    d = (0.3 * rng.standard_normal((B, C))).astype(np.float32)
    raw, act = ops.fir_resample(nhwc(x, DT[prec]), direction, affine=dev(np.stack([a, d], -1)))
    f = O.upsample_2d if direction > 0 else O.downsample_2d
    ref_raw = f(x.astype(np.float64))
    ref_act = f(O.silu(x.astype(np.float64) * a[:, :, None, None] + d[:, :, None, None]))
    tol = 4e-3 if prec == "bf16" else 2e-6
    check(f"fir_raw[{direction},{C},{prec}]", from_nhwc(raw), ref_raw, tol)
    check(f"fir_act[{direction},{C},{prec}]", from_nhwc(act), ref_act, tol)
    raw2, act2 = ops.fir_resample(nhwc(x, DT[prec]), direction)
    assert act2 is None
    check(f"fir_raw_only[{direction},{C},{prec}]", from_nhwc(raw2), ref_raw, tol)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("direction", [1, -1])
def test_fir_resample_block_variants(ops, direction, prec):
    """The fused FIR kernels pick rows / output block per thread by the grid size (big blocks on big grids); every output is the
    same fma sequence in every variant: a batch large enough for the big-block variant gives, clip by clip, the bits of the
    one-clip launches (small-block variants), which test_fir_resample pins against the oracle."""
    g = torch.Generator(device="cuda").manual_seed(7 + direction)
    B, H, W, C = 6, 256, 128, 256
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(DT[prec])
    aff = torch.stack([1 + 0.2 * torch.randn(B, C, device="cuda", generator=g), 0.3 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
    raw, act = ops.fir_resample(x, direction, affine=aff)
    for b in (0, B - 1):
        raw1, act1 = ops.fir_resample(x[b:b + 1].contiguous(), direction, affine=aff[b:b + 1].contiguous())
        assert torch.equal(raw[b], raw1[0]) and torch.equal(act[b], act1[0])
    # a crop small enough for the one-row / one-pixel kernels: away from the cut edges it must give the big launch's bits (this is the
    # cross-check of the full-strip variants -- unconditional stores, next row requested early -- against the plain kernels)
    small = x[:1, :24, :16].contiguous()
    raw_s, act_s = ops.fir_resample(small, direction, affine=aff[:1].contiguous())
    assert torch.isfinite(raw_s.float()).all() and torch.isfinite(act_s.float()).all()
    if direction > 0:
        assert torch.equal(raw_s[0, :44, :28], raw[0, :44, :28]) and torch.equal(act_s[0, :44, :28], act[0, :44, :28])
    else:
        assert torch.equal(raw_s[0, :11, :7], raw[0, :11, :7]) and torch.equal(act_s[0, :11, :7], act[0, :11, :7])


@pytest.mark.parametrize("B,H,W", [(9, 488, 264), (9, 512, 256), (8, 384, 128)], ids=["ragged", "full_strips_16", "full_strips_8"])
def test_fir_down_marching_strips_bit_identical(ops, B, H, W):
    """Big grids take fir_down_march_kernel (strips of 16 / 8 output rows x 4 columns per thread, each input row activated once): same
    fma sequence per output as the block kernels -- clip by clip the bits of one-clip launches.  ragged: 244 output rows, 132 columns
    (guarded stores); full strips: the variant with unconditional stores and peeled row pairs."""
    g = torch.Generator(device="cuda").manual_seed(3)
    C = 256
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    aff = torch.stack([1 + 0.2 * torch.randn(B, C, device="cuda", generator=g), 0.3 * torch.randn(B, C, device="cuda", generator=g)], -1).contiguous()
    raw, act = ops.fir_resample(x, -1, affine=aff)
    for b in (0, 4, B - 1):
        raw1, act1 = ops.fir_resample(x[b:b + 1].contiguous(), -1, affine=aff[b:b + 1].contiguous())
        assert torch.equal(raw[b], raw1[0]) and torch.equal(act[b], act1[0])


def test_upfirdn2d_golden():
    from flowdec_amd import op
    g = load_golden("g4_upfirdn2d.npz")
    k = torch.from_numpy(O.setup_fir_kernel((1, 3, 3, 1))).cuda()
    for nm in ("a", "b"):
        x = dev(g["x" + nm])
        up = op.upfirdn2d(x, k * 4, up=2, pad=(2, 1))
        dn = op.upfirdn2d(x, k, down=2, pad=(1, 1))
        check(f"upfirdn2d_up[{nm}]", up.cpu().numpy(), g["up_" + nm], 1e-6)
        check(f"upfirdn2d_down[{nm}]", dn.cpu().numpy(), g["down_" + nm], 1e-6)
    # asymmetric generic call (up 2x3, down 1x2, pads x(1,2) y(0,1), 3x2 kernel) straight through the raw ABI
    from flowdec_amd import ops
    xa = dev(g["xa"])
    n, c, h, w = xa.shape
    out = ops.upfirdn2d_raw(xa.reshape(-1, h, w, 1), dev(g["k2"]), 2, 3, 1, 2, 1, 2, 0, 1)
    check("upfirdn2d_generic", out.reshape(n, c, out.shape[1], out.shape[2]).cpu().numpy(), g["generic_a"], 1e-6)


def test_upfirdn2d_errors():
    from flowdec_amd import op
    with pytest.raises(RuntimeError):
        op.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(4, 4))  # CPU tensor -> refuses (no CPU fallback)


def test_fused_bias_act():
    from flowdec_amd import op
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 5, 7)).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    got = op.fused_leaky_relu(dev(x), dev(b)).cpu().numpy()
    v = x + b[None, :, None, None]
    ref = np.where(v > 0, v, 0.2 * v) * np.sqrt(2.0)
    check("fused_leaky_relu", got, ref, 1e-6)


# ---------------------------------------------------------------------------------------------------------
# time embedding
# ---------------------------------------------------------------------------------------------------------
def test_time_embedding_golden(ops):
    g = load_golden("g8_ncsnpp_nf8.npz")
    sd = O.random_state_dict(seed=int(g["seed"]), nf=8)
    p = lambda k: dev(sd["backbone.all_modules." + k])
    temb = ops.time_embedding(dev(g["t"]), p("0.W"), p("1.weight"), p("1.bias"), p("2.weight"), p("2.bias"))
    check("time_embedding", temb.cpu().numpy(), g["temb"], 5e-5)
    rb = "backbone.all_modules.4."
    tb = ops.temb_bias(temb, dev(sd[rb + "Dense_0.weight"]), dev(sd[rb + "Dense_0.bias"]), dev(sd[rb + "Conv_0.bias"])).cpu().numpy()
    ref = O.linear(O.silu(g["temb"]), sd[rb + "Dense_0.weight"], sd[rb + "Dense_0.bias"]) + sd[rb + "Conv_0.bias"]
    check("temb_bias", tb, ref, 5e-5)


# ---------------------------------------------------------------------------------------------------------
# STFT front / back end
# ---------------------------------------------------------------------------------------------------------
def test_stft_golden(ops):
    g = load_golden("g1_stft.npz")
    y = dev(g["y"][:, 0])
    Y, nf, T = ops.stft_compress(y, normalize=False)
    assert T == 13 and Y.shape == (2, 1, 768, 64)
    check("stft_compress", Y[..., :T].cpu().numpy(), g["compressed"], 2e-5)
    assert float(Y[..., T:].abs().max()) == 0.0            # zero padding to a multiple of 64
    assert np.array_equal(nf.cpu().numpy(), np.ones(2, np.float32))
    yi = ops.decompress_istft(dev(np.pad(g["compressed"], ((0, 0), (0, 0), (0, 0), (0, 51)))), T, 4800)
    check("istft_roundtrip", yi.cpu().numpy(), g["roundtrip"][:, 0], 2e-5)
    Z = dev(np.pad(O.compress(O.decompress(g["Z"])), ((0, 0), (0, 0), (0, 0), (0, 55))))
    za = ops.decompress_istft(dev(np.pad(g["Z"], ((0, 0), (0, 0), (0, 0), (0, 55)))), 9, 3000)
    zb = ops.decompress_istft(dev(np.pad(g["Z"], ((0, 0), (0, 0), (0, 0), (0, 55)))), 9, 3072)
    check("istft_len3000", za.cpu().numpy(), g["istft_len3000"][:, 0], 2e-5)
    check("istft_len3072", zb.cpu().numpy(), g["istft_len3072"][:, 0], 2e-5)


def test_stft_normalize_and_silence(ops):
    g = load_golden("g3_pad_norm.npz")
    y = dev(g["y"][:, 0])
    Y, nf, T = ops.stft_compress(y, normalize=True)
    assert np.array_equal(nf.cpu().numpy(), g["normfac"].reshape(-1))  # bit exact, silent clip -> 1
    ref = O.pad_spec(O.compress(O.stft(g["y_norm"])))[0]
    check("stft_normalized", Y.cpu().numpy(), ref, 2e-5)
    assert float(Y[1].abs().max()) == 0.0


def test_stft_roundtrip_full_size(ops):
    """Size-independent property at the BASELINE config-2 shape: iSTFT(STFT(y)) == y (feature_extractors.py:21-22)."""
    gen = torch.Generator(device="cuda").manual_seed(0)
    y = 0.1 * torch.randn(8, 96000, device="cuda", generator=gen)
    Y, nf, T = ops.stft_compress(y, normalize=True)
    assert T == 251 and Y.shape[-1] == 256
    yr = ops.decompress_istft(Y, T, 96000, nf)
    e = float((yr - y).norm() / y.norm())
    report("stft_roundtrip_8x96000", e, 2e-4)
    assert e < 2e-4


# ---- one ResnetBlockBigGANpp through fd_resblock against the reference golden G6 --------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(O.RESBLOCK_CASES))
def test_resblock_golden(ops, name, prec):
    g = load_golden("g6_resblock.npz")
    seed, ci, co, up, down = O.RESBLOCK_CASES[name]
    p = {k: dev(v) for k, v in O.random_resblock_params(seed, ci, co, has_conv2=(ci != co or up or down)).items()}
    x = nhwc(g[f"{name}_x"], DT[prec])
    x0, x1 = (x[..., :256].contiguous(), x[..., 256:].contiguous()) if name == "cat" else (x, None)    # virtual concat 256 + 64
    out = ops.resblock(x0, x1, p, dev(g[f"{name}_temb"]), up=up, down=down)
    assert tuple(out.shape) == (2, g[f"{name}_out"].shape[2], g[f"{name}_out"].shape[3], co)
    check(f"resblock[{name},{prec}]", from_nhwc(out), g[f"{name}_out"], 1e-4 if prec == "fp32" else 3e-2)


def test_gn_silu_apply_golden(ops):
    g = load_golden("g5_groupnorm_silu.npz")
    x = nhwc(g["x64"], torch.float32)
    aff = ops.gn_affine(x, None, dev(g["gamma64"]), dev(g["beta64"]))
    check("gn_silu_apply[64]", from_nhwc(ops.gn_silu_apply(x, aff)), g["out64"], 2e-5)
    # GroupNorm(32, 320) over the virtual concat 256 + 64 (ncsnpp.py:337): one statistics pass, applied per tensor
    x = nhwc(g["x320"], torch.float32)
    x0, x1 = x[..., :256].contiguous(), x[..., 256:].contiguous()
    aff = ops.gn_affine(x0, x1, dev(g["gamma320"]), dev(g["beta320"]))
    out = torch.cat([ops.gn_silu_apply(x0, aff[:, :256].contiguous()), ops.gn_silu_apply(x1, aff[:, 256:].contiguous())], dim=-1)
    check("gn_silu_apply[256+64]", from_nhwc(out), g["out320"], 2e-5)
