"""Ragged batches (round 6): clips of DIFFERENT lengths whose spectrograms pad to the same T_pad run as ONE native call
(fd_stft_compress_ragged / fd_decompress_istft_ragged / fd_enhance_ragged, FlowModel.enhance_batch, the CLI's --batch-files).

The reference's driver enhances a directory file by file (enhance.py:96-137; model.py:129-163,476-528; util/other.py:25-52).  The
contract here is stronger than a tolerance: every clip of a ragged batch is BIT-IDENTICAL to the one-clip call on that clip --
which the goldens G17 / G18 / G21 pin to the reference's own FlowModel.enhance."""
import numpy as np
import pytest
import torch

from oracle import flowdec_oracle as O

pytestmark = pytest.mark.gpu

HOP, NFFT = 384, 1534
# one T_pad = 128 bucket: T = 1 + L // 384 in 65..128 <=> 24576 <= L <= 49151 (both ends included below)
BUCKET128 = [24576, 30000, 41234, 48000, 49151]


def _model(nf, precision, seed=8):
    import flowdec_amd
    m = flowdec_amd.from_preset("flowdec_75m", precision=precision, nf=nf)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=seed, nf=nf).items()}, strict=False)
    return m.cuda()


def _clips(lengths, seed=0):
    rng = np.random.default_rng(seed)
    return [torch.from_numpy((0.1 * (1 + i) * rng.standard_normal(n)).astype(np.float32)) for i, n in enumerate(lengths)]


def test_stft_ragged_bit_identical_to_single_clips():
    """Front end and back end alone: per-clip normalisation factor, reflect padding at the clip's own end, its own frame count (zero
    frames behind it, as pad_spec leaves them) and torch.istft(length = its own length) with its own overlap-add envelope."""
    from flowdec_amd import ops
    clips = _clips(BUCKET128, seed=1)
    clips[2] = torch.zeros_like(clips[2])       # an all-zero clip: the silence guard of normalize_noisy (util/other.py:77)
    Lrow = max(BUCKET128)
    y = torch.zeros(len(clips), Lrow)
    for b, c in enumerate(clips):
        y[b, :c.numel()] = c
    Y, nf, T = ops.stft_compress(y.cuda(), lengths=BUCKET128)
    assert T == 1 + Lrow // HOP and Y.shape[-1] == 128
    back = ops.decompress_istft(Y, T, Lrow, nf, lengths=BUCKET128)
    for b, c in enumerate(clips):
        Yb, nfb, Tb = ops.stft_compress(c[None].cuda())
        assert Tb == 1 + c.numel() // HOP
        assert torch.equal(nfb, nf[b:b + 1]), f"clip {b}: normfac differs"
        assert torch.equal(torch.view_as_real(Yb), torch.view_as_real(Y[b:b + 1])), f"clip {b}: spectrogram differs from the one-clip call"
        assert not torch.view_as_real(Y[b, :, :, Tb:]).any(), f"clip {b}: frames behind its own T = {Tb} must be zero"
        xb = ops.decompress_istft(Yb, Tb, c.numel(), nfb)
        assert torch.equal(xb[0], back[b, :c.numel()]), f"clip {b}: iSTFT differs from the one-clip call"
        assert not back[b, c.numel():].any()
    # the round trip reproduces the clips (the docstring claim of feature_extractors.py:21-22), the zero clip stays zero
    for b, c in enumerate(clips):
        ref = c.numpy()
        err = np.abs(back[b, :c.numel()].cpu().numpy() - ref).max()
        assert err < 2e-5 * max(np.abs(ref).max(), 1.0), (b, err)


def test_stft_ragged_rejects_mixed_buckets():
    from flowdec_amd import ops
    y = torch.zeros(2, 49151).cuda()
    with pytest.raises(RuntimeError, match="T_pad bucket"):
        ops.stft_compress(y, lengths=[49151, 20000])       # 20000 samples pad to 64 frames, not 128
    with pytest.raises(RuntimeError, match="row length"):
        ops.stft_compress(y, lengths=[49151, 60000])
    with pytest.raises(RuntimeError, match="row length"):
        ops.stft_compress(y, lengths=[49151, 700])          # shorter than the reflect padding


@pytest.mark.parametrize("nf,precision,solver,N", [(8, "fp32", "midpoint", 2), (8, "bf16", "euler", 3), (64, "bf16", "euler", 2)])
def test_enhance_batch_bit_identical_to_one_by_one(nf, precision, solver, N):
    """FlowModel.enhance_batch == [FlowModel.enhance(c) for c in clips], bit for bit, with injected noise and with per-clip generators;
    a second batch of OTHER lengths in the same bucket replays the captured graph (the lengths are device data, not graph constants)."""
    m = _model(nf, precision)
    F, Tp = 768, 128
    for trial, lengths in enumerate((BUCKET128, [49151, 24576, 33333, 25000, 47999], BUCKET128[::-1])):
        clips = _clips(lengths, seed=10 + trial)
        g = torch.Generator().manual_seed(100 + trial)
        noise = [torch.view_as_complex(torch.randn(1, 1, F, Tp, 2, generator=g) / np.sqrt(2)) for _ in clips]
        outs = m.enhance_batch([c.cuda() if trial == 0 else c for c in clips], N=N, solver=solver, noise=noise)
        for b, c in enumerate(clips):
            ref = m.enhance(c, N=N, solver=solver, noise=noise[b], use_graph=False)
            assert outs[b].shape == c.shape and outs[b].device.type == ("cuda" if trial == 0 else "cpu")
            assert torch.isfinite(ref).all() and ref.abs().max() > 0
            assert torch.equal(outs[b].cpu(), ref.cpu()), f"trial {trial}, clip {b} ({lengths[b]} samples): batch != one-by-one"
    # per-clip generators = what the CLI does under --seed (file i: seed + i)
    clips = _clips(BUCKET128, seed=3)
    gens = [torch.Generator(device="cuda").manual_seed(7 + i) for i in range(len(clips))]
    outs = m.enhance_batch(clips, N=N, solver=solver, generator=gens)
    for b, c in enumerate(clips):
        ref = m.enhance(c, N=N, solver=solver, generator=torch.Generator(device="cuda").manual_seed(7 + b), use_graph=False)
        assert torch.equal(outs[b], ref), f"clip {b}: seeded batch != seeded one-by-one"
    # shapes [1, L] and [1, 1, L] keep their rank
    outs = m.enhance_batch([clips[0][None], clips[1][None, None]], N=1, solver="euler")
    assert outs[0].shape == (1, BUCKET128[0]) and outs[1].shape == (1, 1, BUCKET128[1])
    with pytest.raises(RuntimeError, match="bucket"):
        m.enhance_batch([clips[0], torch.zeros(20000)], N=1)
    with pytest.raises(ValueError):
        m.enhance_batch(clips, N=1, solver="dopri5")


def test_enhance_batch_equal_lengths_is_enhance():
    """A ragged batch whose clips all have the row's length is the ordinary batched enhance()."""
    m = _model(8, "bf16")
    L = 48000
    clips = _clips([L] * 3, seed=5)
    g = torch.Generator().manual_seed(1)
    noise = torch.view_as_complex(torch.randn(3, 1, 768, 128, 2, generator=g) / np.sqrt(2))
    ref = m.enhance(torch.stack(clips)[:, None], N=2, solver="euler", noise=noise)
    outs = m.enhance_batch(clips, N=2, solver="euler", noise=[noise[i:i + 1] for i in range(3)])
    for b in range(3):
        assert torch.equal(outs[b], ref[b, 0])


def test_cli_batches_files_bit_identical_to_one_file_per_call(tmp_path):
    """`enhance_cli --batch-files 4` against `--batch-files 1` (the reference's loop) on a corpus of eight files in two T_pad buckets (three batches),
    one of them resampled from 16 kHz, plus one that is too long: identical bytes in every output file, the same rtfs.csv
    rows (paths), one aggregate rtf line."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).parent))
    from test_cli import synthetic_ckpt
    from flowdec_amd import enhance_cli
    torch.save(synthetic_ckpt(), tmp_path / "m.ckpt")
    ind = tmp_path / "in"
    ind.mkdir()
    rng = np.random.default_rng(2)
    spec = [("a", 30000, 48000, 1), ("b", 41234, 48000, 1), ("c", 24576, 48000, 1), ("d", 49151, 48000, 1), ("e", 48000, 48000, 1),
            ("f", 20000, 48000, 1), ("g", 23000, 48000, 1), ("h", 12000, 16000, 1), ("long", 31 * 8000, 8000, 1)]
    for name, n, sr, ch in spec:
        enhance_cli.save_wav(str(ind / f"{name}.wav"), torch.from_numpy((0.1 * rng.standard_normal((ch, n))).astype(np.float32)), sr)
    common = ["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind), "--N", "2", "--solver", "midpoint", "--rtf", "--seed", "11"]
    r4 = enhance_cli.run(common + ["--outdir", str(tmp_path / "o4"), "--batch-files", "4"])
    r1 = enhance_cli.run(common + ["--outdir", str(tmp_path / "o1"), "--batch-files", "1"])
    assert r4.n_done == r1.n_done == 8 and r4.n_too_long == r1.n_too_long == 1
    for name, n, sr, ch in spec[:-1]:
        a = (tmp_path / "o4" / f"{name}.wav").read_bytes()
        b = (tmp_path / "o1" / f"{name}.wav").read_bytes()
        assert a == b, f"{name}.wav: batched output differs from the one-file-per-call output"
    rows4 = sorted(l.split(",")[0].split("/")[-1] for l in (tmp_path / "o4" / "rtfs.csv").read_text().strip().splitlines()[1:])
    rows1 = sorted(l.split(",")[0].split("/")[-1] for l in (tmp_path / "o1" / "rtfs.csv").read_text().strip().splitlines()[1:])
    assert rows4 == rows1 and len(rows4) == 8
    assert r4.audio_seconds == pytest.approx(r1.audio_seconds) and r4.gpu_seconds > 0
    # the bucket plan: {a, b, c, d} + {e, h} in the 128-frame bucket (h: 12000 @ 16 kHz -> 36000 samples), {f, g} in the 64-frame one
    m = enhance_cli.load_from_checkpoint(str(tmp_path / "m.ckpt"), map_location="cuda:0")
    jobs = list(enhance_cli.plan_jobs(sorted(str(p) for p in ind.glob("*.wav")), None, str(tmp_path / "o9"), None, None, True))
    plan = [[j.src.split("/")[-1][:-4] for j in b] for b in enhance_cli.plan_batches(m, jobs, 4)]
    assert plan == [["f", "g"], ["a", "b", "c", "d"], ["e", "h"], ["long"]], plan
