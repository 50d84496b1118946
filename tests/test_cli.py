"""enhance.py-equivalent driver and checkpoint reader."""
import os
from collections.abc import Mapping, Sequence

import numpy as np
import pytest
import torch

from oracle import flowdec_oracle as O
from conftest import ROOT


def synthetic_ckpt(nf=8, seed=8, with_hp=True):
    sd = {k: torch.from_numpy(v) for k, v in O.random_state_dict(seed=seed, nf=nf).items()}
    ema = {k: v.clone() for k, v in sd.items()}
    ema["backbone.all_modules.3.bias"] = ema["backbone.all_modules.3.bias"] + 1.0   # make EMA != raw weights
    sig = torch.full((768, 1), 0.4, dtype=torch.float64)
    for d in (sd, ema):
        d["sigma_y"] = sig.clone(); d["sigma_x"] = torch.tensor(0.0)
        d["feature_extractor.complex_stft.window"] = torch.signal.windows.hann(1534)
    ckpt = {"state_dict": sd, "_pl_ema_state_dict": ema, "epoch": 3}
    if with_hp:
        ckpt["hyper_parameters"] = {"sampling_rate": 48000, "model": {
            "backbone": {"_target_": "flowdec.backbones.ncsnpp.NCSNpp", "nf": nf, "ch_mult": [4, 4, 4, 2], "num_res_blocks": 1,
                         "attn_resolutions": [], "bottleneck_attn": False, "image_size": 768},
            "feature_extractor": {"n_fft": 1534, "n_hops": 4, "window_fn": "hann", "alpha": 0.3, "beta": 0.33}}}
    return ckpt


def test_read_list_formats(tmp_path):
    from flowdec_amd.enhance_cli import collect_files, read_list
    a = tmp_path / "plain.txt"; a.write_text("x/a.wav\n\nx/b.wav\n")
    fl = read_list(str(a))
    assert fl.inputs == ["x/a.wav", "x/b.wav"] and fl.clean is None and not fl.from_pairs
    b = tmp_path / "pairs.txt"; b.write_text("c/a.wav ---> n/a.wav\nc/b.wav,n/b.wav\n")
    fl = read_list(str(b))
    assert fl.from_pairs and fl.inputs == ["n/a.wav", "n/b.wav"] and fl.clean == ["c/a.wav", "c/b.wav"]
    assert collect_files(str(b), False) == (["n/a.wav", "n/b.wav"], ["c/a.wav", "c/b.wav"])   # second entry is the input
    assert collect_files(str(a), True) == ([str(a)], None)
    for name, text in (("bad1.txt", "c/a.wav,n/a.wav\nplain.wav\n"), ("bad2.txt", "plain.wav\nc/a.wav ---> n/a.wav\n")):
        bad = tmp_path / name; bad.write_text(text)
        with pytest.raises(ValueError, match=r":2: .*inconsistent"):     # the reference asserts (enhance.py:162)
            read_list(str(bad))
    # the arrow wins over commas (reference precedence, enhance.py:153-158): a path with a comma in an arrow line is not cut
    c = tmp_path / "comma.txt"; c.write_text("c/a,1.wav ---> n/a,1.wav\n")
    fl = read_list(str(c))
    assert fl.clean == ["c/a,1.wav"] and fl.inputs == ["n/a,1.wav"]
    # more than two fields: the reference keeps the whole split and uses fields 0 and 1 (enhance.py:153-158), so the tool's own
    # triples_list output (`clean ---> noisy ---> out`) or a CSV with extra columns can be fed back in; here the same, with a warning
    for name, text in (("tri1.txt", "a.wav,b.wav,c.wav\n"), ("tri2.txt", "a.wav ---> b.wav ---> c.wav\n")):
        tri = tmp_path / name; tri.write_text(text)
        fl = read_list(str(tri))
        assert fl.clean == ["a.wav"] and fl.inputs == ["b.wav"]
    (tmp_path / "d").mkdir()
    for n in ("b.wav", "a.wav", "c.txt"):
        (tmp_path / "d" / n).write_bytes(b"")
    assert [os.path.basename(f) for f in collect_files(str(tmp_path / "d"), False)[0]] == ["a.wav", "b.wav"]


def test_plan_jobs_filters(tmp_path):
    """The work list: exclusion pattern (renumbers), inclusive index window, exists-check under --skip-existing; pair lists carry the
    clean path along."""
    from flowdec_amd.enhance_cli import plan_jobs
    out = tmp_path / "out"; out.mkdir()
    noisy = [f"n/{k}.wav" for k in "abcde"]
    clean = [f"c/{k}.wav" for k in "abcde"]
    (out / "c.wav").write_bytes(b"")
    jobs = list(plan_jobs(noisy, clean, str(out), None, None, True))
    assert [j.index for j in jobs] == [0, 1, 2, 3, 4] and [j.pending for j in jobs] == [True, True, False, True, True]
    assert jobs[1].src == "n/b.wav" and jobs[1].clean == "c/b.wav" and jobs[1].dst == os.path.join(str(out), "b.wav")
    assert all(j.pending for j in plan_jobs(noisy, None, str(out), None, None, False)) and next(plan_jobs(noisy, None, str(out), None, None, False)).clean is None
    assert [j.src for j in plan_jobs(noisy, clean, str(out), 1, 3, True)] == ["n/b.wav", "n/c.wav", "n/d.wav"]
    assert [j.src for j in plan_jobs(noisy, clean, str(out), 3, 99, True)] == ["n/d.wav", "n/e.wav"]
    ex = list(plan_jobs(noisy, clean, str(out), 1, 2, True, exclude="b.wav"))      # the window counts what the exclusion left
    assert [(j.index, j.src, j.clean) for j in ex] == [(1, "n/c.wav", "c/c.wav"), (2, "n/d.wav", "c/d.wav")]
    assert list(plan_jobs([], None, str(out), None, None, True)) == []


def test_wav_roundtrip_and_resample(tmp_path):
    from flowdec_amd.enhance_cli import load_wav, resample, save_wav
    x = torch.from_numpy((0.3 * np.sin(np.arange(4800) * 0.05)).astype(np.float32))[None]
    save_wav(str(tmp_path / "f.wav"), x, 48000)
    y, sr = load_wav(str(tmp_path / "f.wav"))
    assert sr == 48000 and y.shape == (1, 4800) and torch.equal(x, y)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "i16.wav"), 16000, (x[0].numpy() * 32767).astype(np.int16))
    z, sr2 = load_wav(str(tmp_path / "i16.wav"))
    assert sr2 == 16000 and float((z - x).abs().max()) < 1e-4
    assert resample(z, 16000, 48000).shape == (1, 14400)


def test_sinc_resampler_matches_torchaudio_definition():
    """enhance.py:118: torchaudio.functional.resample(y, sr, 48000, lowpass_filter_width=64).  torchaudio is absent, so the
    restated filter bank is checked (a) against the closed-form Hann-windowed sinc evaluated independently in float64 at
    every (phase, tap), (b) for its structural constants (width, taps, DC gain), and (c) end to end: a band-limited tone
    resampled 16 k -> 48 k and 44.1 k -> 48 k must equal the analytically resampled tone, and the output length must be
    ceil(n * L / o)."""
    import math
    from flowdec_amd.enhance_cli import resample, sinc_resample_kernel
    for sr, lpw in ((16000, 64), (44100, 64), (96000, 64), (16000, 6)):
        k, width, o, n = sinc_resample_kernel(sr, 48000, lowpass_filter_width=lpw)
        g = math.gcd(sr, 48000)
        assert (o, n) == (sr // g, 48000 // g)
        f = min(o, n) * 0.99
        assert width == math.ceil(lpw * o / f) and k.shape == (n, 2 * width + o) and k.dtype == np.float32
        ref = np.zeros(k.shape)
        for i in range(n):                                  # the definition, one scalar at a time (float64)
            for j in range(0, k.shape[1], 7 if k.size > 40000 else 1):
                t = ((j - width) / o - i / n) * f
                t = max(-lpw, min(lpw, t))
                sinc = 1.0 if t == 0 else math.sin(math.pi * t) / (math.pi * t)
                ref[i, j] = sinc * math.cos(math.pi * t / lpw / 2) ** 2 * f / o
                assert abs(k[i, j] - ref[i, j]) <= 1e-7, (sr, i, j)
        assert np.allclose(k.sum(1), 1.0, atol=2e-3)        # every phase passes DC
    for sr, L in ((16000, 4000), (44100, 4410), (8000, 999)):
        tt = np.arange(L) / sr
        x = torch.from_numpy((0.5 * np.sin(2 * np.pi * 440.0 * tt)).astype(np.float32))[None]
        y = resample(x, sr, 48000)
        g = math.gcd(sr, 48000)
        assert y.shape == (1, math.ceil((48000 // g) * L / (sr // g)))
        want = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(y.shape[-1]) / 48000)
        m = slice(600, y.shape[-1] - 600)                    # away from the zero-padded ends
        assert float(np.abs(y[0].numpy()[m] - want[m]).max()) < 2e-3
    assert resample(x, 48000, 48000) is x


def test_checkpoint_reader_cpu():
    from flowdec_amd.enhance_cli import model_from_checkpoint
    ckpt = synthetic_ckpt()
    m_ema = model_from_checkpoint(ckpt, ema=True)
    m_raw = model_from_checkpoint(ckpt, ema=False)
    assert m_ema.backbone.nf == 8 and m_ema.sigma_y.dtype == torch.float64 and float(m_ema.sigma_y[0]) == pytest.approx(0.4)
    b_e = m_ema.state_dict()["backbone.all_modules.3.bias"]; b_r = m_raw.state_dict()["backbone.all_modules.3.bias"]
    assert torch.allclose(b_e, b_r + 1.0)                      # --ema picks _pl_ema_state_dict (callbacks/ema.py:201-215)
    m2 = model_from_checkpoint(synthetic_ckpt(with_hp=False))  # no hyper_parameters: width inferred from the weights
    assert m2.backbone.nf == 8
    m3 = model_from_checkpoint(ckpt["_pl_ema_state_dict"])      # bare state_dict
    assert torch.equal(m3.state_dict()["backbone.all_modules.3.bias"], b_e)
    broken = {k: v for k, v in ckpt["state_dict"].items() if "all_modules.4.Conv_0" not in k}
    with pytest.raises(RuntimeError):
        model_from_checkpoint({"state_dict": broken}, ema=False)


class _FakeDictConfig(Mapping):
    """Stands in for omegaconf.DictConfig (what Lightning stores for save_hyperparameters(full_config), model.py:60-63):
    a Mapping that is NOT a dict, with nested nodes of the same kind and list-like nodes that are not lists."""

    def __init__(self, d):
        self._d = {k: (_FakeDictConfig(v) if isinstance(v, dict) else (_FakeList(v) if isinstance(v, list) else v)) for k, v in d.items()}

    def __getitem__(self, k): return self._d[k]
    def __iter__(self): return iter(self._d)
    def __len__(self): return len(self._d)


class _FakeList(Sequence):
    def __init__(self, v): self._v = list(v)
    def __getitem__(self, i): return self._v[i]
    def __len__(self): return len(self._v)


def test_checkpoint_reader_non_dict_hyper_parameters():
    """ablation_higheralpha_75s-style checkpoint: alpha / beta differ from the defaults and hyper_parameters is a DictConfig-like
    Mapping.  The reader must pick the values up (it used to fall back to alpha = 0.3 / beta = 0.33 silently)."""
    from flowdec_amd.enhance_cli import model_from_checkpoint
    ckpt = synthetic_ckpt()
    hp = ckpt["hyper_parameters"]
    hp["model"]["feature_extractor"].update(alpha=0.5, beta=0.16)
    ckpt["hyper_parameters"] = _FakeDictConfig(hp)
    m = model_from_checkpoint(ckpt)
    cfg = m.feature_extractor._cfg()
    assert cfg["alpha"] == pytest.approx(0.5) and cfg["beta"] == pytest.approx(0.16) and m.backbone.nf == 8
    assert tuple(m.backbone.ch_mult) == (4, 4, 4, 2)
    ckpt["hyper_parameters"] = 12345        # present but unreadable: refuse instead of guessing
    with pytest.raises(RuntimeError):
        model_from_checkpoint(ckpt)


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    from flowdec_amd import enhance_cli
    torch.save(synthetic_ckpt(), tmp_path / "m.ckpt")
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    rng = np.random.default_rng(0)
    for name, n, sr in (("a.wav", 24000, 48000), ("b.wav", 16000, 16000), ("long.wav", 31 * 8000, 8000)):
        enhance_cli.save_wav(str(ind / name), torch.from_numpy((0.1 * rng.standard_normal(n)).astype(np.float32))[None], sr)
    n = enhance_cli.main(["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind), "--outdir", str(outd), "--N", "2", "--solver", "midpoint",
                          "--rtf", "--seed", "3"])
    assert n == 2 and not (outd / "long.wav").exists()          # > 30 s files are skipped (enhance.py:115,139)
    a, sr = enhance_cli.load_wav(str(outd / "a.wav"))
    assert sr == 48000 and a.shape == (1, 24000) and torch.isfinite(a).all()
    b, sr = enhance_cli.load_wav(str(outd / "b.wav"))
    assert sr == 48000 and b.shape == (1, 48000)                 # resampled 16 k -> 48 k
    rows = (outd / "rtfs.csv").read_text().strip().splitlines()
    assert rows[0] == "path,runtime,filetime,rtf" and len(rows) == 3
    path, runtime, filetime, rtf = rows[1].split(",")
    assert float(filetime) == pytest.approx(0.5) and float(rtf) == pytest.approx(float(runtime) / 0.5, rel=1e-3)
    assert enhance_cli.main(["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind), "--outdir", str(outd), "--N", "2"]) == 0  # skip-existing
    # the CLI result equals a direct enhance() call with the same seed
    m = enhance_cli.load_from_checkpoint(str(tmp_path / "m.ckpt"), map_location="cuda:0")
    y, _ = enhance_cli.load_wav(str(ind / "a.wav"))
    ref = m.enhance(y, N=2, solver="midpoint", generator=torch.Generator(device="cuda:0").manual_seed(3))
    assert torch.equal(ref, a)
    # ... and the ORACLE's result on the same checkpoint: the CLI draws its noise with a seeded device generator, file a.wav
    # (first in sorted order) gets the first draw, so the same tensor can be handed to the oracle (fp32 run, tolerance of the
    # fp32 waveform parity tests)
    out32 = tmp_path / "out32"
    assert enhance_cli.main(["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind / "a.wav"), "--single-file", "--outdir", str(out32), "--N", "2",
                             "--solver", "midpoint", "--seed", "3", "--precision", "fp32"]) == 1
    a32, _ = enhance_cli.load_wav(str(out32 / "a.wav"))
    Tp = O.padded_frames(O.num_frames(24000))
    noise = torch.randn((1, 1, 768, Tp), dtype=torch.complex64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3))
    sd = {k: v.numpy() for k, v in synthetic_ckpt()["_pl_ema_state_dict"].items() if k.startswith("backbone.")}
    ref_o = O.enhance(O.NCSNppOracle(sd, nf=8), y.numpy()[None], noise.cpu().numpy(), np.full((768, 1), 0.4), N=2, solver="midpoint")
    err = float(np.linalg.norm(a32.numpy() - ref_o[0]) / np.linalg.norm(ref_o[0]))
    assert err < 5e-4, f"CLI vs oracle: {err:.3e}"


@pytest.mark.gpu
def test_real_checkpoint_check_script(tmp_path):
    """scripts/real_checkpoint_check.py (the first-run-on-a-real-checkpoint one-liner) on a synthetic Lightning checkpoint: the three
    precisions run on the same noise, the report carries per-file errors, PASS / FAIL follows the thresholds."""
    import importlib.util
    from flowdec_amd import enhance_cli
    spec = importlib.util.spec_from_file_location("rcc", os.path.join(ROOT, "scripts", "real_checkpoint_check.py"))
    rcc = importlib.util.module_from_spec(spec); spec.loader.exec_module(rcc)
    torch.save(synthetic_ckpt(), tmp_path / "m.ckpt")
    ind = tmp_path / "in"; ind.mkdir()
    rng = np.random.default_rng(1)
    for name, n, sr in (("a.wav", 24000, 48000), ("b.wav", 12000, 16000)):
        enhance_cli.save_wav(str(ind / name), torch.from_numpy((0.1 * rng.standard_normal(n)).astype(np.float32))[None], sr)
    rep = rcc.run(["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind), "--N", "2", "--solver", "midpoint", "--tol", "0.3",
                   "--report", str(tmp_path / "r.json")])
    assert rep["verdict"] == "PASS" and len(rep["files"]) == 2 and os.path.exists(tmp_path / "r.json")
    assert all(r["bf16x3_rel_l2"] < 5e-4 and 1e-4 < r["bf16_rel_l2"] < 0.3 for r in rep["files"])   # (nf = 8 toy model: bf16 ~ 1e-1)
    rep = rcc.run(["--ckpt", str(tmp_path / "m.ckpt"), "--files", str(ind), "--N", "2", "--solver", "midpoint", "--tol", "1e-4"])
    assert rep["verdict"] == "FAIL"


def test_plan_batches_buckets_by_padded_frames(tmp_path):
    """(CPU) The CLI's batching plan: files are bucketed by the frame count their spectrogram pads to -- computed from the wav HEADERS
    and the resampler's output length -- and cut into batches of at most --batch-files, in work-list order; files the length rule
    skips, stereo files and unreadable ones go through the one-file path."""
    from types import SimpleNamespace
    from flowdec_amd import enhance_cli
    from flowdec_amd.model import padded_frames_of
    assert [padded_frames_of(n) for n in (24575, 24576, 49151, 49152, 96000)] == [64, 128, 128, 192, 256]
    assert enhance_cli.resampled_length(12000, 16000, 48000) == 36000 and enhance_cli.resampled_length(1001, 44100, 48000) == 1090
    assert enhance_cli.resample(torch.zeros(1, 1001), 44100, 48000).shape[-1] == 1090
    rng = np.random.default_rng(0)
    spec = [("a", 30000, 48000, 1), ("b", 41234, 48000, 1), ("c", 20000, 48000, 1), ("d", 12000, 16000, 1), ("e", 30000, 48000, 2),
            ("f", 49151, 48000, 1), ("g", 31 * 8000, 8000, 1)]
    for name, n, sr, ch in spec:
        enhance_cli.save_wav(str(tmp_path / f"{name}.wav"), torch.from_numpy((0.1 * rng.standard_normal((ch, n))).astype(np.float32)), sr)
    (tmp_path / "h.wav").write_bytes(b"not a wav file")
    assert enhance_cli.wav_info(str(tmp_path / "e.wav")) == (30000, 48000, 2)
    model = SimpleNamespace(sampling_rate=48000, feature_extractor=SimpleNamespace(_cfg=lambda: dict(n_fft=1534, hop=384, alpha=0.3, beta=0.33)))
    jobs = list(enhance_cli.plan_jobs(sorted(str(p) for p in tmp_path.glob("*.wav")), None, str(tmp_path / "out"), None, None, True))
    names = lambda plan: [[j.src.split("/")[-1][:-4] for j in b] for b in plan]
    assert names(enhance_cli.plan_batches(model, jobs, 2)) == [["c"], ["a", "b"], ["d", "f"], ["e"], ["g"], ["h"]]
    assert names(enhance_cli.plan_batches(model, jobs, 8)) == [["c"], ["a", "b", "d", "f"], ["e"], ["g"], ["h"]]
    assert names(enhance_cli.plan_batches(model, jobs, 1)) == [[n] for n in "abcdefgh"]
    # RunLog opens its files on __enter__; a failing second open must not leak the first handle
    log = enhance_cli.RunLog(str(tmp_path / "missing_dir"), "", want_rtf=True, want_triples=True)
    with pytest.raises(OSError):
        log.__enter__()
