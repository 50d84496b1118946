"""Command-line driver equivalent to the reference's `enhance.py`.

    python -m flowdec_amd.enhance_cli --ckpt flowdec_75m.ckpt --files noisy_dir/ --outdir out/ --N 3 --solver midpoint [--rtf]

Same arguments and file conventions as the reference (enhance.py:24-49): `--files` is a directory of *.wav, a file list
(one path per line, or `clean ---> noisy` / `clean,noisy` pair lines, enhance.py:146-164) or, with `--single-file`, one
wav; files longer than 30 s are skipped (:115,:139); `--rtf` writes `path,runtime,filetime,rtf` rows (:94,:135) with
rtf = runtime / filetime like the reference.  The score-model-only options are accepted and ignored.

Checkpoints: a Lightning `.ckpt` (dict with `_pl_ema_state_dict` and/or `state_dict`, optionally `hyper_parameters`
holding the resolved config: callbacks/ema.py:201-215, model.py:100,119) or a bare state_dict.  `--ema` (default)
selects the EMA weights exactly like `demo.ipynb` cell 2.

Non-48 kHz input is resampled with a restatement of torchaudio.functional.resample(..., lowpass_filter_width=64)
(enhance.py:118; torchaudio itself is not a dependency): `sinc_resample_kernel` / `resample` below.

Batching (round 6): the reference enhances one file per call (enhance.py:96-137).  This driver reads the lengths first, buckets the
files by the frame count their spectrogram pads to (T_pad, util/other.py:25-52) and runs up to `--batch-files` files of a bucket as
ONE ragged native call (FlowModel.enhance_batch -> fd_enhance_ragged): every file's waveform is bit-identical to the one-file call,
the GPU sees a batch.  With `--seed S` file i of the work list draws its noise from its own generator seeded S + i, so the result
of a file does not depend on the batching (`--batch-files 1` = the reference's loop).  Under `--rtf` a batch is timed as a whole and
its time is split over its files in proportion to their duration (every file of a batch gets the batch's rtf).

Length limit: the reference skips files longer than 30 s; so does this driver, in every precision (one image of the conv kernels
must stay below 2 GiB in bf16 and below 4 GiB with f32 storage -- 32-bit byte offsets inside an image -- both ~43 s of audio).
`PRECISION_MAX_SECONDS` is where a precision with a shorter reach would say so: such files are then skipped WITH a message and
the exit status is 3 (the reference would have processed them).
"""
import argparse
import contextlib
import glob
import math
import os
import re
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

from .model import BACKBONE_FINAL_NO_ATTN, AmplitudeCompressedComplexSTFT, FlowModel, NCSNpp, from_preset

MAX_SECONDS = 30.0  # enhance.py:115
PRECISION_MAX_SECONDS = {}   # precision -> clip length it can take, if shorter than MAX_SECONDS (none since round 4: every mode reaches ~43 s)
PRECISION_NOTE = {   # printed at start-up so that a log says which arithmetic produced the files
    "bf16": "bf16 storage and MFMA operands, f32 accumulation; ~2e-2 relative waveform error vs the fp32 reference on random weights",
    "mixed": "f32 residual stream, bf16 MFMA operands; ~1.3e-2",
    "bf16x3": "f32 storage, split-bf16 operands (3 MFMAs per product): meets the fp32 tolerances (~3e-5)",
    "fp32": "exact f32 MFMA (~6e-6); slowest",
}


# ------------------------------------------------------------------------------------------------
# file lists / wav I/O
# ------------------------------------------------------------------------------------------------
@dataclass
class FileList:
    """A parsed `--files` list: the inputs to enhance and, for pair lists, the clean reference each one belongs to."""
    inputs: List[str] = field(default_factory=list)
    clean: Optional[List[str]] = None     # None for a plain list

    @property
    def from_pairs(self) -> bool:
        return self.clean is not None


def _split_pair(entry: str) -> List[str]:
    """`clean ---> coded` if the arrow is present, else `clean,coded` if a comma is (the reference's precedence, enhance.py:153-158:
    a path with commas in an arrow line is not cut at the comma)."""
    if " ---> " in entry:
        return entry.split(" ---> ")
    if "," in entry:
        return entry.split(",")
    return [entry]


def read_list(listfile: str) -> FileList:
    """The list formats of the reference (enhance.py:146-164): one path per line, or pair lines `clean ---> coded` /
    `clean,coded` of which the SECOND entry is the file to enhance; further fields are ignored like there (the tool's own
    `triples_list` output `clean ---> noisy ---> out` can be fed back in), with a warning.  A plain line after a pair line is an
    error there (an assert); here it is a ValueError naming the line."""
    out = FileList()
    with open(listfile, "r") as f:
        for lineno, raw in enumerate(f, 1):
            entry = raw.strip()
            if not entry:
                continue
            parts = _split_pair(entry)
            if len(parts) > 2:
                print(f"warning: {listfile}:{lineno}: {len(parts)} fields in a pair line, using the first two (clean, coded)", file=sys.stderr)
            if len(parts) >= 2:
                if out.clean is None:
                    if out.inputs:
                        raise ValueError(f"{listfile}:{lineno}: pair line after plain paths -- inconsistent file list format")
                    out.clean = []
                out.clean.append(parts[0]); out.inputs.append(parts[1])
            elif out.from_pairs:
                raise ValueError(f"{listfile}:{lineno}: plain path after pair lines -- inconsistent file list format")
            else:
                out.inputs.append(entry)
    return out


def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """-> (float32 tensor [C, L] in [-1, 1], sampling rate), like torchaudio.load."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 1:
        x = x[:, None]
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def wav_info(path: str) -> Tuple[int, int, int]:
    """-> (samples per channel, sampling rate, channels) from the header only (memory-mapped: the data is not read)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path, mmap=True)
    return int(data.shape[0]), int(sr), (1 if data.ndim == 1 else int(data.shape[1]))


def resampled_length(length: int, sr: int, target: int) -> int:
    """Samples `resample` returns for `length` input samples: ceil(n * L / o) with o, n = sr, target over their gcd."""
    if int(sr) == int(target):
        return int(length)
    g = math.gcd(int(sr), int(target))
    return int(math.ceil((int(target) // g) * length / (int(sr) // g)))


def save_wav(path: str, x: torch.Tensor, sr: int) -> None:
    """float32 wav, [C, L] or [L] (the reference saves float tensors through torchaudio.save)."""
    from scipy.io import wavfile
    a = x.detach().cpu().float().numpy()
    if a.ndim == 2:
        a = a.T
    wavfile.write(path, sr, np.ascontiguousarray(a, dtype=np.float32))


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 64, rolloff: float = 0.99):
    """The polyphase filter bank of torchaudio.functional.resample (resampling_method='sinc_interp_hann', the default), restated
    from its published definition: with o = orig/gcd, n = new/gcd, f = min(o, n) * rolloff and width = ceil(lpw * o / f),
    phase i in [0, n) and tap k in [-width, width + o):
        t = clamp((k / o - i / n) * f, -lpw, lpw);   h[i, k] = sinc(t) * cos^2(pi * t / (2 * lpw)) * f / o      (sinc(t) = sin(pi t)/(pi t))
    Returns (kernel [n, 2 * width + o] float32 computed in float64 like torchaudio, width, o, n)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    f = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / f)
    k = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    i = np.arange(0, -n, -1, dtype=np.float64)[:, None] / n
    t = np.clip((i + k) * f, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    tp = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(tp == 0, 1.0, np.sin(tp) / tp)
    return (sinc * window * (f / o)).astype(np.float32), width, o, n


def resample(y: torch.Tensor, sr: int, target: int, lowpass_filter_width: int = 64, rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.functional.resample(y, sr, target, lowpass_filter_width=64) as the reference calls it (enhance.py:118):
    zero-pad by (width, width + o), correlate with the n-phase filter bank at stride o, interleave the phases, keep
    ceil(n * L / o) samples.  Host-side float32 (file pre-processing, not on the hot path)."""
    if int(sr) == int(target):
        return y
    kern, width, o, n = sinc_resample_kernel(sr, target, lowpass_filter_width, rolloff)
    shape = y.shape
    w = y.reshape(-1, shape[-1]).float()
    length = w.shape[-1]
    w = torch.nn.functional.pad(w, (width, width + o))
    r = torch.nn.functional.conv1d(w[:, None], torch.from_numpy(kern)[:, None], stride=o)       # [num, n, frames]
    r = r.transpose(1, 2).reshape(w.shape[0], -1)[:, : int(math.ceil(n * length / o))]
    return r.reshape(*shape[:-1], r.shape[-1])


# ------------------------------------------------------------------------------------------------
# checkpoint reader
# ------------------------------------------------------------------------------------------------
def _to_plain(obj):
    """hyper_parameters of a real Lightning checkpoint are an OmegaConf DictConfig (the reference calls
    save_hyperparameters(self.full_config), model.py:60-63), not a dict: convert any Mapping / sequence tree to plain
    Python containers (OmegaConf.to_container when omegaconf is importable, so that interpolations are resolved)."""
    from collections.abc import Mapping, Sequence
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(obj):
            return OmegaConf.to_container(obj, resolve=True)
    except ImportError:
        pass
    if isinstance(obj, Mapping):
        return {k: _to_plain(v) for k, v in obj.items()}
    if isinstance(obj, Sequence) and not isinstance(obj, (str, bytes)):
        return [_to_plain(v) for v in obj]
    return obj


def _cfg_get(cfg, *path, default=None):
    for k in path:
        if isinstance(cfg, dict) and k in cfg:
            cfg = cfg[k]
        else:
            return default
    return cfg


def model_from_checkpoint(ckpt, ema: bool = True, precision: str = "bf16", preset: str = "flowdec_75m") -> FlowModel:
    """Build a FlowModel from a loaded checkpoint object (see module docstring for the accepted layouts)."""
    if not isinstance(ckpt, dict):
        raise RuntimeError("checkpoint must be a dict")
    if "_pl_ema_state_dict" in ckpt or "state_dict" in ckpt:
        key = "_pl_ema_state_dict" if (ema and "_pl_ema_state_dict" in ckpt) else "state_dict"
        if key not in ckpt:
            raise RuntimeError(f"checkpoint has no '{key}' (available: {[k for k in ckpt if 'state_dict' in k]})")
        sd = ckpt[key]
    else:
        sd = ckpt  # bare state_dict
    hp = None
    if ckpt.get("hyper_parameters", None) is not None:
        hp = _to_plain(ckpt["hyper_parameters"])
        if not isinstance(hp, dict):   # silently falling back to the defaults would produce wrong audio for e.g. the ablation configs
            raise RuntimeError(f"checkpoint 'hyper_parameters' of type {type(ckpt['hyper_parameters']).__name__} cannot be read as a mapping")
    mcfg = _cfg_get(hp, "model") or {}
    bb_cfg = dict(BACKBONE_FINAL_NO_ATTN)
    for k, v in (mcfg.get("backbone") or {}).items():
        if k != "_target_":
            bb_cfg[k] = tuple(v) if isinstance(v, list) else v
    if not mcfg.get("backbone") and "backbone.all_modules.0.W" in sd:      # infer the width from the weights
        bb_cfg["nf"] = int(sd["backbone.all_modules.0.W"].shape[0])
    fe_cfg = mcfg.get("feature_extractor") or {}
    fe = AmplitudeCompressedComplexSTFT(window_fn=fe_cfg.get("window_fn", "hann"), n_fft=int(fe_cfg.get("n_fft", 1534)),
                                        n_hops=int(fe_cfg.get("n_hops", 4)) if "hop_length" not in fe_cfg else None,
                                        hop_length=fe_cfg.get("hop_length"), sampling_rate=int(_cfg_get(hp, "sampling_rate", default=48000)),
                                        alpha=float(fe_cfg.get("alpha", 0.3)), beta=float(fe_cfg.get("beta", 0.33)))
    sigma_y = sd["sigma_y"] if "sigma_y" in sd else from_preset(preset, nf=8).sigma_y.data
    model = FlowModel(backbone=NCSNpp(precision=precision, **bb_cfg), feature_extractor=fe,
                      sampling_rate=int(_cfg_get(hp, "sampling_rate", default=48000)), sigma_x=0.0, sigma_y=sigma_y.clone())
    res = model.load_state_dict(sd, strict=False)  # strict_loading = False in the reference (model.py:397)
    missing = [k for k in res.missing_keys if k.startswith("backbone.")]
    if missing:
        raise RuntimeError(f"checkpoint is missing backbone parameters, e.g. {missing[:3]}")
    return model.eval()


def load_from_checkpoint(path: str, map_location="cpu", ema: bool = True, precision: str = "bf16") -> FlowModel:
    """Replacement for `EnhancementModel.load_from_checkpoint(ckpt, map_location=..., ema=...)` (enhance.py:66)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model = model_from_checkpoint(ckpt, ema=ema, precision=precision)
    return model.to(map_location) if map_location is not None else model


# ------------------------------------------------------------------------------------------------
# main loop
# ------------------------------------------------------------------------------------------------
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Enhance wav files with a FlowDec postfilter on MI355X")
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--files", type=str, required=True)
    p.add_argument("--outdir", type=str, required=True)
    p.add_argument("--N", type=int, required=True)
    p.add_argument("--single-file", action="store_true")
    p.add_argument("--exclude-files-matching", type=str, required=False)
    p.add_argument("--predictor", type=str, default="reverse_diffusion")   # score model only (ignored)
    p.add_argument("--corrector", type=str, default="ald")
    p.add_argument("--snr", type=float, default=0.5)
    p.add_argument("--solver", type=str, default="midpoint")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--ema", type=lambda s: str(s).lower() not in ("0", "false", "no"), default=True)
    p.add_argument("--skip-existing", type=lambda s: str(s).lower() not in ("0", "false", "no"), default=True)
    p.add_argument("--i-min", type=int, default=None)
    p.add_argument("--i-max", type=int, default=None)
    p.add_argument("--rtf", action="store_true")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32", "mixed", "bf16x3"])
    p.add_argument("--seed", type=int, default=None, help="file i of the work list draws its initial noise from a generator seeded SEED + i "
                                                          "(default: nondeterministic like the reference)")
    p.add_argument("--batch-files", type=int, default=8, help="files of one T_pad bucket per native call (1 = one file per call, the reference's loop)")
    return p


def collect_files(files: str, single_file: bool) -> Tuple[List[str], Optional[List[str]]]:
    """-> (noisy paths, clean paths or None)."""
    if os.path.isfile(files):
        if single_file:
            return [files], None
        fl = read_list(files)
        return fl.inputs, fl.clean
    return sorted(glob.glob(f"{files}/*.wav")), None


@dataclass
class RunResult:
    """What a CLI run did: files enhanced, files skipped because they exceed the length limit of the chosen precision (but not the
    reference's 30 s), files skipped as too long for the reference as well."""
    n_done: int = 0
    n_over_precision_limit: int = 0
    n_too_long: int = 0
    gpu_seconds: float = 0.0      # --rtf: GPU time and audio duration over the files enhanced in this run
    audio_seconds: float = 0.0

    @property
    def exit_code(self) -> int:
        return 3 if self.n_over_precision_limit else 0


@dataclass
class FileJob:
    """One entry of the work list: where the input is, where the output goes, the clean reference of a pair list, and whether there
    is anything to compute (`pending` is False when the output exists and --skip-existing holds)."""
    index: int
    src: str
    dst: str
    clean: Optional[str]
    pending: bool


def plan_jobs(noisy: List[str], clean: Optional[List[str]], outdir: str, i_min: Optional[int], i_max: Optional[int],
              skip_existing: bool, exclude: Optional[str] = None):
    """The work list of a run as a generator of FileJob: the exclusion pattern first (it renumbers the list), then the inclusive
    index window [i_min, i_max] over what is left, then the exists-check."""
    entries = [(n, clean[k] if clean is not None else None) for k, n in enumerate(noisy) if exclude is None or exclude not in n]
    lo = 0 if i_min is None else max(i_min, 0)
    hi = len(entries) - 1 if i_max is None else min(i_max, len(entries) - 1)
    for index in range(lo, hi + 1):
        src, cl = entries[index]
        dst = os.path.join(outdir, os.path.basename(src))
        yield FileJob(index, src, dst, cl, pending=not (skip_existing and os.path.exists(dst)))


class GpuTimer:
    """`with GpuTimer(enabled) as t: ...` -> t.seconds = device time between entry and exit on the current stream (None when disabled)."""

    def __init__(self, enabled: bool):
        self.enabled, self.seconds = enabled, None

    def __enter__(self):
        if self.enabled:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.enabled and exc[0] is None:
            self._ev[1].record()
            torch.cuda.synchronize()
            self.seconds = self._ev[0].elapsed_time(self._ev[1]) / 1000.0
        return False


class RunLog:
    """The two side files of a run, in the reference's FORMATS (enhance.py:94,135,143): `rtfs{suffix}.csv` with the header
    path,runtime,filetime,rtf (--rtf) and `triples_list{suffix}.txt` with `clean ---> noisy ---> enhanced` lines (pair lists).
    The files are opened on __enter__ (a failing second open closes the first)."""

    def __init__(self, outdir: str, suffix: str, want_rtf: bool, want_triples: bool):
        self._paths = (os.path.join(outdir, f"rtfs{suffix}.csv") if want_rtf else None,
                       os.path.join(outdir, f"triples_list{suffix}.txt") if want_triples else None)
        self._stack = contextlib.ExitStack()
        self._rtf = self._tri = None
        self.runtime = self.filetime = 0.0

    def __enter__(self):
        with contextlib.ExitStack() as guard:
            self._rtf = guard.enter_context(open(self._paths[0], "w")) if self._paths[0] else None
            self._tri = guard.enter_context(open(self._paths[1], "w")) if self._paths[1] else None
            self._stack = guard.pop_all()
        if self._rtf:
            print("path,runtime,filetime,rtf", file=self._rtf)
        return self

    def __exit__(self, *exc):
        return self._stack.__exit__(*exc)

    def rtf(self, dst: str, runtime: float, filetime: float):
        print(runtime, filetime, "-> rtf =", runtime / filetime)
        self.runtime += runtime
        self.filetime += filetime
        if self._rtf:
            print(f"{dst},{runtime:.5f},{filetime:.5f},{runtime / filetime:.5f}", file=self._rtf)

    def triple(self, job: FileJob):
        if self._tri:
            print(f"{job.clean} ---> {job.src} ---> {job.dst}", file=self._tri)


def file_generator(model: FlowModel, seed: Optional[int], index: int):
    """The noise generator of file `index` of the work list: seeded SEED + index, so that a file's result does not depend on which
    files share its batch; None (the global device RNG, like the reference: model.py:512) without --seed."""
    return None if seed is None else torch.Generator(device=model.device).manual_seed(int(seed) + int(index))


def load_for_model(model: FlowModel, job: FileJob, res: RunResult, max_seconds: float, precision: str):
    """Load -> the reference's length rule (enhance.py:115,139) -> resample to the model rate.  -> waveform [C, L] or None (skipped)."""
    y, sr = load_wav(job.src)
    seconds = y.shape[-1] / sr
    if seconds > MAX_SECONDS:
        res.n_too_long += 1
        print("Skipping file due to length:", job.src)
        return None
    if seconds > max_seconds:
        res.n_over_precision_limit += 1
        print(f"Skipping file: {seconds:.1f} s exceeds the {max_seconds:g} s limit of precision={precision} "
              f"(the reference's limit is {MAX_SECONDS:g} s; use --precision bf16 for files up to it):", job.src)
        return None
    if sr != model.sampling_rate:
        print("RESAMPLING from", sr, "to", model.sampling_rate)
        y = resample(y, sr, model.sampling_rate)
    return y


def enhance_file(model: FlowModel, job: FileJob, args, log: RunLog, res: RunResult, max_seconds: float) -> None:
    """One file per native call (the reference's loop): load -> enhance (timed under --rtf) -> save.  Updates `res`."""
    y = load_for_model(model, job, res, max_seconds, args.precision)
    if y is None:
        return
    sr = model.sampling_rate
    # use_graph=False: every file has its own length, and a replay would not be faster anyway -- a one-clip solve is bound by the GPU,
    # not by the host's launches (profiles/r02_graph_cost.txt: eager 18.06 ms, replay 18.10 ms; capture + instantiate 2.4 ms)
    with GpuTimer(args.rtf) as timer:
        x_hat = model.enhance(y, N=args.N, solver=args.solver, generator=file_generator(model, args.seed, job.index), use_graph=False)
    if timer.seconds is not None:
        log.rtf(job.dst, timer.seconds, y.shape[-1] / sr)
    save_wav(job.dst, x_hat.cpu(), sr)
    res.n_done += 1


def plan_batches(model: FlowModel, jobs: List[FileJob], batch_files: int):
    """Buckets the pending jobs by the frame count their spectrogram pads to (from the wav HEADERS: nothing is decoded here) and cuts
    every bucket into batches of at most `batch_files` files, in work-list order.  Multi-channel files, unreadable headers and files
    the length rule will skip go through the one-file path (their own messages).  -> list of lists of FileJob."""
    from .model import padded_frames_of
    hop = model.feature_extractor._cfg()["hop"]
    buckets, singles = {}, []
    for job in jobs:
        try:
            n, sr, channels = wav_info(job.src)
        except Exception:   # an unreadable file fails in its own one-file call, with the reference's behaviour (an exception)
            singles.append([job]); continue
        if channels != 1 or n / sr > MAX_SECONDS or batch_files <= 1:
            singles.append([job]); continue
        buckets.setdefault(padded_frames_of(resampled_length(n, sr, model.sampling_rate), hop), []).append(job)
    batches = []
    for tp in sorted(buckets):
        group = buckets[tp]
        batches += [group[i:i + batch_files] for i in range(0, len(group), batch_files)]
    return batches + singles


def enhance_batch_files(model: FlowModel, batch: List[FileJob], args, log: RunLog, res: RunResult, max_seconds: float) -> None:
    """One ragged native call for the files of `batch` (same T_pad bucket).  Every output equals the one-file call bit for bit."""
    loaded = [(job, load_for_model(model, job, res, max_seconds, args.precision)) for job in batch]
    loaded = [(job, y) for job, y in loaded if y is not None]
    if not loaded:
        return
    sr = model.sampling_rate
    gens = [file_generator(model, args.seed, job.index) for job, _ in loaded]
    with GpuTimer(args.rtf) as timer:
        outs = model.enhance_batch([y for _, y in loaded], N=args.N, solver=args.solver, generator=gens)
    total = sum(y.shape[-1] for _, y in loaded) / sr
    for (job, y), x_hat in zip(loaded, outs):
        if timer.seconds is not None:   # the batch's time, split in proportion to the files' durations
            log.rtf(job.dst, timer.seconds * (y.shape[-1] / sr) / total, y.shape[-1] / sr)
        save_wav(job.dst, x_hat.cpu(), sr)
        res.n_done += 1


def main(argv=None, model: Optional[FlowModel] = None) -> int:
    """Runs the CLI and returns the number of files enhanced (the detailed result: `run()`)."""
    return run(argv, model).n_done


def cli(argv=None) -> int:
    """Console entry point: exit status 0, or 3 when files were skipped only because of the chosen precision's length limit."""
    return run(argv).exit_code


def run(argv=None, model: Optional[FlowModel] = None) -> RunResult:
    args = build_parser().parse_args(argv)
    os.makedirs(args.outdir, exist_ok=True)
    if model is None:
        print("Loading model from checkpoint...")
        model = load_from_checkpoint(args.ckpt, map_location=args.device, ema=args.ema, precision=args.precision)
        print("Done loading model.")
    noisy, clean = collect_files(args.files, args.single_file)
    max_seconds = min(MAX_SECONDS, PRECISION_MAX_SECONDS.get(args.precision, MAX_SECONDS))
    print(f"flowdec_amd: precision={args.precision} ({PRECISION_NOTE[args.precision]}), solver={args.solver}, N={args.N}, "
          f"files per native call <= {max(args.batch_files, 1)}")
    res = RunResult()
    suffix = f"_{args.i_min}-{args.i_max}" if args.i_max else ""
    jobs = list(plan_jobs(noisy, clean, args.outdir, args.i_min, args.i_max, args.skip_existing, args.exclude_files_matching))
    batchable = args.solver in ("euler", "midpoint", "heun2", "heun2_eulerlast")   # (the adaptive solvers step clip by clip)
    with RunLog(args.outdir, suffix, want_rtf=args.rtf, want_triples=clean is not None) as log:
        for batch in plan_batches(model, [j for j in jobs if j.pending], args.batch_files if batchable else 1):
            if len(batch) == 1:
                enhance_file(model, batch[0], args, log, res, max_seconds)
            else:
                enhance_batch_files(model, batch, args, log, res, max_seconds)
        for job in jobs:      # the pair list names every file of the window, enhanced now or found existing (enhance.py:143)
            log.triple(job)
        if args.rtf and log.runtime > 0:
            print(f"total: {log.filetime:.2f} s of audio in {log.runtime:.3f} s of GPU time -> rtf = {log.runtime / log.filetime:.5f} "
                  f"({log.filetime / log.runtime:.1f} x real time)")
    res.gpu_seconds, res.audio_seconds = log.runtime, log.filetime
    return res


if __name__ == "__main__":
    sys.exit(cli())
