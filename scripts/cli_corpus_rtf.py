#!/usr/bin/env python
"""The reference's use case -- a DIRECTORY of files of different lengths (enhance.py:96-137) -- through enhance_cli with and without
ragged batching: a synthetic corpus (default 64 files, 1-4 s, 48 kHz), a full-width synthetic Lightning checkpoint, `--rtf`, and the
aggregate audio-seconds per GPU-second of   --batch-files 1 (one file per call = the reference's loop)   vs   --batch-files 8 / 16.
Every output file of the batched runs is compared byte for byte with the one-file-per-call run.

    python scripts/cli_corpus_rtf.py [--files 64] [--min-s 1] [--max-s 4] [--N 6] [--solver euler] [--out gpurun_out/r06_cli_corpus_rtf.json]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--min-s", type=float, default=1.0)
    ap.add_argument("--max-s", type=float, default=4.0)
    ap.add_argument("--N", type=int, default=6)
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 16])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "mixed"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_cli_corpus_rtf.json"))
    args = ap.parse_args()
    import flowdec_amd
    from flowdec_amd import enhance_cli
    from flowdec_amd.model import padded_frames_of

    tmp = tempfile.mkdtemp(prefix="fd_corpus_")
    try:
        # full-width FlowDec-75m with seeded random weights, saved in the Lightning layout the reference's checkpoints have
        m = flowdec_amd.from_preset("flowdec_75m", precision=args.precision)
        g = torch.Generator().manual_seed(1234)
        sd = {}
        for k, v in m.state_dict().items():
            if k.endswith(".W"):
                sd[k] = torch.randn(v.shape, generator=g) * 16.0
            elif k.startswith("backbone.") and v.ndim == 1 and k.endswith("weight"):
                sd[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
            elif k.startswith("backbone.") and k.endswith("bias"):
                sd[k] = 0.05 * torch.randn(v.shape, generator=g)
            elif k.startswith("backbone."):
                sd[k] = torch.randn(v.shape, generator=g) / v[0].numel() ** 0.5
            else:
                sd[k] = v.clone()
        ckpt = os.path.join(tmp, "flowdec_75m_synthetic.ckpt")
        torch.save({"_pl_ema_state_dict": sd, "state_dict": sd}, ckpt)
        del m
        ind = os.path.join(tmp, "in")
        os.makedirs(ind)
        rng = np.random.default_rng(0)
        lens = rng.integers(int(args.min_s * 48000), int(args.max_s * 48000) + 1, size=args.files)
        for i, n in enumerate(lens):
            enhance_cli.save_wav(os.path.join(ind, f"clip{i:03d}.wav"), torch.from_numpy((0.1 * rng.standard_normal((1, int(n)))).astype(np.float32)), 48000)
        buckets = {}
        for n in lens:
            buckets[padded_frames_of(int(n))] = buckets.get(padded_frames_of(int(n)), 0) + 1
        audio = float(lens.sum()) / 48000
        fill = float(sum(lens)) / sum(384.0 * padded_frames_of(int(n)) for n in lens)
        model = enhance_cli.load_from_checkpoint(ckpt, map_location="cuda:0", precision=args.precision)
        res = {"files": int(args.files), "audio_seconds": audio, "lengths_s": [float(args.min_s), float(args.max_s)], "N": args.N, "solver": args.solver,
               "precision": args.precision, "files_per_T_pad_bucket": {str(k): v for k, v in sorted(buckets.items())},
               "samples_over_padded_frames": fill, "runs": {}}
        ref_dir = None
        for bf in args.batches:
            outd = os.path.join(tmp, f"out{bf}")
            argv = ["--ckpt", ckpt, "--files", ind, "--outdir", outd, "--N", str(args.N), "--solver", args.solver, "--rtf", "--seed", "5", "--batch-files", str(bf)]
            enhance_cli.run(argv, model=model)                 # warm-up pass (graph capture of every (B, T_pad) bucket, allocator)
            shutil.rmtree(outd)
            t0 = time.perf_counter()
            r = enhance_cli.run(argv, model=model)
            wall = time.perf_counter() - t0
            same = None
            if ref_dir is None:
                ref_dir = outd
            else:
                same = all(open(os.path.join(ref_dir, f), "rb").read() == open(os.path.join(outd, f), "rb").read() for f in sorted(os.listdir(ind)))
            res["runs"][f"batch_files_{bf}"] = {"files_done": r.n_done, "gpu_seconds": r.gpu_seconds, "audio_seconds_per_gpu_second": r.audio_seconds / r.gpu_seconds,
                                               "wall_seconds_incl_file_io": wall, "audio_seconds_per_wall_second": r.audio_seconds / wall,
                                               "outputs_identical_to_one_file_per_call": same}
            print(f"--batch-files {bf}: {r.n_done} files, {r.audio_seconds:.1f} s of audio in {r.gpu_seconds:.3f} GPU-s = "
                  f"{r.audio_seconds / r.gpu_seconds:.1f} x real time (wall incl. wav I/O: {wall:.2f} s); identical to one-file-per-call: {same}", flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps(res))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
