#!/usr/bin/env python
"""What error SHOULD the bf16 mode have?  One full-width (nf = 64) forward of the oracle on the golden G10 input with the bf16 mode's
roundings modelled in float32 NumPy arithmetic -- conv operands rounded to bf16 (operand_round) and every tensor the mode stores
between kernels rounded to bf16 (storage_round) -- against the reference output stored in G10.  Writes g19_bf16_prediction.json;
tests/test_hip_model.py::test_bf16_error_is_the_predicted_one holds the HIP bf16 forward to 1.3 x the prediction (the factor covers
summation order, the fused SiLU's transcendental ulps and the fp16-operand Winograd launches, which round LESS than the model here).
CPU only, ~5 min:  python tests/golden/make_golden_bf16_prediction.py
`--cfg2clip` / `--cfg3clip` (round 5, ~15 min each): ONLY the prediction for G21 (one 2 s clip, T_pad = 256, Euler-6: the image size
bench.py times) / G23 (the same for FlowDec-25s, midpoint N = 3), merged into the existing json as enhance_rel_l2_cfg2clip / _cfg3clip."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flowdec_oracle as O  # noqa: E402

rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


def g21_inputs(g21):
    """(y, noise) of G21: y is stored, the noise is re-drawn in the generator's order (make_golden_nf64_enhance.py --cfg2clip)."""
    rng = np.random.default_rng(int(g21["rng_seed"]))
    y = (0.1 * rng.standard_normal(g21["y"].shape)).astype(np.float32)
    assert np.array_equal(y, g21["y"])
    shape = (1, 1, 768, O.padded_frames(O.num_frames(y.shape[-1])))
    noise = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)
    assert abs(noise.astype(np.complex128).sum() - complex(g21["noise_sum"])) < 1e-6
    return y, noise


CLIPS = {"--cfg2clip": ("g21_enhance_nf64_cfg2clip.npz", "euler", 6, "enhance_rel_l2_cfg2clip"),
         "--cfg3clip": ("g23_enhance_nf64_cfg3clip.npz", "midpoint", 3, "enhance_rel_l2_cfg3clip"),
         "--cfg4clip": ("g24_enhance_nf64_cfg4clip.npz", "midpoint", 3, "enhance_rel_l2_cfg4clip")}
for flag, (fname, solver, N, table) in CLIPS.items():
    if flag in sys.argv:
        gc = np.load(os.path.join(HERE, fname))
        yc, nzc = g21_inputs(gc)
        t0 = time.time()
        net = O.NCSNppOracle(O.random_state_dict(seed=int(gc["seed"]), nf=64), nf=64, operand_round="bf16", storage_round="bf16")
        xh = O.enhance(net, yc, nzc, gc["sigma_y"], N=N, solver=solver)
        e = rel(xh, gc[f"{solver}_N{N}"])
        print(f"{flag[2:]} enhance {solver} N={N}: predicted waveform rel L2 err {e:.3e}  ({time.time() - t0:.0f} s)", flush=True)
        path = os.path.join(HERE, "g19_bf16_prediction.json")
        j = json.load(open(path))
        j[table] = {f"{solver}_N{N}": e}
        json.dump(j, open(path, "w"), indent=1)
        sys.exit(0)

g = np.load(os.path.join(HERE, "g10_ncsnpp_nf64.npz"))
sd = O.random_state_dict(seed=int(g["seed"]), nf=64)
t = np.array([float(g["t"])] if "t" in g.files else [0.5], np.float32)
out = {}
for name, kw in (("f32", {}), ("bf16_operands", dict(operand_round="bf16")), ("bf16_operands_and_storage", dict(operand_round="bf16", storage_round="bf16"))):
    t0 = time.time()
    y = O.NCSNppOracle(sd, nf=64, **kw).forward(g["x"], g["y"], t)
    out[name] = rel(y, g["out"])
    print(f"{name:28s} rel L2 err vs reference {out[name]:.3e}  ({time.time() - t0:.0f} s)", flush=True)
# the same model through the whole path: enhance() on the 0.5 s clip of G17 (the reference's own FlowModel.enhance at full width)
g17 = np.load(os.path.join(HERE, "g17_enhance_nf64.npz"))
sd17 = O.random_state_dict(seed=int(g17["seed"]), nf=64)
wave = {}
for solver, N in (("euler", 6), ("midpoint", 3)):
    t0 = time.time()
    net = O.NCSNppOracle(sd17, nf=64, operand_round="bf16", storage_round="bf16")
    xh = O.enhance(net, g17["y"], g17["noise"], g17["sigma_y"], N=N, solver=solver)
    wave[f"{solver}_N{N}"] = rel(xh, g17[f"{solver}_N{N}"])
    print(f"enhance {solver} N={N}: predicted waveform rel L2 err {wave[f'{solver}_N{N}']:.3e}  ({time.time() - t0:.0f} s)", flush=True)
json.dump(dict(golden="g10_ncsnpp_nf64.npz / g17_enhance_nf64.npz", forward_rel_l2=out, enhance_rel_l2=wave,
               note="oracle/flowdec_oracle.py NCSNppOracle(operand_round, storage_round)"),
          open(os.path.join(HERE, "g19_bf16_prediction.json"), "w"), indent=1)
