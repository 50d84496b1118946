#!/usr/bin/env python
"""The bf16 mode's DERIVED tolerance for the nf = 8 goldens (G8: one forward at two time settings, G9: enhance with four solvers), like
make_golden_bf16_prediction.py does for the full-width ones: the oracle with bf16-rounded conv operands and bf16-rounded stored tensors
in float32 NumPy arithmetic, against the reference outputs stored in the goldens.  Writes g22_bf16_prediction_nf8.json;
tests/test_hip_model.py holds the HIP bf16 mode to 1.3 x (one forward) / 1.6 x (one waveform draw) of these numbers instead of the
fitted 3e-2 / 0.13 of rounds 1-4.  CPU only, a few minutes:  python tests/golden/make_golden_bf16_prediction_nf8.py && python tests/golden/make_golden_bf16_prediction_nf8.py --25s"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flowdec_oracle as O  # noqa: E402

rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
KW = dict(operand_round="bf16", storage_round="bf16")
PATH = os.path.join(HERE, "g22_bf16_prediction_nf8.json")
if "--25s" in sys.argv:   # only the draw of tests/test_hip_configs.py::test_flowdec_25s_midpoint_vs_oracle (its truth is the f32 oracle), merged into the json
    sig = np.load(os.path.join(HERE, "g12_sigma_y.npz"))["25s"]
    sd = O.random_state_dict(seed=8, nf=8)
    rng = np.random.default_rng(25)
    L = 19200
    y = (0.1 * rng.standard_normal((2, 1, L))).astype(np.float32)
    Tp = O.padded_frames(O.num_frames(L))
    noise = ((rng.standard_normal((2, 1, 768, Tp)) + 1j * rng.standard_normal((2, 1, 768, Tp))) / np.sqrt(2)).astype(np.complex64)
    ref = O.enhance(O.NCSNppOracle(sd, nf=8), y, noise, sig, N=3, solver="midpoint")
    xh = O.enhance(O.NCSNppOracle(sd, nf=8, **KW), y, noise, sig, N=3, solver="midpoint")
    j = json.load(open(PATH))
    j["enhance_rel_l2"]["flowdec_25s_midpoint_N3"] = rel(xh, ref)
    print("flowdec_25s midpoint N=3: predicted", j["enhance_rel_l2"]["flowdec_25s_midpoint_N3"], flush=True)
    json.dump(j, open(PATH, "w"), indent=1)
    sys.exit(0)
out = {"forward_rel_l2": {}, "enhance_rel_l2": {}}
g8 = np.load(os.path.join(HERE, "g8_ncsnpp_nf8.npz"))
sd8 = O.random_state_dict(seed=int(g8["seed"]), nf=8)
for key, t in (("out_t025", np.array([0.25, 0.25], np.float32)), ("out_t01_09", np.array([0.1, 0.9], np.float32))):
    t0 = time.time()
    y = O.NCSNppOracle(sd8, nf=8, **KW).forward(g8["x"], g8["y"], t)
    out["forward_rel_l2"][key] = rel(y, g8[key])
    print(f"forward {key}: predicted rel L2 err {out['forward_rel_l2'][key]:.3e}  ({time.time() - t0:.0f} s)", flush=True)
g9 = np.load(os.path.join(HERE, "g9_enhance_nf8.npz"))
sd9 = O.random_state_dict(seed=int(g9["seed"]), nf=8)
for solver, N in (("euler", 6), ("midpoint", 3), ("heun2", 3), ("heun2_eulerlast", 3)):
    t0 = time.time()
    xh = O.enhance(O.NCSNppOracle(sd9, nf=8, **KW), g9["y"], g9["noise"], g9["sigma_y"],
                   N=N, solver=solver)
    out["enhance_rel_l2"][f"{solver}_N{N}"] = rel(xh, g9[f"{solver}_N{N}"])
    print(f"enhance {solver} N={N}: predicted waveform rel L2 err {out['enhance_rel_l2'][f'{solver}_N{N}']:.3e}  ({time.time() - t0:.0f} s)", flush=True)
out["golden"] = "g8_ncsnpp_nf8.npz / g9_enhance_nf8.npz"
out["note"] = "oracle/flowdec_oracle.py NCSNppOracle(operand_round='bf16', storage_round='bf16')"
json.dump(out, open(PATH, "w"), indent=1)
