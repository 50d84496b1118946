#!/usr/bin/env python
"""Where the bf16 mode's error comes from (CPU, NumPy oracle, full width): one forward of the nf = 64 network on the golden G10
input with (a) f32 everywhere, (b) ONLY the conv operands rounded to bf16 = what a 'bf16 operands / f32 residual stream'
storage mode would give, against the reference output; the HIP bf16 mode (bf16 operands AND bf16 activation storage) and the
Winograd path (fp16 operands, bf16 storage) are measured by tests/test_hip_*.py on the same golden."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flowdec_oracle as O

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g10_ncsnpp_nf64.npz"))
sd = O.random_state_dict(seed=int(g["seed"]), nf=64)
rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
for name, rnd in (("f32 everywhere", None), ("bf16 conv operands only (f32 storage)", "bf16")):
    t0 = time.time()
    out = O.NCSNppOracle(sd, nf=64, operand_round=rnd).forward(g["x"], g["y"], np.array([0.5], np.float32))
    print(f"{name:42s} rel err vs reference {rel(out, g['out']):.3e}   ({time.time() - t0:.0f} s)", flush=True)
