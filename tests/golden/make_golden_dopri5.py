#!/usr/bin/env python
"""Fixture for solver='dopri5' / 'tsit5' (adaptive 5(4) pairs).  NOT a reference output: torchdyn is not installed, so this is
produced by the ORACLE's restatement of torchdyn's controller (oracle.odeint_dopri5, parity unpinned) on the inputs of
g9_enhance_nf8.npz; it lets the GPU test check the HIP driver without running ~60 oracle forward passes on the GPU box.

    python tests/golden/make_golden_dopri5.py     # writes tests/golden/g16_dopri5_oracle_nf8.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flowdec_oracle as O  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "g9_enhance_nf8.npz"))
    y, nz = g["y"][:1], g["noise"][:1]
    net = O.NCSNppOracle(O.random_state_dict(seed=int(g["seed"]), nf=8), nf=8)
    Y, info = O.preprocess(y)
    x0 = O.initial_state(Y, g["sigma_y"], nz)
    f = lambda t, X: net.forward(X, Y, np.asarray([t], dtype=np.float32))
    out = {}
    for tol in (1e-3,):
        traj, nfe = O.odeint_dopri5(f, x0, O.t_span_linspace(2), atol=tol, rtol=tol, return_traj=True)
        out["wave_tol1e-3"] = O.postprocess(traj[-1], info)
        out["mid_feat_norm"] = np.float64(np.linalg.norm(traj[1].ravel()))
        out["nfe_tol1e-3"] = np.int64(nfe)
        print("tol", tol, "nfe", nfe)
        traj, nfe = O.odeint_adaptive(f, x0, O.t_span_linspace(2), "tsit5", atol=tol, rtol=tol, return_traj=True)
        out["tsit5_wave_tol1e-3"] = O.postprocess(traj[-1], info)
        out["tsit5_mid_feat_norm"] = np.float64(np.linalg.norm(traj[1].ravel()))
        out["tsit5_nfe_tol1e-3"] = np.int64(nfe)
        print("tsit5 tol", tol, "nfe", nfe)
    np.savez_compressed(os.path.join(HERE, "g16_dopri5_oracle_nf8.npz"), **out)


if __name__ == "__main__":
    main()
