#!/usr/bin/env python
"""Feasibility check for DESIGN.md section 8 item 1 (CPU, NumPy): does Winograd F(2x2, 3x3) with bf16 GEMM operands stay
inside the bf16 forward tolerance?  Runs the nf = 8 NCSN++ oracle against the reference golden G8 with three 3x3 conv
implementations: direct f32, direct with bf16-rounded operands (what the shipped HIP bf16 mode does), and Winograd with
the transformed input V = B^T d B and the transformed weights U = G g G^T rounded to bf16 (f32 transforms, f32 accumulate)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flowdec_oracle as O

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
direct = O.conv2d

def winograd_bf16(x, w, b, operand_round=None):
    if w.shape[2] != 3 or x.shape[2] % 2 or x.shape[3] % 2:
        return direct(x, w, b, operand_round="bf16")
    Bn, Ci, H, W = x.shape
    Co = w.shape[0]
    xb = O.round_bf16(x.astype(np.float32))                              # activations are stored as bf16
    xp = np.pad(xb, ((0, 0), (0, 0), (1, 1), (1, 1)))
    th, tw = H // 2, W // 2
    d = np.empty((Bn, Ci, th, tw, 4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            d[..., i, j] = xp[:, :, i:i + H:2, j:j + W:2]
    V = O.round_bf16(np.einsum("ij,bctujk,lk->bctuil", BT, d, BT, optimize=True))
    U = O.round_bf16(np.einsum("ij,ocjk,lk->ocil", G, w.astype(np.float32), G, optimize=True))
    M = np.einsum("ocil,bctuil->botuil", U, V, optimize=True)            # f32 accumulate
    Y = np.einsum("ij,botujk,lk->botuil", AT, M, AT, optimize=True)      # [B,Co,th,tw,2,2]
    out = Y.transpose(0, 1, 2, 4, 3, 5).reshape(Bn, Co, H, W)
    return out + (b[None, :, None, None] if b is not None else 0)

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g8_ncsnpp_nf8.npz"))
sd = O.random_state_dict(seed=int(g["seed"]), nf=8)
rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
for name, fn in (("direct f32", None), ("direct, bf16 operands", lambda x, w, b, operand_round=None: direct(x, w, b, operand_round="bf16")),
                 ("Winograd F(2x2,3x3), bf16 U and V", winograd_bf16)):
    O.conv2d = fn if fn is not None else direct
    net = O.NCSNppOracle(sd, nf=8)
    out = net.forward(g["x"], g["y"], np.array([0.25], np.float32))
    print(f"{name:36s} rel L2 error vs reference golden = {rel(out, g['out_t025']):.3e}")
O.conv2d = direct
