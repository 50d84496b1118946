"""Drop-in for `flowdec.backbones.ncsnpp_utils.op` (op/__init__.py:24-25): the reference's two native
operators, backed by the HIP C ABI instead of JIT-built CUDA."""
import math

import torch
from torch import nn

from .. import ops as _ops


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:169-180 -- input [N, C, H, W] on the GPU, kernel [kh, kw]."""
    if input.device.type != "cuda":
        raise RuntimeError("flowdec_amd.op.upfirdn2d: input must be a GPU tensor")
    n, c, h, w = input.shape
    x = input.reshape(-1, h, w, 1)  # op/upfirdn2d.py:123
    out = _ops.upfirdn2d_raw(x, kernel.to(input.device), up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return out.view(-1, c, out.shape[1], out.shape[2])


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """op/fused_act.py:110-121 (forward only)."""
    return _ops.fused_bias_act(input, bias, act=3, alpha=negative_slope, scale=scale).to(input.dtype)


class FusedLeakyReLU(nn.Module):
    """op/fused_act.py:97-107."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
