"""SwiGLU gate: silu(x[..., :I]) * x[..., I:] (reference: nanovllm/layers/activation.py:36-58);
kernel nvh_silu_mul in csrc/elementwise.hip."""
import torch
from torch import nn

from .. import ops


class SiluAndMul(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_mul(x)
