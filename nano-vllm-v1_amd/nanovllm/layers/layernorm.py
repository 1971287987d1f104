"""RMSNorm operator module; arithmetic in csrc/elementwise.hip (nvh_rmsnorm / nvh_add_rmsnorm).

Interface of the reference's nanovllm/layers/layernorm.py:48-165: ``RMSNorm(hidden_size,
eps)``, ``.weight``, ``forward(x)`` -> y, ``forward(x, residual)`` -> (y, new_residual).
"""
import torch
from torch import nn

from .. import ops


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size), requires_grad=False)

    def forward(self, x, residual: torch.Tensor | None = None):
        """x: tensor, or a SplitKOutput (unreduced projection; only with a residual)."""
        slabs = getattr(x, "slabs", None)
        if residual is None:
            if slabs is not None:
                x = x.reduce()
            return ops.rmsnorm(x, self.weight.data, self.eps)
        # the residual stream is updated in place: new_residual aliases `residual`
        if slabs is not None:
            return ops.add_rmsnorm(None, residual, self.weight.data, self.eps, partials=slabs)
        return ops.add_rmsnorm(x, residual, self.weight.data, self.eps)
