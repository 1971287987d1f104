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

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is None:
            return ops.rmsnorm(x, self.weight.data, self.eps)
        # the residual stream is updated in place: new_residual aliases `residual`
        return ops.add_rmsnorm(x, residual, self.weight.data, self.eps)

    def forward_from_partials(self, partials: torch.Tensor, residual: torch.Tensor):
        """Same as forward(x, residual) with x = bf16(sum of fp32 split-K slabs [S, rows, dim])."""
        return ops.add_rmsnorm(None, residual, self.weight.data, self.eps, partials=partials)
