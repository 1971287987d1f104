"""Tensor-parallel linear layers over the gfx950 GEMM (nvh_gemm_bf16_nt, csrc/gemm.hip).

Same classes, constructor arguments, ``.weight [out, in]`` / ``.bias`` parameters and
per-parameter ``weight_loader(param, loaded_weight[, shard_id])`` hooks as the reference's
nanovllm/layers/linear.py:44-411, so checkpoints load the same way:

  column-parallel  rows of W are split over ranks (QKV: per-rank layout [Q | K | V];
                   merged gate/up: [gate shard | up shard]);
  row-parallel     columns of W are split; every rank produces a partial sum that is
                   all-reduced over RCCL (bias added on rank 0 only).

Unlike the reference, a process group is optional: without one the layer is rank 0 of 1.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from .. import comm, ops


class SplitKOutput:
    """Unreduced result of a split-K projection: fp32 slabs [S, M, N] whose sum (rounded to bf16) is
    the layer output.  RMSNorm.forward accepts it in place of x and sums the slabs in its own
    prologue, which saves the separate reduction launch."""

    __slots__ = ("slabs",)

    def __init__(self, slabs: torch.Tensor):
        self.slabs = slabs

    def reduce(self) -> torch.Tensor:
        return ops.splitk_reduce(self.slabs)


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0
    return numerator // denominator


def tp_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def tp_world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim: int | None = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank = tp_rank()
        self.tp_size = tp_world()
        self.weight = nn.Parameter(torch.empty(output_size, input_size), requires_grad=False)
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size), requires_grad=False)
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def _bias(self):
        return None if self.bias is None else self.bias.data

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x, self.weight.data, self._bias())


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, divide(output_size, tp_world()), bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        rows = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * rows, rows))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x, self.weight.data, self._bias())


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices sharing one input, stored as one weight
    (gate_proj + up_proj -> gate_up_proj); shard ids are the 0-based matrix index."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class QKVParallelLinear(ColumnParallelLinear):
    """Fused Q/K/V projection; per-rank rows [Hq/tp * D | Hkv/tp * D | Hkv/tp * D], shard ids 'q','k','v'."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        tp = tp_world()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp)
        self.num_kv_heads = divide(total_num_kv_heads, tp)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        q_rows, kv_rows = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class RowParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(divide(input_size, tp_world()), output_size, bias, 1)
        # set by the model wiring when the consumer is an RMSNorm that can sum split-K slabs
        self.defer_reduce = False

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        cols = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * cols, cols))

    def forward(self, x: torch.Tensor):
        if self.defer_reduce and self.tp_size == 1 and self.bias is None:
            m = x.numel() // x.shape[-1]
            split_k = ops.suggest_split_k(m, self.weight.shape[0], self.weight.shape[1])
            if split_k > 1:
                return SplitKOutput(ops.linear_splitk(x, self.weight.data, split_k))
        y = ops.linear(x, self.weight.data, self._bias() if self.tp_rank == 0 else None)
        if self.tp_size > 1:
            y = comm.all_reduce(y)
        return y
