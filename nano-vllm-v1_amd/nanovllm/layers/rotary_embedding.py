"""Rotary position embedding (half-split rotation), reference interface
nanovllm/layers/rotary_embedding.py:92-205; kernel nvh_rope in csrc/elementwise.hip.

The cos|sin table is built on the HOST in fp32 with the reference's formula and then
moved to the device, so every rank and the CPU oracle rotate with bit-identical tables.
"""
from functools import lru_cache

import torch
from torch import nn

from .. import ops


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float) -> None:
        super().__init__()
        assert rotary_dim == head_size
        self.head_size = head_size
        cpu = torch.device("cpu")
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float, device=cpu) / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float, device=cpu)
        freqs = torch.outer(t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze_(1)      # [max_pos, 1, D]
        self.register_buffer("cos_sin_cache", cache.to(torch.empty(0).device), persistent=False)

    def table2d(self) -> torch.Tensor:
        return self.cos_sin_cache.view(self.cos_sin_cache.shape[0], self.head_size)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """Rotates query [T,Hq,D] and key [T,Hkv,D] IN PLACE and returns them."""
        ops.rope_(positions, query, key, self.table2d())
        return query, key


@lru_cache(1)
def _cached_rope(head_size: int, rotary_dim: int, max_position: int, base: float):
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, rope_scaling: dict | None = None):
    """One shared module per process (all layers use the same table)."""
    if rope_scaling is not None:
        kind = rope_scaling.get("rope_type", "default")
        assert kind == "default", f"rope_type '{kind}' is not supported"
    return _cached_rope(head_size, rotary_dim, max_position, float(base))
