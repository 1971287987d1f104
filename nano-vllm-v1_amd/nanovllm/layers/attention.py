"""Paged attention operator module (reference interface: nanovllm/layers/attention.py:93-249).

``Attention(num_heads, head_dim, scale, num_kv_heads)`` with ``k_cache`` / ``v_cache``
attributes that the runner binds to per-layer views [pages, 256, Hkv, D] of the KV pool, and
``forward(q, k, v) -> o`` reading the per-step Context:

  cu_seqlens_q is None        one query per sequence over the cache (graph-replay decode)
                              -> nvh_attn_decode (csrc/attention_decode.hip)
  block_tables is not None    var-len queries over the paged cache (chunked prefill, prefix
                              hits, mixed batches, eager decode) -> nvh_attn_varlen, or
                              nvh_attn_decode when every sequence has exactly one query
  block_tables is None        var-len self-attention over the packed new tokens (warm-up,
                              fresh un-chunked prefill) -> nvh_attn_varlen

New K/V rows are written to the cache first (nvh_store_kvcache) whenever a cache is bound.
"""
import torch
from torch import nn

from .. import ops
from ..utils.context import get_context


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor) -> None:
    """key/value [N, Hkv, D] (token stride free) -> cache rows slot_mapping[n]; -1 skips."""
    ops.store_kvcache(key, value, k_cache, v_cache, slot_mapping)


class Attention(nn.Module):
    def __init__(self, num_heads, head_dim, scale, num_kv_heads):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale
        self.num_kv_heads = num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])
        self.decode_splits = 0          # 0 = heuristic per call
        self.max_context = None
        self.split_fn = None            # (batch, longest context) -> (key ranges, minimum keys per range); the runner
                                        # installs one policy for eager and graph steps

    def attend(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """Attention proper, with the new K/V already in the cache (or no cache bound)."""
        ctx = get_context()
        splits, min_range = (self.split_fn(q.shape[0], ctx.max_seqlen_k or 0) if self.split_fn is not None
                             else (self.decode_splits, 0))
        if ctx.cu_seqlens_q is None:
            return ops.attn_decode(q, self.k_cache, self.v_cache, ctx.context_lens, ctx.block_tables, self.scale,
                                   splits, self.max_context, min_range)
        if ctx.block_tables is not None:
            if ctx.max_seqlen_q == 1 and ctx.context_lens is not None:
                return ops.attn_decode(q, self.k_cache, self.v_cache, ctx.context_lens, ctx.block_tables, self.scale,
                                       splits, ctx.max_seqlen_k, min_range)
            return ops.attn_varlen(q, self.k_cache, self.v_cache, ctx.cu_seqlens_q, ctx.cu_seqlens_k,
                                   ctx.max_seqlen_q, self.scale, ctx.block_tables, ctx.max_seqlen_k or None)
        return ops.attn_varlen(q, k, v, ctx.cu_seqlens_q, ctx.cu_seqlens_k, ctx.max_seqlen_q, self.scale,
                               None, ctx.max_seqlen_k or None)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        ctx = get_context()
        if self.k_cache.numel() and self.v_cache.numel() and ctx.slot_mapping is not None and ctx.slot_mapping.numel():
            store_kvcache(k, v, self.k_cache, self.v_cache, ctx.slot_mapping)
        return self.attend(q, k, v)
