"""Per-step attention metadata shared between the runner and the operator modules.

Mirrors nanovllm/utils/context.py:22-111 of the reference (same eight fields, same
positional order in set_context) so callers and operator modules written against
the reference keep working.  One record per process, replaced every step.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

FIELDS = ("cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "slot_mapping", "context_lens",
          "block_tables", "seq_need_compute_logits")


@dataclass
class Context:
    cu_seqlens_q: torch.Tensor | None = None      # int32 [B+1]; None selects the one-query-per-sequence decode mode
    cu_seqlens_k: torch.Tensor | None = None      # int32 [B+1]
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    slot_mapping: torch.Tensor | None = None      # int32 [T], -1 = no store
    context_lens: torch.Tensor | None = None      # int32 [B]
    block_tables: torch.Tensor | None = None      # int32 [B, W], -1 padded (0 padded in graph buffers)
    seq_need_compute_logits: torch.Tensor | None = None   # int32 [B']


_current = Context()


def get_context() -> Context:
    return _current


def set_context(*args, **kwargs) -> None:
    """set_context(cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, slot_mapping,
    context_lens, block_tables, seq_need_compute_logits); every argument optional."""
    global _current
    if len(args) > len(FIELDS):
        raise TypeError("set_context takes at most 8 positional arguments")
    values = dict(zip(FIELDS, args))
    for key, val in kwargs.items():
        if key not in FIELDS or key in values:
            raise TypeError(f"set_context: bad argument {key!r}")
        values[key] = val
    _current = Context(**values)


def reset_context() -> None:
    global _current
    _current = Context()
