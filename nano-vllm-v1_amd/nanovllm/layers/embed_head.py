"""Vocabulary-parallel embedding and LM head (reference: nanovllm/layers/embed_head.py:33-211).

Embedding: each rank holds vocab rows [start, end); ids outside produce zero rows and the
partial results are summed with an all-reduce.  LM head: only the LAST scheduled token of each
sequence that finishes its prompt this step is projected (rows cu_seqlens_q[j+1]-1 for j in
the context's need list; an empty list selects every sequence), then vocab shards are gathered
on rank 0.  Kernels: nvh_embedding, nvh_gather_last_rows (csrc/elementwise.hip), nvh_gemm_bf16_nt.
"""
import torch
from torch import nn

from .. import comm, ops
from ..utils.context import get_context
from .linear import tp_rank, tp_world


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank, self.tp_size = tp_rank(), tp_world()
        assert num_embeddings % self.tp_size == 0
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = num_embeddings // self.tp_size
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim), requires_grad=False)
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        rows = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * rows, rows))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = ops.embedding(x, self.weight.data, self.vocab_start_idx)
        if self.tp_size > 1:
            y = comm.all_reduce(y)
        return y


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        assert not bias
        super().__init__(num_embeddings, embedding_dim)

    def forward(self, x: torch.Tensor):
        ctx = get_context()
        need = ctx.seq_need_compute_logits
        if need is not None and need.numel():
            rows = ops.gather_last_rows(x, ctx.cu_seqlens_q, need, need.numel())
        else:
            rows = ops.gather_last_rows(x, ctx.cu_seqlens_q, None, ctx.cu_seqlens_q.numel() - 1)
        return self.forward_rows(rows)

    def forward_rows(self, rows: torch.Tensor):
        """Logits of already selected hidden rows [B', H] (one-token-per-sequence steps skip the gather)."""
        logits = ops.linear(rows, self.weight.data)
        if self.tp_size > 1:
            logits = comm.gather_cols(logits, 0)
        return logits
