"""CPU restatement of the nano-vllm runner + Qwen3 wiring (TEST INFRASTRUCTURE ONLY).

Imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg.  It drives ``oracle.ref_ops`` in the operator order of the
reference model (models/qwen3.py) over the inputs the reference runner builds
(engine/model_runner.py:400-495), on plain torch-CPU tensors, single rank.

Weights are a flat ``dict[str, Tensor]`` keyed by the reference's parameter
names (``model.layers.{i}.self_attn.qkv_proj.weight`` ...), so a state_dict of
the reference model loads directly.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import ref_ops as R

BLOCK = R.BLOCK


@dataclass
class ModelShape:
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    max_position_embeddings: int = 40960
    tie_word_embeddings: bool = True

    @property
    def q_size(self):
        return self.num_attention_heads * self.head_dim

    @property
    def kv_size(self):
        return self.num_key_value_heads * self.head_dim


@dataclass
class SeqState:
    """The fields of engine/sequence.py that the runner reads (its pickled state,
    sequence.py:224-235)."""
    token_ids: list
    num_cached_tokens: int = 0
    num_new_tokens: int = 0
    block_table: list = field(default_factory=list)
    temperature: float = 1.0

    def __len__(self):
        return len(self.token_ids)


@dataclass
class StepInput:
    input_ids: torch.Tensor
    positions: torch.Tensor
    cu_seqlens_q: torch.Tensor
    cu_seqlens_k: torch.Tensor
    max_seqlen_q: int
    max_seqlen_k: int
    slot_mapping: torch.Tensor
    context_lens: torch.Tensor
    block_tables: torch.Tensor | None
    need_logits: torch.Tensor


def prepare_model_input(seqs, block_size: int = BLOCK) -> StepInput:
    """engine/model_runner.py:438-495 restated without device transfers.

    need_logits: a sequence whose scheduled tokens reach its end AND that owns a
    block table (:451).  slot ranges: first new block starts at the cached offset,
    last block ends at the context length (:466-481).  block_tables only when the
    summed key length exceeds the summed query length (:483), -1 padded (:394-395).
    """
    ids, pos, slots, ctx_lens, need = [], [], [], [], []
    cu_q, cu_k = [0], [0]
    mq = mk = 0
    for i, s in enumerate(seqs):
        cached, new = s.num_cached_tokens, s.num_new_tokens
        ctx = cached + new
        if len(s) == ctx and s.block_table:
            need.append(i)
        ctx_lens.append(ctx)
        ids.extend(s.token_ids[cached:ctx])
        pos.extend(range(cached, ctx))
        cu_q.append(cu_q[-1] + new)
        cu_k.append(cu_k[-1] + ctx)
        mq, mk = max(mq, new), max(mk, ctx)
        if not s.block_table:
            continue
        first = cached // block_size
        for b in range(first, len(s.block_table)):
            base = s.block_table[b] * block_size
            lo = base + (cached % block_size if b == first else 0)
            if b == len(s.block_table) - 1 and ctx % block_size:
                hi = base + ctx % block_size
            else:
                hi = base + block_size
            slots.extend(range(lo, hi))
    tables = None
    if cu_k[-1] > cu_q[-1]:
        width = max(len(s.block_table) for s in seqs)
        tables = torch.tensor([s.block_table + [-1] * (width - len(s.block_table)) for s in seqs],
                              dtype=torch.int32)
    t = torch.tensor
    return StepInput(t(ids, dtype=torch.int64), t(pos, dtype=torch.int64),
                     t(cu_q, dtype=torch.int32), t(cu_k, dtype=torch.int32), mq, mk,
                     t(slots, dtype=torch.int32), t(ctx_lens, dtype=torch.int32), tables,
                     t(need, dtype=torch.int32))


def random_weights(shape: ModelShape, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02) -> dict:
    """Synthetic weights: N(0, std) matrices, ones for norm weights (layernorm.py:55)."""
    g = torch.Generator().manual_seed(seed)

    def mat(*s):
        return (torch.randn(*s, generator=g) * std).to(dtype)

    w = {"model.embed_tokens.weight": mat(shape.vocab_size, shape.hidden_size)}
    for i in range(shape.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "self_attn.qkv_proj.weight"] = mat(shape.q_size + 2 * shape.kv_size, shape.hidden_size)
        w[p + "self_attn.o_proj.weight"] = mat(shape.hidden_size, shape.q_size)
        w[p + "self_attn.q_norm.weight"] = torch.ones(shape.head_dim, dtype=dtype)
        w[p + "self_attn.k_norm.weight"] = torch.ones(shape.head_dim, dtype=dtype)
        w[p + "mlp.gate_up_proj.weight"] = mat(2 * shape.intermediate_size, shape.hidden_size)
        w[p + "mlp.down_proj.weight"] = mat(shape.hidden_size, shape.intermediate_size)
        w[p + "input_layernorm.weight"] = torch.ones(shape.hidden_size, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = torch.ones(shape.hidden_size, dtype=dtype)
    w["model.norm.weight"] = torch.ones(shape.hidden_size, dtype=dtype)
    if shape.tie_word_embeddings:
        w["lm_head.weight"] = w["model.embed_tokens.weight"]
    else:
        w["lm_head.weight"] = mat(shape.vocab_size, shape.hidden_size)
    return w


class RefRunner:
    """Single-rank CPU runner: KV cache [2, L, blocks, 256, Hkv, D]
    (engine/model_runner.py:368) + forward in the reference's op order."""

    def __init__(self, shape: ModelShape, weights: dict, num_blocks: int, dtype=torch.bfloat16):
        self.shape, self.w, self.dtype = shape, weights, dtype
        self.table = R.rope_table(shape.head_dim, shape.max_position_embeddings, shape.rope_theta)
        self.kv = torch.zeros(2, shape.num_hidden_layers, max(num_blocks, 0), BLOCK,
                              shape.num_key_value_heads, shape.head_dim, dtype=dtype)
        self.has_cache = num_blocks > 0

    # models/qwen3.py:183-199
    def _attention(self, i, positions, h, inp: StepInput, graph_decode: bool):
        sh, w = self.shape, self.w
        p = f"model.layers.{i}.self_attn."
        qkv = R.linear(h, w[p + "qkv_proj.weight"])
        q, k, v = qkv.split([sh.q_size, sh.kv_size, sh.kv_size], dim=-1)
        q = q.view(-1, sh.num_attention_heads, sh.head_dim)
        k = k.view(-1, sh.num_key_value_heads, sh.head_dim)
        v = v.view(-1, sh.num_key_value_heads, sh.head_dim)
        q = R.rms_norm(q, w[p + "q_norm.weight"], sh.rms_norm_eps)
        k = R.rms_norm(k, w[p + "k_norm.weight"], sh.rms_norm_eps)
        q, k = R.rope(positions, q, k, self.table)
        scale = sh.head_dim ** -0.5
        kc, vc = self.kv[0, i], self.kv[1, i]
        if self.has_cache and inp.slot_mapping.numel():
            R.store_kvcache(k, v, kc, vc, inp.slot_mapping)       # attention.py:219-220
        if graph_decode:
            o = R.attention_decode(q, kc, vc, inp.context_lens, inp.block_tables, scale)
        elif inp.block_tables is not None:
            o = R.attention_varlen(q, None, None, inp.cu_seqlens_q, inp.cu_seqlens_k, scale,
                                   kc, vc, inp.block_tables)
        else:
            o = R.attention_varlen(q, k, v, inp.cu_seqlens_q, inp.cu_seqlens_k, scale)
        return R.linear(o.flatten(1), w[p + "o_proj.weight"])

    # models/qwen3.py:365-376, 434-441
    def forward(self, inp: StepInput, graph_decode: bool = False, return_layers: bool = False):
        sh, w = self.shape, self.w
        h = R.embedding(inp.input_ids, w["model.embed_tokens.weight"])
        res = None
        layers_out = []
        for i in range(sh.num_hidden_layers):
            p = f"model.layers.{i}."
            if res is None:
                res, h = h, R.rms_norm(h, w[p + "input_layernorm.weight"], sh.rms_norm_eps)
            else:
                h, res = R.add_rms_norm(h, res, w[p + "input_layernorm.weight"], sh.rms_norm_eps)
            h = self._attention(i, inp.positions, h, inp, graph_decode)
            h, res = R.add_rms_norm(h, res, w[p + "post_attention_layernorm.weight"], sh.rms_norm_eps)
            gu = R.linear(h, w[p + "mlp.gate_up_proj.weight"])
            h = R.linear(R.silu_mul(gu), w[p + "mlp.down_proj.weight"])
            if return_layers:
                layers_out.append(h.clone())
        h, _ = R.add_rms_norm(h, res, w["model.norm.weight"], sh.rms_norm_eps)
        return (h, layers_out) if return_layers else h

    def compute_logits(self, hidden, inp: StepInput):
        rows = R.last_token_rows(hidden, inp.cu_seqlens_q, inp.need_logits)
        return R.linear(rows, self.w["lm_head.weight"])

    def run(self, seqs):
        """engine/model_runner.py:602-638 up to (not including) the sampler: returns
        (logits [B', V], need_logits)."""
        inp = prepare_model_input(seqs)
        return self.compute_logits(self.forward(inp), inp), inp.need_logits


def greedy_generate(runner: RefRunner, prompt: list, max_tokens: int, chunk: int | None = None,
                    blocks: list | None = None):
    """Greedy decode of ONE prompt through paged attention: optional chunked prefill
    (``chunk`` tokens per step), then one token per step.  Returns (token ids, list
    of per-step logits rows)."""
    toks = list(prompt)
    nblk_needed = (len(prompt) + max_tokens + BLOCK - 1) // BLOCK
    blocks = blocks if blocks is not None else list(range(nblk_needed))
    s = SeqState(toks)
    out, rows = [], []
    while len(out) < max_tokens:
        remaining = len(s) - s.num_cached_tokens
        s.num_new_tokens = min(remaining, chunk) if chunk else remaining
        ctx = s.num_cached_tokens + s.num_new_tokens
        s.block_table = blocks[:(ctx + BLOCK - 1) // BLOCK]
        logits, need = runner.run([s])
        s.num_cached_tokens = ctx
        if need.numel():
            tok = int(R.greedy(logits)[0])
            rows.append(logits[0].clone())
            out.append(tok)
            s.token_ids.append(tok)
        s.num_new_tokens = 0
    return out, rows
