"""Qwen3 dense decoder wiring over the gfx950 operator modules.

Class names, constructor arguments, forward signatures and ``packed_modules_mapping`` follow
the reference's nanovllm/models/qwen3.py:41-533 (so its loader and runner conventions apply),
and the operator ORDER per layer is the reference's:

    add+RMSNorm -> QKV GEMM -> per-head RMSNorm(q), RMSNorm(k) -> RoPE -> KV store -> attention
    -> O GEMM -> add+RMSNorm -> gate/up GEMM -> SiLU*mul -> down GEMM

Two launches-saving fusions keep that arithmetic (same rounding points):
  * q/k norm + RoPE + KV store run as ONE kernel over the fused qkv buffer (nvh_qknorm_rope_store);
  * on a single rank the O and down projections may return their split-K slabs unreduced and the
    following add+RMSNorm sums them in its prologue.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..layers.activation import SiluAndMul
from ..layers.attention import Attention
from ..layers.embed_head import ParallelLMHead, VocabParallelEmbedding
from ..layers.layernorm import RMSNorm
from ..layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear, tp_world
from ..layers.rotary_embedding import get_rope
from ..utils.context import get_context


def rope_theta_of(config) -> float:
    """rope base from either config generation (``rope_theta`` attribute or ``rope_parameters``)."""
    theta = getattr(config, "rope_theta", None)
    if theta is None:
        params = getattr(config, "rope_parameters", None) or {}
        theta = params.get("rope_theta", 1000000)
    return float(theta)


class Qwen3Attention(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, max_position: int = 4096 * 32,
                 head_dim: int | None = None, rms_norm_eps: float = 1e-06, qkv_bias: bool = False,
                 rope_theta: float = 10000, rope_scaling: tuple | None = None) -> None:
        super().__init__()
        tp = tp_world()
        self.total_num_heads = num_heads
        self.total_num_kv_heads = num_kv_heads
        assert num_heads % tp == 0 and num_kv_heads % tp == 0
        self.num_heads = num_heads // tp
        self.num_kv_heads = num_kv_heads // tp
        self.head_dim = head_dim or hidden_size // num_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_bias = qkv_bias
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, num_heads, num_kv_heads, bias=qkv_bias)
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=False)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=max_position,
                                   base=rope_theta, rope_scaling=rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads)
        if not qkv_bias:
            self.q_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)
            self.k_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)
        self.fused_qk_epilogue = True

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor):
        qkv = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q = q.view(-1, self.num_heads, self.head_dim)
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        if not self.qkv_bias and self.fused_qk_epilogue:
            attn = self.attn
            ops.qknorm_rope_store_(qkv, self.num_heads, self.num_kv_heads, self.q_norm.weight.data,
                                   self.k_norm.weight.data, self.q_norm.eps, positions, self.rotary_emb.table2d(),
                                   attn.k_cache, attn.v_cache, get_context().slot_mapping)
            o = attn.attend(q, k, v)
        else:
            if not self.qkv_bias:
                q = self.q_norm(q)
                k = self.k_norm(k)
            q, k = self.rotary_emb(positions, q, k)
            o = self.attn(q, k, v)
        return self.o_proj(o.flatten(1, -1))


class Qwen3MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str) -> None:
        super().__init__()
        assert hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False)
        self.act_fn = SiluAndMul()

    def forward(self, x):
        return self.down_proj(self.act_fn(self.gate_up_proj(x)))


class Qwen3DecoderLayer(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.self_attn = Qwen3Attention(
            hidden_size=config.hidden_size, num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads, max_position=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps, qkv_bias=getattr(config, "attention_bias", True),
            head_dim=getattr(config, "head_dim", None), rope_theta=rope_theta_of(config),
            rope_scaling=getattr(config, "rope_scaling", None))
        self.mlp = Qwen3MLP(config.hidden_size, config.intermediate_size, config.hidden_act)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        # both projections feed an add+RMSNorm, which can sum split-K slabs itself
        self.self_attn.o_proj.defer_reduce = True
        self.mlp.down_proj.defer_reduce = True

    def forward(self, positions: torch.Tensor, hidden_states, residual: torch.Tensor | None):
        if residual is None:
            hidden_states, residual = self.input_layernorm(hidden_states), hidden_states
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class Qwen3Model(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Qwen3ForCausalLM(nn.Module):
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config) -> None:
        super().__init__()
        self.model = Qwen3Model(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if config.tie_word_embeddings:
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.lm_head(hidden_states)
