"""CPU restatement of the nano-vllm operator arithmetic (TEST INFRASTRUCTURE ONLY).

This module is the parity oracle for the HIP kernels.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; the product package never imports it.

Every function restates, in plain torch-CPU eager arithmetic, what one
reference operator computes, with the same fp32 intermediates and the same
bf16 rounding points as the reference *source* (eager semantics: the explicit
``.to(dtype)`` casts in the source are rounding points).  Citations are
``/root/reference`` paths.

Pinning status
  * rms_norm / add_rms_norm / rope / silu_mul / linear / embedding / lm-head row
    selection / store_kvcache: pinned against golden vectors produced by the
    reference's own modules (``oracle/gen_golden.py`` -> ``tests/golden``).
  * paged attention arithmetic: the reference delegates it to the third-party
    ``flash-attn`` wheel (unpinned in its pyproject.toml:18, absent here), so
    the dense fp32 softmax below restates flash-attn's published semantics
    (causal mask aligned to the bottom-right corner, GQA head mapping
    ``kv = q_head // group``, fp32 softmax, scale = head_dim**-0.5).  The
    reference holds no fixture for it: attention parity is UNPINNED by
    reference outputs and is anchored by call-site semantics
    (layers/attention.py:168-249) and by the chunked == un-chunked property.
"""
from __future__ import annotations

import torch

BLOCK = 256  # the only page size the reference supports (engine/sequence.py:73)


# --------------------------------------------------------------------------
# K4 / K5  RMSNorm                       (layers/layernorm.py:57-93, 95-141)
# --------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """y = round_bf16(f32(x) * rsqrt(mean(f32(x)^2) + eps)) * w   (two roundings).

    layers/layernorm.py:87-92: the normalised value is cast back to the input
    dtype *before* the multiply by the (model-dtype) weight.
    """
    dt = x.dtype
    xf = x.float()
    inv = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (xf * inv).to(dt) * weight


def add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """(y, new_residual); variance from the UNROUNDED fp32 sum (layernorm.py:134-140)."""
    dt = x.dtype
    s = x.float() + residual.float()
    new_res = s.to(dt)
    inv = torch.rsqrt(s.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (s * inv).to(dt) * weight, new_res


# --------------------------------------------------------------------------
# K6  rotary embedding         (layers/rotary_embedding.py:27-65, 92-153)
# --------------------------------------------------------------------------
def rope_table(head_size: int, max_position: int, base: float) -> torch.Tensor:
    """fp32 [max_position, 1, head_size]: cos in the first half, sin in the second
    (rotary_embedding.py:103-111)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_size, 2, dtype=torch.float) / head_size))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.outer(t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze(1)


def rope_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Half-split ("NeoX") rotation in fp32, one rounding at the end
    (rotary_embedding.py:61-65).  x: [T, heads, D]; cos/sin: [T, 1, D/2]."""
    half = x.shape[-1] // 2
    xf = x.float()
    a, b = xf[..., :half], xf[..., half:]
    return torch.cat((a * cos - b * sin, b * cos + a * sin), dim=-1).to(x.dtype)


def rope(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, table: torch.Tensor):
    """rotary_embedding.py:148-153: gather table rows by position, rotate q and k."""
    cs = table[positions]
    half = cs.shape[-1] // 2
    cos, sin = cs[..., :half], cs[..., half:]
    return rope_apply(q, cos, sin), rope_apply(k, cos, sin)


# --------------------------------------------------------------------------
# K7  SiLU(gate) * up                         (layers/activation.py:36-58)
# --------------------------------------------------------------------------
def silu_mul(x: torch.Tensor) -> torch.Tensor:
    """Eager semantics: silu rounds to the model dtype, then the product rounds again."""
    half = x.shape[-1] // 2
    g, u = x[..., :half], x[..., half:]
    return torch.nn.functional.silu(g) * u


# --------------------------------------------------------------------------
# G1-G5  linear                      (layers/linear.py:198, 407; embed_head.py:204)
# --------------------------------------------------------------------------
def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y = round(x @ W^T (+ b)) with fp32 accumulation (the cuBLAS bf16 contract)."""
    y = x.float() @ weight.float().t()
    if bias is not None:
        y = y + bias.float()
    return y.to(x.dtype)


# --------------------------------------------------------------------------
# E1  embedding with vocab-parallel masking        (layers/embed_head.py:119-128)
# --------------------------------------------------------------------------
def embedding(ids: torch.Tensor, weight: torch.Tensor, vocab_start: int = 0, tp_size: int = 1) -> torch.Tensor:
    if tp_size > 1:
        n = weight.shape[0]
        mask = (ids >= vocab_start) & (ids < vocab_start + n)
        rows = weight[(ids - vocab_start) * mask]
        return rows * mask.unsqueeze(1)
    return weight[ids]


def last_token_rows(hidden: torch.Tensor, cu_seqlens_q: torch.Tensor, need: torch.Tensor) -> torch.Tensor:
    """embed_head.py:197-202: rows cu_seqlens_q[1:]-1, filtered by ``need`` when it
    is non-empty (an EMPTY need list selects every sequence)."""
    last = cu_seqlens_q[1:].long() - 1
    if need.numel():
        last = last[need.long()]
    return hidden[last].contiguous()


# --------------------------------------------------------------------------
# K1  KV-cache scatter store                  (layers/attention.py:75-90)
# --------------------------------------------------------------------------
def store_kvcache(key, value, k_cache, v_cache, slot_mapping):
    """Row ``slot`` of the flattened [slots, Hkv*D] cache <- token row; slot -1 skipped."""
    n, hkv, d = key.shape
    kc = k_cache.view(-1, hkv * d)
    vc = v_cache.view(-1, hkv * d)
    sl = slot_mapping.long()
    keep = sl >= 0
    kc[sl[keep]] = key.reshape(n, hkv * d)[keep]
    vc[sl[keep]] = value.reshape(n, hkv * d)[keep]


# --------------------------------------------------------------------------
# K2 / K3  paged causal GQA attention        (layers/attention.py:168-249)
# --------------------------------------------------------------------------
def _attend_one(q, k, v, scale):
    """q [Lq,Hq,D], k/v [Lk,Hkv,D] -> [Lq,Hq,D]; mask aligned bottom-right:
    query i sees keys j <= i + (Lk - Lq)."""
    lq, hq, d = q.shape
    lk, hkv, _ = k.shape
    g = hq // hkv
    qf = q.float().transpose(0, 1)                                  # [Hq,Lq,D]
    kf = k.float().transpose(0, 1).repeat_interleave(g, dim=0)      # [Hq,Lk,D]
    vf = v.float().transpose(0, 1).repeat_interleave(g, dim=0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale                # [Hq,Lq,Lk]
    qi = torch.arange(lq).unsqueeze(1)
    kj = torch.arange(lk).unsqueeze(0)
    s = s.masked_fill(kj > qi + (lk - lq), float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)     # a row with no visible key yields zeros
    return torch.matmul(p, vf).transpose(0, 1).to(q.dtype)


def gather_paged(cache: torch.Tensor, table_row, length: int) -> torch.Tensor:
    """cache [blocks, 256, Hkv, D]; first ``length`` tokens of a sequence."""
    nblk = (length + BLOCK - 1) // BLOCK
    idx = torch.as_tensor(table_row[:nblk], dtype=torch.long)
    return cache[idx].reshape(-1, cache.shape[-2], cache.shape[-1])[:length]


def attention_varlen(q, k, v, cu_q, cu_k, scale, k_cache=None, v_cache=None, block_tables=None):
    """Modes W (block_tables None: K/V are the packed new tokens) and P (paged:
    K/V come from the cache through the block table).  attention.py:236-248."""
    out = torch.empty_like(q)
    nseq = len(cu_q) - 1
    for i in range(nseq):
        q0, q1 = int(cu_q[i]), int(cu_q[i + 1])
        lk = int(cu_k[i + 1]) - int(cu_k[i])
        if q1 == q0:
            continue
        if block_tables is None:
            k0 = int(cu_k[i])
            ki, vi = k[k0:k0 + lk], v[k0:k0 + lk]
        else:
            ki = gather_paged(k_cache, block_tables[i].tolist(), lk)
            vi = gather_paged(v_cache, block_tables[i].tolist(), lk)
        out[q0:q1] = _attend_one(q[q0:q1], ki, vi, scale)
    return out


def attention_decode(q, k_cache, v_cache, context_lens, block_tables, scale):
    """Mode G (attention.py:222-234): one query per sequence, cache length
    ``context_lens[i]``; a padded row (length 0) produces zeros."""
    out = torch.zeros_like(q)
    for i in range(q.shape[0]):
        lk = int(context_lens[i])
        if lk == 0:
            continue
        ki = gather_paged(k_cache, block_tables[i].tolist(), lk)
        vi = gather_paged(v_cache, block_tables[i].tolist(), lk)
        out[i:i + 1] = _attend_one(q[i:i + 1], ki, vi, scale)
    return out


# --------------------------------------------------------------------------
# K8  sampler                                   (layers/sampler.py:41-81)
# --------------------------------------------------------------------------
def sampler_probs(logits: torch.Tensor, temperatures: torch.Tensor) -> torch.Tensor:
    return torch.softmax(logits.float() / temperatures.unsqueeze(1), dim=-1)


def sample_with_noise(logits, temperatures, noise):
    """argmax(p / clamp_min(noise, 1e-10)) for a GIVEN Exp(1) noise tensor; the RNG
    stream itself is not reproducible across devices, so parity is stated on this
    deterministic form and on the sampled distribution."""
    return (sampler_probs(logits, temperatures) / noise.clamp_min(1e-10)).argmax(dim=-1)


def greedy(logits: torch.Tensor) -> torch.Tensor:
    """Pre-sampler argmax, lowest index wins ties."""
    return logits.float().argmax(dim=-1)
