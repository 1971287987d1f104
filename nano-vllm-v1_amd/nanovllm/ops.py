"""Tensor-level wrappers over the C ABI (one function per entry point).

Each wrapper checks what the kernels assume (device, dtype, contiguity of the
innermost dimension), allocates outputs / workspaces through torch's caching
allocator (so calls are graph-capturable), and launches on torch's current
stream.  bf16 only: the operator path has no other precision.
"""
from __future__ import annotations

import torch

from . import _hip

BF16 = torch.bfloat16
KV_PAGE = 256


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gpu_bf16(*tensors):
    for t in tensors:
        if not (t.is_cuda and t.dtype == BF16):
            raise TypeError(f"expected a bf16 GPU tensor, got {t.dtype} on {t.device}")


def _p(t):
    return 0 if t is None else t.data_ptr()


# ---------------------------------------------------------------- norms
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """x [..., dim] -> normalised copy (a strided view, e.g. q inside the fused qkv buffer, is
    compacted first; the model's own forward uses the fused qknorm_rope_store_ instead)."""
    _gpu_bf16(x, weight)
    dim = x.shape[-1]
    if not x.is_contiguous():
        x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    rows = x.numel() // dim
    _hip.call("nvh_rmsnorm", x.data_ptr(), dim, weight.data_ptr(), out.data_ptr(), dim, rows, dim, float(eps), _stream())
    return out


def add_rmsnorm(x: torch.Tensor | None, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                partials: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """(y, residual): residual is updated IN PLACE to bf16(x + residual).  With ``partials``
    (fp32 [S, rows, dim] split-K slabs) x is bf16(sum of slabs)."""
    _gpu_bf16(residual, weight)
    dim = residual.shape[-1]
    rows = residual.numel() // dim
    assert residual.is_contiguous()
    y = torch.empty_like(residual)
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.shape[-2:] == (rows, dim)
        _hip.call("nvh_add_rmsnorm", 0, residual.data_ptr(), weight.data_ptr(), y.data_ptr(), rows, dim, float(eps),
                  partials.data_ptr(), partials.shape[0], _stream())
    else:
        _gpu_bf16(x)
        if not x.is_contiguous():
            x = x.contiguous()
        _hip.call("nvh_add_rmsnorm", x.data_ptr(), residual.data_ptr(), weight.data_ptr(), y.data_ptr(), rows, dim,
                  float(eps), 0, 0, _stream())
    return y, residual


# ---------------------------------------------------------------- rotary / fused qkv epilogue
def rope_(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, cos_sin: torch.Tensor) -> None:
    """In-place rotation of q [T,Hq,D] and k [T,Hkv,D] (views with head stride D allowed)."""
    _gpu_bf16(q, k)
    assert positions.dtype == torch.int64 and cos_sin.dtype == torch.float32 and cos_sin.is_contiguous()
    t, hq, d = q.shape
    hkv = k.shape[1]
    assert q.stride(2) == 1 and q.stride(1) == d and k.stride(2) == 1 and k.stride(1) == d
    _hip.call("nvh_rope", positions.data_ptr(), q.data_ptr(), q.stride(0), hq, k.data_ptr(), k.stride(0), hkv,
              cos_sin.data_ptr(), cos_sin.shape[0], t, d, _stream())


def qknorm_rope_store_(qkv: torch.Tensor, num_q_heads: int, num_kv_heads: int, q_weight, k_weight, eps: float,
                       positions, cos_sin, k_cache, v_cache, slot_mapping) -> None:
    """Fused per-head RMSNorm + RoPE on the q/k parts of qkv [T, (Hq+2Hkv)*128] in place,
    and K/V scatter into the caches (skipped when the caches are None)."""
    _gpu_bf16(qkv, q_weight, k_weight)
    assert qkv.stride(1) == 1 and positions.dtype == torch.int64
    store = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None and slot_mapping.numel() > 0
    _hip.call("nvh_qknorm_rope_store", qkv.data_ptr(), qkv.stride(0), num_q_heads, num_kv_heads, q_weight.data_ptr(),
              k_weight.data_ptr(), float(eps), positions.data_ptr(), cos_sin.data_ptr(), cos_sin.shape[0],
              k_cache.data_ptr() if store else 0, v_cache.data_ptr() if store else 0,
              slot_mapping.data_ptr() if store else 0, qkv.shape[0], _stream())


def linear_qkv_rope_store(x, w, num_q_heads: int, num_kv_heads: int, q_weight, k_weight, eps: float, positions, cos_sin,
                          k_cache=None, v_cache=None, slot_mapping=None, raw_rows: int = 0) -> torch.Tensor:
    """qkv [T, (Hq+2Hkv)*128] = x w^T with the q/k epilogue and KV store of rows raw_rows.. applied (in the GEMM's
    epilogue where the shape allows, as two launches otherwise; same bits)."""
    _gpu_bf16(x, w, q_weight, k_weight)
    t, k = x.shape
    n = (num_q_heads + 2 * num_kv_heads) * 128
    assert w.shape == (n, k) and w.is_contiguous() and x.stride(1) == 1 and positions.dtype == torch.int64
    store = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None and slot_mapping.numel() > 0
    qkv = torch.empty(t, n, dtype=BF16, device=x.device)
    _hip.call("nvh_gemm_qkv_rope_store", x.data_ptr(), x.stride(0), w.data_ptr(), qkv.data_ptr(), n, t, k, num_q_heads,
              num_kv_heads, 128, q_weight.data_ptr(), k_weight.data_ptr(), float(eps), positions.data_ptr(), cos_sin.data_ptr(),
              cos_sin.shape[0], k_cache.data_ptr() if store else 0, v_cache.data_ptr() if store else 0,
              slot_mapping.data_ptr() if store else 0, raw_rows, _stream())
    return qkv


# ---------------------------------------------------------------- RMSNorm as a row scale (prefill-sized steps)
def residual_rowsq_parts(m: int, n: int, k: int) -> int:
    """Partial sums of squares per row that `linear_add_residual_` leaves for this shape; 0: shape not supported."""
    return int(_hip.lib().nvh_gemm_residual_rowsq_parts(m, n, k))


def fold_norm_weight(weight: torch.Tensor, norm_weight: torch.Tensor) -> torch.Tensor:
    """w[n][k] * norm[k] rounded to bf16: the projection weight the row-scaled entry points take."""
    return (weight.float() * norm_weight.float()[None, :]).to(weight.dtype)


def linear_add_residual_(x: torch.Tensor, weight: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    """residual += x weight^T in place (product rounded to bf16, sum in fp32, rounded to bf16); returns the fp32 partial
    sums of squares of the un-rounded sums, [M, parts]."""
    _gpu_bf16(x, weight, residual)
    m, k = x.shape
    n = weight.shape[0]
    parts = residual_rowsq_parts(m, n, k)
    assert parts > 0 and residual.shape == (m, n) and residual.is_contiguous() and weight.is_contiguous() and x.stride(1) == 1
    rowsq = torch.empty(m, parts, dtype=torch.float32, device=x.device)
    _hip.call("nvh_gemm_residual_rowsq_bf16_nt", x.data_ptr(), x.stride(0), weight.data_ptr(), residual.data_ptr(), rowsq.data_ptr(),
              m, n, k, _stream())
    return rowsq


def linear_swiglu_rowscaled(x: torch.Tensor, weight_folded_interleaved: torch.Tensor, rowsq: torch.Tensor, eps: float) -> torch.Tensor:
    """silu(g) * u of (x w^T) * 1/sqrt(sum(rowsq[m]) / K + eps): RMSNorm + gate/up projection + SwiGLU over the
    un-normalised rows, the norm weight folded into `weight_folded_interleaved`."""
    _gpu_bf16(x, weight_folded_interleaved)
    m, k = x.shape
    n = weight_folded_interleaved.shape[0]
    assert rowsq.dtype == torch.float32 and rowsq.shape[0] == m and rowsq.is_contiguous() and x.stride(1) == 1
    y = torch.empty(m, n // 2, dtype=BF16, device=x.device)
    _hip.call("nvh_gemm_swiglu_rowscaled_bf16_nt", x.data_ptr(), x.stride(0), weight_folded_interleaved.data_ptr(), y.data_ptr(), n // 2,
              m, n, k, rowsq.data_ptr(), rowsq.shape[1], float(eps), _stream())
    return y


def linear_qkv_rope_store_rowscaled(x, w_folded, num_q_heads: int, num_kv_heads: int, q_weight, k_weight, eps: float, positions, cos_sin,
                                    rowsq: torch.Tensor, norm_eps: float, k_cache=None, v_cache=None, slot_mapping=None,
                                    raw_rows: int = 0) -> torch.Tensor:
    """`linear_qkv_rope_store` over un-normalised rows (see linear_swiglu_rowscaled); only for shapes whose epilogue runs
    inside the GEMM (HipError otherwise)."""
    _gpu_bf16(x, w_folded, q_weight, k_weight)
    t, k = x.shape
    n = (num_q_heads + 2 * num_kv_heads) * 128
    assert w_folded.shape == (n, k) and w_folded.is_contiguous() and x.stride(1) == 1 and positions.dtype == torch.int64
    assert rowsq.dtype == torch.float32 and rowsq.shape[0] == t and rowsq.is_contiguous()
    store = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None and slot_mapping.numel() > 0
    qkv = torch.empty(t, n, dtype=BF16, device=x.device)
    _hip.call("nvh_gemm_qkv_rope_store_rowscaled", x.data_ptr(), x.stride(0), w_folded.data_ptr(), qkv.data_ptr(), n, t, k, num_q_heads,
              num_kv_heads, 128, q_weight.data_ptr(), k_weight.data_ptr(), float(eps), positions.data_ptr(), cos_sin.data_ptr(),
              cos_sin.shape[0], k_cache.data_ptr() if store else 0, v_cache.data_ptr() if store else 0,
              slot_mapping.data_ptr() if store else 0, raw_rows, rowsq.data_ptr(), rowsq.shape[1], float(norm_eps), _stream())
    return qkv


def store_kvcache(key, value, k_cache, v_cache, slot_mapping) -> None:
    _gpu_bf16(key, value, k_cache, v_cache)
    n, h, d = key.shape
    assert key.stride(-1) == 1 and value.stride(-1) == 1 and key.stride(1) == d and value.stride(1) == d
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and slot_mapping.dtype == torch.int32
    assert slot_mapping.numel() == n
    _hip.call("nvh_store_kvcache", key.data_ptr(), key.stride(0), value.data_ptr(), value.stride(0), k_cache.data_ptr(),
              v_cache.data_ptr(), slot_mapping.data_ptr(), n, h * d, _stream())


# ---------------------------------------------------------------- activation / gathers
def silu_mul(x: torch.Tensor) -> torch.Tensor:
    _gpu_bf16(x)
    if not x.is_contiguous():
        x = x.contiguous()
    inter = x.shape[-1] // 2
    rows = x.numel() // (2 * inter)
    y = torch.empty(*x.shape[:-1], inter, dtype=BF16, device=x.device)
    _hip.call("nvh_silu_mul", x.data_ptr(), y.data_ptr(), rows, inter, _stream())
    return y


def embedding(ids: torch.Tensor, table: torch.Tensor, vocab_start: int = 0) -> torch.Tensor:
    _gpu_bf16(table)
    assert ids.dtype == torch.int64 and ids.is_cuda and table.is_contiguous()
    out = torch.empty(ids.numel(), table.shape[1], dtype=BF16, device=table.device)
    _hip.call("nvh_embedding", ids.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), table.shape[1],
              vocab_start, table.shape[0], _stream())
    return out


def feed_token_ids(input_ids: torch.Tensor, dst_index: torch.Tensor, sampled: torch.Tensor, src_index: torch.Tensor) -> None:
    """input_ids[dst_index[i]] = sampled[src_index[i]], in place (ids a step takes from the previous step's sampler
    output before the host has read them)."""
    assert input_ids.dtype == torch.int64 and sampled.dtype == torch.int64 and input_ids.is_cuda and sampled.is_cuda
    assert dst_index.dtype == torch.int32 and src_index.dtype == torch.int32 and dst_index.numel() == src_index.numel()
    assert input_ids.is_contiguous() and sampled.is_contiguous()
    _hip.call("nvh_feed_token_ids", input_ids.data_ptr(), dst_index.data_ptr(), sampled.data_ptr(), src_index.data_ptr(),
              dst_index.numel(), _stream())


def gather_last_rows(hidden: torch.Tensor, cu_seqlens_q: torch.Tensor, need: torch.Tensor | None, num_rows: int) -> torch.Tensor:
    _gpu_bf16(hidden)
    assert hidden.is_contiguous() and cu_seqlens_q.dtype == torch.int32
    out = torch.empty(num_rows, hidden.shape[1], dtype=BF16, device=hidden.device)
    _hip.call("nvh_gather_last_rows", hidden.data_ptr(), cu_seqlens_q.data_ptr(), _p(need), out.data_ptr(), num_rows,
              hidden.shape[1], _stream())
    return out


# ---------------------------------------------------------------- GEMM
def suggest_split_k(m: int, n: int, k: int) -> int:
    return _hip.lib().nvh_gemm_suggest_split_k(m, n, k)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y = x @ W^T (+ bias); x [..., K] with contiguous rows, W [N, K]."""
    _gpu_bf16(x, weight)
    k = x.shape[-1]
    n = weight.shape[0]
    assert weight.is_contiguous() and weight.shape[1] == k
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    y = torch.empty(m, n, dtype=BF16, device=x.device)
    _hip.call("nvh_gemm_bf16_nt", x2.data_ptr(), x2.stride(0), weight.data_ptr(), _p(bias), y.data_ptr(), n, m, n, k, 1,
              0, _stream())
    return y.view(*x.shape[:-1], n)


def linear_splitk(x: torch.Tensor, weight: torch.Tensor, split_k: int) -> torch.Tensor:
    """fp32 partial slabs [split_k, M, N]; the consumer (add_rmsnorm / splitk_reduce) sums them."""
    _gpu_bf16(x, weight)
    k = x.shape[-1]
    n = weight.shape[0]
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    ws = torch.empty(split_k, m, n, dtype=torch.float32, device=x.device)
    _hip.call("nvh_gemm_bf16_nt", x2.data_ptr(), x2.stride(0), weight.data_ptr(), 0, 0, n, m, n, k, split_k,
              ws.data_ptr(), _stream())
    return ws


def interleave_gate_up(weight: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """[gate (I rows) | up (I rows)] -> blocks of 16 gate rows followed by their 16 up rows (for linear_swiglu)."""
    _gpu_bf16(weight)
    assert weight.is_contiguous() and weight.shape[0] % 32 == 0
    if out is None:
        out = torch.empty_like(weight)
    assert out.shape == weight.shape and out.is_contiguous() and out.data_ptr() != weight.data_ptr()
    _hip.call("nvh_interleave_gate_up", weight.data_ptr(), out.data_ptr(), weight.shape[0] // 2, weight.shape[1], _stream())
    return out


def linear_swiglu(x: torch.Tensor, weight_interleaved: torch.Tensor) -> torch.Tensor:
    """silu(x Wg^T) * (x Wu^T) in one kernel; bit-identical to silu_mul(linear(x, W))."""
    _gpu_bf16(x, weight_interleaved)
    k = x.shape[-1]
    n = weight_interleaved.shape[0]
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m = x2.shape[0]
    y = torch.empty(m, n // 2, dtype=BF16, device=x.device)
    _hip.call("nvh_gemm_swiglu_bf16_nt", x2.data_ptr(), x2.stride(0), weight_interleaved.data_ptr(), y.data_ptr(), n // 2,
              m, n, k, _stream())
    return y.view(*x.shape[:-1], n // 2)


def add_rmsnorm_linear(x: torch.Tensor | None, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float, weight: torch.Tensor,
                       partials: torch.Tensor | None = None, swiglu: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """(y, new_residual) of add_rmsnorm followed by linear (or linear_swiglu over an interleaved weight) in ONE launch:
    decode batches of 1..64 rows with rows of at most 1024 elements.  ``residual`` is NOT modified; bit-identical to the
    two calls.  ``fused_add_rmsnorm_linear`` tells whether the step executor would take it."""
    _gpu_bf16(residual, norm_weight, weight)
    k = residual.shape[-1]
    m = residual.numel() // k
    n = weight.shape[0]
    assert residual.is_contiguous() and weight.is_contiguous() and weight.shape[1] == k
    new_res = torch.empty_like(residual)
    y = torch.empty(m, n // 2 if swiglu else n, dtype=BF16, device=residual.device)
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.shape[-2:] == (m, k)
        prev, slabs, ns = 0, partials.data_ptr(), partials.shape[0]
    else:
        _gpu_bf16(x)
        x = x.contiguous()
        prev, slabs, ns = x.data_ptr(), 0, 0
    _hip.call("nvh_add_rmsnorm_gemm_bf16_nt", prev, slabs, ns, residual.data_ptr(), new_res.data_ptr(), norm_weight.data_ptr(),
              float(eps), weight.data_ptr(), y.data_ptr(), y.shape[1], m, n, k, 1 if swiglu else 0, _stream())
    return y, new_res


def fused_add_rmsnorm_linear(m: int, n: int, k: int, num_partials: int = 0, swiglu: bool = False) -> bool:
    return _hip.lib().nvh_add_rmsnorm_gemm_fused(m, n, k, num_partials, 1 if swiglu else 0) == 1


def splitk_reduce(partials: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    s, m, n = partials.shape
    y = torch.empty(m, n, dtype=BF16, device=partials.device)
    _hip.call("nvh_splitk_reduce", partials.data_ptr(), s, _p(bias), y.data_ptr(), n, m, n, _stream())
    return y


# ---------------------------------------------------------------- attention
def attn_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q: int, scale: float, block_tables=None,
                max_seqlen_k: int | None = None, num_splits: int = 0) -> torch.Tensor:
    """q [T,Hq,128]; (k, v) packed [Tk,Hkv,128] or page pools [P,256,Hkv,128] with block_tables [B,W] int32."""
    _gpu_bf16(q, k, v)
    t, hq, d = q.shape
    assert q.stride(2) == 1 and q.stride(1) == d
    if block_tables is None:
        hkv = k.shape[1]
        assert k.stride(2) == 1 and k.stride(1) == d and v.stride(1) == d and k.stride(0) == v.stride(0)
        kv_stride, tbl, tbl_stride = k.stride(0), 0, 0
    else:
        hkv = k.shape[2]
        assert k.is_contiguous() and v.is_contiguous() and k.shape[1] == KV_PAGE
        assert block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
        kv_stride, tbl, tbl_stride = hkv * d, block_tables.data_ptr(), block_tables.stride(0)
    out = torch.empty(t, hq, d, dtype=BF16, device=q.device)
    batch = cu_seqlens_q.numel() - 1
    if max_seqlen_k is None:        # upper bound: every key of the pool / packed buffer
        max_seqlen_k = block_tables.shape[1] * KV_PAGE if block_tables is not None else k.shape[0]
    lib = _hip.lib()
    if num_splits <= 0:
        num_splits = lib.nvh_attn_varlen_suggest_splits(t, batch, hq, hkv, max_seqlen_k)
    ws = 0
    if num_splits > 1:
        ws_t = torch.empty(lib.nvh_attn_varlen_workspace_bytes(t, hq, d, num_splits) // 4, dtype=torch.float32, device=q.device)
        ws = ws_t.data_ptr()
    _hip.call("nvh_attn_varlen", q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), kv_stride,
              cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), tbl, tbl_stride, batch,
              max_seqlen_q, max_seqlen_k, t, hq, hkv, d, float(scale), out.data_ptr(), num_splits, ws, _stream())
    return out


def suggest_decode_splits(batch: int, num_kv_heads: int, max_context: int) -> int:
    return _hip.lib().nvh_attn_decode_suggest_splits(batch, num_kv_heads, max_context)


def plan_decode_items(context_lens, max_ranges: int = 32, range_keys: int = 0, capacity: int = 0) -> dict:
    """Work list of the decode attention for a batch of uneven context lengths (host side, numpy): one item per
    key range, where only sequences that hold at least two ranges' worth of keys are cut into ranges.  A range
    is `range_keys` keys (0: total keys / 512, kept within 512..2048) and a sequence has at most `max_ranges`;
    `capacity` > 0 fixes the list length (unused slots -1) and widens the ranges until the list fits it.
    Returns int32 arrays item_seq, item_range [items], seq_active, seq_chunk [B] and max_ranges."""
    import numpy as np
    ctx = np.maximum(np.asarray(context_lens, dtype=np.int64), 0)
    b = ctx.size
    assert capacity <= 0 or capacity >= b
    if range_keys <= 0:
        range_keys = min(2048, max(512, (-(-int(ctx.sum()) // 512) + 63) // 64 * 64))
    while True:
        active = np.clip(ctx // range_keys, 1, max_ranges)
        if capacity <= 0 or int(active.sum()) <= capacity:
            break
        range_keys *= 2
    chunk = np.maximum(64, (-(-ctx // active) + 63) // 64 * 64)
    active = np.maximum(1, -(-ctx // chunk))              # rounding the range up to 64 keys can empty the last one
    n = capacity if capacity > 0 else int(active.sum())
    # longest items first: workgroups are dispatched in list order, and a long range started last would run on alone
    order = np.argsort(-chunk * (active > 1), kind="stable")
    reps = active[order]
    total = int(reps.sum())
    item_seq, item_range = np.full(n, -1, np.int32), np.zeros(n, np.int32)
    item_seq[:total] = np.repeat(order, reps)
    item_range[:total] = np.arange(total) - np.repeat(np.cumsum(reps) - reps, reps)
    return {"item_seq": item_seq, "item_range": item_range, "seq_active": active.astype(np.int32),
            "seq_chunk": chunk.astype(np.int32), "max_ranges": max_ranges}


def upload_decode_items(plan: dict, device, out: torch.Tensor | None = None) -> dict:
    """The plan's four arrays in one int32 device tensor (one copy); `out` reuses a buffer of the right length."""
    import numpy as np
    packed = np.concatenate([plan["item_seq"], plan["item_range"], plan["seq_active"], plan["seq_chunk"]])
    host = torch.from_numpy(packed)
    if out is None:
        out = host.to(device, non_blocking=False)
    else:
        assert out.numel() == packed.size and out.dtype == torch.int32
        out.copy_(host, non_blocking=False)
    n, b = plan["item_seq"].size, plan["seq_active"].size
    return {"item_seq": out[:n], "item_range": out[n:2 * n], "seq_active": out[2 * n:2 * n + b],
            "seq_chunk": out[2 * n + b:2 * n + 2 * b], "max_ranges": plan["max_ranges"]}


def _items_args(items, batch):
    assert all(items[k].dtype == torch.int32 and items[k].is_cuda and items[k].is_contiguous()
               for k in ("item_seq", "item_range", "seq_active", "seq_chunk"))
    assert items["item_seq"].numel() == items["item_range"].numel() >= batch
    assert items["seq_active"].numel() == items["seq_chunk"].numel() == batch
    return (items["item_seq"].data_ptr(), items["item_range"].data_ptr(), items["seq_active"].data_ptr(),
            items["seq_chunk"].data_ptr(), items["item_seq"].numel(), items["max_ranges"])


def attn_decode(q, k_cache, v_cache, context_lens, block_tables, scale: float, num_splits: int = 0,
                max_context: int | None = None, min_range: int = 0, items: dict | None = None) -> torch.Tensor:
    """q [B,Hq,128] (token stride free); caches [P,256,Hkv,128]; one query per sequence.  `num_splits` key ranges
    are offered to every sequence, of at least `min_range` keys each (0 = 64); or `items`, an uploaded
    plan_decode_items work list, names the ranges one by one."""
    _gpu_bf16(q, k_cache, v_cache)
    b, hq, d = q.shape
    hkv = k_cache.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == d and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert context_lens.dtype == torch.int32 and block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    if items is not None:
        num_splits = items["max_ranges"]
    elif num_splits <= 0:
        mc = max_context if max_context is not None else block_tables.shape[1] * KV_PAGE
        num_splits = suggest_decode_splits(b, hkv, mc)
    out = torch.empty(b, hq, d, dtype=BF16, device=q.device)
    ws = 0
    if num_splits > 1:
        nbytes = _hip.lib().nvh_attn_decode_workspace_bytes(b, hq, d, num_splits)
        ws_t = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
        ws = ws_t.data_ptr()
    if items is not None:
        _hip.call("nvh_attn_decode_items", q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
                  context_lens.data_ptr(), block_tables.data_ptr(), block_tables.stride(0), b, hq, hkv, d, float(scale),
                  out.data_ptr(), *_items_args(items, b), ws, _stream())
        return out
    _hip.call("nvh_attn_decode", q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
              context_lens.data_ptr(), block_tables.data_ptr(), block_tables.stride(0), b, hq, hkv, d, float(scale),
              out.data_ptr(), num_splits, min_range, ws, _stream())
    return out


def attn_decode_fused_qkv(qkv, num_q_heads: int, num_kv_heads: int, q_weight, k_weight, eps: float, positions, cos_sin,
                          k_cache, v_cache, context_lens, block_tables, scale: float, num_splits: int = 0,
                          max_context: int | None = None, min_range: int = 0, items: dict | None = None) -> torch.Tensor:
    """Decode attention straight from the raw fused qkv rows [B, (Hq+2Hkv)*128]: q/k norm + RoPE + KV store +
    attention in one launch."""
    _gpu_bf16(qkv, q_weight, k_weight, k_cache, v_cache)
    b, d = qkv.shape[0], 128
    assert qkv.stride(1) == 1 and k_cache.is_contiguous() and v_cache.is_contiguous() and positions.dtype == torch.int64
    if items is not None:
        num_splits = items["max_ranges"]
    elif num_splits <= 0:
        mc = max_context if max_context is not None else block_tables.shape[1] * KV_PAGE
        num_splits = suggest_decode_splits(b, num_kv_heads, mc)
    out = torch.empty(b, num_q_heads, d, dtype=BF16, device=qkv.device)
    ws = 0
    if num_splits > 1:
        ws_t = torch.empty(_hip.lib().nvh_attn_decode_workspace_bytes(b, num_q_heads, d, num_splits) // 4,
                           dtype=torch.float32, device=qkv.device)
        ws = ws_t.data_ptr()
    if items is not None:
        _hip.call("nvh_attn_decode_fused_qkv_items", qkv.data_ptr(), qkv.stride(0), q_weight.data_ptr(), k_weight.data_ptr(),
                  float(eps), positions.data_ptr(), cos_sin.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                  context_lens.data_ptr(), block_tables.data_ptr(), block_tables.stride(0), b, num_q_heads, num_kv_heads, d,
                  float(scale), out.data_ptr(), *_items_args(items, b), ws, _stream())
        return out
    _hip.call("nvh_attn_decode_fused_qkv", qkv.data_ptr(), qkv.stride(0), q_weight.data_ptr(), k_weight.data_ptr(), float(eps),
              positions.data_ptr(), cos_sin.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), context_lens.data_ptr(),
              block_tables.data_ptr(), block_tables.stride(0), b, num_q_heads, num_kv_heads, d, float(scale), out.data_ptr(),
              num_splits, min_range, ws, _stream())
    return out


# ---------------------------------------------------------------- sampling
def _logits_args(logits):
    assert logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1
    if logits.dtype == torch.float32:
        return 1
    if logits.dtype == BF16:
        return 0
    raise TypeError(f"logits dtype {logits.dtype} unsupported")


def _sample_ws(logits):
    n = _hip.lib().nvh_sample_workspace_bytes(logits.shape[0], logits.shape[1])
    return torch.empty(n, dtype=torch.uint8, device=logits.device) if n else None


def argmax(logits: torch.Tensor) -> torch.Tensor:
    f = _logits_args(logits)
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    ws = _sample_ws(logits)
    _hip.call("nvh_argmax", logits.data_ptr(), f, logits.stride(0), out.data_ptr(), logits.shape[0], logits.shape[1],
              _p(ws), _stream())
    return out


def sample(logits: torch.Tensor, temperatures: torch.Tensor, seed: int, offset: int) -> torch.Tensor:
    f = _logits_args(logits)
    assert temperatures.dtype == torch.float32 and temperatures.numel() == logits.shape[0]
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    ws = _sample_ws(logits)
    _hip.call("nvh_sample", logits.data_ptr(), f, logits.stride(0), temperatures.data_ptr(), out.data_ptr(),
              logits.shape[0], logits.shape[1], seed, offset, _p(ws), _stream())
    return out


def lm_head_sample_fused(m: int, n: int, k: int) -> bool:
    """Whether `lm_head_sample` covers an [m, k] x [n, k]^T projection (it takes the 256 x 256 GEMM tile)."""
    return bool(_hip.lib().nvh_lm_head_sample_fused(m, n, k))


def lm_head_sample(x: torch.Tensor, weight: torch.Tensor, temperatures: torch.Tensor, seed: int, offset: int) -> torch.Tensor:
    """sample(x @ W^T, temperatures, seed, offset) in one GEMM launch + a pick: the logits are never written."""
    _gpu_bf16(x, weight)
    m, k = x.shape
    n = weight.shape[0]
    assert x.stride(1) == 1 and weight.is_contiguous() and weight.shape[1] == k
    assert temperatures.dtype == torch.float32 and temperatures.numel() == m
    out = torch.empty(m, dtype=torch.int64, device=x.device)
    ws = torch.empty(_hip.lib().nvh_lm_head_sample_workspace_bytes(m, n), dtype=torch.uint8, device=x.device)
    _hip.call("nvh_lm_head_sample", x.data_ptr(), x.stride(0), weight.data_ptr(), temperatures.data_ptr(), out.data_ptr(), m, n, k,
              seed, offset, ws.data_ptr(), _stream())
    return out


def sample_shard(logits: torch.Tensor, temperatures: torch.Tensor, seed: int, offset: int, col_offset: int, rank: int,
                 world: int) -> torch.Tensor:
    """Best candidate of every row of a vocabulary shard [rows, shard_vocab] (columns col_offset.. of the full row), as
    bf16 bytes in this rank's segment of a [world, rows, 8] tensor (zeros elsewhere): sum it over the ranks, then
    `sample_pick`."""
    f = _logits_args(logits)
    rows, vocab = logits.shape
    assert temperatures.dtype == torch.float32 and temperatures.numel() == rows
    enc = torch.empty(world, rows, 8, dtype=BF16, device=logits.device)
    ws = torch.empty(max(16, _hip.lib().nvh_sample_shard_workspace_bytes(rows, vocab)), dtype=torch.uint8, device=logits.device)
    _hip.call("nvh_sample_shard", logits.data_ptr(), f, logits.stride(0), temperatures.data_ptr(), enc.data_ptr(), rows, vocab,
              col_offset, rank, world, seed, offset, ws.data_ptr(), _stream())
    return enc


def sample_pick(enc: torch.Tensor) -> torch.Tensor:
    """The winner per row among the ranks' candidates (`sample_shard` summed over the ranks): int64 [rows]."""
    world, rows, _ = enc.shape
    assert enc.dtype == BF16 and enc.is_contiguous()
    out = torch.empty(rows, dtype=torch.int64, device=enc.device)
    _hip.call("nvh_sample_pick", enc.data_ptr(), out.data_ptr(), rows, world, _stream())
    return out


def sample_with_noise(logits: torch.Tensor, temperatures: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    f = _logits_args(logits)
    assert noise.dtype == torch.float32 and noise.is_contiguous() and noise.shape == logits.shape
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    ws = _sample_ws(logits)
    _hip.call("nvh_sample_with_noise", logits.data_ptr(), f, logits.stride(0), temperatures.data_ptr(), noise.data_ptr(),
              out.data_ptr(), logits.shape[0], logits.shape[1], _p(ws), _stream())
    return out
