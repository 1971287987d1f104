"""Binding of the native step executor (nvh_qwen3_forward / nvh_qwen3_logits): builds the
`nvh_qwen3_model` descriptor from a Qwen3ForCausalLM module once, and runs one step per call.

The descriptor holds raw device pointers into the module's parameters and KV-cache views; the
module keeps owning them.  With an `XgmiComm` the same executor runs one tensor-parallel rank: the
descriptor then holds the rank's shard sizes and the exchange handle."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _hip, ops

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
PP = C.POINTER(vp)

MODE_PACKED, MODE_PAGED, MODE_DECODE = 0, 1, 2


class ModelDesc(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("num_q_heads", i32), ("num_kv_heads", i32), ("head_dim", i32),
                ("inter", i32), ("vocab_rows", i32), ("vocab_start", i32), ("eps", f32), ("reserved2", i32),
                ("max_position", i64), ("embed", vp), ("final_norm", vp), ("lm_head", vp), ("cos_sin", vp),
                ("qkv_w", PP), ("o_w", PP), ("q_norm", PP), ("k_norm", PP), ("gate_up_w", PP), ("gate_up_inter", PP), ("down_w", PP),
                ("input_norm", PP), ("post_norm", PP), ("k_cache", PP), ("v_cache", PP), ("comm", vp),
                ("qkv_w_scaled", PP), ("gate_up_scaled", PP)]


class StepDesc(C.Structure):
    _fields_ = [("num_tokens", i64), ("batch", i64), ("input_ids", vp), ("positions", vp), ("cu_seqlens_q", vp),
                ("cu_seqlens_k", vp), ("slot_mapping", vp), ("context_lens", vp), ("block_tables", vp),
                ("table_stride", i64), ("max_seqlen_q", i32), ("max_seqlen_k", i32), ("mode", i32), ("decode_splits", i32),
                ("decode_min_range", i32), ("decode_prefix", i64), ("cu_seqlens_q_tail", vp), ("cu_seqlens_k_tail", vp),
                ("decode_items", i64), ("decode_item_seq", vp), ("decode_item_range", vp), ("decode_seq_active", vp),
                ("decode_seq_chunk", vp), ("layer_events_wait", i32), ("layer_events_record", i32)]


def _ptr(t):
    return 0 if t is None or t.numel() == 0 else t.data_ptr()


def _raw_bf16(ptr: int, numel: int) -> torch.Tensor:
    """bf16 view of device memory owned by someone else (an area of the step workspace)."""
    class _Raw:
        __cuda_array_interface__ = {"shape": (numel,), "typestr": "<u2", "data": (ptr, False), "version": 3}
    return torch.as_tensor(_Raw(), device="cuda").view(torch.bfloat16)


class NativeQwen3:
    def __init__(self, model, fuse_swiglu: bool = True, comm=None, reduce_fn=None):
        """`comm`: the peer exchange of a tensor-parallel rank (sums happen inside the native step).  `reduce_fn`: for a
        tensor-parallel rank WITHOUT that exchange - an in-place sum over the ranks (torch.distributed / RCCL) that the
        step calls between its parts; the launches still come from native code and still capture into a hipGraph."""
        self.lib = _hip.lib()
        self.model = model
        self.comm = comm
        self.reduce_fn = reduce_fn if comm is None else None
        self.fuse_swiglu = fuse_swiglu
        self.rebuild()

    def rebuild(self):
        """(Re)read every pointer - call again after the KV cache views are bound."""
        m = self.model
        layers = list(m.model.layers)
        a0 = layers[0].self_attn
        assert not a0.qkv_bias, "the native executor covers the q/k-norm (bias-free) Qwen3 layout"
        self._arrays = {}

        def arr(name, tensors):
            a = (vp * len(tensors))(*[_ptr(t) for t in tensors])
            self._arrays[name] = a
            return C.cast(a, PP)

        d = ModelDesc()
        d.hidden = m.model.embed_tokens.weight.shape[1]
        d.layers = len(layers)
        d.num_q_heads, d.num_kv_heads, d.head_dim = a0.num_heads, a0.num_kv_heads, a0.head_dim
        d.inter = layers[0].mlp.down_proj.weight.shape[1]
        d.vocab_rows = m.lm_head.weight.shape[0]
        d.vocab_start = m.lm_head.vocab_start_idx
        d.comm = self.comm.handle if self.comm is not None else None
        d.eps = layers[0].input_layernorm.eps
        table = a0.rotary_emb.table2d()
        self._keep = [table]
        d.max_position = table.shape[0]
        d.embed, d.final_norm = m.model.embed_tokens.weight.data_ptr(), m.model.norm.weight.data_ptr()
        d.lm_head, d.cos_sin = m.lm_head.weight.data_ptr(), table.data_ptr()
        d.qkv_w = arr("qkv", [l.self_attn.qkv_proj.weight for l in layers])
        d.o_w = arr("o", [l.self_attn.o_proj.weight for l in layers])
        d.q_norm = arr("qn", [l.self_attn.q_norm.weight for l in layers])
        d.k_norm = arr("kn", [l.self_attn.k_norm.weight for l in layers])
        d.gate_up_w = arr("gu", [l.mlp.gate_up_proj.weight for l in layers])
        # interleaved copies for the fused SwiGLU epilogue (one extra copy of the gate/up weights)
        if self.fuse_swiglu and layers[0].mlp.gate_up_proj.weight.shape[0] % 32 == 0:
            # refreshed IN PLACE on later rebuilds: captured graphs keep pointing at these buffers
            prev = getattr(self, "_inter", None) or [None] * len(layers)
            self._inter = [ops.interleave_gate_up(l.mlp.gate_up_proj.weight.data, out=p) for l, p in zip(layers, prev)]
            d.gate_up_inter = arr("gui", self._inter)
        else:
            self._inter = None
            d.gate_up_inter = None
        # RMSNorm as a row scale (prefill-sized steps on one rank, hidden sizes whose O / down projections take the kernel that
        # leaves the row sums): copies of the QKV and gate/up weights with the norm weight of their input folded in
        self.rowscale = (self.comm is None and self.reduce_fn is None and self._inter is not None and d.hidden <= 2048
                         and os.environ.get("NANOVLLM_ROWSCALE", "1") != "0"
                         and any(self.lib.nvh_gemm_residual_rowsq_parts(mm, d.hidden, d.num_q_heads * d.head_dim) > 0 for mm in range(64, 4097, 64)))
        if self.rowscale:
            prev_q = getattr(self, "_qkv_scaled", None) or [None] * len(layers)
            prev_g = getattr(self, "_gu_scaled", None) or [None] * len(layers)

            def folded(weight, norm, out):
                f = (weight.float() * norm.float()[None, :]).to(weight.dtype)
                if out is None:
                    return f
                out.copy_(f)              # refreshed IN PLACE on later rebuilds, like the interleaved copies
                return out

            self._qkv_scaled = [folded(l.self_attn.qkv_proj.weight.data, l.input_layernorm.weight.data, p) for l, p in zip(layers, prev_q)]
            self._gu_scaled = [folded(w, l.post_attention_layernorm.weight.data, p) for w, l, p in zip(self._inter, layers, prev_g)]
            d.qkv_w_scaled = arr("qkvs", self._qkv_scaled)
            d.gate_up_scaled = arr("gus", self._gu_scaled)
        else:
            self._qkv_scaled = self._gu_scaled = None
            d.qkv_w_scaled = None
            d.gate_up_scaled = None
        d.down_w = arr("dn", [l.mlp.down_proj.weight for l in layers])
        d.input_norm = arr("in", [l.input_layernorm.weight for l in layers])
        d.post_norm = arr("pn", [l.post_attention_layernorm.weight for l in layers])
        d.k_cache = arr("kc", [l.self_attn.attn.k_cache for l in layers])
        d.v_cache = arr("vc", [l.self_attn.attn.v_cache for l in layers])
        self.desc = d
        self.device = m.lm_head.weight.device
        # per-layer events that order two steps in flight on two streams (created once; nothing is created inside a step)
        self.layer_events = self.comm is None and self.reduce_fn is None and self.lib.nvh_qwen3_prepare_layer_events(d.layers) >= d.layers

    def forward(self, input_ids, positions, *, mode, batch, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=0,
                max_seqlen_k=0, slot_mapping=None, context_lens=None, block_tables=None, decode_splits=1,
                decode_prefix=0, cu_seqlens_q_tail=None, cu_seqlens_k_tail=None, decode_min_range=0,
                decode_items=None, layer_events=(0, 0), out=None, workspace=None):
        """One step: final-norm hidden states [T, hidden].  `out` / `workspace` (uint8, at least workspace_bytes()):
        buffers of the caller's instead of fresh ones - a capture that runs two steps side by side on two streams
        allocates them beforehand."""
        t = input_ids.numel()
        st = StepDesc()
        st.num_tokens, st.batch = t, batch
        st.input_ids, st.positions = input_ids.data_ptr(), positions.data_ptr()
        st.cu_seqlens_q, st.cu_seqlens_k = _ptr(cu_seqlens_q), _ptr(cu_seqlens_k)
        st.slot_mapping, st.context_lens = _ptr(slot_mapping), _ptr(context_lens)
        st.block_tables = _ptr(block_tables)
        st.table_stride = block_tables.stride(0) if block_tables is not None else 0
        st.max_seqlen_q, st.max_seqlen_k, st.mode, st.decode_splits = max_seqlen_q, max_seqlen_k, mode, max(1, decode_splits)
        st.decode_min_range = decode_min_range
        if decode_items is not None:               # work list of the decode attention: dict of int32 device tensors
            st.decode_items = decode_items["item_seq"].numel()
            st.decode_item_seq, st.decode_item_range = decode_items["item_seq"].data_ptr(), decode_items["item_range"].data_ptr()
            st.decode_seq_active, st.decode_seq_chunk = decode_items["seq_active"].data_ptr(), decode_items["seq_chunk"].data_ptr()
        if decode_prefix:
            st.decode_prefix = decode_prefix
            st.cu_seqlens_q_tail, st.cu_seqlens_k_tail = _ptr(cu_seqlens_q_tail), _ptr(cu_seqlens_k_tail)
        st.layer_events_wait, st.layer_events_record = layer_events     # two steps in flight (nvh_step.layer_events_wait)
        nbytes = self.lib.nvh_qwen3_workspace_bytes(C.byref(self.desc), t, batch, st.decode_splits)
        ws = workspace if workspace is not None else torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        assert ws.numel() >= nbytes
        if out is None:
            out = torch.empty(t, self.desc.hidden, dtype=torch.bfloat16, device=self.device)
        assert out.shape == (t, self.desc.hidden) and out.is_contiguous()
        stream = torch.cuda.current_stream().cuda_stream
        if self.reduce_fn is None:
            _hip.check(self.lib.nvh_qwen3_forward(C.byref(self.desc), C.byref(st), out.data_ptr(), ws.data_ptr(), nbytes, stream),
                       "nvh_qwen3_forward")
            return out
        # one native call per part of the step, the caller's sum over the ranks between them
        emb, proj = vp(), vp()
        _hip.check(self.lib.nvh_qwen3_exchange_buffers(C.byref(self.desc), t, batch, st.decode_splits, ws.data_ptr(),
                                                       C.byref(emb), C.byref(proj)), "nvh_qwen3_exchange_buffers")
        n = t * self.desc.hidden
        sums = (_raw_bf16(emb.value, n), _raw_bf16(proj.value, n))
        last = 2 * self.desc.layers + 1
        for part in range(last + 1):
            _hip.check(self.lib.nvh_qwen3_forward_parts(C.byref(self.desc), C.byref(st), out.data_ptr(), ws.data_ptr(), nbytes,
                                                        part, part, stream), "nvh_qwen3_forward_parts")
            if part < last:
                self.reduce_fn(sums[0] if part == 0 else sums[1])
        return out

    def workspace_bytes(self, num_tokens: int, batch: int, decode_splits: int) -> int:
        return max(16, self.lib.nvh_qwen3_workspace_bytes(C.byref(self.desc), num_tokens, batch, max(1, decode_splits)))

    def logits(self, hidden, cu_seqlens_q=None, need=None, num_rows=None, gather=True):
        """lm_head over the last scheduled row of the selected sequences (cu_seqlens_q None: `hidden`
        rows are used directly).  `gather=False` (tensor-parallel ranks): this rank's vocabulary shard [n, vocab_rows]
        stays where it is - for sampling over the shards (ops.sample_shard)."""
        n = hidden.shape[0] if cu_seqlens_q is None else num_rows
        if not gather:
            rows = hidden if cu_seqlens_q is None else ops.gather_last_rows(hidden, cu_seqlens_q, need, n)
            return ops.linear(rows, self.model.lm_head.weight)
        if self.reduce_fn is not None:
            # this rank's vocabulary shard, then the gather to rank 0 through torch.distributed (layers/embed_head.py:205-209)
            from . import comm as comm_mod
            shard = torch.empty(n, self.desc.vocab_rows, dtype=torch.bfloat16, device=self.device)
            rows = torch.empty(n, self.desc.hidden, dtype=torch.bfloat16, device=self.device) if cu_seqlens_q is not None else None
            _hip.check(self.lib.nvh_qwen3_logits(C.byref(self.desc), hidden.data_ptr(), _ptr(cu_seqlens_q), _ptr(need), n,
                                                 _ptr(rows), shard.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "nvh_qwen3_logits")
            return comm_mod.gather_cols(shard, 0)
        if self.comm is None:
            out = torch.empty(n, self.desc.vocab_rows, dtype=torch.bfloat16, device=self.device)
        elif self.comm.rank == 0:
            out = torch.empty(n, self.comm.world * self.desc.vocab_rows, dtype=torch.bfloat16, device=self.device)
        else:
            out = None
        rows = torch.empty(n, self.desc.hidden, dtype=torch.bfloat16, device=self.device) if cu_seqlens_q is not None else None
        _hip.check(self.lib.nvh_qwen3_logits(C.byref(self.desc), hidden.data_ptr(), _ptr(cu_seqlens_q), _ptr(need), n,
                                             _ptr(rows), _ptr(out), torch.cuda.current_stream().cuda_stream),
                   "nvh_qwen3_logits")
        return out
