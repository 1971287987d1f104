"""Model runner: turns the scheduler's sequence list into one forward pass on an MI355X.

Same contract as the reference's engine/model_runner.py:49-715 -
``ModelRunner(config, rank, event)``, ``call(method, *args)`` with methods ``"run"`` and
``"exit"``, ``run(seqs) -> (token_ids | None, seq_need_compute_logits)``, side effect
``config.num_kvcache_blocks`` - with a different inside:

  * batch flattening is vectorised numpy over per-sequence scalars (no per-token Python), and all
    step metadata crosses PCIe in ONE pinned-buffer copy instead of eight;
  * the forward is the gfx950 kernel path (nanovllm.ops); one-token-per-sequence steps replay a
    hipGraph whose input buffers are the copy's destination;
  * a step in which no sequence finishes its prompt (a lone chunked prefill) returns ``([], need)``
    instead of tripping the scheduler's length assertion (the reference samples every sequence
    there and fails in postprocess);
  * an already initialised process group (torchrun, one rank per GPU) is used as is: every rank
    then calls ``run`` with the same sequences and no shared-memory control plane is needed.
"""
from __future__ import annotations

import collections
import contextlib
import os
import sys
import pickle
import time
from multiprocessing.shared_memory import SharedMemory

import numpy as np
import torch
import torch.distributed as dist

from .. import _hip, comm
from ..config import Config, model_dtype
from ..layers.sampler import Sampler
from ..models.qwen3 import Qwen3ForCausalLM
from ..native import MODE_DECODE, MODE_PACKED, MODE_PAGED, NativeQwen3
from .. import ops
from ..utils.context import get_context, reset_context, set_context
from ..utils.loader import has_checkpoint, load_model, load_tensors, synthetic_tensors
from .sequence import Sequence

try:
    from .. import _hostpath
except ImportError:                       # not built: same results from the numpy implementation
    _hostpath = None

SHM_NAME = "nanovllm"
SHM_BYTES = 2 ** 20


# ------------------------------------------------------------------------------------------
# host-side batch flattening (device independent; unit-tested on CPU)
# ------------------------------------------------------------------------------------------
def flatten_batch(seqs, block_size: int = 256) -> dict:
    """Per-step tensors of reference model_runner.py:438-495 as numpy arrays: the C helper
    (csrc/host/hostpath.c) when it is built, else the numpy implementation below (same results)."""
    if _hostpath is not None:
        return _hostpath.flatten_batch(seqs, block_size)
    return _flatten_batch_py(seqs, block_size)


def _flatten_batch_py(seqs, block_size: int = 256) -> dict:
    """Per-step tensors of reference model_runner.py:438-495 as numpy arrays.

    input_ids / positions int64 [T]; cu_seqlens_q / cu_seqlens_k int32 [B+1]; slot_mapping int32
    (one entry per new token of every sequence that owns a block table); context_lens int32 [B];
    need int32 [B'] (sequences whose scheduled tokens reach their end and that own a block table);
    block_tables int32 [B, W] padded with -1, or None when no sequence has cached keys.
    """
    nseq = len(seqs)
    cached = np.fromiter((s.num_cached_tokens for s in seqs), dtype=np.int64, count=nseq)
    new = np.fromiter((s.num_new_tokens for s in seqs), dtype=np.int64, count=nseq)
    total = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=nseq)
    ntab = np.fromiter((len(s.block_table) for s in seqs), dtype=np.int64, count=nseq)
    ctx = cached + new
    cu_q = np.zeros(nseq + 1, dtype=np.int32)
    cu_k = np.zeros(nseq + 1, dtype=np.int32)
    np.cumsum(new, out=cu_q[1:])
    np.cumsum(ctx, out=cu_k[1:])
    t = int(cu_q[-1])
    parts = [s[int(c):int(e)] for s, c, e in zip(seqs, cached, ctx)]
    input_ids = np.fromiter((tok for part in parts for tok in part), dtype=np.int64, count=t) if t else np.zeros(0, np.int64)
    seq_of_tok = np.repeat(np.arange(nseq), new)
    positions = np.arange(t, dtype=np.int64) - np.repeat(cu_q[:-1].astype(np.int64), new) + np.repeat(cached, new)
    need = np.nonzero((total == ctx) & (ntab > 0))[0].astype(np.int32)
    width = int(ntab.max()) if nseq else 0
    tables = None
    if width:
        tables = np.full((nseq, width), -1, dtype=np.int32)
        for i, s in enumerate(seqs):
            tables[i, :len(s.block_table)] = s.block_table
        has = ntab[seq_of_tok] > 0
        pos_h, seq_h = positions[has], seq_of_tok[has]
        slot_mapping = (tables[seq_h, pos_h // block_size].astype(np.int64) * block_size + pos_h % block_size).astype(np.int32)
    else:
        slot_mapping = np.zeros(0, dtype=np.int32)
    return dict(input_ids=input_ids, positions=positions, cu_seqlens_q=cu_q, cu_seqlens_k=cu_k,
                max_seqlen_q=int(new.max()) if nseq else 0, max_seqlen_k=int(ctx.max()) if nseq else 0,
                slot_mapping=slot_mapping, context_lens=ctx.astype(np.int32), need=need,
                block_tables=tables if int(cu_k[-1]) > int(cu_q[-1]) else None)


class _Staging:
    """One pinned host buffer + one device buffer per step: named arrays are packed back to back
    (16-byte aligned), copied with a single async H2D, and handed out as typed device views."""

    def __init__(self, device):
        self.device = device
        self.host = torch.empty(1 << 16, dtype=torch.uint8, device="cpu", pin_memory=device.type == "cuda")
        self.event = torch.cuda.Event() if device.type == "cuda" else None

    def upload(self, arrays: dict) -> dict:
        offs, n = {}, 0
        for name, a in arrays.items():
            offs[name] = n
            n += (a.nbytes + 15) & ~15
        n = max(n, 16)
        if self.host.numel() < n:
            self.host = torch.empty(max(n, 2 * self.host.numel()), dtype=torch.uint8, device="cpu",
                                    pin_memory=self.device.type == "cuda")
        if self.event is not None:
            self.event.synchronize()            # the previous step's copy has left the buffer
        hnp = self.host.numpy()
        for name, a in arrays.items():
            hnp[offs[name]:offs[name] + a.nbytes] = a.reshape(-1).view(np.uint8)
        dev = torch.empty(n, dtype=torch.uint8, device=self.device)
        dev.copy_(self.host[:n], non_blocking=True)
        if self.event is not None:
            self.event.record()
        out = {}
        for name, a in arrays.items():
            view = dev[offs[name]:offs[name] + a.nbytes].view(_TORCH_DTYPE[a.dtype.type])
            out[name] = view.view(a.shape)
        return out


_TORCH_DTYPE = {np.int64: torch.int64, np.int32: torch.int32, np.float32: torch.float32}


class PendingTokens:
    """Sampled ids of a launched step on their way to the host (ModelRunner.run_async)."""

    def __init__(self, buf, event, need: list[int], count: int):
        self.buf, self.event, self.need, self.count = buf, event, need, count

    def result(self) -> list[int]:
        """The step's ids, waiting for them.  A step that sampled nothing (a chunk that does not end its prompt) has nothing
        to wait for: how far the host may run ahead of the GPU over such steps is bounded where steps are launched
        (ModelRunner._step: LANES steps)."""
        if not self.count:
            return []
        if self.event is not None:
            self.event.synchronize()
        return self.buf[:self.count].tolist()


class ModelRunner:
    def __init__(self, config: Config, rank: int, event):
        self.config = config
        hf = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.event = event
        self.external_group = dist.is_initialized()
        self.shm = None
        self._msg_seq = 0                       # control channel: messages posted (rank 0) / taken (workers) so far
        self.shard_sampling = False
        # one GPU: batches from about 230 rows sample inside the LM head's GEMM (NANOVLLM_FUSED_HEAD_SAMPLER=0: two launches)
        self.fused_head = os.environ.get("NANOVLLM_FUSED_HEAD_SAMPLER", "1") != "0"
        self.prev_sampled = None                # run_async: the last sampler output, kept on the device for the next step
        self._token_bufs, self._token_flip = [None, None], 0    # pinned landing buffers of two steps in flight

        # NANOVLLM_SHARE_DEVICE=1: every rank on device 0 with gloo as the control plane (rehearsal on a one-GPU box)
        share_device = os.environ.get("NANOVLLM_SHARE_DEVICE", "0") == "1"
        if self.external_group:
            assert dist.get_world_size() == self.world_size and dist.get_rank() == rank
        elif self.world_size > 1:
            port = int(os.environ.get("NANOVLLM_PORT", "2333"))
            dist.init_process_group("gloo" if share_device else "nccl", f"tcp://127.0.0.1:{port}",
                                    world_size=self.world_size, rank=rank)
        device_index = int(os.environ.get("LOCAL_RANK", rank)) if self.external_group else (0 if share_device else rank)
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        _hip.check(_hip.lib().nvh_device_check(device_index), "nvh_device_check")

        self.split_mixed = os.environ.get("NANOVLLM_SPLIT_MIXED", "1") != "0"     # A/B switch: decode prefix of mixed steps
        # Two steps in flight (see _step): consecutive prefill-sized steps that do not wait for each other's sampled ids
        # alternate between the caller's stream and a side stream (Qwen3-0.6B: a 1024-token chunk 2.77 -> 2.18 ms; the
        # Qwen3-14B shape 27.7 -> 26.4 ms).  NANOVLLM_STEP_OVERLAP=0 turns it off.
        self.step_overlap = self.world_size == 1 and os.environ.get("NANOVLLM_STEP_OVERLAP", "1") != "0"
        self._side_streams = []                 # created with the first steps that take them
        self._last_step = None                  # lane, completion event and event set of the step launched last
        self._lane_done = {}                    # side lane -> completion event of the last step launched on it
        self._recent_steps = collections.deque()   # completion events of the steps launched last (bounds the run-ahead)
        self._step_depth, self._step_fb = 0, None
        self.comm = None
        if self.world_size > 1:
            # staging: the widest all-reduce (a full token budget of hidden rows) or logits gather (every sequence)
            vocab_shard = hf.vocab_size // self.world_size
            nbytes = 2 * max(config.max_num_batched_tokens * hf.hidden_size, config.max_num_seqs * vocab_shard)
            self.comm = comm.install(rank, self.world_size, nbytes)
            if self.comm is None:
                # No peer exchange: the native step still issues every launch, the sums between its parts go through
                # torch.distributed.  RCCL collectives capture into the decode hipGraphs with the launches; any other
                # backend (gloo rehearsals stage through the host) cannot be captured and runs eagerly.
                print(f"[nanovllm] rank {rank}: no peer exchange ({comm.fallback_reason}); sums over the ranks go through "
                      f"torch.distributed ({dist.get_backend()})", file=sys.stderr)
                if dist.get_backend() != "nccl":
                    self.enforce_eager = True

        dtype = model_dtype(hf)
        if dtype != torch.bfloat16:
            raise _hip.HipError(f"model dtype {dtype} unsupported: the kernel path is bf16 only")
        prev_dtype = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        torch.set_default_device(self.device)
        try:
            self.model = Qwen3ForCausalLM(hf)
            if has_checkpoint(config.model):
                load_model(self.model, config.model)
            else:
                load_tensors(self.model, synthetic_tensors(hf, seed=0, device=self.device, dtype=dtype))
            self.sampler = Sampler()
            self.staging = _Staging(self.device)
            self.graphs, self.graph_bs = {}, []
            # one native call per step (per rank; ranks meet through the xGMI exchange); the operator modules,
            # with torch.distributed collectives between them when the exchange is unavailable, otherwise.
            # NANOVLLM_NATIVE=0 forces the module path (parity tests use it).
            use_native = os.environ.get("NANOVLLM_NATIVE", "1") != "0" and not getattr(hf, "attention_bias", True)
            external_sum = comm.all_reduce if (self.world_size > 1 and self.comm is None) else None
            self.native = NativeQwen3(self.model, comm=self.comm, reduce_fn=external_sum) if use_native else None
            # tensor-parallel ranks sample over their vocabulary shards instead of gathering the logits on rank 0
            # (NANOVLLM_SHARD_SAMPLING=0: the reference's gather); needs shards that start on a multiple of 4 columns
            self.shard_sampling = (self.world_size > 1 and self.native is not None and (hf.vocab_size // self.world_size) % 4 == 0
                                   and os.environ.get("NANOVLLM_SHARD_SAMPLING", "1") != "0")
            self.warmup_model()
            self.allocate_kv_cache()
            if self.native is not None:
                self.native.rebuild()
            if not self.enforce_eager:
                try:
                    self.capture_graphs()
                except Exception as e:                              # noqa: BLE001
                    if self.world_size == 1 or self.comm is not None:
                        raise
                    # a collective that would not capture: keep the native step, run it eagerly
                    print(f"[nanovllm] rank {rank}: decode graphs unavailable with torch.distributed sums ({e}); eager steps",
                          file=sys.stderr)
                    self.graphs, self.graph_bs, self.enforce_eager = {}, [], True
        finally:
            torch.set_default_device("cpu")
            torch.set_default_dtype(prev_dtype)

        if self.world_size > 1 and not self.external_group:
            if rank == 0:
                try:
                    stale = SharedMemory(name=SHM_NAME, create=False)
                    stale.close()
                    stale.unlink()
                except FileNotFoundError:
                    pass
                self.shm = SharedMemory(name=SHM_NAME, create=True, size=SHM_BYTES)
                dist.barrier()
            else:
                dist.barrier()
                self.shm = SharedMemory(name=SHM_NAME)
                self.loop()

    def load_weights(self, named_tensors) -> int:
        """(Re)load parameters from (name, tensor) pairs in checkpoint naming and refresh everything derived
        from them (the native executor's pointer table and its interleaved gate/up copies)."""
        n = load_tensors(self.model, named_tensors)
        if self.native is not None:
            self.native.rebuild()
        return n

    # -------------------------------------------------------------- control plane (tp > 1)
    def exit(self):
        if self.shm is not None:
            self.shm.close()
            dist.barrier()
            if self.rank == 0:
                self.shm.unlink()
        self.graphs = {}
        torch.cuda.synchronize()
        if self.comm is not None:
            if dist.is_initialized():
                dist.barrier()                  # no rank unmaps a region a peer's kernel may still read
            comm.uninstall()
            self.comm = None
        if dist.is_initialized() and not self.external_group:
            dist.destroy_process_group()

    def loop(self):
        while True:
            method_name, args = self.read_shm()
            self.call(method_name, *args)
            if method_name == "exit":
                break

    # Control channel: one message slot + one event per worker, as in the reference (engine/model_runner.py:195-288),
    # plus an acknowledgement: the last 64 bytes of the segment hold, per worker, the number of the last message it has
    # taken.  Rank 0 posts message k+1 only when every worker has taken message k - the reference gets that for free
    # because its `run` waits for the GPU (and so for every rank) before the next call; a step launched ahead of its
    # results (`run_async`) returns at once.
    ACK_BYTES = 64

    def read_shm(self):
        assert self.world_size > 1 and self.rank > 0
        self.event.wait()
        n = int.from_bytes(self.shm.buf[0:4], "little")
        method_name, *args = pickle.loads(self.shm.buf[4:n + 4])
        self.event.clear()
        self._msg_seq = (self._msg_seq + 1) & 0xFFFFFFFF                      # a 4-byte counter on both sides: wraps, never overflows
        at = SHM_BYTES - self.ACK_BYTES + 4 * self.rank
        self.shm.buf[at:at + 4] = self._msg_seq.to_bytes(4, "little")        # after the clear: the next set cannot be lost
        return method_name, args

    def write_shm(self, method_name, *args):
        assert self.world_size > 1 and self.rank == 0
        data = pickle.dumps([method_name, *args])
        n = len(data)
        if n + 4 > SHM_BYTES - self.ACK_BYTES:
            raise RuntimeError(f"control message of {n} bytes exceeds the {SHM_BYTES - self.ACK_BYTES}-byte shared segment")
        deadline = time.perf_counter() + 120.0
        for r in range(1, self.world_size):                                   # everyone has taken the previous message
            at = SHM_BYTES - self.ACK_BYTES + 4 * r
            while int.from_bytes(self.shm.buf[at:at + 4], "little") != self._msg_seq:
                if time.perf_counter() > deadline:
                    raise RuntimeError(f"rank {r} has not taken control message {self._msg_seq} for 120 s")
                time.sleep(0)
        self._msg_seq = (self._msg_seq + 1) & 0xFFFFFFFF
        self.shm.buf[0:4] = n.to_bytes(4, "little")
        self.shm.buf[4:n + 4] = data
        for event in self.event:
            event.set()

    def call(self, method_name, *args):
        if self.shm is not None and self.rank == 0:
            self.write_shm(method_name, *args)
        return getattr(self, method_name)(*args)

    # -------------------------------------------------------------- init-time passes
    def warmup_model(self):
        """One maximum-shape forward without a cache, to measure peak activation memory."""
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        cfg = self.config
        num_seqs = max(min(cfg.max_num_batched_tokens // cfg.max_model_len, cfg.max_num_seqs), 1)
        length = min(cfg.max_model_len, cfg.max_num_batched_tokens)
        seqs = [Sequence([0] * length) for _ in range(num_seqs)]
        for seq in seqs:
            seq.num_new_tokens = length
        self.run(seqs)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def kv_block_bytes(self) -> int:
        hf = self.config.hf_config
        num_kv_heads = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        return 2 * hf.num_hidden_layers * self.block_size * num_kv_heads * head_dim * 2

    def allocate_kv_cache(self):
        """KV pool [2, L, blocks, 256, Hkv/tp, D]; per-layer views are bound to the attention modules.
        A positive ``config.num_kvcache_blocks`` is honoured (tests, benchmarks); otherwise the pool
        takes what ``gpu_memory_utilization`` leaves after weights and peak activations."""
        cfg, hf = self.config, self.config.hf_config
        if cfg.num_kvcache_blocks <= 0:
            free, total = torch.cuda.mem_get_info()
            used = total - free
            stats = torch.cuda.memory_stats()
            peak, current = stats["allocated_bytes.all.peak"], stats["allocated_bytes.all.current"]
            cfg.num_kvcache_blocks = int(total * cfg.gpu_memory_utilization - used - peak + current) // self.kv_block_bytes()
        assert cfg.num_kvcache_blocks > 0
        num_kv_heads = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        self.kv_cache = torch.zeros(2, hf.num_hidden_layers, cfg.num_kvcache_blocks, self.block_size, num_kv_heads, head_dim)
        layer_id = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer_id]
                module.v_cache = self.kv_cache[1, layer_id]
                module.max_context = cfg.max_model_len
                module.split_fn = self.module_decode_plan
                layer_id += 1

    # -------------------------------------------------------------- per-step input
    def prepare_model_input(self, seqs, fb: dict | None = None):
        fb = fb if fb is not None else flatten_batch(seqs, self.block_size)
        arrays = {k: fb[k] for k in ("input_ids", "positions", "cu_seqlens_q", "cu_seqlens_k", "slot_mapping",
                                     "context_lens", "need")}
        if fb["block_tables"] is not None:
            arrays["block_tables"] = fb["block_tables"]
        if self.rank == 0 or self.shard_sampling:
            temps = np.fromiter((s.temperature for s in seqs), dtype=np.float32, count=len(seqs))
            arrays["temperatures"] = temps[fb["need"]] if fb["need"].size else temps
        # Mixed step: the running sequences come first with one token each; the native executor sends that prefix
        # through the decode kernel and needs the remaining sequences' offsets rebased to zero.
        prefix = 0
        if self.native is not None and fb["block_tables"] is not None and fb["max_seqlen_q"] > 1 and self.split_mixed:
            new = np.diff(fb["cu_seqlens_q"])
            other = np.flatnonzero(new != 1)
            prefix = int(other[0]) if other.size else len(new)
            # worth two more launches per layer from 8 sequences on, or for any number of them once their contexts are
            # long (each would otherwise walk its whole context as a 64-row block of the var-len kernel)
            if prefix >= 8 or (prefix >= 1 and int(fb["context_lens"][:prefix].max()) >= 1024):
                arrays["cu_seqlens_q_tail"] = fb["cu_seqlens_q"][prefix:] - fb["cu_seqlens_q"][prefix]
                arrays["cu_seqlens_k_tail"] = fb["cu_seqlens_k"][prefix:] - fb["cu_seqlens_k"][prefix]
            else:
                prefix = 0
        fb["decode_prefix"] = prefix
        # Work list of the decode attention when long and short contexts share the launch (see decode_plan).
        rows = prefix if prefix else (len(seqs) if fb["block_tables"] is not None and fb["max_seqlen_q"] == 1 else 0)
        plan = None
        if self.native is not None and rows:
            lens = fb["context_lens"][:rows]
            if self.decode_plan(rows, int(lens.max()))[1]:
                plan = ops.plan_decode_items(lens, max_ranges=self.MAX_RANGES)
                arrays.update({k: plan[k] for k in ("item_seq", "item_range", "seq_active", "seq_chunk")})
        fed = fb.get("fed")
        if fed is not None:
            arrays["fed_dst"], arrays["fed_src"] = fed
        d = self.staging.upload(arrays)
        if fed is not None:
            self.prev_sampled.record_stream(torch.cuda.current_stream())       # (it may come from a step on the other stream)
            ops.feed_token_ids(d["input_ids"], d["fed_dst"], self.prev_sampled, d["fed_src"])
        fb["items"] = None if plan is None else {k: d[k] for k in ("item_seq", "item_range", "seq_active", "seq_chunk")}
        fb["tails"] = (d.get("cu_seqlens_q_tail"), d.get("cu_seqlens_k_tail"))
        set_context(d["cu_seqlens_q"], d["cu_seqlens_k"], fb["max_seqlen_q"], fb["max_seqlen_k"], d["slot_mapping"],
                    d["context_lens"], d.get("block_tables"), d["need"])
        return d["input_ids"], d["positions"], d.get("temperatures"), fb

    # -------------------------------------------------------------- forward
    @torch.inference_mode()
    def run_model(self, input_ids: torch.Tensor, positions: torch.Tensor):
        return self.model.compute_logits(self.model(input_ids, positions))

    def forward_hidden(self, input_ids, positions, fb):
        """Final-norm hidden states [T, H] for the step whose Context is set."""
        if self.native is None:
            return self.model(input_ids, positions)
        ctx = get_context()
        bs = len(fb["context_lens"])
        if ctx.block_tables is None:
            mode, splits = MODE_PACKED, 1
        elif fb["max_seqlen_q"] == 1:
            mode = MODE_DECODE
            splits = self.decode_plan(bs, fb["max_seqlen_k"])[0]
        else:
            mode, splits = MODE_PAGED, 1
        prefix = fb.get("decode_prefix", 0) if mode == MODE_PAGED else 0
        tails = fb.get("tails", (None, None))
        if prefix:
            longest = int(fb["context_lens"][:prefix].max())
            splits, listed = self.decode_plan(prefix, longest)
            if not listed:                  # an eager launch: ranges for the contexts at hand, not for max_model_len
                splits = ops.suggest_decode_splits(prefix, self.model.model.layers[0].self_attn.num_kv_heads, longest)
        items = fb.get("items")
        assert items is None or splits == self.MAX_RANGES
        return self.native.forward(input_ids, positions, mode=mode, batch=bs, cu_seqlens_q=ctx.cu_seqlens_q,
                                   cu_seqlens_k=ctx.cu_seqlens_k, max_seqlen_q=fb["max_seqlen_q"],
                                   max_seqlen_k=fb["max_seqlen_k"], slot_mapping=ctx.slot_mapping,
                                   context_lens=ctx.context_lens, block_tables=ctx.block_tables, decode_splits=splits,
                                   decode_prefix=prefix, cu_seqlens_q_tail=tails[0], cu_seqlens_k_tail=tails[1],
                                   decode_items=items, layer_events=fb.get("layer_events", (0, 0)))

    def decode_buckets(self) -> list[int]:
        max_bs = min(self.config.max_num_seqs, 512)
        return [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs + 1, 16))

    def decode_splits(self, bs: int) -> int:
        """Key-range count of the decode attention for a batch of `bs`: chosen per graph bucket and for the longest
        context a sequence may reach, so a captured graph and an eager step of the same batch sum in the same
        order (enforce_eager does not change a single bit of the output)."""
        bucket = next((b for b in self.decode_buckets() if b >= bs), bs)
        hkv = self.model.model.layers[0].self_attn.num_kv_heads
        return ops.suggest_decode_splits(bucket, hkv, self.config.max_model_len)

    LONG_CONTEXT = 2048          # a batch whose longest context exceeds this takes the long-context decode plan
    MAX_RANGES = 32              # key ranges a sequence can be cut into under that plan
    ITEM_SLACK = 64              # work-list slots a captured graph holds beyond one per sequence

    def decode_plan(self, bs: int, max_context: int) -> tuple[int, bool]:
        """(key ranges per sequence, work list?) of the decode attention for a batch of `bs` whose longest context is
        `max_context`.  Small batches already get many ranges each (to fill the chip) and need nothing else.  A large
        batch gets few or one - right as long as every context is short; with a long context among them that
        sequence would walk its keys in ONE workgroup per kv head and hold the whole step (256 sequences at 128 keys
        and one at 8192: 6.2 ms instead of 2.5), while offering every sequence many ranges launches tens of
        thousands of idle workgroups per layer.  Such a batch takes a work list (ops.plan_decode_items): the short
        sequences stay in one range, each long one is cut into up to MAX_RANGES."""
        short = self.decode_splits(bs)
        if max_context <= self.LONG_CONTEXT or short >= 16:
            return short, False
        return self.MAX_RANGES, True

    def module_decode_plan(self, bs: int, max_context: int) -> tuple[int, int]:
        """(key ranges, minimum keys per range) for the operator-module path, which has no work list: up to 64
        sequences with a long context among them are offered 16 ranges of at least 512 keys (short ones still use
        one); larger batches keep the uniform split, where that many idle workgroups would cost more than they save."""
        short = self.decode_splits(bs)
        if max_context <= self.LONG_CONTEXT or short >= 16 or bs > 64:
            return short, 0
        return 16, 512

    def _graph_for(self, fb):
        """The captured graph that fits this step, or None (eager)."""
        bs = len(fb["context_lens"])
        if (self.enforce_eager or not self.graphs or fb["block_tables"] is None or fb["max_seqlen_q"] != 1
                or fb["input_ids"].size != bs or fb["need"].size != bs or bs > self.graph_bs[-1]
                or fb["block_tables"].shape[1] > self.graph_width):
            return None
        graph = self.graphs[next(x for x in self.graph_bs if x >= bs)]
        if self.native is None:
            if self.module_decode_plan(bs, fb["max_seqlen_k"]) != (graph.splits, 0):
                return None                     # the graph was captured with the uniform split: eager
        elif self.decode_plan(bs, fb["max_seqlen_k"])[1] and graph.graph_long is None:
            return None
        return graph

    OVERLAP_MIN_TOKENS = 256     # steps with fewer new tokens are not worth another stream
    # streams that consecutive prefill-sized steps rotate over (the caller's + side streams).  Measured on Qwen3-0.6B / the 14B
    # shape, 1024-token chunks: 2 lanes 2.19 / 27.1 ms per chunk, 3 lanes 2.11 / 25.2, 4 lanes 2.15 / 25.2, 8 lanes 2.13; an
    # 8192-token prompt 24.7 % (2), 26.4 % (3), 25.0 % (4) of the MFMA peak
    LANES = max(1, int(os.environ.get("NANOVLLM_STEP_LANES", "3")))             # (1: every step on the caller's stream)

    @contextlib.contextmanager
    def _step(self, seqs, join: bool = False):
        """Everything a step enqueues happens inside this context: it flattens the batch, picks the stream the step runs
        on and orders it against the steps launched before it.

        Normally that is the caller's stream and stream order.  But a prefill-sized step (a chunk of a prompt, a wave of
        new prompts) is ~200 dependent launches that cannot fill the chip one at a time, and the next such step is often
        launched while it is still running (chunk steps return no ids, so no loop waits for them; the loop that runs
        one step ahead never waits before it has launched the next step).  Such steps in a row - eager native steps of one
        rank that take no token from the sampler output of a step still in flight - therefore rotate over LANES streams
        (the caller's first) and fill each other's idle time.  What orders two consecutive ones is per layer: the later
        step's QKV projection + KV store of layer l waits for the earlier step's attention of layer l
        (nvh_step.layer_events_wait), which covers keys the later step reads (the next chunk of the same prompt, shared
        prefix pages) and pages the earlier step still reads and the later one overwrites; the chain of these waits
        orders it against the steps before that one, and stream order against the step LANES back.  Any other step runs
        on the caller's stream behind everything launched before it.  At most LANES steps are ever enqueued ahead of the
        GPU.  `join`: the caller's stream also waits for this step (for entry points that hand device tensors back to
        their caller)."""
        if self._step_depth:                        # nested (run -> forward_sampled): the outer context owns the step
            self._step_depth += 1
            try:
                yield self._step_fb
            finally:
                self._step_depth -= 1
            return
        while len(self._recent_steps) >= max(self.LANES, 2):   # (two: what the loop that runs one step ahead needs)
            self._recent_steps.popleft().synchronize()
        fb = flatten_batch(seqs, self.block_size)
        fb["fed"] = self._fed_indices(seqs, fb)
        fb["graph"] = self._graph_for(fb)
        last = self._last_step
        eligible = (self.step_overlap and self.native is not None and self.native.layer_events and fb["graph"] is None
                    and fb["fed"] is None and fb["max_seqlen_q"] > 1 and fb["input_ids"].size >= self.OVERLAP_MIN_TOKENS
                    and all(len(s.block_table) for s in seqs))
        lane, wait, record = 0, 0, 0
        if eligible:
            record = 1
            if last is not None and last["record"]:
                lane, wait, record = (last["lane"] + 1) % self.LANES, last["record"], 3 - last["record"]
        fb["layer_events"] = (wait, record)
        main = torch.cuda.current_stream()
        while lane > len(self._side_streams):
            self._side_streams.append(torch.cuda.Stream(device=self.device))
        stream = self._side_streams[lane - 1] if lane else main
        if not wait:
            # not the next of a run of overlapping steps (then lane 0): the whole of everything on the side streams first
            for ev in self._lane_done.values():
                main.wait_event(ev)
            self._lane_done.clear()
        self._step_depth, self._step_fb = 1, fb
        try:
            with torch.cuda.stream(stream):
                yield fb
                done = torch.cuda.Event()
                done.record(stream)
        finally:
            self._step_depth, self._step_fb = 0, None
        self._last_step = dict(lane=lane, done=done, record=record)
        self._recent_steps.append(done)
        if lane:
            self._lane_done[lane] = done
            if join:
                main.wait_event(done)

    def _forward_rows(self, seqs, want_temps):
        """A step up to the final norm: (hidden, cu_seqlens_q | None, rows to keep | None, row count, temperatures, need).
        cu_seqlens_q None: the rows of `hidden` are the LM head's rows as they are (a graph replay); a row count of 0: no
        sequence finishes its scheduled prompt in this step and nothing is to be projected."""
        fb = self._step_fb
        assert fb is not None, "_forward_rows runs inside _step()"
        graph = fb["graph"]
        need_host = torch.from_numpy(fb["need"])
        if graph is not None:
            if fb["fed"] is not None:
                self.prev_sampled.record_stream(torch.cuda.current_stream())
            hidden, temps = graph.replay(fb, seqs, want_temps, self.prev_sampled)
            out = (hidden[:len(seqs)], None, None, len(seqs), temps, need_host)
        else:
            input_ids, positions, temps, _ = self.prepare_model_input(seqs, fb)
            hidden = self.forward_hidden(input_ids, positions, fb)
            ctx = get_context()
            if fb["need"].size == 0 and any(len(s.block_table) for s in seqs):
                out = (hidden, None, None, 0, temps, need_host)
            elif self.native is not None:
                out = (hidden, ctx.cu_seqlens_q, ctx.seq_need_compute_logits if fb["need"].size else None,
                       fb["need"].size or len(seqs), temps, need_host)
            else:                               # operator modules: the rows ParallelLMHead.forward would select
                keep = ctx.seq_need_compute_logits if fb["need"].size else None
                rows = ops.gather_last_rows(hidden, ctx.cu_seqlens_q, keep, fb["need"].size or len(seqs))
                out = (rows, None, None, rows.shape[0], temps, need_host)
        reset_context()
        return out

    @torch.inference_mode()
    def forward_logits(self, seqs, gather=True):
        """Everything of a step up to the sampler: (logits [B', V] | None, temperatures | None, need)
        with ``need`` a HOST int32 tensor (indices into ``seqs`` of the rows of ``logits``).  logits is
        None on ranks > 0 and when no sequence finishes its scheduled prompt in this step.  ``gather=False`` (tensor
        parallelism with the native executor): every rank keeps the logits of ITS vocabulary shard, [B', V / tp]."""
        with self._step(seqs, join=True):
            hidden, cu, keep, n, temps, need_host = self._forward_rows(seqs, self.rank == 0 or not gather)
            if n == 0:
                logits = None
            elif self.native is not None:
                logits = self.native.logits(hidden, cu, keep, n, gather=gather)
            else:
                logits = self.model.lm_head.forward_rows(hidden)
        return logits, temps, need_host

    @torch.inference_mode()
    def forward_sampled(self, seqs):
        """A whole step: (sampled ids - int64 on the device - | None, need).  None when nothing was to be sampled and,
        without sampling over the vocabulary shards, on ranks > 0.  Three routes to the same ids: the LM head with
        the sampler in its epilogue (one GPU, batches that take its 256 x 256 tile: the logits are never written),
        sampling over the ranks' shards, or the sampler over the (gathered) logits."""
        with self._step(seqs, join=True):
            return self._forward_sampled(seqs)

    def _forward_sampled(self, seqs):
        if self.fused_head and self.native is not None and self.world_size == 1:
            hidden, cu, keep, n, temps, need_host = self._forward_rows(seqs, True)
            if n == 0:
                return None, need_host
            w = self.model.lm_head.weight
            if ops.lm_head_sample_fused(n, w.shape[0], w.shape[1]):
                rows = hidden if cu is None else ops.gather_last_rows(hidden, cu, keep, n)
                self.sampler.calls += 1
                return ops.lm_head_sample(rows, w, temps, self.sampler.seed, self.sampler.calls), need_host
            logits = self.native.logits(hidden, cu, keep, n)
        else:
            logits, temps, need_host = self.forward_logits(seqs, gather=not self.shard_sampling)
        if logits is None:
            return None, need_host
        if self.shard_sampling:
            return self.sample_over_shards(logits, temps), need_host
        return self.sampler(logits, temps), need_host

    def sample_over_shards(self, shard_logits, temps):
        """Token ids (int64 on the device, the same on every rank) from the ranks' vocabulary shards of the logits,
        without gathering them (the reference gathers [B, V] on rank 0, layers/embed_head.py:205-209): every rank
        offers the best candidate of its shard - drawn with the noise of the full row's columns - and the candidates
        take the activations' sum over the ranks.  Same ids as the sampler over the gathered logits."""
        self.sampler.calls += 1
        shard = shard_logits.shape[1]
        enc = ops.sample_shard(shard_logits, temps, self.sampler.seed, self.sampler.calls, self.rank * shard, self.rank, self.world_size)
        comm.all_reduce(enc.view(-1))
        return ops.sample_pick(enc)

    @torch.inference_mode()
    def run(self, seqs):
        t0 = time.perf_counter()
        with self._step(seqs):
            ids, need_host = self.forward_sampled(seqs)
            token_ids = None
            if self.rank == 0:                  # (with shard sampling every rank holds the ids; the contract hands them to rank 0's caller)
                token_ids = [] if ids is None else ids.tolist()
        if self.comm is not None and time.perf_counter() - t0 > 2.0:
            self.comm.check()                   # a step this slow may have sat through an exchange timeout: raise, do not go on
        return token_ids, need_host

    def _fed_indices(self, seqs, fb):
        """(positions in the flat input_ids, rows of the previous step's sampler output) of the tokens whose ids have
        not been read back yet - placeholders left by an engine that runs one step ahead - or None."""
        dst = np.flatnonzero(fb["input_ids"] < 0).astype(np.int32)
        if dst.size == 0:
            return None
        assert self.prev_sampled is not None
        owner = np.searchsorted(fb["cu_seqlens_q"][1:], dst, side="right")
        src = np.fromiter((seqs[int(i)].pending_slot for i in owner), dtype=np.int32, count=dst.size)
        assert src.min() >= 0 and src.max() < self.prev_sampled.numel()
        return dst, src

    @torch.inference_mode()
    def run_async(self, seqs) -> "PendingTokens":
        """`run()` without the wait: the step is launched, the sampled ids start their way to a pinned host buffer and
        stay on the device for the next step (whose placeholder input ids are filled from them there); the caller reads
        them with `.result()` when it needs them - normally after it has launched the next step.  Under tensor
        parallelism every rank calls it with the same sequences: rank 0 samples, the ids reach the other ranks on the
        device (comm.broadcast_ids), and every rank can read them back."""
        with self._step(seqs):
            sampled, need_host = self.forward_sampled(seqs)
            need = need_host.tolist()
            count = len(need) if (need or any(len(s.block_table) for s in seqs)) else len(seqs)
            if count == 0:
                event = torch.cuda.Event()
                event.record()
                return PendingTokens(None, event, need, 0)
            if self.world_size > 1 and not self.shard_sampling:
                sampled = comm.broadcast_ids(sampled, count, self.device)
            n = sampled.numel()
            self._token_flip ^= 1
            buf = self._token_bufs[self._token_flip]
            if buf is None or buf.numel() < n:
                buf = self._token_bufs[self._token_flip] = torch.empty(max(n, 1024), dtype=torch.int64, device="cpu", pin_memory=True)
            buf[:n].copy_(sampled, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            self.prev_sampled = sampled
            return PendingTokens(buf, event, need, n)

    # -------------------------------------------------------------- hipGraph decode
    @torch.inference_mode()
    def capture_graphs(self):
        cfg = self.config
        width = self.graph_width = (cfg.max_model_len + self.block_size - 1) // self.block_size
        self.graph_bs = self.decode_buckets()
        pool = None
        for bs in reversed(self.graph_bs):
            g = _DecodeGraph(self, bs, width, pool)
            pool = g.pool
            self.graphs[bs] = g
        torch.cuda.synchronize()


class _DecodeGraph:
    """One captured forward for ``bs`` one-token sequences.  Its inputs live in ONE device buffer laid
    out like the pinned staging buffer, so a step is: fill staging, one H2D copy, one graph launch.
    Padding rows carry slot -1 (no KV store), context length 0 (attention writes zeros)."""

    def __init__(self, runner: ModelRunner, bs: int, width: int, pool):
        self.bs, self.width = bs, width
        dev = runner.device
        self.items_cap = bs + runner.ITEM_SLACK
        sizes = dict(input_ids=bs * 8, positions=bs * 8, slot_mapping=bs * 4, context_lens=bs * 4,
                     block_tables=bs * width * 4, temperatures=bs * 4, item_seq=self.items_cap * 4,
                     item_range=self.items_cap * 4, seq_active=bs * 4, seq_chunk=bs * 4, fed_dst=bs * 4, fed_src=bs * 4)
        self.offs, n = {}, 0
        for name, nbytes in sizes.items():
            self.offs[name] = (n, nbytes)
            n += (nbytes + 15) & ~15
        self.host = torch.zeros(n, dtype=torch.uint8, device="cpu", pin_memory=True)
        self.dev = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.event = torch.cuda.Event()
        hv, dv = self._views(self.host), self._views(self.dev)
        self.h = {k: v.numpy() for k, v in hv.items()}
        self.d = dv
        self.h["slot_mapping"][:] = -1
        self._plan_items()
        self.dev.copy_(self.host)
        set_context(slot_mapping=dv["slot_mapping"], context_lens=dv["context_lens"], block_tables=dv["block_tables"])
        self.splits = runner.decode_splits(bs)                        # batches without a long context
        has_long = runner.native is not None and runner.decode_plan(bs, 1 << 30)[1]
        self.graph_long = self.out_long = None
        if runner.native is not None:
            items = {k: dv[k] for k in ("item_seq", "item_range", "seq_active", "seq_chunk")}
            def fwd(long=False):
                return runner.native.forward(dv["input_ids"], dv["positions"], mode=MODE_DECODE, batch=bs,
                                             slot_mapping=dv["slot_mapping"], context_lens=dv["context_lens"],
                                             block_tables=dv["block_tables"],
                                             decode_splits=runner.MAX_RANGES if long else self.splits,
                                             decode_items=items if long else None)
        else:
            def fwd(long=False):
                return runner.model(dv["input_ids"], dv["positions"])
        fwd()                                                         # warm-up (allocator, lazy state)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool):
            self.out = fwd()
        self.pool = self.graph.pool()
        if has_long:                                                  # second capture: attention over the work list
            fwd(True)
            torch.cuda.synchronize()
            self.graph_long = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_long, self.pool):
                self.out_long = fwd(True)
        self.runner = runner
        reset_context()

    def _plan_items(self):
        """Work list for the context lengths now in the host buffer (padding rows: one empty range each)."""
        h = self.h
        plan = ops.plan_decode_items(h["context_lens"], max_ranges=ModelRunner.MAX_RANGES, capacity=self.items_cap)
        for k in ("item_seq", "item_range", "seq_active", "seq_chunk"):
            h[k][:] = plan[k]

    def _views(self, buf):
        bs, w = self.bs, self.width
        def v(name, dtype, *shape):
            o, nb = self.offs[name]
            return buf[o:o + nb].view(dtype).view(*shape)
        return dict(input_ids=v("input_ids", torch.int64, bs), positions=v("positions", torch.int64, bs),
                    slot_mapping=v("slot_mapping", torch.int32, bs), context_lens=v("context_lens", torch.int32, bs),
                    block_tables=v("block_tables", torch.int32, bs, w), temperatures=v("temperatures", torch.float32, bs),
                    item_seq=v("item_seq", torch.int32, self.items_cap), item_range=v("item_range", torch.int32, self.items_cap),
                    seq_active=v("seq_active", torch.int32, bs), seq_chunk=v("seq_chunk", torch.int32, bs),
                    fed_dst=v("fed_dst", torch.int32, bs), fed_src=v("fed_src", torch.int32, bs))

    def replay(self, fb, seqs, want_temps: bool, prev_sampled=None):
        n = len(seqs)
        if not self.event.query():              # the previous step's copy has left the pinned buffer (almost always: the
            self.event.synchronize()            # caller read that step's ids back since)
        h = self.h
        h["input_ids"][:n] = fb["input_ids"]
        h["positions"][:n] = fb["positions"]
        h["slot_mapping"][:n] = fb["slot_mapping"]
        h["slot_mapping"][n:] = -1
        h["context_lens"][:n] = fb["context_lens"]
        h["context_lens"][n:] = 0
        tb = fb["block_tables"]
        h["block_tables"][:n, :tb.shape[1]] = tb
        if want_temps:
            h["temperatures"][:n] = np.fromiter((s.temperature for s in seqs), dtype=np.float32, count=n)
        long = self.graph_long is not None and self.runner.decode_plan(n, fb["max_seqlen_k"])[1]
        if long:
            self._plan_items()
        fed = fb.get("fed")
        if fed is not None:
            k = fed[0].size
            h["fed_dst"][:k], h["fed_src"][:k] = fed
        self.dev.copy_(self.host, non_blocking=True)
        self.event.record()
        if fed is not None:                     # ids that are still on the GPU: from the previous sampler output
            ops.feed_token_ids(self.d["input_ids"], self.d["fed_dst"][:k], prev_sampled, self.d["fed_src"][:k])
        temps = self.d["temperatures"][:n] if want_temps else None
        if long:
            self.graph_long.replay()
            return self.out_long, temps
        self.graph.replay()
        return self.out, temps
