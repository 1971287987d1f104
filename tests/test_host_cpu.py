"""Host-side logic of the product package, on CPU: batch flattening against the reference's
recorded scheduler traces, tensor-parallel shard loaders against the reference's loaders, the
C-ABI surface (library loads, every declared symbol is exported), Context and Sequence contracts."""
import ctypes
import multiprocessing as mp
import pickle
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_cabi_exports_every_declared_symbol():
    from nanovllm import _hip
    header = (ROOT / "include" / "nanovllm_hip.h").read_text()
    declared = set(re.findall(r"\b(nvh_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    lib = _hip.lib()                                     # raises if the .so is missing
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nvh_version() >= 100
    # pure host helpers may be called without a GPU
    assert lib.nvh_gemm_workspace_bytes(5, 1024, 4) == 4 * 5 * 1024 * 4
    assert lib.nvh_gemm_suggest_split_k(1, 1024, 2048) > 1 and lib.nvh_gemm_suggest_split_k(1024, 4096, 1024) == 1
    assert lib.nvh_attn_decode_workspace_bytes(2, 16, 128, 1) == 0
    assert 1 <= lib.nvh_attn_decode_suggest_splits(1, 8, 40960) <= 64
    assert lib.nvh_attn_decode_suggest_splits(256, 8, 1024) == 1
    # argument validation happens before any launch: errors carry a message, nothing throws
    rc = lib.nvh_gemm_bf16_nt(None, 0, None, None, None, 0, 1, 8, 64, 1, None, None)
    assert rc == -22 and b"null pointer" in lib.nvh_last_error()
    rc = lib.nvh_rope(ctypes.c_void_p(16), ctypes.c_void_p(16), 0, 1, ctypes.c_void_p(16), 0, 1, ctypes.c_void_p(16), 1, 1, 64, None)
    assert rc == -38 and b"head_dim" in lib.nvh_last_error()
    p = ctypes.c_void_p(4096)                            # a well-aligned non-null address; never dereferenced
    # a work list shorter than the batch, or whose range count differs from the workspace's
    rc = lib.nvh_attn_decode_items(p, 4096, p, p, p, p, 8, 4, 16, 8, 128, 0.1, p, p, p, p, p, 2, 4, p, None)
    assert rc == -22 and b"work list" in lib.nvh_last_error()
    rc = lib.nvh_attn_decode_items(p, 4096, p, p, p, p, 8, 4, 16, 8, 128, 0.1, p, p, p, p, p, 6, 4, None, None)
    assert rc == -22 and b"workspace" in lib.nvh_last_error()
    rc = lib.nvh_gemm_qkv_rope_store(p, 1024, p, p, 4096, 8, 1024, 16, 8, 64, p, p, 1e-6, p, p, 4096, None, None, None, 0, None)
    assert rc == -38 and b"head_dim" in lib.nvh_last_error()
    rc = lib.nvh_gemm_qkv_rope_store(p, 1024, p, p, 4096, 8, 1024, 16, 8, 128, p, p, 1e-6, p, p, 4096, None, None, None, 9, None)
    assert rc == -22 and b"raw_rows" in lib.nvh_last_error()
    rc = lib.nvh_gemm_qkv_rope_store(p, 1024, p, p, 4096, 8, 1024, 16, 8, 128, p, p, 1e-6, p, p, 4096, p, None, None, 0, None)
    assert rc == -22 and b"k_cache and v_cache" in lib.nvh_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nanovllm import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", tmp_path / "absent.so")
    with pytest.raises(_hip.HipError, match="no CPU or PyTorch fallback"):
        _hip.lib()


def _seqs_from_step(st):
    from nanovllm.engine.sequence import Sequence
    seqs = []
    o = 0
    for ln, c, n, tb in zip(st["lens"], st["cached"], st["new"], st["tables"]):
        toks = [0] * ln
        toks[c:c + n] = st["input_ids"][o:o + n]
        o += n
        s = Sequence(toks)
        s.num_cached_tokens, s.num_new_tokens, s.block_table = c, n, list(tb)
        seqs.append(s)
    return seqs


def test_flatten_batch_matches_reference_traces(f4):
    """Integer work: exact equality with what the reference's prepare_model_input produced on every
    step of four scheduler traces (chunked prefill, prefix reuse, pre-emption, un-chunked)."""
    from nanovllm.engine.model_runner import flatten_batch
    steps = 0
    for name, trace in f4.items():
        for st in trace:
            fb = flatten_batch(_seqs_from_step(st))
            assert fb["input_ids"].tolist() == st["input_ids"], name
            assert fb["positions"].tolist() == st["positions"]
            assert fb["cu_seqlens_q"].tolist() == st["cu_q"] and fb["cu_seqlens_k"].tolist() == st["cu_k"]
            assert (fb["max_seqlen_q"], fb["max_seqlen_k"]) == (st["max_q"], st["max_k"])
            assert fb["slot_mapping"].tolist() == st["slots"]
            assert fb["context_lens"].tolist() == st["ctx_lens"]
            assert fb["need"].tolist() == st["need"]
            got = None if fb["block_tables"] is None else fb["block_tables"].tolist()
            assert got == st["block_tables"]
            for k in ("input_ids", "positions"):
                assert fb[k].dtype == np.int64
            for k in ("cu_seqlens_q", "cu_seqlens_k", "slot_mapping", "context_lens", "need"):
                assert fb[k].dtype == np.int32
            steps += 1
    assert steps >= 30


def test_flatten_batch_edge_cases():
    from nanovllm.engine.model_runner import flatten_batch
    from nanovllm.engine.sequence import Sequence
    # warm-up shape: no block table -> no slots, no need, no tables
    s = Sequence([0] * 7)
    s.num_new_tokens = 7
    fb = flatten_batch([s])
    assert fb["slot_mapping"].size == 0 and fb["need"].size == 0 and fb["block_tables"] is None
    # chunk that ends exactly on a page edge, then one that starts on it
    a = Sequence(list(range(600)))
    a.num_cached_tokens, a.num_new_tokens, a.block_table = 0, 256, [9]
    fb = flatten_batch([a])
    assert fb["slot_mapping"].tolist() == list(range(9 * 256, 10 * 256)) and fb["need"].size == 0
    a.num_cached_tokens, a.num_new_tokens, a.block_table = 256, 344, [9, 4, 2]
    fb = flatten_batch([a])
    assert fb["slot_mapping"].tolist() == list(range(4 * 256, 5 * 256)) + list(range(2 * 256, 2 * 256 + 88))
    assert fb["need"].tolist() == [0] and fb["block_tables"].tolist() == [[9, 4, 2]]


def test_staging_single_copy_roundtrip():
    from nanovllm.engine.model_runner import _Staging
    st = _Staging(torch.device("cpu"))
    arrays = dict(a=np.arange(5, dtype=np.int64), b=np.arange(7, dtype=np.int32), c=np.zeros(0, np.int32),
                  d=np.arange(6, dtype=np.int32).reshape(2, 3), e=np.array([0.5, 1.5], dtype=np.float32))
    out = st.upload(arrays)
    for k, a in arrays.items():
        assert out[k].shape == a.shape and np.array_equal(out[k].numpy(), a)
        assert out[k].data_ptr() % 16 == 0 or a.size == 0


def test_context_contract():
    from nanovllm.utils.context import get_context, reset_context, set_context
    t = torch.arange(3, dtype=torch.int32)
    set_context(t, t, 2, 3, t, t, None, t)
    c = get_context()
    assert c.cu_seqlens_q is t and c.max_seqlen_q == 2 and c.max_seqlen_k == 3 and c.block_tables is None
    assert c.seq_need_compute_logits is t
    set_context(slot_mapping=t, context_lens=t, block_tables=t)      # the graph-capture form
    c = get_context()
    assert c.cu_seqlens_q is None and c.block_tables is t
    reset_context()
    assert get_context().slot_mapping is None


def test_sequence_pickle_carries_only_the_scheduled_window():
    from nanovllm.engine.sequence import Sequence
    s = Sequence(list(range(5000)))
    s.num_cached_tokens, s.num_new_tokens, s.block_table, s.temperature = 4096, 300, list(range(18, 0, -1)), 0.6
    blob = pickle.dumps(s)
    assert len(blob) < 2000                                          # not O(context)
    r = pickle.loads(blob)
    assert len(r) == 5000 and r[4096:4396] == list(range(4096, 4396))
    assert (r.num_cached_tokens, r.num_new_tokens, r.block_table, r.temperature) == (4096, 300, list(range(18, 0, -1)), 0.6)
    from nanovllm.engine.model_runner import flatten_batch
    assert np.array_equal(flatten_batch([r])["input_ids"], flatten_batch([s])["input_ids"])


@pytest.mark.parametrize("ws", [1, 2, 4, 8])
def test_shard_loaders_match_reference(f1b, ws, monkeypatch):
    """Every rank's packed parameters equal what the reference's weight_loaders produced."""
    import nanovllm.layers.linear as L
    import nanovllm.layers.embed_head as E
    bf = torch.bfloat16
    for rk in range(ws):
        for mod in (L, E):
            monkeypatch.setattr(mod, "tp_rank", lambda rk=rk: rk)
            monkeypatch.setattr(mod, "tp_world", lambda ws=ws: ws)
        qkv = L.QKVParallelLinear(32, 16, 8, 8).to(bf)
        for sid in "qkv":
            qkv.weight.weight_loader(qkv.weight, f1b.bf(sid), sid)
        mc = L.MergedColumnParallelLinear(32, [64, 64]).to(bf)
        mc.weight.weight_loader(mc.weight, f1b.bf("gate"), 0)
        mc.weight.weight_loader(mc.weight, f1b.bf("up"), 1)
        rp = L.RowParallelLinear(64, 32).to(bf)
        rp.weight.weight_loader(rp.weight, f1b.bf("row"))
        ve = E.VocabParallelEmbedding(64, 16).to(bf)
        ve.weight.weight_loader(ve.weight, f1b.bf("emb"))
        t = f"ws{ws}_r{rk}_"
        assert torch.equal(qkv.weight.data, f1b.bf(t + "qkv"))
        assert torch.equal(mc.weight.data, f1b.bf(t + "gate_up"))
        assert torch.equal(rp.weight.data, f1b.bf(t + "row"))
        assert torch.equal(ve.weight.data, f1b.bf(t + "emb"))
        assert (ve.vocab_start_idx, ve.vocab_end_idx) == (rk * 64 // ws, (rk + 1) * 64 // ws)


def test_loader_packed_mapping_and_tied_head():
    from types import SimpleNamespace
    from nanovllm.models.qwen3 import Qwen3ForCausalLM
    from nanovllm.utils.loader import load_tensors, synthetic_tensors
    from conftest import TINY
    hf = SimpleNamespace(**TINY, rms_norm_eps=1e-6, rope_theta=1e6, rope_scaling=None, attention_bias=False,
                         hidden_act="silu", tie_word_embeddings=True)
    model = Qwen3ForCausalLM(hf)
    tensors = dict(synthetic_tensors(hf, seed=3, device="cpu", dtype=torch.float32))
    assert load_tensors(model, tensors.items()) == len(tensors)
    a = model.model.layers[1].self_attn
    assert torch.equal(a.qkv_proj.weight.data[:512], tensors["model.layers.1.self_attn.q_proj.weight"])
    assert torch.equal(a.qkv_proj.weight.data[512:768], tensors["model.layers.1.self_attn.k_proj.weight"])
    assert torch.equal(a.qkv_proj.weight.data[768:], tensors["model.layers.1.self_attn.v_proj.weight"])
    mlp = model.model.layers[0].mlp
    assert torch.equal(mlp.gate_up_proj.weight.data[512:], tensors["model.layers.0.mlp.up_proj.weight"])
    assert model.lm_head.weight.data_ptr() == model.model.embed_tokens.weight.data_ptr()   # tied
    # every rank draws the same full tensors (name-keyed streams)
    again = dict(synthetic_tensors(hf, seed=3, device="cpu", dtype=torch.float32))
    assert all(torch.equal(tensors[k], again[k]) for k in tensors)


def test_loader_refuses_a_checkpoint_that_lacks_one_shard_of_a_packed_parameter():
    """qkv_proj / gate_up_proj are filled from three / two checkpoint tensors each; a checkpoint without one of them
    (k_proj of one layer, up_proj of another) must not pass for complete, nor must one without a whole parameter."""
    from types import SimpleNamespace
    from nanovllm.models.qwen3 import Qwen3ForCausalLM
    from nanovllm.utils.loader import check_all_loaded, load_tensors, synthetic_tensors
    from conftest import TINY
    hf = SimpleNamespace(**TINY, rms_norm_eps=1e-6, rope_theta=1e6, rope_scaling=None, attention_bias=False,
                         hidden_act="silu", tie_word_embeddings=True)
    tensors = dict(synthetic_tensors(hf, seed=3, device="cpu", dtype=torch.float32))
    loaded = set()
    load_tensors(Qwen3ForCausalLM(hf), tensors.items(), loaded)
    check_all_loaded(Qwen3ForCausalLM(hf), loaded)                       # complete: passes
    for gone, word in (("model.layers.1.self_attn.k_proj.weight", "shards k missing"), ("model.layers.0.mlp.up_proj.weight", "shards 1 missing"),
                       ("model.layers.0.mlp.down_proj.weight", "not loaded")):
        model, loaded = Qwen3ForCausalLM(hf), set()
        load_tensors(model, ((k, v) for k, v in tensors.items() if k != gone), loaded)
        with pytest.raises(RuntimeError, match=word):
            check_all_loaded(model, loaded)


# ------------------------------------------------------------------ world_size-2 over gloo
def _tp_worker(rank, ws, port, q):
    """Tensor-parallel layer algebra on CPU: the GEMM / embedding kernels are replaced BY THE TEST with
    the oracle's arithmetic, the sharding, collectives and rank-0 conventions are the product's."""
    import sys
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "nano-vllm-v1_amd"))
    import torch.distributed as dist
    from oracle import ref_ops as R
    from nanovllm import ops
    ops.linear = lambda x, w, b=None: R.linear(x, w, b)
    ops.embedding = lambda ids, table, vocab_start=0: R.embedding(ids, table, vocab_start, ws)
    ops.gather_last_rows = lambda h, cu, need, n: R.last_token_rows(h, cu, need if need is not None else torch.zeros(0, dtype=torch.int32))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=ws, rank=rank)
    from nanovllm.layers.linear import MergedColumnParallelLinear, RowParallelLinear
    from nanovllm.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
    from nanovllm.utils.context import set_context
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    w_up, w_gate, w_down = [(torch.randn(*s, generator=g) * 0.1).to(bf) for s in ((64, 32), (64, 32), (32, 64))]
    emb_w = (torch.randn(128, 32, generator=g)).to(bf)
    x = torch.randn(6, 32, generator=g).to(bf)
    col = MergedColumnParallelLinear(32, [64, 64]).to(bf)
    col.weight.weight_loader(col.weight, w_gate, 0)
    col.weight.weight_loader(col.weight, w_up, 1)
    row = RowParallelLinear(64, 32).to(bf)
    row.weight.weight_loader(row.weight, w_down)
    y = row(R.silu_mul(col(x)))                                  # column -> row pairing, one all-reduce
    emb = VocabParallelEmbedding(128, 32).to(bf)
    emb.weight.weight_loader(emb.weight, emb_w)
    ids = torch.tensor([0, 127, 63, 64, 5, 100])
    e = emb(ids)
    head = ParallelLMHead(128, 32).to(bf)
    head.weight.weight_loader(head.weight, emb_w)
    cu = torch.tensor([0, 2, 6], dtype=torch.int32)
    set_context(cu, cu, 4, 4, None, None, None, torch.tensor([1], dtype=torch.int32))
    logits = head(x)
    q.put((rank, y.float(), e.float(), None if logits is None else logits.float()))
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_parallel_layers_world2_gloo():
    from oracle import ref_ops as R
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sock:                      # a port nobody holds right now (a pid-derived one collided now and then)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, y, e, logits = q.get(timeout=120)
        got[rank] = (y, e, logits)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    w_up, w_gate, w_down = [(torch.randn(*s, generator=g) * 0.1).to(bf) for s in ((64, 32), (64, 32), (32, 64))]
    emb_w = (torch.randn(128, 32, generator=g)).to(bf)
    x = torch.randn(6, 32, generator=g).to(bf)
    ref_y = R.linear(R.silu_mul(R.linear(x, torch.cat([w_gate, w_up]))), w_down).float()
    ids = torch.tensor([0, 127, 63, 64, 5, 100])
    ref_logits = R.linear(x[5:6], emb_w).float()
    for rank in (0, 1):
        y, e, logits = got[rank]
        assert torch.allclose(y, ref_y, atol=2e-2, rtol=2e-2)        # partial sums round separately per rank
        assert torch.equal(e, emb_w[ids].float())                    # exactly one rank contributes each row
    assert got[1][2] is None                                         # only rank 0 receives logits
    assert got[0][2].shape == (1, 128) and torch.allclose(got[0][2], ref_logits, atol=1e-2, rtol=1e-2)


def os_getpid():
    import os
    return os.getpid()


def test_shm_control_plane_roundtrip():
    """pickle-over-shared-memory RPC used between rank 0 and the tensor-parallel workers."""
    import threading
    from multiprocessing.shared_memory import SharedMemory
    from nanovllm.engine.model_runner import ModelRunner
    from nanovllm.engine.sequence import Sequence
    name = f"nvh_test_{os_getpid()}"
    shm = SharedMemory(name=name, create=True, size=2 ** 20)
    try:
        ev = threading.Event()
        r0, r1 = ModelRunner.__new__(ModelRunner), ModelRunner.__new__(ModelRunner)
        r0.world_size = r1.world_size = 2
        r0.rank, r1.rank = 0, 1
        r0.shm = r1.shm = shm
        r0.event, r1.event = [ev], ev
        r0._msg_seq = r1._msg_seq = 0
        s = Sequence(list(range(300)))
        s.num_new_tokens, s.block_table, s.pending_slot = 300, [1, 2], 5
        r0.write_shm("run", [s])
        method, args = r1.read_shm()
        assert method == "run" and not ev.is_set()
        (seqs,) = args
        assert seqs[0][0:300] == list(range(300)) and seqs[0].block_table == [1, 2] and seqs[0].pending_slot == 5
        # the next message waits until the worker has taken the previous one: post it from a thread, take it here
        posted = threading.Thread(target=lambda: r0.write_shm("run_async", [s]))
        posted.start()
        posted.join(5)
        assert not posted.is_alive() and ev.is_set()         # message 1 was acknowledged, so message 2 went out at once
        blocked = threading.Thread(target=lambda: r0.write_shm("exit"))
        blocked.start()
        blocked.join(0.3)
        assert blocked.is_alive()                             # message 2 has not been taken yet
        assert r1.read_shm()[0] == "run_async"
        blocked.join(5)
        assert not blocked.is_alive() and r1.read_shm()[0] == "exit"
        with pytest.raises(RuntimeError, match="exceeds"):
            r0.write_shm("run", [Sequence(list(range(10))) for _ in range(40000)])
    finally:
        shm.close()
        shm.unlink()


# ------------------------------------------------------------------ C host path == numpy host path
def _same_fb(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray) or isinstance(b[k], np.ndarray):
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_c_flatten_batch_equals_numpy_version():
    from nanovllm.engine import model_runner as MR
    from nanovllm.engine.sequence import Sequence
    if MR._hostpath is None:
        pytest.skip("csrc/host/hostpath.c not built")
    rng = np.random.default_rng(0)
    cases = []
    # decode batch, mixed prefill/decode with page crossings, no block tables (warm-up), zero-token sequences
    def mk(n_tok, cached, new, table):
        s = Sequence(rng.integers(0, 10000, n_tok).tolist())
        s.num_cached_tokens, s.num_new_tokens, s.block_table = cached, new, list(table)
        return s
    cases.append([mk(700, 699, 1, [3, 9, 4]), mk(257, 256, 1, [0, 1]), mk(5, 4, 1, [7])])
    cases.append([mk(700, 300, 250, [3, 9, 4]), mk(1000, 0, 512, [5, 6]), mk(30, 29, 1, [2]), mk(600, 512, 88, [10, 11, 12])])
    cases.append([mk(64, 0, 64, []), mk(64, 0, 64, [])])
    cases.append([mk(300, 0, 300, [1, 2]), mk(10, 10, 0, [8])])
    cases.append([])
    for seqs in cases:
        _same_fb(MR._hostpath.flatten_batch(seqs, 256), MR._flatten_batch_py(seqs, 256))
    # the pickled (windowed) form that tensor-parallel workers receive
    import pickle
    seqs = pickle.loads(pickle.dumps(cases[1]))
    _same_fb(MR._hostpath.flatten_batch(seqs, 256), MR._flatten_batch_py(seqs, 256))
    with pytest.raises(IndexError):
        MR._hostpath.flatten_batch([mk(700, 300, 250, [3])], 256)


def test_decode_work_list_covers_every_key_exactly_once():
    """ops.plan_decode_items on random batches: every (sequence, range) appears once, the ranges tile the context
    (64-key multiples, none empty but a padded row's), no sequence exceeds max_ranges, a fixed capacity is kept by
    widening ranges, and one-range sequences are exactly those below two ranges' worth of keys."""
    from nanovllm import ops
    rng = np.random.default_rng(5)
    for trial in range(200):
        b = int(rng.integers(1, 300))
        ctx = rng.integers(0, 400, b)
        for i in rng.integers(0, b, int(rng.integers(0, 5))):
            ctx[i] = int(rng.integers(400, 70000))
        max_ranges = int(rng.choice([1, 4, 32, 256]))
        capacity = 0 if trial % 2 else b + int(rng.integers(0, 130))
        plan = ops.plan_decode_items(ctx, max_ranges=max_ranges, capacity=capacity)
        seq, rg, active, chunk = plan["item_seq"], plan["item_range"], plan["seq_active"], plan["seq_chunk"]
        assert all(a.dtype == np.int32 for a in (seq, rg, active, chunk))
        assert seq.size == (capacity or int(active.sum())) and active.size == chunk.size == b
        used = seq >= 0
        assert int(used.sum()) == int(active.sum()) and not used[int(used.sum()):].any()      # unused slots at the end
        pairs = set(zip(seq[used].tolist(), rg[used].tolist()))
        assert len(pairs) == int(used.sum()) == sum(int(a) for a in active)
        assert pairs == {(s, r) for s in range(b) for r in range(int(active[s]))}
        assert (active >= 1).all() and (active <= max_ranges).all() and (chunk % 64 == 0).all() and (chunk >= 64).all()
        assert (active.astype(np.int64) * chunk >= ctx).all()
        assert ((active.astype(np.int64) - 1) * chunk < np.maximum(ctx, 1)).all()                # no empty last range
        first = chunk[seq[used]] * (active[seq[used]] > 1)
        assert (np.diff(first) <= 0).all()                                                      # longest ranges first
    plan = ops.plan_decode_items([100, 1023, 1024, 5000], max_ranges=32, range_keys=512)
    assert plan["seq_active"].tolist() == [1, 1, 2, 9] and plan["seq_chunk"].tolist() == [128, 1024, 512, 576]

