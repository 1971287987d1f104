"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs /root/reference); elsewhere it exits
with a message.  Nothing from the reference is written out except numeric
inputs/outputs (arrays; bf16 stored as uint16 bit patterns).

How the reference is made to run on a CPU-only box (no source is modified):
  * ``flash_attn`` (third-party CUDA wheel, absent) is replaced in sys.modules by
    a stub whose two entry points call oracle.ref_ops' dense-softmax attention,
    so everything AROUND the attention arithmetic (module wiring, context
    handling, K/V source selection, LM-head row selection) is the reference's;
  * ``store_kvcache`` (a Triton launch) is replaced by the oracle's scatter;
  * a 1-rank gloo process group exists because LinearBase.__init__ reads the
    rank (layers/linear.py:84-85); tensor-parallel shards are produced by
    overriding ``dist.get_rank/get_world_size`` around module construction;
  * TORCHDYNAMO_DISABLE=1: the decorated methods run as written (eager).

Usage:  python -m oracle.gen_golden     (from the repo root)
"""
from __future__ import annotations

import os
import sys
import types
from pathlib import Path
from types import SimpleNamespace

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16 -> uint16 bit pattern; other dtypes -> numpy as is."""
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def main():
    if not REF.exists():
        sys.exit("gen_golden: /root/reference not present - fixtures can only be generated in the build container")
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle import ref_ops as R
    from oracle import ref_model as M

    # ---- stub the absent third-party attention + the Triton launch -------------
    stub = types.ModuleType("flash_attn")

    def varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
               softmax_scale, causal, block_table=None):
        assert causal
        if block_table is None:
            return R.attention_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale)
        return R.attention_varlen(q, None, None, cu_seqlens_q, cu_seqlens_k, softmax_scale, k, v, block_table)

    def with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, causal):
        o = R.attention_decode(q[:, 0], k_cache, v_cache, cache_seqlens, block_table, softmax_scale)
        return o.unsqueeze(1)

    stub.flash_attn_varlen_func = varlen
    stub.flash_attn_with_kvcache = with_kvcache
    sys.modules["flash_attn"] = stub
    sys.path.insert(0, str(REF))

    import torch.distributed as dist
    pg_file = "/tmp/_gen_golden_pg"
    if os.path.exists(pg_file):
        os.remove(pg_file)
    dist.init_process_group("gloo", init_method=f"file://{pg_file}", world_size=1, rank=0)

    import nanovllm.layers.attention as ref_attn
    from nanovllm.layers.layernorm import RMSNorm
    from nanovllm.layers.rotary_embedding import RotaryEmbedding
    from nanovllm.layers.activation import SiluAndMul
    from nanovllm.layers import linear as ref_linear
    from nanovllm.layers.embed_head import VocabParallelEmbedding, ParallelLMHead
    from nanovllm.layers.sampler import Sampler
    from nanovllm.models.qwen3 import Qwen3ForCausalLM
    from nanovllm.utils.context import set_context, reset_context
    from nanovllm.engine.sequence import Sequence
    from nanovllm.engine.scheduler import Scheduler
    from nanovllm.engine.block_manager import BlockManager
    from nanovllm.sampling_params import SamplingParams
    import nanovllm.engine.model_runner as ref_runner

    ref_attn.store_kvcache = R.store_kvcache
    OUT.mkdir(parents=True, exist_ok=True)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(1234)

    def rnd(*s, scale=1.0, dtype=bf):
        return (torch.randn(*s, generator=g) * scale).to(dtype)

    # ======================= F1: per-operator known answers =====================
    f1 = {}
    # K4 RMSNorm on [5,256] and on heads [5,4,128]
    for tag, shp in (("rms_2d", (5, 256)), ("rms_heads", (5, 4, 128))):
        m = RMSNorm(shp[-1], eps=1e-6)
        m.weight.data = rnd(shp[-1]).float().add(1.0).to(bf)
        x = rnd(*shp, scale=3.0)
        with torch.no_grad():
            y = m.to(bf)(x.clone())
        f1[tag + "_x"], f1[tag + "_w"], f1[tag + "_y"] = bits(x), bits(m.weight.data), bits(y)
    # K5 add + RMSNorm
    m = RMSNorm(256, eps=1e-6).to(bf)
    m.weight.data = rnd(256).float().add(1.0).to(bf)
    x, r = rnd(7, 256, scale=2.0), rnd(7, 256, scale=2.0)
    with torch.no_grad():
        y, nr = m(x.clone(), r.clone())
    f1.update(addrms_x=bits(x), addrms_r=bits(r), addrms_w=bits(m.weight.data), addrms_y=bits(y), addrms_res=bits(nr))
    # K6 RoPE, theta 1e6, positions incl. page edges and the last table row
    rope = RotaryEmbedding(128, 128, 40960, 1e6)
    pos = torch.tensor([0, 1, 255, 256, 4095, 40959, 17, 1024], dtype=torch.int64)
    q, k = rnd(8, 4, 128), rnd(8, 2, 128)
    with torch.no_grad():
        qo, ko = rope(pos, q.clone(), k.clone())
    f1.update(rope_pos=pos.numpy(), rope_q=bits(q), rope_k=bits(k), rope_qo=bits(qo), rope_ko=bits(ko),
              rope_table_rows=rope.cos_sin_cache[pos].squeeze(1).numpy())
    # K7 SiLU*mul
    x = rnd(6, 512, scale=2.5)
    with torch.no_grad():
        y = SiluAndMul()(x.clone())
    f1.update(silu_x=bits(x), silu_y=bits(y))
    # G: replicated linear, bias on/off
    for tag, bias in (("lin", False), ("lin_bias", True)):
        lin = ref_linear.ReplicatedLinear(192, 160, bias=bias).to(bf)
        lin.weight.data = rnd(160, 192, scale=0.1)
        if bias:
            lin.bias.data = rnd(160, scale=0.5)
        x = rnd(9, 192)
        with torch.no_grad():
            y = lin(x)
        f1[tag + "_x"], f1[tag + "_w"], f1[tag + "_y"] = bits(x), bits(lin.weight.data), bits(y)
        if bias:
            f1[tag + "_b"] = bits(lin.bias.data)
    # E1 embedding (tp=1) and LM-head row selection with a need subset and the
    # empty-need fall-through
    emb = VocabParallelEmbedding(300, 64).to(bf)
    emb.weight.data = rnd(300, 64)
    ids = torch.tensor([0, 299, 5, 5, 123], dtype=torch.int64)
    with torch.no_grad():
        f1.update(emb_w=bits(emb.weight.data), emb_ids=ids.numpy(), emb_y=bits(emb(ids)))
    head = ParallelLMHead(300, 64).to(bf)
    head.weight.data = emb.weight.data
    hidden = rnd(10, 64)
    cu = torch.tensor([0, 3, 4, 10], dtype=torch.int32)
    for tag, need in (("head_subset", [0, 2]), ("head_empty", [])):
        set_context(cu, cu, 6, 6, None, None, None, torch.tensor(need, dtype=torch.int32))
        with torch.no_grad():
            f1[tag + "_logits"] = bits(head(hidden))
        f1[tag + "_need"] = np.array(need, dtype=np.int32)
        reset_context()
    f1.update(head_hidden=bits(hidden), head_cu=cu.numpy())
    # K8 sampler with the noise made explicit: the module draws
    # empty_like(probs).exponential_(1); re-seeding reproduces its draw on CPU.
    logits = rnd(4, 300, scale=2.0, dtype=torch.float32)
    temps = torch.tensor([0.6, 1.0, 1e-9, 2.0])
    torch.manual_seed(77)
    with torch.no_grad():
        tok = Sampler()(logits.clone(), temps)
    torch.manual_seed(77)
    noise = torch.empty(4, 300).exponential_(1)
    f1.update(samp_logits=logits.numpy(), samp_temps=temps.numpy(), samp_noise=noise.numpy(), samp_tok=tok.numpy())
    np.savez_compressed(OUT / "f1_ops.npz", **f1)

    # ======================= F1b: tensor-parallel shard loaders =================
    f1b = {}
    real_rank, real_ws = dist.get_rank, dist.get_world_size
    full_qkv = rnd((8 + 2 * 8) * 16, 32)          # Hq=8, Hkv=8, D=16 rows: q | k | v
    fq, fk, fv = full_qkv.split([128, 128, 128])
    full_gate, full_up = rnd(64, 32), rnd(64, 32)
    full_row = rnd(32, 64)
    full_emb = rnd(64, 16)
    f1b.update(q=bits(fq), k=bits(fk), v=bits(fv), gate=bits(full_gate), up=bits(full_up), row=bits(full_row), emb=bits(full_emb))
    for ws in (1, 2, 4, 8):
        for rk in range(ws):
            dist.get_rank, dist.get_world_size = (lambda rk=rk: rk), (lambda ws=ws: ws)
            try:
                qkv = ref_linear.QKVParallelLinear(32, 16, 8, 8).to(bf)
                for sid, src in (("q", fq), ("k", fk), ("v", fv)):
                    qkv.weight.weight_loader(qkv.weight, src, sid)
                mc = ref_linear.MergedColumnParallelLinear(32, [64, 64]).to(bf)
                mc.weight.weight_loader(mc.weight, full_gate, 0)
                mc.weight.weight_loader(mc.weight, full_up, 1)
                rp = ref_linear.RowParallelLinear(64, 32).to(bf)
                rp.weight.weight_loader(rp.weight, full_row)
                ve = VocabParallelEmbedding(64, 16).to(bf)
                ve.weight.weight_loader(ve.weight, full_emb)
                ids = torch.tensor([0, 63, 17, 40], dtype=torch.int64)
                # the masked gather BEFORE the all-reduce (embed_head.py:119-128)
                mask = (ids >= ve.vocab_start_idx) & (ids < ve.vocab_end_idx)
                part = mask.unsqueeze(1) * torch.nn.functional.embedding(mask * (ids - ve.vocab_start_idx), ve.weight.data)
            finally:
                dist.get_rank, dist.get_world_size = real_rank, real_ws
            t = f"ws{ws}_r{rk}_"
            f1b[t + "qkv"], f1b[t + "gate_up"] = bits(qkv.weight.data), bits(mc.weight.data)
            f1b[t + "row"], f1b[t + "emb"], f1b[t + "emb_part"] = bits(rp.weight.data), bits(ve.weight.data), bits(part)
    np.savez_compressed(OUT / "f1b_tp_shards.npz", **f1b)

    # ======================= F2/F3: tiny Qwen3 end to end =======================
    shape = M.ModelShape(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                         head_dim=128, intermediate_size=512, vocab_size=1024, max_position_embeddings=4096)
    hf = SimpleNamespace(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                         head_dim=128, intermediate_size=512, vocab_size=1024, max_position_embeddings=4096,
                         rms_norm_eps=1e-6, rope_theta=1e6, rope_scaling=None, attention_bias=False,
                         hidden_act="silu", tie_word_embeddings=True, torch_dtype=bf)
    weights = M.random_weights(shape, seed=0, std=0.05)
    torch.set_default_dtype(bf)
    model = Qwen3ForCausalLM(hf)
    torch.set_default_dtype(torch.float32)
    sd = {k: v for k, v in weights.items() if k != "lm_head.weight"}
    missing = model.load_state_dict(sd, strict=False)
    assert set(missing.missing_keys) <= {"lm_head.weight"}, missing
    model.lm_head.weight.data = model.model.embed_tokens.weight.data
    num_blocks = 12
    kv = torch.zeros(2, 2, num_blocks, 256, 2, 128, dtype=bf)

    def bind(cache):
        li = 0
        for mod in model.modules():
            if hasattr(mod, "k_cache") and hasattr(mod, "v_cache"):
                if cache is None:
                    mod.k_cache = mod.v_cache = torch.tensor([])
                else:
                    mod.k_cache, mod.v_cache = cache[0, li], cache[1, li]
                li += 1

    def ref_step(seqs):
        """One reference forward through ITS prepare_model_input (cuda/pin neutralised)."""
        inp = ref_prepare(seqs)
        from nanovllm.utils.context import get_context
        with torch.no_grad():
            hidden = model(*inp)
            logits = model.compute_logits(hidden)
        ctx = get_context()
        need = ctx.seq_need_compute_logits.clone()
        reset_context()
        return hidden, logits, need

    # neutralise pinned-memory / .cuda() inside the reference's prepare_model_input
    class _TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def tensor(data, dtype=None, pin_memory=False):
            return torch.tensor(data, dtype=dtype)

    ref_runner.torch = _TorchProxy()
    torch.Tensor.cuda = lambda self, *a, **k: self
    fake_self = SimpleNamespace(block_size=256)
    fake_self.prepare_block_tables = lambda seqs: ref_runner.ModelRunner.prepare_block_tables(fake_self, seqs)

    def ref_prepare(seqs):
        return ref_runner.ModelRunner.prepare_model_input(fake_self, seqs)

    tg = torch.Generator().manual_seed(5)

    def prompt(n):
        return torch.randint(0, 1024, (n,), generator=tg).tolist()

    f2 = {"weights_checksum": np.array([float(sum(v.float().abs().sum() for v in weights.values()))])}
    # mode W: ragged batch, no cache, empty need list -> logits for every sequence
    bind(None)
    prompts = [prompt(5), prompt(1), prompt(9)]
    seqs = []
    for p in prompts:
        s = Sequence(p)
        s.num_new_tokens = len(p)
        seqs.append(s)
    hidden, logits, need = ref_step(seqs)
    f2.update(w_prompts=np.array(sum(prompts, []), dtype=np.int64), w_lens=np.array([5, 1, 9]),
              w_hidden=bits(hidden), w_logits=bits(logits))

    # mode P: chunked prefill across a page edge (300 tokens in chunks of 3/256/41),
    # then decode steps; greedy ids and the logits rows of every sampling step
    bind(kv)
    p = prompt(300)
    s = Sequence(p, SamplingParams(temperature=1.0, max_tokens=8))
    blocks = [7, 2, 9]
    got_ids, got_rows, trace = [], [], []
    for chunk in (3, 256, 41):
        s.num_new_tokens = chunk
        ctxlen = s.num_cached_tokens + chunk
        s.block_table = blocks[:(ctxlen + 255) // 256]
        hidden, logits, need = ref_step([s])
        trace.append((s.num_cached_tokens, chunk, int(need.numel())))
        s.num_cached_tokens = ctxlen
    assert need.numel() == 1
    for step in range(6):
        tok = int(logits.float().argmax(-1)[0])
        got_ids.append(tok)
        got_rows.append(logits[0].clone())
        s.append_token(tok)
        s.num_new_tokens = 1
        s.block_table = blocks[:(len(s) + 255) // 256]
        hidden, logits, need = ref_step([s])
        s.num_cached_tokens += 1
    f2.update(p_prompt=np.array(p, dtype=np.int64), p_blocks=np.array(blocks), p_ids=np.array(got_ids),
              p_rows=bits(torch.stack(got_rows)), p_trace=np.array(trace))

    # mixed batch: one sequence decoding (ctx 300+6), one prefix-hit prefill
    # (256 cached via a shared block, 40 new), one fresh prefill of 17 tokens
    s2 = Sequence(p[:256] + prompt(40))
    s2.num_cached_tokens, s2.num_new_tokens, s2.block_table = 256, 40, [7, 4]
    s3 = Sequence(prompt(17))
    s3.num_cached_tokens, s3.num_new_tokens, s3.block_table = 0, 17, [5]
    s.num_new_tokens = 1
    s.append_token(int(logits.float().argmax(-1)[0]))
    s.block_table = blocks[:(len(s) + 255) // 256]
    hidden, logits, need = ref_step([s, s2, s3])
    f2.update(m_tok_s2=np.array(s2.token_ids, dtype=np.int64), m_tok_s3=np.array(s3.token_ids, dtype=np.int64),
              m_last_s=np.array([s.token_ids[-1]]), m_need=need.numpy(), m_logits=bits(logits),
              m_hidden=bits(hidden))
    np.savez_compressed(OUT / "f2_tiny_qwen3.npz", **f2)

    # ======================= F4: scheduler -> runner-input traces ===============
    f4 = {}

    def run_trace(tag, cfg, arrivals, steps):
        """arrivals: {step: [(prompt, max_tokens)]}.  Token ids are produced by a
        deterministic stand-in for the sampler: (sum of last 3 ids + step) % 1000."""
        sched = Scheduler(SimpleNamespace(**cfg))
        rec = []
        for st in range(steps):
            for pr, mt in arrivals.get(st, []):
                sched.add(Sequence(pr, SamplingParams(temperature=1.0, max_tokens=mt, ignore_eos=True)))
            if sched.is_finished():
                continue
            seqs = sched.schedule()
            from nanovllm.utils.context import get_context
            ids, pos = ref_prepare(seqs)
            c = get_context()
            step_rec = dict(
                seq_ids=[x.seq_id for x in seqs], cached=[x.num_cached_tokens for x in seqs],
                new=[x.num_new_tokens for x in seqs], lens=[len(x) for x in seqs],
                tables=[list(x.block_table) for x in seqs],
                input_ids=ids.tolist(), positions=pos.tolist(), cu_q=c.cu_seqlens_q.tolist(),
                cu_k=c.cu_seqlens_k.tolist(), max_q=c.max_seqlen_q, max_k=c.max_seqlen_k,
                slots=c.slot_mapping.tolist(), ctx_lens=c.context_lens.tolist(),
                need=c.seq_need_compute_logits.tolist(),
                block_tables=None if c.block_tables is None else c.block_tables.tolist())
            need = c.seq_need_compute_logits.tolist()
            reset_context()
            toks = [(sum(seqs[i].token_ids[-3:]) + st) % 1000 for i in need]
            sched.postprocess(seqs, toks, need)
            step_rec["sampled"] = toks
            rec.append(step_rec)
        return rec

    import json
    base = dict(chunked_prefill=True, max_model_len=4096, max_num_seqs=8, max_num_batched_tokens=512,
                eos=-1, num_kvcache_blocks=16, kvcache_block_size=256)
    pg = torch.Generator().manual_seed(9)

    def pr(n):
        return torch.randint(0, 10000, (n,), generator=pg).tolist()

    shared = pr(600)
    Sequence.counter = __import__("itertools").count()
    traces = {
        # chunked prefill of a 700-token prompt next to short ones, then decode
        "chunked": run_trace("chunked", base, {0: [(pr(700), 5), (pr(30), 4)], 2: [(pr(10), 3)]}, 14),
        # prefix reuse: the second arrival shares 2 full blocks with the first
        "prefix": run_trace("prefix", base, {0: [(shared + pr(20), 3)], 1: [(shared[:512] + pr(33), 3)],
                                              6: [(shared[:512] + pr(5), 2)]}, 14),
        # few blocks -> LIFO pre-emption while decoding across a page edge
        "preempt": run_trace("preempt", dict(base, num_kvcache_blocks=4, max_num_batched_tokens=1024),
                             {0: [(pr(250), 12), (pr(252), 12), (pr(255), 12)]}, 60),
        # no chunking: admission stops at the token budget
        "unchunked": run_trace("unchunked", dict(base, chunked_prefill=False),
                               {0: [(pr(300), 2), (pr(300), 2), (pr(100), 2)]}, 10),
    }
    (OUT / "f4_sched_traces.json").write_text(json.dumps(traces, separators=(",", ":")))

    # ======================= F5: prefix-hash chain ==============================
    blk0, blk1 = list(range(256)), list(range(1000, 1256))
    h0 = BlockManager.compute_hash(blk0)
    h1 = BlockManager.compute_hash(blk1, h0)
    (OUT / "f5_hashes.json").write_text(json.dumps({"h0": h0, "h1": h1, "h_short": BlockManager.compute_hash([1, 2, 3])}))

    dist.destroy_process_group()
    for f in sorted(OUT.iterdir()):
        print(f"{f.name:28s} {f.stat().st_size:9d} B")


if __name__ == "__main__":
    main()
