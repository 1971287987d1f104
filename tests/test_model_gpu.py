"""End-to-end parity of the HIP runner (tiny Qwen3, random weights) against
  * golden logits / greedy ids produced by the REFERENCE model wiring (tests/golden/f2), and
  * the CPU oracle on the same inputs,
through ModelRunner.forward_logits (everything of a step except the random sampler).

Tolerance: logits |err| <= 3e-2 + 2e-2*|ref| (bf16 activations through 2 layers); greedy token
ids bit-exact.
"""
import pytest
import torch

from conftest import ROOT, TINY
from oracle import ref_model as M
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
TINY_DIR = str(ROOT / "nano-vllm-v1_amd" / "model_configs" / "qwen3-tiny")


def hf_named(weights, shape):
    """Oracle weights (packed names) -> Hugging Face checkpoint names, so the per-parameter shard
    loaders (q/k/v, gate/up) are what fills the model."""
    for name, t in weights.items():
        if name == "lm_head.weight":
            continue
        if "qkv_proj" in name:
            q, k, v = t.split([shape.q_size, shape.kv_size, shape.kv_size])
            for tag, part in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                yield name.replace("qkv_proj", tag), part
        elif "gate_up_proj" in name:
            g, u = t.chunk(2)
            yield name.replace("gate_up_proj", "gate_proj"), g
            yield name.replace("gate_up_proj", "up_proj"), u
        else:
            yield name, t


def make_runner(enforce_eager, num_blocks=12, native=True, **kw):
    import os
    os.environ["NANOVLLM_NATIVE"] = "1" if native else "0"
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    cfg = Config(TINY_DIR, max_num_batched_tokens=1024, max_num_seqs=8, max_model_len=2048,
                 enforce_eager=enforce_eager, num_kvcache_blocks=num_blocks, chunked_prefill=True, **kw)
    runner = ModelRunner(cfg, 0, [])
    shape = M.ModelShape(**TINY)
    weights = M.random_weights(shape, seed=0, std=0.05)
    n = runner.load_weights(hf_named(weights, shape))
    assert n == 1 + 2 * 11 + 1
    assert (runner.native is not None) == native
    return runner, M.RefRunner(shape, weights, num_blocks)


@pytest.fixture(scope="module")
def eager():
    r = make_runner(True)
    yield r
    r[0].exit()


@pytest.fixture(scope="module")
def modules():
    """The operator-module path (what tensor-parallel runs use), eager."""
    r = make_runner(True, native=False)
    yield r
    r[0].exit()


@pytest.fixture(scope="module")
def graphed():
    r = make_runner(False)
    yield r
    r[0].exit()


def seq(tokens, cached=0, new=None, table=()):
    from nanovllm.engine.sequence import Sequence
    s = Sequence(list(tokens))
    s.num_cached_tokens, s.num_new_tokens, s.block_table = cached, len(tokens) - cached if new is None else new, list(table)
    return s


def close(got, ref, atol=3e-2, rtol=2e-2):
    got, ref = got.float().cpu(), ref.float()
    err = (got - ref).abs()
    assert bool((err <= atol + rtol * ref.abs()).all()), f"max err {float(err.max()):.4g}"


def test_mode_w_ragged_batch_matches_reference(eager, f2):
    runner, oracle = eager
    lens, toks = f2.z["w_lens"].tolist(), f2.z["w_prompts"].tolist()
    seqs, o = [], 0
    for n in lens:
        seqs.append(seq(toks[o:o + n]))
        o += n
    logits, _, need = runner.forward_logits(seqs)
    assert need.numel() == 0 and logits.shape == (3, 1024)      # no block table: every sequence gets logits
    close(logits, f2.bf("w_logits"))
    assert torch.equal(logits.float().argmax(-1).cpu(), f2.bf("w_logits").float().argmax(-1))


@pytest.mark.parametrize("which", ["eager", "graphed", "modules"])
def test_chunked_paged_greedy_ids_bit_exact(which, request, f2):
    runner, oracle = request.getfixturevalue(which)
    runner.kv_cache.zero_()
    prompt, blocks = f2.z["p_prompt"].tolist(), f2.z["p_blocks"].tolist()
    s = seq(prompt, 0, 0)
    logits = None
    for chunk, n_need in zip((3, 256, 41), (0, 0, 1)):
        s.num_new_tokens = chunk
        ctx = s.num_cached_tokens + chunk
        s.block_table = blocks[:(ctx + 255) // 256]
        logits, _, need = runner.forward_logits([s])
        assert need.numel() == n_need and (logits is None) == (n_need == 0)   # lone chunk: no logits, no crash
        s.num_cached_tokens = ctx
    ids, rows = [], []
    for _ in range(6):
        tok = int(logits.float().argmax(-1)[0])
        ids.append(tok)
        rows.append(logits[0].cpu())
        s.append_token(tok)
        s.num_new_tokens = 1
        s.block_table = blocks[:(len(s) + 255) // 256]
        logits, _, need = runner.forward_logits([s])
        s.num_cached_tokens += 1
    assert ids == f2.z["p_ids"].tolist()
    close(torch.stack(rows), f2.bf("p_rows"))


def test_mixed_batch_matches_reference_and_oracle(eager, f2):
    """decode row + prefix-hit prefill + fresh prefill in one step (block tables shared)."""
    runner, oracle = eager
    runner.kv_cache.zero_()
    oracle.kv.zero_()
    prompt, blocks = f2.z["p_prompt"].tolist(), f2.z["p_blocks"].tolist()
    ids = f2.z["p_ids"].tolist()
    full = prompt + ids + [int(f2.z["m_last_s"][0])]
    # rebuild the cache state: prefill everything but the last token in one go, on both sides
    pre = full[:-1]
    for r in (runner, oracle):
        s0 = seq(pre, 0, len(pre), blocks[:(len(pre) + 255) // 256]) if r is runner else M.SeqState(list(pre), 0, len(pre), blocks[:(len(pre) + 255) // 256])
        (r.forward_logits([s0]) if r is runner else r.run([s0]))
    t2, t3 = f2.z["m_tok_s2"].tolist(), f2.z["m_tok_s3"].tolist()
    tb = blocks[:(len(full) + 255) // 256]
    batch = [seq(full, len(full) - 1, 1, tb), seq(t2, 256, 40, [7, 4]), seq(t3, 0, 17, [5])]
    logits, _, need = runner.forward_logits(batch)
    assert need.tolist() == f2.z["m_need"].tolist() == [0, 1, 2]
    close(logits, f2.bf("m_logits"))
    ob = [M.SeqState(list(full), len(full) - 1, 1, tb), M.SeqState(list(t2), 256, 40, [7, 4]), M.SeqState(list(t3), 0, 17, [5])]
    ol, _ = oracle.run(ob)
    close(logits, ol)
    assert torch.equal(logits.float().argmax(-1).cpu(), ol.float().argmax(-1))


def test_graph_replay_equals_eager_decode_batch(eager, graphed):
    """Three sequences decoding together (graph bucket 4, one padded row) vs the eager path."""
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in (5, 300, 257)]
    tables = [[0], [1, 2], [3, 4]]
    outs = []
    for runner, _ in (eager, graphed):
        runner.kv_cache.zero_()
        seqs = [seq(p, 0, len(p), t) for p, t in zip(prompts, tables)]
        logits, _, need = runner.forward_logits(seqs)
        assert need.tolist() == [0, 1, 2]
        steps = []
        for _ in range(4):
            toks = logits.float().argmax(-1).tolist()
            for s, t in zip(seqs, toks):
                s.num_cached_tokens += s.num_new_tokens
                s.append_token(t)
                s.num_new_tokens = 1
                s.block_table = tables[seqs.index(s)][:(len(s) + 255) // 256]
            logits, _, _ = runner.forward_logits(seqs)
            steps.append(logits.cpu())
        outs.append(torch.stack(steps))
    assert graphed[0].graphs and not eager[0].graphs
    assert torch.equal(outs[0], outs[1])                 # same kernels, same order: bit-identical


def test_run_samples_and_reports_need(eager):
    runner, _ = eager
    runner.kv_cache.zero_()
    s = seq(list(range(10)), 0, 10, [6])
    s.temperature = 1e-9
    toks, need = runner.run([s])
    assert need.tolist() == [0] and len(toks) == 1 and 0 <= toks[0] < 1024
    logits, _, _ = runner.forward_logits([s])
    assert toks[0] == int(logits.float().argmax(-1)[0])  # near-zero temperature == greedy


def test_native_executor_equals_operator_modules(eager, modules):
    """One native call per step vs the nn.Module path: same kernels in the same order -> identical bits,
    for a packed prefill, a paged mixed batch and eager decode."""
    g = torch.Generator().manual_seed(21)
    p1, p2 = (torch.randint(0, 1024, (n,), generator=g).tolist() for n in (300, 40))
    outs = []
    for runner, _ in (eager, modules):
        runner.kv_cache.zero_()
        got = []
        a = seq(p1, 0, 300, [1, 2])
        l, _, _ = runner.forward_logits([a])                       # packed (no cached keys)
        got.append(l.cpu())
        a.num_cached_tokens, a.num_new_tokens = 300, 1
        a.append_token(int(l.float().argmax(-1)[0]))
        b = seq(p2, 0, 40, [3])
        l, _, need = runner.forward_logits([a, b])                 # paged mixed: decode row + prefill
        assert need.tolist() == [0, 1]
        got.append(l.cpu())
        a.num_cached_tokens, b.num_cached_tokens = 301, 40
        for s_, t_ in zip((a, b), l.float().argmax(-1).tolist()):
            s_.append_token(t_)
            s_.num_new_tokens = 1
        l, _, _ = runner.forward_logits([a, b])                    # eager decode
        got.append(l.cpu())
        outs.append(got)
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("batch", [1, 3, 8])
def test_decode_with_the_norm_inside_the_projection_equals_separate_launches(eager, switches, batch):
    """Decode batches fold the residual add + RMSNorm into the prologue of the projection that consumes it (two launches
    less per layer, the residual alternating between two buffers).  With the previous projection's K-slabs kept
    (NVH_DECODE_SLABS) the step must give the bits of the separate launches (NVH_NO_NORM_GEMM); the default route (no
    slabs at all) must agree with the oracle like every other step."""
    runner, oracle = eager
    g = torch.Generator().manual_seed(40 + batch)
    prompts = [torch.randint(0, 1024, (20 + 7 * i,), generator=g).tolist() for i in range(batch)]
    forced = torch.randint(0, 1024, (3, batch), generator=g).tolist()        # the same next tokens on every route
    outs = {}
    for mode in ("default", "slabs", "separate"):
        switches("NVH_DECODE_SLABS", 1 if mode == "slabs" else -1)
        switches("NVH_NO_NORM_GEMM", 1 if mode == "separate" else -1)
        runner.kv_cache.zero_()
        seqs = [seq(p, 0, len(p), [i]) for i, p in enumerate(prompts)]
        logits, _, _ = runner.forward_logits(seqs)
        steps = []
        for k in range(3):
            for s, t in zip(seqs, forced[k]):
                s.num_cached_tokens += s.num_new_tokens
                s.append_token(t)
                s.num_new_tokens = 1
            logits, _, _ = runner.forward_logits(seqs)
            steps.append(logits.cpu())
        outs[mode] = torch.stack(steps)
    assert torch.equal(outs["slabs"], outs["separate"])
    oracle.kv.zero_()
    ob = [M.SeqState(list(p), 0, len(p), [i]) for i, p in enumerate(prompts)]
    ol, _ = oracle.run(ob)
    for k in range(3):
        for s, t in zip(ob, forced[k]):
            s.num_cached_tokens += s.num_new_tokens
            s.token_ids.append(t)
            s.num_new_tokens = 1
        ol, _ = oracle.run(ob)
        close(outs["default"][k], ol)
        close(outs["separate"][k], ol)


def test_mixed_step_decode_prefix_matches_oracle_and_unsplit_path():
    """A serving-shaped step: ten running sequences with one token each (contexts that sit on, before and after page
    boundaries), then a continuing chunk and a fresh prompt.  The native executor sends the one-token prefix through
    the decode kernel and the rest through the var-len kernel; logits must match the oracle, and the other route
    (everything through the var-len kernel) within the same tolerance, with the same K/V pages."""
    import os
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    shape = M.ModelShape(**TINY)
    weights = M.random_weights(shape, seed=0, std=0.05)
    g = torch.Generator().manual_seed(77)
    ctxs = [2, 17, 255, 256, 257, 300, 511, 512, 513, 40]              # context after the new token
    decode = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in ctxs]
    chunk = torch.randint(0, 1024, (400,), generator=g).tolist()         # 256 cached, 100 scheduled now
    fresh = torch.randint(0, 1024, (70,), generator=g).tolist()
    tables, page = [], 0
    for n in ctxs + [356, 70]:
        nb = (n + 255) // 256
        tables.append(list(range(page, page + nb)))
        page += nb
    results = {}
    for split in ("1", "0"):
        os.environ["NANOVLLM_NATIVE"] = "1"
        os.environ["NANOVLLM_SPLIT_MIXED"] = split
        cfg = Config(TINY_DIR, max_num_batched_tokens=1024, max_num_seqs=16, max_model_len=2048, enforce_eager=True,
                     num_kvcache_blocks=page + 1, chunked_prefill=True)
        runner = ModelRunner(cfg, 0, [])
        try:
            runner.load_weights(hf_named(weights, shape))
            runner.kv_cache.zero_()
            oracle = M.RefRunner(shape, weights, page + 1)
            # fill the caches: every decode sequence but its last token, and the first 256 tokens of the chunked prompt
            warm = [(d[:-1], t) for d, t in zip(decode, tables) if len(d) > 1] + [(chunk[:256], tables[10])]
            for toks, tb in warm:
                tb = tb[:(len(toks) + 255) // 256]                         # the pages those tokens need, no more
                runner.forward_logits([seq(toks, 0, len(toks), tb)])
                oracle.run([M.SeqState(list(toks), 0, len(toks), list(tb))])
            batch = [seq(d, len(d) - 1, 1, t) for d, t in zip(decode, tables)]
            batch += [seq(chunk, 256, 100, tables[10]), seq(fresh, 0, 70, tables[11])]
            fb_prefix = []
            orig = runner.native.forward
            runner.native.forward = lambda *a, **k: (fb_prefix.append(k.get("decode_prefix", 0)), orig(*a, **k))[1]
            logits, _, need = runner.forward_logits(batch)
            assert fb_prefix == [10 if split == "1" else 0]
            assert need.tolist() == list(range(10)) + [11]                 # the chunk does not reach its prompt's end
            ob = [M.SeqState(list(d), len(d) - 1, 1, list(t)) for d, t in zip(decode, tables)]
            ob += [M.SeqState(list(chunk), 256, 100, list(tables[10])), M.SeqState(list(fresh), 0, 70, list(tables[11]))]
            ol, _ = oracle.run(ob)
            close(logits, ol)
            got = logits.float().cpu()
            top = got.max(-1).values
            picked = got.gather(1, ol.float().argmax(-1, keepdim=True)).squeeze(1)
            assert bool((picked >= top - (3e-2 + 2e-2 * top.abs())).all())  # the oracle's greedy id wins, up to bf16 ties
            results[split] = (logits.cpu(), runner.kv_cache.cpu().clone())
        finally:
            runner.exit()
            os.environ.pop("NANOVLLM_SPLIT_MIXED", None)
    close(results["1"][0], results["0"][0])
    kv1, kv0 = results["1"][1], results["0"][1]                            # [2, layers, pages, 256, kv heads, 128]
    assert torch.equal(kv1[:, 0], kv0[:, 0])                               # first layer: the same K/V rows, bit for bit
    close(kv1, kv0)                                                        # deeper layers inherit the attention outputs' last bits


def test_wide_hidden_decode_uses_split_qkv_and_matches_oracle(tmp_path):
    """hidden_size 4096: at a few rows the native executor splits the QKV projection over K and reduces the slabs
    (and sums 3+ slabs of the O / down projections inside the norms).  Prefill then three decode steps, two
    sequences, against the CPU oracle."""
    import json
    import os
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    wide = dict(TINY, hidden_size=4096, num_hidden_layers=1, num_attention_heads=8, num_key_value_heads=2)
    cfg_json = json.load(open(os.path.join(TINY_DIR, "config.json")))
    cfg_json.update(wide)
    d = tmp_path / "wide"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg_json))
    shape = M.ModelShape(**wide)
    weights = M.random_weights(shape, seed=3, std=0.02)
    os.environ["NANOVLLM_NATIVE"] = "1"
    runner = ModelRunner(Config(str(d), max_num_batched_tokens=512, max_num_seqs=8, max_model_len=1024, enforce_eager=True,
                                num_kvcache_blocks=4, chunked_prefill=True), 0, [])
    try:
        runner.load_weights(hf_named(weights, shape))
        oracle = M.RefRunner(shape, weights, 4)
        g = torch.Generator().manual_seed(5)
        prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in (9, 40)]
        seqs = [seq(p, 0, len(p), [i]) for i, p in enumerate(prompts)]
        states = [M.SeqState(list(p), 0, len(p), [i]) for i, p in enumerate(prompts)]
        for step in range(4):
            logits, _, need = runner.forward_logits(seqs)
            ref, _ = oracle.run(states)
            close(logits, ref)
            toks = ref.float().argmax(-1).tolist()                          # both sides continue with the oracle's ids
            for s, st, t in zip(seqs, states, toks):
                s.num_cached_tokens += s.num_new_tokens
                s.append_token(t)
                s.num_new_tokens = 1
                st.token_ids.append(t)
                st.num_cached_tokens, st.num_new_tokens = s.num_cached_tokens, 1
    finally:
        runner.exit()


def test_uneven_contexts_decode_takes_the_work_list(monkeypatch):
    """Sequences of very different context lengths decoding together.  With the batch treated as a large one (one key
    range per sequence) and a low long-context threshold, the runner plans a work list for the decode attention:
    in an eager step, in the captured graph (fixed-capacity list) and for the one-token prefix of a mixed step.
    Logits against the CPU oracle; eager and graph bit-identical."""
    import os
    from nanovllm.config import Config
    from nanovllm.engine.model_runner import ModelRunner
    monkeypatch.setattr(ModelRunner, "decode_splits", lambda self, bs: 1)
    monkeypatch.setattr(ModelRunner, "LONG_CONTEXT", 512)
    monkeypatch.setattr(ModelRunner, "MAX_RANGES", 4)
    shape = M.ModelShape(**TINY)
    weights = M.random_weights(shape, seed=0, std=0.05)
    g = torch.Generator().manual_seed(123)
    ctxs = [1500, 40, 600, 257, 1]
    prompts = [torch.randint(0, 1024, (n,), generator=g).tolist() for n in ctxs]
    fresh = torch.randint(0, 1024, (90,), generator=g).tolist()
    tables, page = [], 0
    for n in ctxs + [90]:
        nb = (n + 1 + 255) // 256
        tables.append(list(range(page, page + nb)))
        page += nb
    os.environ["NANOVLLM_NATIVE"] = "1"
    outs = {}
    for which in ("eager", "graph"):
        cfg = Config(TINY_DIR, max_num_batched_tokens=2048, max_num_seqs=8, max_model_len=2048, enforce_eager=which == "eager",
                     num_kvcache_blocks=page + 1, chunked_prefill=True)
        runner = ModelRunner(cfg, 0, [])
        try:
            runner.load_weights(hf_named(weights, shape))
            runner.kv_cache.zero_()
            oracle = M.RefRunner(shape, weights, page + 1)
            seen = []
            orig = runner.native.forward
            runner.native.forward = lambda *a, **k: (seen.append((k.get("decode_items") is not None, k.get("decode_prefix", 0))),
                                                      orig(*a, **k))[1]
            seqs = [seq(p, 0, len(p), t[:(len(p) + 255) // 256]) for p, t in zip(prompts, tables)]
            states = [M.SeqState(list(p), 0, len(p), list(s.block_table)) for p, s in zip(prompts, seqs)]
            got = []
            for step in range(3):
                logits, _, need = runner.forward_logits(seqs)
                ref, _ = oracle.run(states)
                close(logits, ref, atol=3e-2 if step else 5e-2)             # step 0 (packed prefill, tested elsewhere) fills the caches
                got.append(logits.cpu())
                toks = ref.float().argmax(-1).tolist()                      # both sides continue with the oracle's ids
                for s, st, t, tb in zip(seqs, states, toks, tables):
                    s.num_cached_tokens += s.num_new_tokens
                    s.append_token(t)
                    s.num_new_tokens = 1
                    s.block_table = tb[:(len(s) + 255) // 256]
                    st.token_ids.append(t)
                    st.num_cached_tokens, st.num_new_tokens, st.block_table = s.num_cached_tokens, 1, list(s.block_table)
            if which == "eager":
                assert seen == [(False, 0), (True, 0), (True, 0)]           # prefill, then two decode steps over a work list
            else:
                assert seen == [(False, 0)] and runner.graphs[8].graph_long is not None
            # mixed step: the five running sequences (one token each) and a fresh prompt
            seen.clear()
            batch = seqs + [seq(fresh, 0, 90, tables[5])]
            logits, _, need = runner.forward_logits(batch)
            assert seen == [(True, 5)]
            ref, _ = oracle.run(states + [M.SeqState(list(fresh), 0, 90, list(tables[5]))])
            close(logits, ref)
            got.append(logits.cpu())
            outs[which] = got
        finally:
            runner.exit()
    for a, b in zip(outs["eager"], outs["graph"]):
        assert torch.equal(a, b)


def test_graph_is_skipped_when_a_one_token_chunk_needs_no_logits(graphed):
    """Chunked prefill left with a budget of exactly one token (max_num_batched_tokens = running sequences + 1): every
    sequence of the step brings one token, but the chunk does not finish its prompt and must get no logits row.  The
    captured decode graph computes a row for every sequence, so such a step has to run eagerly; logits rows and the need
    list must line up (Scheduler.postprocess zips them) and match the oracle."""
    runner, oracle = graphed
    g = torch.Generator().manual_seed(123)
    a, b = (torch.randint(0, 1024, (n,), generator=g).tolist() for n in (30, 41))
    long = torch.randint(0, 1024, (20,), generator=g).tolist()
    for toks, tb in ((a[:-1], [0]), (b[:-1], [1]), (long[:5], [2])):
        runner.forward_logits([seq(toks, 0, len(toks), tb)])
        oracle.run([M.SeqState(list(toks), 0, len(toks), list(tb))])
    batch = [seq(a, 29, 1, [0]), seq(b, 40, 1, [1]), seq(long, 5, 1, [2])]
    logits, _, need = runner.forward_logits(batch)
    assert need.tolist() == [0, 1] and logits.shape[0] == 2
    ref, _ = oracle.run([M.SeqState(list(a), 29, 1, [0]), M.SeqState(list(b), 40, 1, [1]), M.SeqState(list(long), 5, 1, [2])])
    close(logits, ref)
    assert torch.equal(logits.float().cpu().argmax(-1), ref.float().argmax(-1))
    toks, need2 = runner.run(batch)
    assert len(toks) == need2.numel() == 2



def test_steps_in_flight_on_several_streams_give_the_bits_of_one_stream(eager):
    """Consecutive prefill-sized steps that need no ids from each other rotate over the caller's stream and the side
    streams, ordered per layer (the later step's QKV projection + KV store of a layer behind the earlier step's attention
    of that layer).  Same kernels in the same order per step, so the logits and the whole KV cache must come out bit for
    bit as with every step on one stream: three chunks of one prompt (each reads the keys of the one before), then two
    unrelated prompts, then a prompt that borrows the pages of the first one's prefix."""
    runner, oracle = eager
    g = torch.Generator().manual_seed(123)
    rnd = lambda n: torch.randint(0, 1024, (n,), generator=g).tolist()       # noqa: E731
    a, b, c = rnd(900), rnd(400), rnd(300)
    d = a[:512] + rnd(280)                                                    # shares two pages with `a`
    lanes = []
    orig = runner.native.forward
    runner.native.forward = lambda *x, **k: (lanes.append((k.get("layer_events"), torch.cuda.current_stream() != torch.cuda.default_stream())),
                                             orig(*x, **k))[1]
    outs = {}
    try:
        for overlap in (False, True):
            runner.step_overlap = overlap
            runner._last_step = None
            torch.cuda.synchronize()
            runner._lane_done.clear()
            runner.kv_cache.zero_()
            lanes.clear()
            got = []
            for cached, new in ((0, 300), (300, 300)):                        # chunk steps: no ids, nothing waits for them
                toks, need = runner.run([seq(a, cached, new, [0, 1, 2, 3][:(cached + new + 255) // 256])])   # the pages scheduled so far
                assert toks == [] and need.numel() == 0
            l, _, _ = runner.forward_logits([seq(a, 600, 300, [0, 1, 2, 3])])
            got.append(l.cpu())
            toks, need = runner.run([seq(b[:-1] + [0] * 50, 0, 399, [4, 5])])  # (does not reach the end of its tokens: no ids)
            assert toks == []
            l, _, _ = runner.forward_logits([seq(c, 0, 300, [6, 7]), seq(d, 512, 280, [0, 1, 8, 9])])
            got.append(l.cpu())
            torch.cuda.synchronize()
            outs[overlap] = (got, runner.kv_cache.cpu().clone(), list(lanes))
    finally:
        runner.native.forward = orig
        runner.step_overlap = True
    assert all(ev == (0, 0) and not side for ev, side in outs[False][2])
    assert [ev for ev, _ in outs[True][2]] == [(0, 1), (1, 2), (2, 1), (1, 2), (2, 1)]
    lanes_n = runner.LANES
    assert [side for _, side in outs[True][2]] == [(i % lanes_n) != 0 for i in range(5)]     # the caller's stream, the side streams, round again
    for x, y in zip(outs[True][0], outs[False][0]):
        assert torch.equal(x, y)
    assert torch.equal(outs[True][1], outs[False][1])
    oracle.kv.zero_()
    for cached, new in ((0, 300), (300, 300)):
        oracle.run([M.SeqState(list(a), cached, new, [0, 1, 2, 3][:(cached + new + 255) // 256])])
    ol, _ = oracle.run([M.SeqState(list(a), 600, 300, [0, 1, 2, 3])])
    close(outs[True][0][0], ol)
