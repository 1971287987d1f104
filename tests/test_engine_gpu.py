"""LLM front-end end to end on the GPU: scheduler + page manager + runner + sampler through `LLM.generate`,
checked against the CPU oracle by teacher forcing (every generated id must be the oracle's argmax at its
position, up to the logit tolerance of the kernel path - exact ties in bf16 may resolve either way)."""
import pytest
import torch

from conftest import TINY
from test_model_gpu import TINY_DIR, hf_named

from oracle import ref_model as M
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    shape = M.ModelShape(**TINY)
    return shape, M.random_weights(shape, seed=0, std=0.05)


def make_llm(weights, **kw):
    from nanovllm import LLM
    shape, w = weights
    cfg = dict(max_num_batched_tokens=512, max_num_seqs=8, max_model_len=2048, num_kvcache_blocks=24,
               chunked_prefill=True, enforce_eager=False)
    cfg.update(kw)
    llm = LLM(TINY_DIR, **cfg)
    llm.model_runner.load_weights(hf_named(w, shape))
    return llm


def check_teacher_forced(weights, prompt, completion, atol=3e-2, rtol=2e-2):
    """Oracle forward over prompt + completion in one packed pass; returns the number of exact argmax matches."""
    shape, w = weights
    oracle = M.RefRunner(shape, w, 0)
    toks = list(prompt) + list(completion[:-1])
    s = M.SeqState(toks, 0, len(toks), [])
    inp = M.prepare_model_input([s])
    hidden = oracle.forward(inp)
    logits = R.linear(hidden[len(prompt) - 1:], w["lm_head.weight"]).float()
    assert logits.shape[0] == len(completion)
    exact = 0
    for row, tok in zip(logits, completion):
        top = float(row.max())
        assert float(row[tok]) >= top - (atol + rtol * abs(top)), f"token {tok}: logit {float(row[tok]):.4f} vs max {top:.4f}"
        exact += int(int(row.argmax()) == tok)
    return exact


def prompts_of(lengths, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 1024, (n,), generator=g).tolist() for n in lengths]


BOTH_LOOPS = pytest.mark.parametrize("ahead", [True, False], ids=["one_step_ahead", "synchronous"])   # Config.async_scheduling


@BOTH_LOOPS
def test_generate_mixed_lengths_chunked(weights, ahead):
    from nanovllm import SamplingParams
    llm = make_llm(weights, max_num_batched_tokens=200, async_scheduling=ahead)       # forces chunking and mixed prefill+decode steps
    try:
        prompts = prompts_of([5, 300, 257, 64, 700], seed=3)
        sps = [SamplingParams(temperature=1e-9, max_tokens=n, ignore_eos=True) for n in (6, 4, 9, 3, 5)]
        outs = llm.generate(prompts, sps, use_tqdm=False)
        assert [len(o["token_ids"]) for o in outs] == [6, 4, 9, 3, 5] and all(o["text"] == "" for o in outs)
        exact = sum(check_teacher_forced(weights, p, o["token_ids"]) for p, o in zip(prompts, outs))
        assert exact >= 26                                     # 27 generated ids; at most one bf16 tie
        assert llm.is_finished() and not llm.scheduler.block_manager.used_block_ids
    finally:
        llm.exit()


@BOTH_LOOPS
def test_generate_prefix_reuse_and_preemption(weights, ahead):
    from nanovllm import SamplingParams
    llm = make_llm(weights, num_kvcache_blocks=4, max_num_batched_tokens=1024, async_scheduling=ahead)
    try:
        shared = prompts_of([520], seed=4)[0]
        tails = prompts_of([10, 20, 250], seed=5)
        prompts = [shared + tails[0], shared[:512] + tails[1], tails[2]]
        sp = SamplingParams(temperature=1e-9, max_tokens=12, ignore_eos=True)
        # first request alone, so its two full pages are hashed before the second one arrives
        first = llm.generate(prompts[:1], sp, use_tqdm=False)
        bm = llm.scheduler.block_manager
        llm.add_request(prompts[1], sp)
        assert bm.get_token_layout(llm.scheduler.waiting[0]) == (0, 512, 20)     # prefix found in released pages
        llm.add_request(prompts[2], sp)
        done = {}
        preempted = False
        while not llm.is_finished():
            before = len(llm.scheduler.waiting)
            outs, _ = llm.step()
            preempted = preempted or len(llm.scheduler.waiting) > before
            done.update(dict(outs))
        rest = [done[k] for k in sorted(done)]
        assert preempted, "4 pages must not be enough for both sequences to finish without pre-emption"
        for p, ids in zip(prompts, [first[0]["token_ids"]] + rest):
            assert len(ids) == 12
            check_teacher_forced(weights, p, ids)
    finally:
        llm.exit()


@BOTH_LOOPS
def test_step_without_finished_prompt_returns_no_tokens(weights, ahead):
    """A lone long prompt under chunked prefill: the first chunks produce no token and must not break the loop."""
    from nanovllm import SamplingParams
    llm = make_llm(weights, max_num_batched_tokens=128, async_scheduling=ahead)
    try:
        prompt = prompts_of([600], seed=6)[0]
        out = llm.generate([prompt], SamplingParams(temperature=1e-9, max_tokens=3, ignore_eos=True), use_tqdm=False)
        check_teacher_forced(weights, prompt, out[0]["token_ids"])
    finally:
        llm.exit()


def test_serving_harness_accounts_every_request(weights):
    """Poisson-arrival harness on the tiny model: all requests complete, token accounting and metric definitions."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import serving_bench as sb
    r = sb.main(["--model", TINY_DIR, "--num-requests", "24", "--request-rate", "400", "--max-num-batched-tokens", "256",
                 "--max-num-seqs", "16", "--max-model-len", "2048", "--random-input-len", "40", "--random-output-len", "7",
                 "--chunked-prefill", "--num-kvcache-blocks", "32", "--warmup-requests", "4"])
    assert r["completed"] == r["requests"] == 24
    assert r["input_tokens"] == 24 * 40 and r["output_tokens"] == 24 * 7
    assert r["mean_ttft_ms"] > 0 and r["mean_tpot_ms"] > 0 and r["mean_latency_s"] * 1e3 >= r["mean_ttft_ms"]
    assert abs(r["throughput_tok_s"] - (24 * 47) / r["total_time_s"]) < 1e-6


def test_generate_text_prompts_with_a_tokenizer(weights, tmp_path):
    """Text in, text out (reference llm_engine.py:113-114 tokenizer load, :131-134 encode, :265-267 decode): a model
    directory with tokenizer files is served through AutoTokenizer; eos comes from the tokenizer and ends a request
    unless ignore_eos.  The tokenizer is a word-level one built here over the tiny model's 1024 ids."""
    import json
    import shutil
    from tokenizers import Tokenizer, models, pre_tokenizers
    from nanovllm import LLM, SamplingParams
    shape, w = weights
    model_dir = tmp_path / "tiny-tok"
    shutil.copytree(TINY_DIR, model_dir)
    tok = Tokenizer(models.WordLevel({f"t{i:04d}": i for i in range(1024)}, unk_token="t0000"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.save(str(model_dir / "tokenizer.json"))
    (model_dir / "tokenizer_config.json").write_text(json.dumps({"tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "t0007", "unk_token": "t0000"}))
    llm = LLM(str(model_dir), max_num_batched_tokens=512, max_num_seqs=8, max_model_len=2048, num_kvcache_blocks=16,
              chunked_prefill=True, enforce_eager=True)
    try:
        llm.model_runner.load_weights(hf_named(w, shape))
        assert llm.tokenizer is not None and llm.config.eos == 7
        ids = prompts_of([12, 40], seed=9)
        texts = [" ".join(f"t{i:04d}" for i in p) for p in ids]      # fixed-width names: no token is a prefix of another
        outs = llm.generate(texts, SamplingParams(temperature=1e-9, max_tokens=5, ignore_eos=True), use_tqdm=False)
        for p, o in zip(ids, outs):
            assert len(o["token_ids"]) == 5
            assert o["text"].split() == [f"t{i:04d}" for i in o["token_ids"]]
            check_teacher_forced(weights, p, o["token_ids"])
        # eos: force it by making the request's first sampled id the eos id, then it must stop after one token
        first = outs[0]["token_ids"][0]
        llm.config.eos = llm.scheduler.eos = first
        again = llm.generate(texts[:1], SamplingParams(temperature=1e-9, max_tokens=5), use_tqdm=False)
        assert again[0]["token_ids"] == [first]
    finally:
        llm.exit()


def test_generate_uneven_contexts_over_the_work_list(weights, monkeypatch):
    """A long prompt next to short ones through `LLM.generate`, with the batch treated as a large one and a low
    long-context threshold, so every decode step (graph replay, fixed-capacity work list) and the one-token prefix of
    every mixed step plan a work list; sequences finish at different times (the graph bucket shrinks) and cross page
    boundaries.  Every generated id is checked against the oracle by teacher forcing."""
    from nanovllm import SamplingParams
    from nanovllm.engine.model_runner import ModelRunner
    monkeypatch.setattr(ModelRunner, "decode_splits", lambda self, bs: 1)
    monkeypatch.setattr(ModelRunner, "LONG_CONTEXT", 300)
    monkeypatch.setattr(ModelRunner, "MAX_RANGES", 4)
    from nanovllm.engine import model_runner as MR
    plans = []
    plan_items = MR._DecodeGraph._plan_items
    monkeypatch.setattr(MR._DecodeGraph, "_plan_items", lambda self: (plans.append(self.bs), plan_items(self))[1])
    llm = make_llm(weights, max_num_batched_tokens=700, num_kvcache_blocks=16)
    try:
        captured = len(plans)                                  # one plan per captured graph (its initial list)
        planned = []
        orig = llm.model_runner.native.forward
        llm.model_runner.native.forward = lambda *a, **k: (planned.append(k.get("decode_items") is not None), orig(*a, **k))[1]
        assert all(g.graph_long is not None for g in llm.model_runner.graphs.values())
        prompts = prompts_of([1500, 20, 250, 5, 600], seed=9)
        sps = [SamplingParams(temperature=1e-9, max_tokens=n, ignore_eos=True) for n in (30, 12, 20, 40, 8)]
        outs = llm.generate(prompts, sps, use_tqdm=False)
        assert [len(o["token_ids"]) for o in outs] == [30, 12, 20, 40, 8]
        exact = sum(check_teacher_forced(weights, p, o["token_ids"]) for p, o in zip(prompts, outs))
        assert exact >= 107                                    # 110 generated ids; a few bf16 ties at most
        assert any(planned), "no eager step planned a work list"
        assert len(plans) - captured >= 20 and len(set(plans[captured:])) >= 2, "graph replays over a work list, in two buckets at least"
        assert llm.is_finished() and not llm.scheduler.block_manager.used_block_ids
    finally:
        llm.exit()


def test_one_step_ahead_engine_draws_the_tokens_of_the_synchronous_engine(weights):
    """Config.async_scheduling launches step n+1 before it reads step n's ids (they are fed to it on the device).  With
    every request present from the start and no eos, both engines run the same batches, so even the sampler's noise
    is the same: identical ids at temperature 0.8, through chunked prefill, mixed steps and decode graphs."""
    from nanovllm import SamplingParams
    prompts = prompts_of([5, 300, 257, 64, 700, 31], seed=11)
    sps = [SamplingParams(temperature=0.8, max_tokens=n, ignore_eos=True) for n in (40, 4, 9, 300, 5, 1)]
    outs = []
    for ahead in (False, True):
        llm = make_llm(weights, max_num_batched_tokens=200, async_scheduling=ahead)
        try:
            outs.append([o["token_ids"] for o in llm.generate(prompts, sps, use_tqdm=False)])
            assert llm.is_finished() and not llm.scheduler.block_manager.used_block_ids
        finally:
            llm.exit()
    assert [len(t) for t in outs[0]] == [40, 4, 9, 300, 5, 1]
    assert outs[1] == outs[0]


def test_one_step_ahead_engine_ends_at_eos_and_survives_preemption(weights):
    """eos is seen one step late when the engine runs ahead: the token computed past it must be dropped, the
    pages released, and every id must still be the oracle's argmax at its position (near-greedy temperature)."""
    from nanovllm import SamplingParams
    prompts = prompts_of([250, 252, 255, 40, 9], seed=12)
    sp = SamplingParams(temperature=1e-9, max_tokens=40, ignore_eos=True)
    llm = make_llm(weights, num_kvcache_blocks=4, max_num_batched_tokens=1024, async_scheduling=False)
    try:
        free_run = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
    finally:
        llm.exit()
    # an id that occurs mid-way in several of the free-running completions becomes eos
    counts = {}
    for ids in free_run:
        for t in set(ids[3:30]):
            counts[t] = counts.get(t, 0) + 1
    eos = max(counts, key=counts.get)
    want = [ids[:ids.index(eos) + 1] if eos in ids else ids for ids in free_run]
    assert sum(len(w) < 40 for w in want) >= 1
    llm = make_llm(weights, num_kvcache_blocks=4, max_num_batched_tokens=1024, async_scheduling=True)
    try:
        llm.scheduler.eos = eos
        sp = SamplingParams(temperature=1e-9, max_tokens=40, ignore_eos=False)
        got = [o["token_ids"] for o in llm.generate(prompts, sp, use_tqdm=False)]
        assert llm.scheduler.num_preemptions > 0
        assert llm.is_finished() and not llm.scheduler.block_manager.used_block_ids and not llm.scheduler.block_manager.deferred
    finally:
        llm.exit()
    for p, ids, w in zip(prompts, got, want):
        assert -1 not in ids and (eos not in ids[:-1]) and (ids[-1] == eos or len(ids) == 40)
        check_teacher_forced(weights, p, ids)
    # (Not asserted: that `got` repeats the synchronous run id for id.  The two loops see different row counts in some
    # steps - a finished sequence costs the ahead loop one wasted row - a row count picks the GEMM tile, the tile the order
    # of a few fp32 sums, and a near-tie of two logits may then resolve the other way.  The invariant is the one checked
    # above: every id is the oracle's choice at its position, eos ends the sequence, nothing follows it.)
    assert sum(len(g) < 40 for g in got) >= 1
