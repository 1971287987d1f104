"""Scheduler + page manager against traces recorded from the reference's own classes
(tests/golden/f4_sched_traces.json, made by oracle/gen_golden.py) and its prefix-hash vectors (f5)."""
import itertools
import json
from types import SimpleNamespace

import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR

from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.model_runner import flatten_batch
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams

BASE = dict(chunked_prefill=True, max_model_len=4096, max_num_seqs=8, max_num_batched_tokens=512,
            eos=-1, num_kvcache_blocks=16, kvcache_block_size=256)


def scenarios():
    """The arrival scripts of the generator, with its prompt stream (one torch generator, drawn in this order)."""
    pg = torch.Generator().manual_seed(9)

    def pr(n):
        return torch.randint(0, 10000, (n,), generator=pg).tolist()

    shared = pr(600)
    out = {}
    out["chunked"] = (BASE, {0: [(pr(700), 5), (pr(30), 4)], 2: [(pr(10), 3)]}, 14)
    out["prefix"] = (BASE, {0: [(shared + pr(20), 3)], 1: [(shared[:512] + pr(33), 3)], 6: [(shared[:512] + pr(5), 2)]}, 14)
    out["preempt"] = (dict(BASE, num_kvcache_blocks=4, max_num_batched_tokens=1024), {0: [(pr(250), 12), (pr(252), 12), (pr(255), 12)]}, 60)
    out["unchunked"] = (dict(BASE, chunked_prefill=False), {0: [(pr(300), 2), (pr(300), 2), (pr(100), 2)]}, 10)
    return out


@pytest.fixture(scope="module")
def traces():
    return json.loads((GOLDEN_DIR / "f4_sched_traces.json").read_text())


def test_scheduler_reproduces_reference_traces(traces):
    Sequence.counter = itertools.count()
    for tag, (cfg, arrivals, steps) in scenarios().items():
        sched = Scheduler(SimpleNamespace(**cfg))
        rec = iter(traces[tag])
        n_steps = 0
        for st in range(steps):
            for prompt, max_tokens in arrivals.get(st, []):
                sched.add(Sequence(prompt, SamplingParams(temperature=1.0, max_tokens=max_tokens, ignore_eos=True)))
            if sched.is_finished():
                continue
            seqs = sched.schedule()
            want = next(rec)
            where = f"{tag} step {st}"
            assert [s.seq_id for s in seqs] == want["seq_ids"], where
            assert [s.num_cached_tokens for s in seqs] == want["cached"], where
            assert [s.num_new_tokens for s in seqs] == want["new"], where
            assert [len(s) for s in seqs] == want["lens"], where
            assert [list(s.block_table) for s in seqs] == want["tables"], where
            fb = flatten_batch(seqs)
            assert fb["input_ids"].tolist() == want["input_ids"], where
            assert fb["slot_mapping"].tolist() == want["slots"], where
            need = fb["need"].tolist()
            assert need == want["need"], where
            toks = [(sum(seqs[i].token_ids[-3:]) + st) % 1000 for i in need]
            assert toks == want["sampled"], where
            sched.postprocess(seqs, toks, need)
            n_steps += 1
        assert next(rec, None) is None and n_steps == len(traces[tag]), tag
        assert sched.is_finished(), tag
        bm = sched.block_manager
        assert not bm.used_block_ids and len(bm.free_block_ids) == cfg["num_kvcache_blocks"]


def test_prefix_hashes_match_reference():
    want = json.loads((GOLDEN_DIR / "f5_hashes.json").read_text())
    h0 = BlockManager.compute_hash(list(range(256)))
    assert h0 == want["h0"]
    assert BlockManager.compute_hash(list(range(1000, 1256)), h0) == want["h1"]
    assert BlockManager.compute_hash([1, 2, 3]) == want["h_short"]


def _mk(tokens, max_tokens=4):
    return Sequence(list(tokens), SamplingParams(temperature=1.0, max_tokens=max_tokens, ignore_eos=True))


def test_prefix_pages_are_shared_and_survive_release():
    bm = BlockManager(8, 256)
    a = _mk(range(600))
    assert bm.get_token_layout(a) == (0, 0, 600)
    a.num_new_tokens = 600
    bm.allocate(a)
    assert a.block_table == [0, 1, 2] and a.num_cached_tokens == 0
    b = _mk(list(range(512)) + [7, 7, 7])
    assert bm.get_token_layout(b) == (512, 0, 3)            # two full pages resident and in use
    b.num_new_tokens = 3
    bm.allocate(b)
    assert b.block_table == [0, 1, 3] and b.num_cached_tokens == 512
    assert bm.blocks[0].ref_count == 2
    bm.deallocate(a)
    assert bm.blocks[0].ref_count == 1 and 2 in bm.free_block_ids
    bm.deallocate(b)
    assert not bm.used_block_ids
    c = _mk(list(range(512)) + [9])
    assert bm.get_token_layout(c) == (0, 512, 1)            # hashes outlive the release
    c.num_new_tokens = 1
    bm.allocate(c)
    assert c.block_table[:2] == [0, 1] and c.num_cached_tokens == 512
    # a sequence that IS exactly its cached pages still recomputes the last page
    d = _mk(range(512))
    assert bm.get_token_layout(d) == (256, 0, 256)


def test_partial_first_chunk_is_not_registered_as_full_page():
    """A prompt longer than a page whose first chunk ends inside the page: the page must stay unhashed until it
    fills (the reference labels it with the full page's hash and then fails its own assertion on the next chunk)."""
    cfg = dict(BASE, max_num_batched_tokens=300)
    sched = Scheduler(SimpleNamespace(**cfg))
    short, long = _mk(range(200), 8), _mk(range(1000, 1700), 2)
    sched.add(short)
    sched.add(long)
    seqs = sched.schedule()                                  # 200 + first 100 tokens of the long prompt
    assert [s.num_new_tokens for s in seqs] == [200, 100]
    assert sched.block_manager.blocks[long.block_table[0]].hash == -1
    sched.postprocess(seqs, [5], [0])
    for _ in range(8):
        seqs = sched.schedule()                              # next chunks extend the partly filled page
        need = flatten_batch(seqs)["need"].tolist()
        sched.postprocess(seqs, [3] * len(need), need)
        if sched.is_finished():
            break
    assert sched.is_finished()


def test_preemption_is_lifo_and_blocks_admission():
    cfg = dict(BASE, num_kvcache_blocks=2, max_num_batched_tokens=1024)
    sched = Scheduler(SimpleNamespace(**cfg))
    a, b, c = _mk(range(255), 4), _mk(range(300, 556), 4), _mk(range(5), 4)
    for s in (a, b):
        sched.add(s)
    seqs = sched.schedule()
    assert seqs == [a, b]
    sched.postprocess(seqs, [1, 1], [0, 1])                  # a: 256 tokens (page full), b: 257 -> needs a 2nd page
    sched.add(c)
    seqs = sched.schedule()
    # b needs a new page, none is free: b (the newest running) is pre-empted and nothing is admitted this step
    assert seqs == [a] and list(sched.waiting) == [b, c] and not b.block_table and b.num_cached_tokens == 0


def test_cpu_plumbing_one_prompt_through_scheduler_and_oracle():
    """The CPU-runnable configuration: one 16-token prompt, 16 new tokens, greedy, with the scheduler and page manager
    in the loop and the CPU oracle as the runner.  The ids must be the ones a plain loop over the oracle produces,
    whether the prompt is prefilled at once or in chunks of 5 tokens; the pages are the ones the scheduler handed out."""
    from conftest import TINY
    from oracle import ref_model as M
    from oracle import ref_ops as R
    shape = M.ModelShape(**TINY)
    weights = M.random_weights(shape, seed=0, std=0.05)
    g = torch.Generator().manual_seed(16)
    prompt = torch.randint(0, 1024, (16,), generator=g).tolist()

    # plain loop: whole prompt, then one token at a time
    oracle = M.RefRunner(shape, weights, 2)
    st = M.SeqState(list(prompt), 0, 16, [1])
    logits, _ = oracle.run([st])
    want = []
    for _ in range(16):
        tok = int(R.greedy(logits)[0])
        want.append(tok)
        st.num_cached_tokens += st.num_new_tokens
        st.token_ids.append(tok)
        st.num_new_tokens = 1
        logits, _ = oracle.run([st])

    for budget in (1024, 5):
        cfg = dict(BASE, num_kvcache_blocks=2, max_num_batched_tokens=budget, chunked_prefill=True)
        sched = Scheduler(SimpleNamespace(**cfg))
        seq = Sequence(list(prompt), SamplingParams(temperature=1.0, max_tokens=16, ignore_eos=True))
        sched.add(seq)
        oracle = M.RefRunner(shape, weights, 2)
        steps = 0
        while not sched.is_finished():
            seqs = sched.schedule()
            states = [M.SeqState(list(s.token_ids), s.num_cached_tokens, s.num_new_tokens, list(s.block_table)) for s in seqs]
            logits, need = oracle.run(states)
            toks = R.greedy(logits).tolist() if need.numel() else []
            sched.postprocess(seqs, toks, need.tolist())
            steps += 1
        assert seq.completion_token_ids == want, f"budget {budget}"
        assert steps == (16 if budget == 1024 else 19)         # 1 (or 4) prefill steps + 15 decode steps
