"""The engine loop that runs one step ahead of its results (Config.async_scheduling) against the synchronous loop of
the reference (engine/llm_engine.py:117-126 -> scheduler.postprocess): same requests, a stand-in runner whose
"sampled" id is a fixed function of (sequence, position), so both loops must finish every sequence with the same
tokens - through eos, length limits, chunked prefill, prefix reuse and pre-emption - and leave the page manager clean."""
import itertools
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from nanovllm.engine.block_manager import NO_HASH
from nanovllm.engine.model_runner import flatten_batch
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import PLACEHOLDER, Sequence
from nanovllm.sampling_params import SamplingParams

EOS = 7
BASE = dict(chunked_prefill=True, max_model_len=2048, max_num_seqs=16, max_num_batched_tokens=512, eos=EOS,
            num_kvcache_blocks=64, kvcache_block_size=256)


def token_for(seq_id: int, position: int) -> int:
    """The id "sampled" after the token at `position` of sequence `seq_id`: eos now and then."""
    h = (seq_id * 1000003 + position * 7919) % 1009
    return EOS if h % 23 == 0 else 10 + h


class FakeRunner:
    """Stands in for ModelRunner: checks what a real step would read (page tables, slots, ids or placeholders where
    they are allowed) and returns the ids the position function dictates."""

    def __init__(self):
        self.prev = None                           # (sequence ids, positions) of the previous step's sampled rows
        self.steps = 0

    def _sample(self, seqs, fb):
        need = fb["need"].tolist()
        rows = [(seqs[i].seq_id, seqs[i].num_cached_tokens + seqs[i].num_new_tokens - 1) for i in need]
        return need, rows

    def run(self, seqs):
        fb = flatten_batch(seqs, 256)
        assert (fb["input_ids"] >= 0).all()
        need, rows = self._sample(seqs, fb)
        self.steps += 1
        return [token_for(s, p) for s, p in rows], need

    def run_async(self, seqs):
        fb = flatten_batch(seqs, 256)
        # a placeholder may only be the LAST scheduled token of a sequence, and must name the row of the previous
        # step's sampler output that holds this very sequence's id for this very position
        ids, cu = fb["input_ids"], fb["cu_seqlens_q"]
        for i, s in enumerate(seqs):
            window = ids[cu[i]:cu[i + 1]]
            assert (window[:-1] >= 0).all()
            if window[-1] == PLACEHOLDER:
                pos = s.num_cached_tokens + s.num_new_tokens - 1
                assert self.prev is not None and self.prev[s.pending_slot] == (s.seq_id, pos - 1)
        if fb["block_tables"] is not None:
            assert (fb["slot_mapping"] >= 0).all()
        need, rows = self._sample(seqs, fb)
        self.prev = rows
        self.steps += 1
        tokens = [token_for(s, p) for s, p in rows]
        return SimpleNamespace(need=need, result=lambda: tokens)


def requests(seed, n, shared_prefix=0):
    g = torch.Generator().manual_seed(seed)
    shared = torch.randint(100, 5000, (shared_prefix,), generator=g).tolist()
    out = []
    for i in range(n):
        length = int(torch.randint(3, 700, (1,), generator=g))
        prompt = shared + torch.randint(100, 5000, (length,), generator=g).tolist()
        out.append((prompt, int(torch.randint(1, 400, (1,), generator=g)), bool(i % 5 == 4)))
    return out


def run_sync(cfg, reqs, arrivals):
    Sequence.counter = itertools.count()
    sched, runner = Scheduler(SimpleNamespace(**cfg)), FakeRunner()
    finished, it, step = {}, iter(reqs), 0
    while True:
        for _ in range(arrivals(step)):
            r = next(it, None)
            if r is not None:
                sched.add(Sequence(r[0], SamplingParams(max_tokens=r[1], ignore_eos=r[2])))
        step += 1
        if sched.is_finished():
            if step > len(reqs) + 5:
                break
            continue
        seqs = sched.schedule()
        tokens, need = runner.run(seqs)
        sched.postprocess(seqs, tokens, need)
        for s in seqs:
            if s.is_finished:
                finished[s.seq_id] = list(s.completion_token_ids)
    return finished, sched, runner


def run_ahead(cfg, reqs, arrivals):
    """The loop of LLMEngine._step_ahead around the stand-in runner."""
    Sequence.counter = itertools.count()
    sched, runner = Scheduler(SimpleNamespace(**cfg)), FakeRunner()
    finished, it, step, inflight = {}, iter(reqs), 0, None
    while True:
        for _ in range(arrivals(step)):
            r = next(it, None)
            if r is not None:
                sched.add(Sequence(r[0], SamplingParams(max_tokens=r[1], ignore_eos=r[2])))
        step += 1
        if sched.is_finished() and inflight is None:
            if step > len(reqs) + 5:
                break
            continue
        launched = None
        if not sched.is_finished():
            seqs = sched.schedule()
            tokens = runner.run_async(seqs)
            launched = (sched.advance(seqs, tokens.need), tokens)
        if inflight is not None:
            for s in sched.commit(inflight[0], inflight[1].result()):
                assert s.seq_id not in finished
                finished[s.seq_id] = list(s.completion_token_ids)
        inflight = launched
    return finished, sched, runner


def check_clean(sched, cfg):
    bm = sched.block_manager
    assert len(bm.free_block_ids) == cfg["num_kvcache_blocks"] and not bm.used_block_ids and not bm.deferred
    assert all(b.ref_count == 0 for b in bm.blocks)
    assert not sched.running and not sched.waiting


@pytest.mark.parametrize("tag,cfg,n,arrive", [
    ("roomy", BASE, 40, lambda step: 3 if step % 4 == 0 else 0),
    ("all at once", BASE, 24, lambda step: 24 if step == 0 else 0),
    ("pages run out", dict(BASE, num_kvcache_blocks=12), 30, lambda step: 2),
    ("whole prompts only", dict(BASE, chunked_prefill=False, max_num_batched_tokens=2048), 24, lambda step: 1),
    ("short model_len", dict(BASE, max_model_len=600), 30, lambda step: 2),
])
def test_one_step_ahead_finishes_every_sequence_with_the_synchronous_tokens(tag, cfg, n, arrive):
    reqs = requests(len(tag), n)
    reqs = [(p[:cfg["max_model_len"] - 2], m, e) for p, m, e in reqs]
    want, s0, r0 = run_sync(cfg, reqs, arrive)
    got, s1, r1 = run_ahead(cfg, reqs, arrive)
    assert set(got) == set(want) == set(range(n))
    for k in want:
        assert got[k] == want[k], (tag, k)
        assert PLACEHOLDER not in got[k]
    assert any(v[-1] == EOS for v in want.values()) and any(v[-1] != EOS for v in want.values())   # both endings occurred
    check_clean(s0, cfg)
    check_clean(s1, cfg)
    if tag == "pages run out":
        assert s0.num_preemptions > 0 and s1.num_preemptions > 0


def test_pages_filled_during_decode_are_hashed_once_their_last_id_is_known():
    """A page that fills up in a decode step is registered for prefix reuse in both loops (one step later when the
    step ran ahead of its ids): afterwards the same full pages are known, under the same hashes."""
    cfg = dict(BASE, num_kvcache_blocks=256)
    reqs = [(p, 300 + m, True) for p, m, _ in requests(3, 12)]          # everyone decodes across at least one page edge
    _, s0, _ = run_sync(cfg, reqs, lambda step: 12 if step == 0 else 0)
    _, s1, _ = run_ahead(cfg, reqs, lambda step: 12 if step == 0 else 0)
    h0, h1 = set(s0.block_manager.hash_to_block_id), set(s1.block_manager.hash_to_block_id)
    assert h0 == h1 and len(h0) >= 12
    for bm in (s0.block_manager, s1.block_manager):
        for h, page in bm.hash_to_block_id.items():
            b = bm.blocks[page]
            assert b.hash == h and len(b.token_ids) == 256 and PLACEHOLDER not in b.token_ids


def test_a_later_request_reuses_pages_that_filled_while_running_ahead():
    cfg = dict(BASE, num_kvcache_blocks=64)
    Sequence.counter = itertools.count()
    first = [(list(range(100, 350)), 300, True)]
    got, sched, runner = run_ahead(cfg, first, lambda step: 1 if step == 0 else 0)
    full = list(range(100, 350)) + got[0]
    seq = Sequence(full[:540], SamplingParams(max_tokens=2, ignore_eos=True))   # two full pages + a tail
    sched.add(seq)
    seqs = sched.schedule()
    assert seqs == [seq] and seq.num_cached_tokens == 512 and seq.num_new_tokens == 28
    assert all(b != NO_HASH for b in (sched.block_manager.blocks[p].hash for p in seq.block_table[:2]))
