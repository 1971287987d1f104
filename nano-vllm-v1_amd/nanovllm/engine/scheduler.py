"""Step scheduler in the v1 style (behaviour of the reference's engine/scheduler.py:58-267): running
sequences first, LIFO pre-emption when pages run out, then admission from the waiting queue - all under one
token budget per step, with optional chunking of long prefills.

What the runner sees of a scheduled sequence is (num_cached_tokens, num_new_tokens, block_table); decode is
simply num_new_tokens == 1.  `schedule()` returns the list (running sequences first, newly admitted after);
`postprocess()` consumes the sampled ids, which exist only for the sequences whose scheduled tokens reach the
end of what they hold (`seq_need_compute_logits` indexes into the scheduled list).

An engine that runs one step ahead (Config.async_scheduling; not in the reference) splits `postprocess()` in two:
`advance()` right after a step is launched - everything that depends on LENGTHS only, with a placeholder for each
token still to come - and `commit()` once the ids have been read back, one step later: it fills the placeholders
in, ends the sequences that drew eos (the step launched in between computed one token too many for them, which is
dropped) and hashes the pages that filled up meanwhile.  The sequences end with the same tokens as under
`postprocess()`.
"""
from __future__ import annotations

from collections import deque

from ..config import Config
from .block_manager import BlockManager
from .sequence import PLACEHOLDER, Sequence, SequenceStatus


class Scheduler:
    def __init__(self, config: Config):
        self.enable_chunked = config.chunked_prefill
        self.max_model_len = config.max_model_len
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()
        self.num_preemptions = 0            # bookkeeping for the serving harness; scheduling never reads it

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence):
        assert len(seq) <= self.max_model_len - 1, "Sequence length exceeds max_model_len"
        self.waiting.append(seq)

    # ------------------------------------------------------------------ one step
    def schedule(self) -> list[Sequence]:
        budget = self.max_num_batched_tokens
        continued, budget, preempted = self._continue_running(budget)
        admitted = [] if preempted else self._admit_waiting(budget)
        scheduled = continued + admitted
        assert scheduled, "nothing could be scheduled"
        return scheduled

    def _continue_running(self, budget: int):
        """Running sequences in queue order.  When pages run out the NEWEST running sequence is pre-empted
        (possibly the one being considered, which ends the pass)."""
        bm = self.block_manager
        bs = bm.block_size
        scheduled: list[Sequence] = []
        preempted = False
        i = 0
        while i < len(self.running) and budget > 0:
            seq = self.running[i]
            n = len(seq) - seq.num_cached_tokens
            if self.enable_chunked:
                n = min(n, budget)
            n = min(n, self.max_model_len - 1 - seq.num_cached_tokens)
            assert n > 0
            if n == 1 and 0 < seq.num_cached_tokens % bs < bs - 1:
                # a decode token that stays inside its partly filled page: no page to take, none to hash
                # (can_append is trivially true and may_append would do nothing)
                seq.num_new_tokens = 1
                scheduled.append(seq)
                budget -= 1
                i += 1
                continue
            while not bm.can_append(seq, n) and len(self.running) > i:
                self.preempt(self.running.pop())
                preempted = True
            if len(self.running) == i:                       # seq itself was the victim
                break
            seq.num_new_tokens = n
            bm.may_append(seq)
            scheduled.append(seq)
            budget -= n
            i += 1
        return scheduled, budget, preempted

    def _admit_waiting(self, budget: int) -> list[Sequence]:
        """Head-of-line admission while tokens, pages and sequence slots last.  Pre-emption in this step
        disables admission (the caller checks), or the freed pages would be taken straight back."""
        bm = self.block_manager
        admitted: list[Sequence] = []
        while self.waiting and budget > 0 and len(self.running) < self.max_num_seqs:
            seq = self.waiting[0]
            assert not seq.block_table
            hit_used, hit_free, n = bm.get_token_layout(seq)
            if self.enable_chunked:
                n = min(n, budget)
            assert n > 0
            if n > budget or not bm.can_allocate(hit_free + n):
                break
            seq.num_new_tokens = n
            bm.allocate(seq)
            assert seq.num_cached_tokens == hit_used + hit_free
            budget -= n
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            admitted.append(seq)
        return admitted

    def preempt(self, seq: Sequence):
        """Drop the sequence's pages and put it at the head of the waiting queue; it is recomputed (or finds its
        pages through the prefix hashes) when it is admitted again."""
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)
        self.num_preemptions += 1

    def postprocess(self, seqs: list[Sequence], token_ids: list[int], seq_need_compute_logits) -> None:
        need = [int(i) for i in seq_need_compute_logits]
        assert len(token_ids) == len(need)
        for index, token_id in zip(need, token_ids):
            seq = seqs[index]
            seq.append_token(token_id)
            hit_eos = not seq.ignore_eos and token_id == self.eos
            if hit_eos or seq.num_completion_tokens == seq.max_tokens or len(seq) >= self.max_model_len:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
        for seq in seqs:
            if seq.status != SequenceStatus.FINISHED:
                seq.num_cached_tokens += seq.num_new_tokens
                seq.num_new_tokens = 0

    # ------------------------------------------------------------------ one step ahead
    def advance(self, seqs: list[Sequence], seq_need_compute_logits) -> list[tuple[Sequence, int]]:
        """Bookkeeping of a step that has been launched but whose sampled ids are still on the GPU.  Returns, per
        entry of the need list, (sequence, index of its placeholder): what `commit()` will fill in."""
        pending = []
        for row, index in enumerate(int(i) for i in seq_need_compute_logits):
            seq = seqs[index]
            seq.append_token(PLACEHOLDER)
            seq.pending_slot = row
            pending.append((seq, len(seq) - 1))
            if seq.num_completion_tokens == seq.max_tokens or len(seq) >= self.max_model_len:
                # ends here whatever the id turns out to be; its pages go back now (a step launched later cannot
                # overtake this one on the GPU)
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
        for seq in seqs:
            if seq.status != SequenceStatus.FINISHED:
                seq.num_cached_tokens += seq.num_new_tokens
                seq.num_new_tokens = 0
        return pending

    def commit(self, pending: list[tuple[Sequence, int]], token_ids: list[int]) -> list[Sequence]:
        """The ids of an `advance()`d step have arrived.  Returns the sequences that are finished with every id known."""
        assert len(pending) == len(token_ids)
        done = []
        for (seq, pos), token_id in zip(pending, token_ids):
            if pos >= len(seq):
                continue                        # drew eos one step earlier: this is the token computed past its end
            assert seq.token_ids[pos] == PLACEHOLDER
            seq.token_ids[pos] = token_id
            hit_eos = not seq.ignore_eos and token_id == self.eos
            if hit_eos:
                # a step launched meanwhile fed eos back in and sampled once more: that token is dropped
                del seq.token_ids[pos + 1:]
                seq.num_tokens = pos + 1
            if pos == len(seq) - 1:
                seq.last_token = token_id
            if seq.status == SequenceStatus.FINISHED:       # ended by length in advance(), in this step or the next
                if pos == len(seq) - 1:
                    done.append(seq)
            elif hit_eos:
                if seq.status == SequenceStatus.RUNNING:
                    self.running.remove(seq)
                else:
                    self.waiting.remove(seq)    # pre-empted in the step between
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                done.append(seq)
            elif seq.status == SequenceStatus.RUNNING:
                self.block_manager.label_deferred(seq)
        return done
