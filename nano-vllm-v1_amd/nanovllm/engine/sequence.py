"""Per-request state.  The runner reads exactly the fields that survive pickling to the
tensor-parallel workers (reference: engine/sequence.py:32-235): token_ids, num_cached_tokens,
num_new_tokens, block_table, temperature, and the length."""
from __future__ import annotations

from copy import copy
from enum import Enum, auto
from itertools import count

from ..sampling_params import SamplingParams


PLACEHOLDER = -1     # a sampled token whose id has not been read back yet (engine steps that run one step ahead)


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    block_size = 256
    counter = count()

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams | None = None):
        sp = sampling_params or SamplingParams()
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = copy(token_ids)
        self.last_token = token_ids[-1]
        self.num_tokens = len(token_ids)
        self.num_prompt_tokens = len(token_ids)
        self.num_cached_tokens = 0          # tokens whose K/V are already in the cache
        self.num_new_tokens = 0             # tokens scheduled for this step
        self.block_table: list[int] = []
        self.temperature = sp.temperature
        self.max_tokens = sp.max_tokens
        self.ignore_eos = sp.ignore_eos
        self.pending_slot = -1              # row of the newest placeholder's id in the sampler output it will come from

    def __len__(self):
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self):
        return self.status == SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    @property
    def num_context_tokens(self):
        return self.num_cached_tokens + self.num_new_tokens

    @property
    def prompt_token_ids(self):
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self):
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self):
        return (self.num_tokens + self.block_size - 1) // self.block_size

    def block(self, i):
        assert 0 <= i < self.num_blocks
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    def append_token(self, token_id: int):
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # Workers only need the scheduled slice of token ids, not the whole history: the payload
    # stays O(new tokens) per step instead of O(context).
    def __getstate__(self):
        lo, hi = self.num_cached_tokens, self.num_cached_tokens + self.num_new_tokens
        return (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.num_new_tokens,
                self.block_table, self.temperature, lo, self.token_ids[lo:hi], self.pending_slot)

    def __setstate__(self, state):
        (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.num_new_tokens,
         self.block_table, self.temperature, lo, window, self.pending_slot) = state
        self.token_ids = _Window(lo, window, self.num_tokens)
        self.last_token = window[-1] if window else -1


class _Window:
    """Read-only view of token_ids[lo:lo+len(window)] addressed with absolute indices."""

    def __init__(self, lo: int, window: list[int], total: int):
        self.lo, self.window, self.total = lo, window, total

    def __len__(self):
        return self.total

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, step = key.indices(self.total)
            assert step == 1 and start >= self.lo and stop <= self.lo + len(self.window)
            return self.window[start - self.lo:stop - self.lo]
        return self.window[key - self.lo]
