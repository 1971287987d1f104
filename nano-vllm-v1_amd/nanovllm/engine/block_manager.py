"""KV-page bookkeeping with prefix reuse (behaviour of the reference's engine/block_manager.py:100-440).

A page holds `block_size` consecutive tokens of one or more sequences.  Full pages are identified by a
chained content hash, xxh64(prefix_hash_le8 || int64 token bytes) (reference :135-165), so a new sequence
whose leading pages are already resident shares them instead of recomputing their K/V.  The page ids a
sequence receives are what the runner turns into slot_mapping and block_tables, so the ORDER in which ids
are handed out is part of the contract with the recorded scheduler traces: fresh pages come from the head
of the free list, released pages go to its tail (last page of the sequence first), and a released page keeps
its hash until it is handed out again.

Differences from the reference, both on purpose:
  * the free list is an insertion-ordered dict, so taking a specific page out of the middle (a prefix hit on
    a released page) is O(1) instead of deque.remove's O(n);
  * a page filled only partly by the first chunk of a prompt is NOT registered under the hash of the full
    page (the reference registers it, :318-329, and then trips its own assertion in may_append, :420, when
    the next chunk extends that page).
"""
from __future__ import annotations

import numpy as np
import xxhash

from .sequence import PLACEHOLDER, Sequence

NO_HASH = -1


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = NO_HASH
        self.token_ids: list[int] = []


class _FreeList:
    """Ordered set of page ids: O(1) head, append and removal of any member."""

    def __init__(self, ids):
        self._d = dict.fromkeys(ids)

    def __len__(self):
        return len(self._d)

    def __contains__(self, i):
        return i in self._d

    def __iter__(self):
        return iter(self._d)

    def __getitem__(self, k):
        if k == 0:
            return next(iter(self._d))
        return list(self._d)[k]

    def append(self, i):
        self._d[i] = None

    def remove(self, i):
        del self._d[i]


class BlockManager:
    def __init__(self, num_blocks: int, block_size: int):
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self.free_block_ids = _FreeList(range(num_blocks))
        self.used_block_ids: set[int] = set()
        # pages that filled up with a token whose id is still on the GPU (seq_id -> (sequence, page index)): they are
        # hashed by label_deferred() once the id is known
        self.deferred: dict[int, tuple[Sequence, int]] = {}

    # ------------------------------------------------------------------ hashing
    @classmethod
    def compute_hash(cls, token_ids: list[int], prefix: int = NO_HASH) -> int:
        h = xxhash.xxh64()
        if prefix != NO_HASH:
            h.update(prefix.to_bytes(8, "little"))
        h.update(np.asarray(token_ids, dtype=np.int64).tobytes())
        return h.intdigest()

    def _chain(self, seq: Sequence):
        """Yields (index, tokens, hash, resident page id | -1) for the pages of `seq` in order; the hash chain
        only continues through full pages."""
        h = NO_HASH
        for i in range(seq.num_blocks):
            tokens = seq.block(i)
            h = self.compute_hash(tokens, h) if len(tokens) == self.block_size else NO_HASH
            page = self.hash_to_block_id.get(h, -1)
            if page != -1 and self.blocks[page].token_ids != tokens:
                page = -1                                   # hash collision: treat as absent
            if i == seq.num_blocks - 1:
                page = -1                                   # the last page is always recomputed (it yields the logits)
            yield i, tokens, h, page

    # ------------------------------------------------------------------ page movement
    def _take(self, page: int) -> Block:
        block = self.blocks[page]
        assert block.ref_count == 0
        if self.hash_to_block_id.get(block.hash) == page:
            del self.hash_to_block_id[block.hash]           # the old content is gone for good
        block.ref_count, block.hash, block.token_ids = 1, NO_HASH, []
        self.free_block_ids.remove(page)
        self.used_block_ids.add(page)
        return block

    def _release(self, page: int):
        assert self.blocks[page].ref_count == 0
        self.used_block_ids.remove(page)
        self.free_block_ids.append(page)

    def _label(self, block: Block, h: int, tokens: list[int]):
        block.hash, block.token_ids = h, tokens
        self.hash_to_block_id[h] = block.block_id

    def _pages_for(self, num_tokens: int) -> int:
        return -(-num_tokens // self.block_size)

    # ------------------------------------------------------------------ admission of a waiting sequence
    def can_allocate(self, num_tokens: int) -> bool:
        return len(self.free_block_ids) >= self._pages_for(num_tokens)

    def get_token_layout(self, seq: Sequence) -> tuple[int, int, int]:
        """(prefix tokens resident in pages in use, prefix tokens resident in released pages, tokens to compute)."""
        assert not seq.block_table
        in_used = in_free = new = 0
        hitting = True
        for _, tokens, _, page in self._chain(seq):
            hitting = hitting and page != -1
            if not hitting:
                new += len(tokens)
            elif page in self.used_block_ids:
                in_used += len(tokens)
            else:
                in_free += len(tokens)
        return in_used, in_free, new

    def allocate(self, seq: Sequence):
        """Give `seq` its shared prefix pages, then fresh pages for the seq.num_new_tokens scheduled tokens."""
        assert not seq.block_table
        bs = self.block_size
        h = NO_HASH
        for _, tokens, h, page in self._chain(seq):
            if page == -1:
                break
            seq.num_cached_tokens += bs
            if page in self.used_block_ids:
                block = self.blocks[page]
                block.ref_count += 1
            else:
                block = self._take(page)
            self._label(block, h, tokens)
            seq.block_table.append(page)
        # `h` is now the chained hash of the first page to compute (or NO_HASH when that page is not full)
        first, end = seq.num_cached_tokens, seq.num_cached_tokens + seq.num_new_tokens
        for lo in range(first, end, bs):
            tokens = seq[lo:min(lo + bs, end)]
            if lo != first:
                h = self.compute_hash(tokens, h) if len(tokens) == bs else NO_HASH
            block = self._take(self.free_block_ids[0])
            if h != NO_HASH and len(tokens) == bs:
                self._label(block, h, tokens)
            seq.block_table.append(block.block_id)

    def deallocate(self, seq: Sequence):
        self.deferred.pop(seq.seq_id, None)
        for page in reversed(seq.block_table):
            block = self.blocks[page]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._release(page)
        seq.num_cached_tokens = 0
        seq.num_new_tokens = 0
        seq.block_table.clear()

    # ------------------------------------------------------------------ growth of a running sequence
    def can_append(self, seq: Sequence, num_new_tokens: int) -> bool:
        room = -seq.num_cached_tokens % self.block_size      # free token slots in the last resident page
        return self._pages_for(num_new_tokens - room) <= len(self.free_block_ids)

    def may_append(self, seq: Sequence):
        """Pages for the seq.num_new_tokens scheduled tokens of a running sequence; pages that become full are
        hashed (chained on their predecessor) so later arrivals can share them."""
        bs = self.block_size
        end = seq.num_cached_tokens + seq.num_new_tokens
        first = seq.num_cached_blocks
        if first < len(seq.block_table) and end < (first + 1) * bs:
            return                                           # the tokens fit the last resident page without filling it
        for lo in range(first * bs, end, bs):
            tokens = seq[lo:min(lo + bs, end)]
            index = lo // bs
            block = self.blocks[seq.block_table[index]] if index < len(seq.block_table) else None
            assert block is None or block.hash == NO_HASH, "a page being extended must not be shareable yet"
            if block is None:
                block = self._take(self.free_block_ids[0])
                seq.block_table.append(block.block_id)
            if len(tokens) == bs:
                if tokens[-1] == PLACEHOLDER:
                    self.deferred[seq.seq_id] = (seq, index)     # hashed when the id arrives
                    continue
                prefix = self.blocks[seq.block_table[index - 1]].hash if index else NO_HASH
                self._label(block, self.compute_hash(tokens, prefix), tokens)

    def label_deferred(self, seq: Sequence):
        """Hash the page of `seq` that filled up while its last token id was unknown (call once the id is in)."""
        entry = self.deferred.pop(seq.seq_id, None)
        if entry is None:
            return
        _, index = entry
        tokens = seq.block(index)
        assert len(tokens) == self.block_size and PLACEHOLDER not in tokens[-2:]
        block = self.blocks[seq.block_table[index]]
        prefix = self.blocks[seq.block_table[index - 1]].hash if index else NO_HASH
        if block.hash == NO_HASH:
            self._label(block, self.compute_hash(tokens, prefix), tokens)
