"""Test infrastructure: a Python restatement of nanovllm.BlockManager (nanovllm/block_manager.go:42-263) — the host
component that stays in Go and computes Sequence.BlockTable / NumCachedTokens.  The paged-KV tests drive the HIP path
with block tables produced exactly the way the scheduler would produce them (Allocate at prefill, MayAppend before
every decode step, Deallocate at the end), including prefix-cache hits."""
import struct

import xxhash


class Block:
    def __init__(self, block_id):
        self.block_id, self.ref_count, self.hash, self.token_ids = block_id, 0, 0, []


class BlockManager:
    def __init__(self, num_blocks, block_size):                       # :52-68
        self.block_size = block_size
        self.blocks = [Block(i) for i in range(num_blocks)]
        self.hash_to_block = {}
        self.free = list(range(num_blocks))
        self.used = set()

    def compute_hash(self, token_ids, prefix_hash):                   # :71-87 (xxhash64, LE u64 prefix + LE u32 tokens)
        h = xxhash.xxh64()
        if prefix_hash != 0:
            h.update(struct.pack("<Q", prefix_hash))
        for t in token_ids:
            h.update(struct.pack("<I", t & 0xFFFFFFFF))
        return h.intdigest()

    def _allocate_block(self, bid):                                   # :90-108
        b = self.blocks[bid]
        assert b.ref_count == 0
        b.ref_count, b.hash, b.token_ids = 1, 0, []
        self.free.remove(bid)
        self.used.add(bid)
        return b

    def can_allocate(self, seq):                                      # :123-125
        return len(self.free) >= self.num_blocks(seq)

    def num_blocks(self, seq):                                        # sequence.go:86-88
        return (len(seq.token_ids) + self.block_size - 1) // self.block_size

    def allocate(self, seq):                                          # :128-203
        assert not seq.block_table
        h, miss = 0, False
        for i in range(self.num_blocks(seq)):
            toks = seq.token_ids[i * self.block_size:(i + 1) * self.block_size]
            h = self.compute_hash(toks, h) if len(toks) == self.block_size else 0
            bid = self.hash_to_block.get(h, -1) if h != 0 else -1
            if bid != -1 and self.blocks[bid].token_ids != toks:
                bid = -1
            if bid == -1:
                miss = True
            if miss:
                bid = self.free[0]
                self._allocate_block(bid)
            else:
                seq.num_cached_tokens += self.block_size
                if bid in self.used:
                    self.blocks[bid].ref_count += 1
                else:
                    self._allocate_block(bid)
            if h != 0:
                self.blocks[bid].hash, self.blocks[bid].token_ids = h, list(toks)
                self.hash_to_block[h] = bid
            seq.block_table.append(bid)

    def deallocate(self, seq):                                        # :206-219
        for bid in reversed(seq.block_table):
            b = self.blocks[bid]
            b.ref_count -= 1
            if b.ref_count == 0:
                self.used.discard(bid)
                self.free.append(bid)
        seq.num_cached_tokens = 0
        seq.block_table = []

    def may_append(self, seq):                                        # :231-263 (called after AppendToken)
        last = self.blocks[seq.block_table[-1]]
        n = len(seq.token_ids)
        if n % self.block_size == 1:
            assert last.hash != 0
            bid = self.free[0]
            self._allocate_block(bid)
            seq.block_table.append(bid)
        elif n % self.block_size == 0:
            assert last.hash == 0
            toks = seq.token_ids[-self.block_size:]
            prefix = self.blocks[seq.block_table[-2]].hash if len(seq.block_table) > 1 else 0
            h = self.compute_hash(toks, prefix)
            last.hash, last.token_ids = h, list(toks)
            self.hash_to_block[h] = last.block_id
        else:
            assert last.hash == 0
