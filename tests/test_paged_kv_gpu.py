"""the KV cache addressed through the host block manager's tables (nanovllm/block_manager.go,
Sequence.BlockTable / NumCachedTokens).  The block tables come from a restatement of that block manager
(tests/block_manager_mirror.py), used the way scheduler.go uses it.

Bars: paged and slab mode run the same kernels over the same tiles, so their logits must be IDENTICAL bit for bit;
against the CPU oracle the usual tolerances (f32 1e-4, bf16 1.5e-2 relative to max) and identical greedy ids."""
import numpy as np
import pytest

from block_manager_mirror import BlockManager
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = {"f32": 1e-4, "bf16": 1.5e-2}
BS = 64            # small blocks so that tiny prompts span several (a multiple of 64, like the default 256)


def models(gpu, oracle, family, precision, blocks=24, **over):
    cfg = gpu.synth.tiny_config(family, **over)
    w = gpu.synth.make_weights(cfg, seed=7, scale=0.05)
    om = oracle.OracleModel(cfg, w)
    slab = gpu.HipTransformerModel(cfg, w, precision=precision, max_seqs=4, max_batch_tokens=512)
    paged = gpu.HipTransformerModel(cfg, w, precision=precision, max_seqs=4, max_batch_tokens=512,
                                    kv_num_blocks=blocks, kv_block_size=BS)
    return cfg, om, slab, paged


@pytest.mark.parametrize("family", ["llama", "gpt2", "falcon", "granite_moe"])
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_paged_equals_slab_and_oracle(gpu, oracle, family, precision):
    """Prefill over scattered blocks + decode steps through MayAppend, batch of ragged sequences."""
    cfg, om, slab, paged = models(gpu, oracle, family, precision)
    r = np.random.default_rng(21)
    bm = BlockManager(24, BS)
    bm.free = list(r.permutation(24))                     # scattered physical blocks
    seqs = [gpu.Sequence(seq_id=i, token_ids=r.integers(0, cfg["vocab_size"], n).tolist(), block_size=BS)
            for i, n in enumerate((70, 130, 5))]
    for s in seqs:
        bm.allocate(s)
        slab.seq_reset(s.seq_id)
    lg_p, am_p = paged.forward_paged([s.token_ids for s in seqs], [0, 0, 0], [s.block_table for s in seqs])
    lg_s, am_s = slab.forward_batch([0, 1, 2], [s.token_ids for s in seqs], [0, 0, 0])
    assert np.array_equal(lg_p, lg_s) and np.array_equal(am_p, am_s)
    caches = []
    for i, s in enumerate(seqs):
        kv = om.new_cache()
        want = om.forward_with_cache(s.token_ids, kv, 0)[-1]
        caches.append(kv)
        assert rel_err(lg_p[i], want) <= TOL[precision]
    # K/V landed where the block table says (layer 1, sequence 1: three blocks)
    k, v = paged.get_kv_paged(seqs[1].block_table, 130, 1)
    ks, vs = slab.get_kv(1, 1)
    assert np.array_equal(k, ks) and np.array_equal(v, vs)
    # decode: append the greedy token, let the block manager extend the table, one step for the whole batch
    for step in range(70):                                # crosses block boundaries (70 -> 140, 130 -> 200, 5 -> 75)
        for s, t in zip(seqs, am_p):
            s.append_token(int(t))
            bm.may_append(s)
        pos = [len(s) - 1 for s in seqs]
        lg_p, am_p = paged.forward_paged([[s.token_ids[-1]] for s in seqs], pos, [s.block_table for s in seqs])
        lg_s, am_s = slab.forward_batch([0, 1, 2], [[s.token_ids[-1]] for s in seqs], pos)
        assert np.array_equal(lg_p, lg_s) and np.array_equal(am_p, am_s), step
        if step % 23 == 0:
            for i, s in enumerate(seqs):
                want = om.forward_with_cache([s.token_ids[-1]], caches[i], pos[i])[-1]
                assert rel_err(lg_p[i], want) <= TOL[precision]
        else:
            for i, s in enumerate(seqs):
                om.forward_with_cache([s.token_ids[-1]], caches[i], pos[i])
    slab.close(); paged.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_prefix_cache_hit_skips_prefill_work(gpu, oracle, precision):
    """Two prompts sharing their first 128 tokens (two full blocks): the second sequence's Allocate reports
    NumCachedTokens = 128 and shares the first sequence's blocks; the runner computes only its suffix, and the result
    equals a full prefill.  Also: both in ONE batch (the shared blocks are written and read within the call)."""
    cfg, om, slab, paged = models(gpu, oracle, "llama", precision)
    r = np.random.default_rng(22)
    common = r.integers(0, cfg["vocab_size"], 128).tolist()
    a = gpu.Sequence(seq_id=0, token_ids=common + r.integers(0, cfg["vocab_size"], 40).tolist(), block_size=BS)
    b = gpu.Sequence(seq_id=1, token_ids=common + r.integers(0, cfg["vocab_size"], 75).tolist(), block_size=BS)
    c = gpu.Sequence(seq_id=2, token_ids=list(common), block_size=BS)      # the whole prompt is cached
    bm = BlockManager(24, BS)
    runner = gpu.HipPagedModelRunner(paged)
    bm.allocate(a)
    assert a.num_cached_tokens == 0
    toks_a, lg_a = runner.run([a], True, return_logits=True)
    bm.allocate(b)
    bm.allocate(c)
    assert b.num_cached_tokens == 128 and b.block_table[:2] == a.block_table[:2]
    assert c.num_cached_tokens == 128 and c.block_table == a.block_table[:2]
    paged.reset_stats()
    toks_bc, lg_bc = runner.run([b, c], True, return_logits=True)
    assert paged.stats()["prefill_tokens"] == 75 + 1      # b's suffix + c's last token, not 203 + 128
    for s, lg in ((a, lg_a[0]), (b, lg_bc[0]), (c, lg_bc[1])):
        want = om.forward_with_cache(s.token_ids, om.new_cache(), 0)[-1]
        assert rel_err(lg, want) <= TOL[precision]
        slab.seq_reset(s.seq_id)
    # bit-identical to the slab path fed the same way (prefix, then the suffix as a second chunk)
    slab.forward_batch([1], [b.token_ids[:128]], [0], want_logits=False)
    lg_chunk, _ = slab.forward_batch([1], [b.token_ids[128:]], [128])
    assert np.array_equal(lg_chunk[0], lg_bc[0])
    # same-batch sharing: fresh pool, a and b allocated back to back, prefilled in one call
    bm2 = BlockManager(24, BS)
    a2 = gpu.Sequence(seq_id=0, token_ids=list(a.token_ids), block_size=BS)
    b2 = gpu.Sequence(seq_id=1, token_ids=list(b.token_ids), block_size=BS)
    bm2.allocate(a2); bm2.allocate(b2)
    assert b2.num_cached_tokens == 128
    paged2 = gpu.HipTransformerModel(cfg, gpu.synth.make_weights(cfg, seed=7, scale=0.05), precision=precision, max_seqs=4,
                                     max_batch_tokens=512, kv_num_blocks=24, kv_block_size=BS)
    _, lg2 = gpu.HipPagedModelRunner(paged2).run([a2, b2], True, return_logits=True)
    assert np.array_equal(lg2[0], lg_a[0]) and np.array_equal(lg2[1], lg_bc[0])
    # blocks of a finished sequence are reused by the next one (Deallocate -> Allocate) without stale reads
    bm.deallocate(a); bm.deallocate(b); bm.deallocate(c)
    d = gpu.Sequence(seq_id=3, token_ids=r.integers(0, cfg["vocab_size"], 100).tolist(), block_size=BS)
    bm.allocate(d)
    assert d.num_cached_tokens == 0
    _, lg_d = runner.run([d], True, return_logits=True)
    assert rel_err(lg_d[0], om.forward_with_cache(d.token_ids, om.new_cache(), 0)[-1]) <= TOL[precision]
    slab.close(); paged.close(); paged2.close()


def test_paged_mode_argument_errors(gpu, oracle):
    cfg, om, slab, paged = models(gpu, oracle, "llama", "f32", blocks=4)
    toks = [1, 2, 3]
    with pytest.raises(gpu.NvlError):                      # slot calls are refused in paged mode, and vice versa
        paged.forward_batch([0], [toks], [0])
    with pytest.raises(gpu.NvlError):
        slab.forward_paged([toks], [0], [[0]])
    with pytest.raises(gpu.NvlError):                      # block id outside the pool
        paged.forward_paged([toks], [0], [[9]])
    with pytest.raises(gpu.NvlError):                      # table does not cover the positions
        paged.forward_paged([list(range(70))], [0], [[1]])
    with pytest.raises(gpu.NvlError):                      # past max_seq_len (rope.go:84-86 would panic)
        paged.forward_paged([toks], [cfg["max_seq_len"] - 1], [[0, 1, 2, 3]])
    slab.close(); paged.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_fused_greedy_loop_over_block_tables(gpu, oracle, precision):
    """nvl_decode_greedy_paged: n steps with the token feedback on the device, KV appended through block tables that the
    block manager extended ahead (MayAppend per step) — ids identical to the slab model's nvl_decode_greedy and to the
    step-by-step paged loop; crosses block boundaries."""
    cfg, om, slab, paged = models(gpu, oracle, "llama", precision)
    r = np.random.default_rng(24)
    bm = BlockManager(24, BS)
    seqs = [gpu.Sequence(seq_id=i, token_ids=r.integers(0, cfg["vocab_size"], n).tolist(), block_size=BS)
            for i, n in enumerate((60, 100, 7))]
    steps = 40
    for s in seqs:
        bm.allocate(s)
        slab.seq_reset(s.seq_id)
    _, first = paged.forward_paged([s.token_ids for s in seqs], [0, 0, 0], [s.block_table for s in seqs], want_logits=False)
    _, first_s = slab.forward_batch([0, 1, 2], [s.token_ids for s in seqs], [0, 0, 0], want_logits=False)
    assert np.array_equal(first, first_s)
    # allocate ahead: the scheduler would call MayAppend after each AppendToken; do it with placeholder tokens
    ahead = []
    for s in seqs:
        t = gpu.Sequence(seq_id=s.seq_id, token_ids=list(s.token_ids), block_table=list(s.block_table), block_size=BS)
        for _ in range(steps):
            t.append_token(0)
            if len(t.token_ids) % BS == 1:                 # only the new-block branch matters for the table
                bid = bm.free[0]
                bm._allocate_block(bid)
                t.block_table.append(bid)
        ahead.append(t.block_table)
    pos = [len(s) for s in seqs]
    got = paged.decode_greedy_paged(first, pos, steps, ahead)
    want = slab.decode_greedy([0, 1, 2], first_s, steps)
    assert np.array_equal(got, want)
    with pytest.raises(gpu.NvlError):                      # tables that stop short of the generated positions
        paged.decode_greedy_paged(first, pos, steps, [s.block_table for s in seqs])
    slab.close(); paged.close()


def test_long_context_paged_decode_equals_slab(gpu, oracle):
    """>= 1024 cached keys in a tiny batch: the decode attention deals its key tiles over several workgroups per
    (sequence, kv head) (attn.h gridDim.z) — through block tables of scattered 64-token blocks exactly as through a slab:
    the same tiles go to the same workgroups, so the logits are identical bit for bit."""
    cfg, om, slab, paged = models(gpu, oracle, "llama", "bf16", blocks=48, max_seq_len=1280)
    slab.close(); paged.close()
    w = gpu.synth.make_weights(cfg, seed=7, scale=0.05)
    slab = gpu.HipTransformerModel(cfg, w, precision="bf16", max_seqs=4, max_batch_tokens=1280)
    paged = gpu.HipTransformerModel(cfg, w, precision="bf16", max_seqs=4, max_batch_tokens=1280, kv_num_blocks=48,
                                    kv_block_size=BS)
    r = np.random.default_rng(5)
    bm = BlockManager(48, BS)
    bm.free = list(r.permutation(48))
    seqs = [gpu.Sequence(seq_id=i, token_ids=r.integers(0, cfg["vocab_size"], n).tolist(), block_size=BS)
            for i, n in enumerate((1090, 130))]
    for s in seqs:
        bm.allocate(s)
        slab.seq_reset(s.seq_id)
    lg_p, am_p = paged.forward_paged([s.token_ids for s in seqs], [0, 0], [s.block_table for s in seqs])
    lg_s, am_s = slab.forward_batch([0, 1], [s.token_ids for s in seqs], [0, 0])
    assert np.array_equal(lg_p, lg_s)
    for step in range(6):
        for s, t in zip(seqs, am_p):
            s.append_token(int(t))
            bm.may_append(s)
        pos = [len(s) - 1 for s in seqs]
        lg_p, am_p = paged.forward_paged([[s.token_ids[-1]] for s in seqs], pos, [s.block_table for s in seqs])
        lg_s, am_s = slab.forward_batch([0, 1], [[s.token_ids[-1]] for s in seqs], pos)
        assert np.array_equal(lg_p, lg_s) and np.array_equal(am_p, am_s), step
    kv = om.new_cache()
    om.forward_with_cache(seqs[0].token_ids[:-1], kv, 0)
    want = om.forward_with_cache([seqs[0].token_ids[-1]], kv, len(seqs[0]) - 1)[-1]
    assert rel_err(lg_p[0], want) <= TOL["bf16"]
    slab.close(); paged.close()
