"""Mamba2 / Granite-4 hybrid layers on the device (csrc/mamba.h) against the CPU oracle's
restatement of purego/tensor/mamba2.go:74-351 and of the hybrid dispatch (generic_model.go:67-202, 285-292, 456-459).

Tolerances as everywhere: fp32 parity mode 1e-4, bf16 product path 1.5e-2 of the largest value.

Two behaviours of the reference are mirrored on purpose: (1) no convolution state is carried across calls, so a decode
call convolves its token with zeros (mamba2.go: ConvCache is never used); (2) the SSM state persists across calls and is
reset when a sequence (re)starts at position 0.  One is NOT mirrored, deliberately: the reference keeps the SSM state on
the LAYER (one state for whichever sequence ran last), the device keeps it per SEQUENCE — checked here with interleaved
sequences against one fresh oracle model per sequence."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = {"f32": 1e-4, "bf16": 1.5e-2}


def build(gpu, oracle, precision, **over):
    cfg = gpu.synth.tiny_config("granite_hybrid", **over)
    w = gpu.synth.make_weights(cfg, seed=17, scale=0.05)
    hm = gpu.HipTransformerModel(cfg, w, precision=precision, max_seqs=4, max_batch_tokens=256)
    return cfg, w, hm


VARIANTS = [dict(),                                                            # heads 8 x 32, state 32, 2 groups
            dict(mamba_num_heads=4, mamba_head_dim=64, mamba_state_size=128, mamba_n_groups=1),   # Granite-4-1b geometry: hd 64, 128 states
            dict(mamba_num_heads=8, mamba_head_dim=0, mamba_state_size=64, mamba_n_groups=4, mamba_conv_kernel=3,
                 hybrid_layers=["attention", "mamba", "mamba", "attention"])]


@pytest.mark.parametrize("variant", range(len(VARIANTS)))
@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("ntok", [23, 150])                                     # decode-sized and prefill-sized kernels
def test_hybrid_prefill_decode_and_state(gpu, oracle, precision, variant, ntok):
    cfg, w, hm = build(gpu, oracle, precision, **VARIANTS[variant])
    om = oracle.OracleModel(cfg, w)
    tol = TOL[precision]
    r = np.random.default_rng(5 + variant)
    toks = r.integers(0, cfg["vocab_size"], ntok).tolist()
    kv = om.new_cache()
    want, want_h = om.forward_with_cache(toks, kv, 0, want_hidden=True)
    hm.set_debug(True)
    got = hm.forward_with_cache(toks, seq_id=3, pos_offset=0)
    got_h = hm.get_hidden(ntok)
    for li in range(cfg["num_layers"]):
        assert rel_err(got_h[li], want_h[li]) <= tol, f"layer {li} ({cfg['hybrid_layers'][li]})"
    assert rel_err(got, want) <= tol
    for li, kind in enumerate(cfg["hybrid_layers"]):
        if kind == "mamba":                                                     # Mamba2Layer.SSMState after the prompt
            assert rel_err(hm.get_mamba_state(3, li), om.mamba_state(li)) <= tol
        else:
            k_ref, v_ref = kv.layer(li, cfg["num_kv_heads"], cfg["head_dim"])
            k_dev, v_dev = hm.get_kv(3, li)
            assert rel_err(k_dev, k_ref) <= tol and rel_err(v_dev, v_ref) <= tol
    hm.set_debug(False)
    # decode, teacher-forced: every step's logits (the conv sees only the step's own token: the reference's behaviour)
    pos = ntok
    for t in r.integers(0, cfg["vocab_size"], 4).tolist():
        want = om.forward_with_cache([t], kv, pos)[-1]
        got = hm.forward_with_cache([t], seq_id=3, pos_offset=pos, all_logits=False)[-1]
        assert rel_err(got, want) <= tol
        pos += 1
    for li, kind in enumerate(cfg["hybrid_layers"]):
        if kind == "mamba":
            assert rel_err(hm.get_mamba_state(3, li), om.mamba_state(li)) <= tol
    # a further multi-token call on the same cache (ForwardWithCache with S > 1 at pos_offset > 0): state carries on
    more = r.integers(0, cfg["vocab_size"], 9).tolist()
    want = om.forward_with_cache(more, kv, pos)
    got = hm.forward_with_cache(more, seq_id=3, pos_offset=pos)
    assert rel_err(got, want) <= tol
    hm.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_per_sequence_state_with_interleaved_sequences(gpu, oracle, precision):
    """Three sequences share batched calls; each must behave as if it were alone.  The reference (state on the layer)
    cannot do this; the oracle stands in with ONE fresh model per sequence."""
    cfg, w, hm = build(gpu, oracle, precision)
    tol = TOL[precision]
    r = np.random.default_rng(9)
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (7, 31, 2)]
    oms = [oracle.OracleModel(cfg, w) for _ in prompts]
    kvs = [om.new_cache() for om in oms]
    for i in range(3):
        hm.seq_reset(10 + i)
    logits, am = hm.forward_batch([10, 11, 12], prompts, [0, 0, 0])
    nxt = []
    for i, p in enumerate(prompts):
        want = oms[i].forward_with_cache(p, kvs[i], 0)[-1]
        assert rel_err(logits[i], want) <= tol
        nxt.append(oracle.argmax(want))
    pos = [len(p) for p in prompts]
    for step in range(3):
        order = [(step + j) % 3 for j in range(3)]                              # batch order changes every step
        lg, _ = hm.forward_batch([10 + i for i in order], [[nxt[i]] for i in order], [pos[i] for i in order])
        for row, i in enumerate(order):
            want = oms[i].forward_with_cache([nxt[i]], kvs[i], pos[i])[-1]
            assert rel_err(lg[row], want) <= tol
            nxt[i] = oracle.argmax(want)
            pos[i] += 1
    # a sequence that restarts at position 0 starts from a zero state again (ResetState, generic_model.go:285-292)
    hm.seq_reset(11)
    again, _ = hm.forward_batch([11], [prompts[1]], [0])
    fresh = oracle.OracleModel(cfg, w)
    assert rel_err(again[0], fresh.forward_with_cache(prompts[1], fresh.new_cache(), 0)[-1]) <= tol
    hm.close()


def test_hybrid_greedy_fused_equals_stepwise_and_runner(gpu, oracle):
    """The fused device decode loop and the ModelRunner path on a hybrid model: same tokens as the oracle's greedy loop."""
    cfg = gpu.synth.tiny_config("granite_hybrid", tied_embedding=False)
    w = gpu.synth.make_weights(cfg, seed=17, scale=0.05, peaked_head=4.0)
    hm = gpu.HipTransformerModel(cfg, w, precision="f32", max_seqs=4, max_batch_tokens=256)
    om = oracle.OracleModel(cfg, w)
    prompt = np.random.default_rng(2).integers(0, cfg["vocab_size"], 12).tolist()
    want = om.greedy(prompt, 8)
    assert hm.greedy(prompt, 8, seq_id=1) == want
    assert hm.greedy_fused(prompt, 8, seq_id=2) == want
    runner = gpu.HipModelRunner(hm)
    seq = gpu.Sequence(seq_id=77, token_ids=list(prompt))
    got = []
    t = runner.run([seq], True)[0]
    for _ in range(8):
        got.append(t)
        seq.append_token(t)
        t = runner.run([seq], False)[0]
    assert got == want
    hm.close()


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("variant", [0, 2])                                     # conv kernel 4 and 3
def test_runner_chunked_prefill_of_a_long_history_matches_one_forward(gpu, oracle, precision, variant):
    """A history longer than max_batch_tokens is prefilled in chunks by the runner; the reference runs ONE Forward over
    it, whose Mamba2 convolution window spans what are chunk seams here (mamba2.go:183-254).  The chunks after the first
    take the previous chunk's last K-1 raw xBC rows (mamba.h MambaArgs::chain): logits and the SSM state equal the
    oracle's one-shot prefill — also with a last chunk shorter than the convolution window."""
    cfg = gpu.synth.tiny_config("granite_hybrid", **VARIANTS[variant])
    w = gpu.synth.make_weights(cfg, seed=17, scale=0.05)
    tol = TOL[precision]
    for ntok in (3 * 64 + 1, 2 * 64 + 37):                                     # last chunk of 1 token / of 37
        hm = gpu.HipTransformerModel(cfg, w, precision=precision, max_seqs=2, max_batch_tokens=64)
        om = oracle.OracleModel(cfg, w)
        toks = np.random.default_rng(40 + ntok).integers(0, cfg["vocab_size"], ntok).tolist()
        want = om.forward_with_cache(toks, om.new_cache(), 0)[-1]
        runner = gpu.HipModelRunner(hm)
        _, lg = runner.run([gpu.Sequence(seq_id=5, token_ids=toks)], True, return_logits=True)
        assert hm.stats()["forward_calls"] == -(-ntok // 64)                   # it really was chunked
        assert rel_err(lg[0], want) <= tol
        for li, kind in enumerate(cfg["hybrid_layers"]):
            if kind == "mamba":
                assert rel_err(hm.get_mamba_state(5, li), om.mamba_state(li)) <= tol
        # a plain multi-call nvl_forward keeps the reference's semantics (each call = one Forward: zero padding)
        hm.seq_reset(6)
        om2 = oracle.OracleModel(cfg, w)
        kv = om2.new_cache()
        om2.forward_with_cache(toks[:64], kv, 0)
        want2 = om2.forward_with_cache(toks[64:100], kv, 64)[-1]
        hm.forward_batch([6], [toks[:64]], [0], want_logits=False)
        got2, _ = hm.forward_batch([6], [toks[64:100]], [64])
        assert rel_err(got2[0], want2) <= tol
        hm.close()


@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_chunked_ssd_scan_matches_oracle_and_the_sequential_scan(gpu, oracle, variant):
    """csrc/mamba.h mamba_ssd_kernel (prefill-sized bf16 calls: 64 tokens per MFMA step) against the oracle's sequential
    recurrence (mamba2.go:256-351) and against the sequential device kernel (tuning key 30 = 0): a ragged batch of three
    sequences (lengths straddling the 64-token chunk), then a continuation of each on its carried state (seq_pos > 0).
    Key 30 = 1 takes the chunk-parallel three-launch form here (four chunks, few heads), 2 the chunk-after-chunk form."""
    cfg = gpu.synth.tiny_config("granite_hybrid", **dict(VARIANTS[variant], max_seq_len=512))
    w = gpu.synth.make_weights(cfg, seed=17, scale=0.05)
    r = np.random.default_rng(50 + variant)
    lens = (200, 64, 129)
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in lens]
    more = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (70, 17, 33)]
    want, want2, states = [], [], []
    for p_, q_ in zip(prompts, more):
        om = oracle.OracleModel(cfg, w)
        kv = om.new_cache()
        want.append(om.forward_with_cache(p_, kv, 0)[-1])
        want2.append(om.forward_with_cache(q_, kv, len(p_))[-1])
        states.append([om.mamba_state(li) for li, k in enumerate(cfg["hybrid_layers"]) if k == "mamba"])
    got = {}
    for ssd in (1, 2, 0):
        old = gpu.lib().nvl_set_tuning(30, ssd)
        try:
            hm = gpu.HipTransformerModel(cfg, w, precision="bf16", max_seqs=4, max_batch_tokens=512)
            ids = [0, 1, 2]
            for i in ids:
                hm.seq_reset(i)
            a, _ = hm.forward_batch(ids, prompts, [0, 0, 0])
            b, _ = hm.forward_batch(ids, more, list(lens))
            st = [[hm.get_mamba_state(i, li) for li, k in enumerate(cfg["hybrid_layers"]) if k == "mamba"] for i in ids]
            got[ssd] = (a, b, st)
            hm.close()
        finally:
            gpu.lib().nvl_set_tuning(30, old)
    for ssd in (1, 2, 0):
        a, b, st = got[ssd]
        for i in range(3):
            assert rel_err(a[i], want[i]) <= TOL["bf16"], (ssd, "prefill", i)
            assert rel_err(b[i], want2[i]) <= TOL["bf16"], (ssd, "continuation", i)
            for x, y in zip(st[i], states[i]):
                assert rel_err(x, y) <= TOL["bf16"], (ssd, "state", i)
    # the two device forms agree much more closely with each other than either needs to with the oracle
    for ssd in (1, 2):
        assert rel_err(got[ssd][0], got[0][0]) <= 6e-3 and rel_err(got[ssd][1], got[0][1]) <= 6e-3


def test_hybrid_is_refused_where_it_cannot_work(gpu):
    cfg = gpu.synth.tiny_config("granite_hybrid")
    with pytest.raises(gpu.NvlError):                       # per-sequence state lives in KV slots: no paged mode
        gpu.HipTransformerModel(cfg, None, kv_num_blocks=8)
    with pytest.raises(gpu.NvlError):                       # tensor parallelism does not shard Mamba2 blocks
        gpu.HipTransformerModel(cfg, None, tp_rank=0, tp_size=2)
    bad = dict(cfg, mamba_num_heads=3)                      # heads x head_dim != expand x hidden
    with pytest.raises(gpu.NvlError):
        gpu.HipTransformerModel(bad, None)
