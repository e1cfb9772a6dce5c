"""tensor.SampleWithHistory (purego/tensor/sampling.go:33-102) on the device (nvl_op_sample, nvl_sample,
nvl_runner_run_sampled) against the CPU oracle's restatement, with the rand.Float32() draw passed to both.

Stated tolerance.  The reference sums V probabilities sequentially in fp32 (softmax denominator, renormalisation,
CDF): that sum itself carries up to V * 2^-24 relative rounding error (7.6e-3 worst case at V = 128256, ~1e-4
observed), while the device sums in a fixed tree order.  So
  * the final distribution must agree within 1e-3 relative per entry on the common support, and the supports may
    differ (at a top-p cut through thousands of near-equal probabilities the reference's cumulative sum is itself
    only that accurate) by entries of at most 1e-3 total probability mass; and
  * the sampled id must own the oracle's CDF interval around r = u * sum up to 2e-3 of probability mass
    (an id can differ from the Go result only when r falls that close to a CDF step).
sort.Slice is unstable in the reference: among EQUAL probabilities at a top-k / top-p cut the survivors are
unspecified there; the oracle and the device both keep the lowest indices, so supports are compared exactly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PARAMS = [
    dict(),                                                              # DefaultSamplingParams: T 1, p 1, k 0, rp 1.2
    dict(temperature=0.7, repetition_penalty=1.0),
    dict(top_k=50),
    dict(top_p=0.9),
    dict(temperature=0.8, top_k=40, top_p=0.95, repetition_penalty=1.3),
    dict(top_k=1),                                                       # greedy through the sampler
    dict(top_p=0.0),                                                     # cutoff after the first element
    dict(temperature=0.0, top_p=0.5),                                    # T <= 0: logits stay unscaled (sampling.go:71)
    dict(top_k=5, top_p=0.999999),
]


def check_rows(oracle, logits, hists, u, got_tok, got_probs, kw, exact_support=False):
    for i in range(logits.shape[0]):
        want_tok, want_p = oracle.sample_with_history(logits[i], hists[i], float(u[i]), return_probs=True, **kw)
        both = (got_probs[i] > 0) & (want_p > 0)
        assert (np.abs(got_probs[i] - want_p)[both] <= 1e-3 * want_p[both] + 1e-12).all(), (i, kw)
        odd = (got_probs[i] > 0) != (want_p > 0)
        assert np.maximum(got_probs[i], want_p)[odd].sum() <= 1e-3, (i, kw, int(odd.sum()))
        if exact_support:
            assert not odd.any(), (i, kw)
        cum = np.cumsum(want_p.astype(np.float64))
        r = float(u[i]) * cum[-1]
        g = int(got_tok[i])
        lo = cum[g - 1] if g > 0 else 0.0
        assert lo - 2e-3 <= r <= cum[g] + 2e-3, (i, kw, g, want_tok, lo, cum[g], r)
        assert want_p[g] > 0 or g == 0 or g == want_tok


@pytest.mark.parametrize("V", [1000, 50257, 128256])
@pytest.mark.parametrize("kw", PARAMS, ids=[str(i) for i in range(len(PARAMS))])
def test_sample_matches_oracle(gpu, oracle, V, kw):
    r = np.random.default_rng(V + len(kw))
    rows = 5
    logits = (r.standard_normal((rows, V)) * np.array([0.5, 2.0, 4.0, 8.0, 1.0])[:, None]).astype(np.float32)
    hists = [r.integers(0, V, n).tolist() for n in (0, 3, 25, 700, 12)]
    hists[4] = [7, 7, 7, 9] + hists[4]                       # repeated tokens: counts multiply the penalty
    hists[2][5] = V + 3                                       # ids >= V are ignored (sampling.go:57)
    u = r.random(rows).astype(np.float32)
    u[0] = 0.0                                                # rand.Float32() can return 0: index 0 wins (sort.Search)
    tok, probs = gpu.ops.sample_with_history(logits, hists, u, return_probs=True, **kw)
    check_rows(oracle, logits, hists, u, tok, probs, kw)


def test_ties_at_the_cut_keep_lowest_indices(gpu, oracle):
    """Quantised logits: many exactly equal probabilities straddle the top-k and the top-p cut."""
    r = np.random.default_rng(3)
    V = 4096
    logits = np.round(r.standard_normal((3, V)) * 2).astype(np.float32)      # ~15 distinct values
    u = np.array([0.3, 0.6, 0.95], np.float32)
    for kw in (dict(top_k=100, repetition_penalty=1.0), dict(top_p=0.5, repetition_penalty=1.0),
               dict(top_k=300, top_p=0.8, repetition_penalty=1.0)):
        tok, probs = gpu.ops.sample_with_history(logits, None, u, return_probs=True, **kw)
        check_rows(oracle, logits, [None] * 3, u, tok, probs, kw, exact_support=True)
    flat = np.zeros((1, 512), np.float32)                                    # uniform distribution: ALL ties
    tok, probs = gpu.ops.sample_with_history(flat, None, [0.5], return_probs=True, top_k=10, repetition_penalty=1.0)
    assert np.flatnonzero(probs[0]).tolist() == list(range(10)) and tok[0] in range(10)


def test_sampled_frequencies_follow_the_distribution(gpu, oracle):
    r = np.random.default_rng(5)
    V, rows, reps = 48, 256, 40
    logits = (r.standard_normal(V) * 1.5).astype(np.float32)
    _, p = oracle.sample_with_history(logits, None, 0.5, return_probs=True, temperature=0.9, repetition_penalty=1.0)
    counts = np.zeros(V)
    for _ in range(reps):
        u = r.random(rows).astype(np.float32)
        tok = gpu.ops.sample_with_history(np.tile(logits, (rows, 1)), None, u, temperature=0.9, repetition_penalty=1.0)
        counts += np.bincount(tok, minlength=V)
    n = rows * reps
    chi2 = (((counts - n * p) ** 2) / np.maximum(n * p, 1e-9))[p * n > 5].sum()
    assert chi2 < 2.5 * V                      # dof ~ V; a wrong CDF gives chi2 in the thousands


def test_model_sampling_and_runner(gpu, oracle):
    """nvl_sample on the logits a forward left on the device == the op on the same logits; nvl_runner_run_sampled ==
    TensorModelRunner.Run with SampleWithHistory(logits, seq.TokenIDs, defaultSampling) (tensor_model_runner.go:93)."""
    cfg = gpu.synth.tiny_config("llama")
    w = gpu.synth.make_weights(cfg, seed=7, scale=0.05)
    hm = gpu.HipTransformerModel(cfg, w, precision="f32", max_seqs=8, max_batch_tokens=256)
    r = np.random.default_rng(9)
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (5, 19, 2)]
    for i in range(3):
        hm.seq_reset(i)
    logits, _ = hm.forward_batch([0, 1, 2], prompts, [0, 0, 0])
    u = r.random(3).astype(np.float32)
    kw = dict(temperature=0.8, top_k=20, top_p=0.95, repetition_penalty=1.2)
    got = hm.sample(prompts, u, **kw)
    assert np.array_equal(got, gpu.ops.sample_with_history(logits, prompts, u, **kw))
    for i in range(3):
        want = oracle.sample_with_history(logits[i], prompts[i], float(u[i]), **kw)
        assert got[i] == want                  # (tiny vocabulary, random u: no CDF-step coincidence)

    runner = gpu.HipModelRunner(hm)
    runner.set_sampling_params_with_repetition(0.8, 0.95, 20, 1.2)
    seqs = [gpu.Sequence(seq_id=10 + i, token_ids=list(p)) for i, p in enumerate(prompts)]
    for step in range(4):
        u = r.random(3).astype(np.float32)
        _, lg = runner.run(seqs, is_prefill=(step == 0), return_logits=True)      # greedy call: logits for the oracle
        toks = runner.run_sampled(seqs, is_prefill=True, uniforms=u) if step == 0 else None
        if toks is None:
            # the greedy call above already advanced the cache to len(seq): sample those same logits on the device
            toks = hm.sample([s.token_ids for s in seqs], u, **kw).tolist()
        for i, s in enumerate(seqs):
            assert toks[i] == oracle.sample_with_history(lg[i], s.token_ids, float(u[i]), **kw)
            s.append_token(toks[i])
    # and the one-call form over several decode steps stays consistent with the cache bookkeeping
    for step in range(3):
        u = r.random(3).astype(np.float32)
        toks = runner.run_sampled(seqs, is_prefill=False, uniforms=u)
        assert all(0 <= t < cfg["vocab_size"] for t in toks)
        for s, t in zip(seqs, toks):
            s.append_token(t)
        assert [hm.seq_len(s.seq_id) for s in seqs] == [len(s) - 1 for s in seqs]
    hm.close()


def test_sampling_argument_errors(gpu):
    lg = np.zeros((1, 16), np.float32)
    with pytest.raises(gpu.NvlError):
        gpu.ops.sample_with_history(lg, [[-1]], [0.5])               # the reference would index out of range
    with pytest.raises(gpu.NvlError):
        gpu.ops.sample_with_history(lg, None, [1.5])
    with pytest.raises(gpu.NvlError):
        gpu.ops.sample_with_history(lg, None, [0.5], repetition_penalty=0.0)


@pytest.mark.parametrize("family", ["llama", "gpt2", "granite_moe"])
def test_fused_sampled_decode_equals_stepwise(gpu, oracle, family):
    """nvl_decode_sampled (sample + token feedback + history append on the device, n steps in one call) == the same loop
    driven step by step: nvl_forward, nvl_sample with the grown history, append (tensor_model_runner.go:55-97)."""
    cfg = gpu.synth.tiny_config(family)
    w = gpu.synth.make_weights(cfg, seed=7, scale=0.05)
    hm = gpu.HipTransformerModel(cfg, w, precision="bf16", max_seqs=4, max_batch_tokens=256)
    r = np.random.default_rng(31)
    prompts = [r.integers(0, cfg["vocab_size"], n).tolist() for n in (9, 33, 2)]
    ids, steps = [0, 1, 2], 14
    kw = dict(temperature=0.9, top_k=50, top_p=0.95, repetition_penalty=1.2)
    U = r.random((steps + 1, 3)).astype(np.float32)
    # step by step through the host
    for i in ids:
        hm.seq_reset(i)
    hm.forward_batch(ids, prompts, [0, 0, 0], want_logits=False)
    hist = [list(p) for p in prompts]
    first = hm.sample(hist, U[0], **kw)
    for h, t in zip(hist, first):
        h.append(int(t))
    want = []
    for s in range(steps):
        hm.forward_batch(ids, [[h[-1]] for h in hist], [len(h) - 1 for h in hist], want_logits=False)
        tok = hm.sample(hist, U[s + 1], **kw)
        want.append(tok.copy())
        for h, t in zip(hist, tok):
            h.append(int(t))
    # fused
    for i in ids:
        hm.seq_reset(i)
    hm.forward_batch(ids, prompts, [0, 0, 0], want_logits=False)
    hist2 = [list(p) for p in prompts]
    first2 = hm.sample(hist2, U[0], **kw)
    assert np.array_equal(first2, first)
    for h, t in zip(hist2, first2):
        h.append(int(t))
    got = hm.decode_sampled(ids, first2, steps, hist2, U[1:], **kw)
    assert np.array_equal(got, np.stack(want))
    assert [hm.seq_len(i) for i in ids] == [len(p) + steps for p in prompts]
    with pytest.raises(gpu.NvlError):                       # a draw outside [0, 1]
        hm.decode_sampled(ids, got[-1], 2, [h + [0] for h in hist2], np.full((2, 3), 1.5, np.float32), **kw)
    hm.close()
