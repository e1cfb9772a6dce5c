"""The oracle's op restatements against float64 numpy formulas, and the properties the domain offers
(softmax rows sum to 1, RoPE preserves norms, causal masking, KV-cache incremental == full)."""
import numpy as np
import pytest

from conftest import rel_err


def rng(s):
    return np.random.default_rng(s)


def test_matmul_layernorm_softmax(oracle):
    r = rng(0)
    a = r.standard_normal((5, 96), dtype=np.float32)
    b = r.standard_normal((96, 33), dtype=np.float32)
    assert rel_err(oracle.matmul(a, b), a.astype(np.float64) @ b.astype(np.float64)) < 1e-5
    x = r.standard_normal((4, 64), dtype=np.float32) * 2 + 1
    w = r.standard_normal(64, dtype=np.float32)
    bias = r.standard_normal(64, dtype=np.float32)
    x64 = x.astype(np.float64)
    rms = np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5)
    assert rel_err(oracle.layernorm(x, w, None, 1e-5), x64 / rms * w) < 1e-5           # bias == nil => RMSNorm
    mu, var = x64.mean(-1, keepdims=True), x64.var(-1, keepdims=True)
    assert rel_err(oracle.layernorm(x, w, bias, 1e-5), (x64 - mu) / np.sqrt(var + 1e-5) * w + bias) < 1e-5
    s = oracle.softmax(x)
    assert np.allclose(s.sum(-1), 1.0, atol=1e-6) and np.all(s > 0)
    assert np.array_equal(oracle.softmax(np.zeros((1, 1), np.float32)), np.ones((1, 1), np.float32))


def test_activations(oracle):
    x = np.linspace(-8, 8, 1001, dtype=np.float32)
    x64 = x.astype(np.float64)
    assert rel_err(oracle.gelu(x), 0.5 * x64 * (1 + np.tanh(np.sqrt(2 / np.pi) * (x64 + 0.044715 * x64 ** 3)))) < 1e-6
    assert rel_err(oracle.silu(x), x64 / (1 + np.exp(-x64))) < 1e-6


def test_rope_tables_and_rotation(oracle):
    c, s = oracle.rope_tables(64, 32, 500000.0)
    assert np.array_equal(c[:, :32], c[:, 32:]) and np.array_equal(s[:, :32], s[:, 32:])   # rope.go:43-46
    assert np.all(c[0] == 1.0) and np.all(s[0] == 0.0)
    t = rng(1).standard_normal((2, 5, 64), dtype=np.float32)
    y = oracle.rope_apply(t, 3, 10000.0, 32)
    assert np.allclose(np.linalg.norm(y, axis=-1), np.linalg.norm(t, axis=-1), rtol=1e-5)  # rotations keep norms
    assert np.array_equal(oracle.rope_apply(t[:, :1], 0, 10000.0, 32), t[:, :1])           # position 0 = identity
    with pytest.raises(RuntimeError):
        oracle.rope_apply(t, 28, 10000.0, 32)                                              # rope.go:176 panic


def test_gqa_core_is_causal_and_matches_float64(oracle):
    r = rng(2)
    nH, nKV, S, T, hd = 4, 2, 6, 9, 64
    q = r.standard_normal((nH, S, hd), dtype=np.float32)
    k = r.standard_normal((nKV, T, hd), dtype=np.float32)
    v = r.standard_normal((nKV, T, hd), dtype=np.float32)
    out = oracle.gqa_core(q, k, v)
    ref = np.zeros_like(out, dtype=np.float64)
    for h in range(nH):
        kk, vv = k[h // 2].astype(np.float64), v[h // 2].astype(np.float64)
        for i in range(S):
            n = T - S + i + 1
            sc = q[h, i].astype(np.float64) @ kk[:n].T / 8.0
            p = np.exp(sc - sc.max())
            ref[h, i] = (p / p.sum()) @ vv[:n]
    assert rel_err(out, ref) < 1e-5
    v2 = v.copy()
    v2[:, -1] += 100.0           # only the last query row may see the last key
    out2 = oracle.gqa_core(q, k, v2)
    assert np.array_equal(out[:, :-1], out2[:, :-1]) and not np.array_equal(out[:, -1], out2[:, -1])


def test_moe_matches_dense_formula(oracle):
    r = rng(3)
    rows, H, E, k, I = 5, 32, 6, 2, 16
    x = r.standard_normal((rows, H), dtype=np.float32)
    router = r.standard_normal((H, E), dtype=np.float32)
    w_in = r.standard_normal((E, 2 * I, H), dtype=np.float32) * 0.2
    w_out = r.standard_normal((E, H, I), dtype=np.float32) * 0.2
    got = oracle.moe(x, router, w_in, w_out, k)
    want = np.zeros((rows, H))
    for i in range(rows):
        lg = x[i].astype(np.float64) @ router
        p = np.exp(lg - lg.max())
        p /= p.sum()
        top = np.argsort(-p, kind="stable")[:k]
        for e in top:
            h = w_in[e].astype(np.float64) @ x[i]
            g, u = h[:I], h[I:]
            want[i] += p[e] / p[top].sum() * (w_out[e].astype(np.float64) @ (g / (1 + np.exp(-g)) * u))
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("family", ["llama", "gpt2", "falcon", "granite_moe"])
def test_incremental_decode_equals_full_forward(oracle, pkg, family):
    """KV-cache path (generic_model.go:276-480 with posOffset) == recomputing the whole sequence."""
    cfg = pkg.synth.tiny_config(family)
    om = oracle.OracleModel(cfg, pkg.synth.make_weights(cfg, seed=2, scale=0.05))
    toks = rng(4).integers(0, cfg["vocab_size"], 14).tolist()
    full = om.forward_with_cache(toks, om.new_cache(), 0)
    kv = om.new_cache()
    first = om.forward_with_cache(toks[:9], kv, 0)
    assert rel_err(first, full[:9]) < 1e-5
    for i in range(9, 14):
        assert rel_err(om.forward_with_cache([toks[i]], kv, i)[-1], full[i]) < 2e-5
    assert len(kv) == 14


def test_oracle_reports_the_reference_panics(oracle, pkg):
    cfg = pkg.synth.tiny_config("llama", max_seq_len=16)
    om = oracle.OracleModel(cfg, pkg.synth.make_weights(cfg, seed=2))
    with pytest.raises(RuntimeError):
        om.forward_with_cache([1] * 17, om.new_cache(), 0)           # rope.go:84-86
    with pytest.raises(RuntimeError):
        om.forward_with_cache([cfg["vocab_size"]], om.new_cache(), 0)   # index out of range in embedWithOffset


def test_sample_with_history_known_answers(oracle):
    """sampling.go:33-102 on a case small enough to work out by hand (numpy float64 as the independent check)."""
    logits = np.array([2.0, 1.0, 0.0, -1.0, 0.5], np.float32)
    hist = [0, 0, 3]                              # all within the last 10: weight 3 each -> counts {0: 6, 3: 3}
    l = logits.astype(np.float64).copy()
    l[0] = l[0] / (1.2 * 6)                       # positive logit: divided (sampling.go:61)
    l[3] = l[3] * (1.2 * 3)                       # negative logit: multiplied (:63)
    l /= 0.5                                      # temperature
    p = np.exp(l - l.max()); p /= p.sum()
    idx, got = oracle.sample_with_history(logits, hist, 0.0, temperature=0.5, repetition_penalty=1.2, return_probs=True)
    assert np.allclose(got, p, atol=1e-6) and idx == 0                     # u = 0 -> first index (sort.Search)
    # the CDF walk: u just below / above a step
    cum = np.cumsum(p)
    assert oracle.sample_with_history(logits, hist, float(cum[1] - 1e-4), temperature=0.5) == 1
    assert oracle.sample_with_history(logits, hist, float(cum[1] + 1e-4), temperature=0.5) == 2
    # top-k keeps the k largest, top-p the shortest descending prefix reaching p; then renormalise (:88-96)
    order = np.argsort(-p)
    _, pk = oracle.sample_with_history(logits, hist, 0.3, temperature=0.5, top_k=2, return_probs=True)
    want = np.zeros_like(p); want[order[:2]] = p[order[:2]]; want /= want.sum()
    assert np.allclose(pk, want, atol=1e-6)
    cs = np.cumsum(p[order]); cut = int(np.argmax(cs >= 0.8)) + 1
    _, pp = oracle.sample_with_history(logits, hist, 0.3, temperature=0.5, top_p=0.8, return_probs=True)
    want = np.zeros_like(p); want[order[:cut]] = p[order[:cut]]; want /= want.sum()
    assert np.allclose(pp, want, atol=1e-6)
    # older history counts once, the last ten three times (:46-52); ids >= V are skipped (:57)
    long_hist = [1] + [4] * 10 + [99]
    _, pl = oracle.sample_with_history(logits, long_hist, 0.3, return_probs=True)
    l = logits.astype(np.float64).copy()
    l[1] /= 1.2 * 1
    l[4] /= 1.2 * (9 * 3 + 1)                     # positions 1..10 hold token 4; the last ten positions are 2..11
    q = np.exp(l - l.max()); q /= q.sum()
    assert np.allclose(pl, q, atol=1e-6)
    # top_k = 1 is greedy whatever u is; T <= 0 leaves the logits unscaled (:71)
    assert all(oracle.sample_with_history(logits, None, u, top_k=1) == 0 for u in (0.0, 0.5, 0.999))
    _, p0 = oracle.sample_with_history(logits, None, 0.5, temperature=0.0, repetition_penalty=1.0, return_probs=True)
    e = np.exp(logits.astype(np.float64) - 2.0)
    assert np.allclose(p0, e / e.sum(), atol=1e-6)


def test_test_time_knobs_do_not_change_a_bit(oracle, pkg):
    """po_set_threads (row-parallel MatMul) and last_only (LM head on the last row) are test-time accelerators of the
    oracle: every value they return is bit-identical to the single-threaded all-rows computation."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((37, 96)).astype(np.float32)
    b = rng.standard_normal((96, 130)).astype(np.float32)
    one = oracle.matmul(a, b)
    oracle.set_threads(4)
    try:
        four = oracle.matmul(a, b)
        cfg = pkg.synth.tiny_config("llama")
        w = pkg.synth.make_weights(cfg, seed=2, scale=0.05)
        om = oracle.OracleModel(cfg, w)
        toks = rng.integers(0, cfg["vocab_size"], 19).tolist()
        thr = om.forward_with_cache(toks, om.new_cache(), 0)
        last = om.forward_with_cache(toks, om.new_cache(), 0, last_only=True)
    finally:
        oracle.set_threads(1)
    ref = om.forward_with_cache(toks, om.new_cache(), 0)
    assert np.array_equal(one, four)
    assert np.array_equal(thr, ref) and last.shape == (1, cfg["vocab_size"]) and np.array_equal(last[0], ref[-1])
