"""Layout contract of the load path, checked on the CPU oracle.

test_split_falcon_qkv* replay the ONLY vectors the reference's own tests pin on the hot path:
purego/tensor/falcon_split_test.go:7-158 (data regenerated in tests/golden/make_golden.py)."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def test_split_falcon_qkv_reference_fixture(oracle):
    g = np.load(GOLD / "falcon_split.npz")
    nH, hd, hidden = 3, 4, 12
    q, k, v = oracle.split_falcon_qkv(g["qkv"], hidden, nH, hd)
    assert q.shape == (hidden, nH * hd) and k.shape == (hidden, hd) and v.shape == (hidden, hd)   # :57-65
    for row in range(3):                                                                         # :68-98
        for h in range(nH):
            assert q[row, h * hd] == g["q_want"][row, h]
        assert k[row, 0] == g["k_want"][row]
        assert v[row, 0] == g["v_want"][row]
    # whole-tensor form of the same statement
    for row in range(hidden):
        for h in range(nH):
            assert np.all(q[row, h * hd:(h + 1) * hd] == 10 * row + h)
        assert np.all(k[row] == 100 * row) and np.all(v[row] == 1000 * row)


def test_split_falcon_qkv_real_falcon7b_dimensions(oracle):
    g = np.load(GOLD / "falcon_split.npz")
    nH, hd, hidden = 71, 64, 4544
    qkv = np.zeros((hidden, (nH + 2) * hd), np.float32)
    qkv[0] = g["row0_7b"]
    q, k, v = oracle.split_falcon_qkv(qkv, hidden, nH, hd)
    assert q.shape == (hidden, nH * hd)                                                          # :141-143
    for h in range(nH):
        assert q[0, h * hd] == float(h)                                                          # :146-152
    assert k[0, 0] == 999.0 and v[0, 0] == 888.0                                                 # :154-160


def test_combine_mqa_kv_and_gpt2_split(oracle):
    r = np.random.default_rng(0)
    k = r.standard_normal((12, 4), dtype=np.float32)
    v = r.standard_normal((12, 4), dtype=np.float32)
    kv = oracle.combine_mqa_kv(k, v)
    assert np.array_equal(kv[:, :4], k) and np.array_equal(kv[:, 4:], v)       # generic_loader.go:751-765
    c_attn = r.standard_normal((8, 24), dtype=np.float32)
    q, kk, vv = oracle.split_gpt2_qkv(c_attn, 8)                               # columns, not rows: :674-702
    assert np.array_equal(q, c_attn[:, :8]) and np.array_equal(kk, c_attn[:, 8:16]) and np.array_equal(vv, c_attn[:, 16:])


def test_transpose_and_concat(oracle):
    a = np.arange(15, dtype=np.float32).reshape(3, 5)
    assert np.array_equal(oracle.transpose(a), a.T)
    b = -np.arange(6, dtype=np.float32).reshape(3, 2)
    assert np.array_equal(oracle.concat_last_dim(a, b), np.concatenate([a, b], axis=1))


def test_dtype_expansion(oracle):
    L = oracle.lib()
    r = np.random.default_rng(1)
    f = r.standard_normal(2000).astype(np.float32)
    bf = (f.view(np.uint32) >> 16).astype(np.uint16)
    for u in bf[:500]:
        want = np.array([int(u) << 16], np.uint32).view(np.float32)[0]
        assert L.po_f32_from_bf16(int(u)) == want                              # generic_loader.go:802-805
    h = f.astype(np.float16)
    for x in list(h[:500]) + [np.float16(6e-8), np.float16(-6e-5), np.float16(0.0), np.float16(65504)]:
        assert L.po_f32_from_f16(int(np.array([x]).view(np.uint16)[0])) == np.float32(x)   # incl. subnormals :783-790


def test_tied_lm_head_is_transposed_embedding(oracle, pkg):
    cfg = pkg.synth.tiny_config("llama")           # tied
    w = pkg.synth.make_weights(cfg, seed=1)
    assert ("lm_head", 0) not in w
    om = oracle.OracleModel(cfg, w)
    logits = om.forward_with_cache([1, 2, 3], om.new_cache(), 0)
    w2 = dict(w)
    w2[("lm_head", 0)] = w[("tok_emb", 0)].T.copy()                            # generic_loader.go:255-259
    cfg2 = dict(cfg, tied_embedding=False)
    om2 = oracle.OracleModel(cfg2, w2)
    assert np.array_equal(logits, om2.forward_with_cache([1, 2, 3], om2.new_cache(), 0))
    assert json.dumps(cfg)  # configs are plain data
