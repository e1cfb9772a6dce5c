"""Host-side mirrors of the reference's plain-data contracts: ModelConfig templates and LoadModelConfig
(purego/tensor/config.go, generic_loader.go:808-1007), Sequence, the DP sharding helper."""
import numpy as np


def test_templates_match_the_reference(pkg):
    c = pkg.config
    g = c.new_gpt2_config()
    assert (g.vocab_size, g.hidden, g.num_layers, g.num_heads, g.head_dim, g.ffn_dim, g.max_seq_len) == \
        (50257, 768, 12, 12, 64, 3072, 1024) and g.tied_embedding and g.attention_type == "mha"      # config.go:125-148
    f = c.new_falcon_config("7b")
    assert (f.hidden, f.num_heads, f.num_kv_heads, f.ffn_dim, f.block_style) == (4544, 71, 1, 18176, "parallel")
    ll = c.new_llama_config("7b")
    assert ll.norm_eps == 1e-6 and ll.max_seq_len == 4096 and ll.rope_base == 10000.0             # config.go:197-224
    m = c.new_granite_moe_config()
    assert not m.use_moe and m.num_experts == 32 and m.num_experts_per_tok == 8                   # config.go:347-349


def test_load_model_config_llama32_1b_with_reference_quirks(pkg):
    hf = {"model_type": "llama", "vocab_size": 128256, "hidden_size": 2048, "num_hidden_layers": 16,
          "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 64, "intermediate_size": 8192,
          "rope_theta": 500000.0, "rms_norm_eps": 1e-5, "tie_word_embeddings": True,
          "max_position_embeddings": 131072, "eos_token_id": [128001, 128008, 128009], "bos_token_id": 128000,
          "rope_scaling": {"factor": 32.0, "rope_type": "llama3"}}
    c = pkg.config.load_model_config(hf)
    assert (c.hidden, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.ffn_dim) == (2048, 16, 32, 8, 64, 8192)
    assert c.rope_base == 500000.0 and c.norm_eps == 1e-5 and c.tied_embedding
    assert c.max_seq_len == 4096          # max_position_embeddings is never read
    assert c.eos_token_id == 2            # a JSON list is not a float64: the template value survives (:881-883)
    assert c.bos_token_id == 128000


def test_load_model_config_other_families(pkg):
    c = pkg.config.load_model_config({"model_type": "gpt2", "n_embd": 768, "n_layer": 12, "n_head": 12, "n_inner": None,
                                      "vocab_size": 50257, "layer_norm_epsilon": 1e-5})
    assert c.ffn_dim == 3072 and c.head_dim == 64
    c = pkg.config.load_model_config({"model_type": "falcon", "hidden_size": 4544, "num_hidden_layers": 32,
                                      "num_attention_heads": 71, "multi_query": True, "vocab_size": 65024})
    assert c.num_kv_heads == 1 and c.head_dim == 64 and c.ffn_dim == 18176
    c = pkg.config.load_model_config({"model_type": "granitemoe", "hidden_size": 1024, "num_hidden_layers": 24,
                                      "num_attention_heads": 16, "num_key_value_heads": 8, "intermediate_size": 512,
                                      "num_local_experts": 32, "num_experts_per_tok": 8, "embedding_multiplier": 12.0,
                                      "attention_multiplier": 0.015625, "residual_multiplier": 0.22,
                                      "logits_scaling": 6.0, "vocab_size": 49155, "tie_word_embeddings": True})
    assert c.use_moe and c.embedding_multiplier == 12.0 and c.logits_scaling == 6.0 and c.head_dim == 64
    assert pkg.config.load_model_config({}).architecture == "gpt2"      # default: GPT-2 style (:1005-1006)


def test_synthetic_weights_are_bf16_exact_and_in_reference_layout(pkg):
    cfg = pkg.synth.tiny_config("llama")
    w = pkg.synth.make_weights(cfg, seed=3)
    assert w[("wq", 0)].shape == (cfg["hidden"], cfg["num_heads"] * cfg["head_dim"])         # [in, out]
    assert w[("w1", 1)].shape == (cfg["hidden"], 2 * cfg["ffn_dim"])                          # gate|up
    for a in w.values():
        assert np.array_equal(a, pkg.synth.round_bf16(a))
    m = pkg.synth.make_weights(pkg.synth.tiny_config("granite_moe"), seed=3)
    assert m[("moe_in", 0)].shape == (8, 128, 256) and m[("moe_out", 0)].shape == (8, 256, 64)   # [E, out, in]
    assert pkg.synth.FULL_CONFIGS["llama-3.2-1b"]["vocab_size"] == 128256


def test_sequence_and_sharding(pkg):
    s = pkg.Sequence(seq_id=7, token_ids=[1, 2, 3])
    s.append_token(9)
    assert len(s) == 4 and s.token_ids[-1] == 9
    ids = [3, 8, 5, 12, 7, 0]
    parts = [pkg.dist.shard(ids, r, 4) for r in range(4)]
    assert sorted(i for p in parts for i in p) == list(range(len(ids)))       # a partition
    assert all(pkg.dist.owner(ids[i], 4) == r for r, p in enumerate(parts) for i in p)
    assert pkg.dist.shard(ids, 0, 1) == list(range(6))


def test_native_config_loader_matches_python_mirror(pkg, tmp_path):
    """nvl_load_config_json (C++: LoadModelConfig generic_loader.go:808-972 over the config.go templates) against the
    Python mirror of the same function, on HF-style configs of every family incl. the reference's quirks."""
    import ctypes as C
    import json
    L = pkg._lib
    cases = [
        dict(model_type="llama", vocab_size=128256, hidden_size=2048, num_hidden_layers=16, num_attention_heads=32,
             num_key_value_heads=8, head_dim=64, intermediate_size=8192, rope_theta=500000.0, rms_norm_eps=1e-5,
             tie_word_embeddings=True, max_position_embeddings=131072, rope_scaling={"factor": 32.0}),
        dict(model_type="gpt2", vocab_size=50257, n_embd=1024, n_layer=24, n_head=16, n_inner=None, layer_norm_epsilon=1e-5),
        dict(model_type="RefinedWebModel", vocab_size=65024, hidden_size=4544, n_layer=32, n_head=71, multi_query=True,
             layer_norm_epsilon=1e-5),
        dict(model_type="granitemoe", vocab_size=49155, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
             num_key_value_heads=8, intermediate_size=512, num_local_experts=32, num_experts_per_tok=8,
             embedding_multiplier=12.0, attention_multiplier=0.015625, residual_multiplier=0.22, logits_scaling=6.0,
             tie_word_embeddings=True, rms_norm_eps=1e-6, rope_theta=10000),
        dict(architecture="llama", hidden_size=512, num_heads=8, num_kv_heads=2, num_layers=4),
        dict(some_unknown_thing=1),                                  # default: GPT-2 template (:1005)
        dict(model_type="granitemoehybrid", vocab_size=100352, hidden_size=768, num_hidden_layers=6, num_attention_heads=12,
             num_key_value_heads=4, intermediate_size=2048, num_local_experts=0, mamba_n_heads=48, mamba_d_head=32,
             mamba_d_state=128, mamba_n_groups=1, mamba_d_conv=4, mamba_expand=2, embedding_multiplier=12.0,
             layer_types=["mamba", "mamba", "attention", "mamba", "attention", "mamba"], tie_word_embeddings=True),
        dict(model_type="granitemoehybrid", hidden_size=1536),       # the "1b" template, no layer_types: all Mamba2
    ]
    enum = dict(attention_type=["mha", "mqa", "gqa"], norm_type=["layernorm", "rmsnorm"],
                position_type=["learned", "rope", "nope"], activation_type=["gelu", "swiglu"],
                block_style=["sequential", "parallel"])
    for i, raw in enumerate(cases):
        p = tmp_path / f"c{i}.json"
        p.write_text(json.dumps(raw) + "\n")
        got = L.ModelConfigC()
        L.check(L.lib().nvl_load_config_json(str(p).encode(), C.byref(got)))
        want = pkg.config.load_model_config(raw)
        for name, _ in L.ModelConfigC._fields_:
            if name.startswith("_"):
                continue
            g, w = getattr(got, name), getattr(want, name)
            if name == "mamba_layer_mask":
                assert [int(g[0]), int(g[1])] == w, (i, name, list(g), w)
                continue
            if name in enum:
                assert enum[name][g] == w, (i, name)
            elif isinstance(w, bool):
                assert bool(g) == w, (i, name)
            elif isinstance(w, float):
                assert abs(g - w) <= 1e-6 * max(1.0, abs(w)), (i, name, g, w)
            else:
                assert g == w, (i, name, g, w)
    bad = tmp_path / "bad.json"
    bad.write_text("{\"a\": [1, 2,")
    assert L.lib().nvl_load_config_json(str(bad).encode(), C.byref(L.ModelConfigC())) < 0
    assert L.lib().nvl_load_config_json(str(tmp_path / "missing.json").encode(), C.byref(L.ModelConfigC())) < 0


def test_block_manager_mirror_meets_the_references_own_test_expectations(pkg):
    """tests/block_manager_mirror.py (the restatement that produces the block tables of the paged-KV GPU tests) against
    the scenarios and expected values of the reference's own nanovllm/block_manager_test.go:7-129 and
    sequence_test.go:53-78 (block arithmetic)."""
    from block_manager_mirror import BlockManager
    Sequence = pkg.Sequence
    bm = BlockManager(100, 256)                                   # TestBlockManagerCreation
    assert len(bm.blocks) == 100 and len(bm.free) == 100 and bm.block_size == 256
    seq = Sequence(seq_id=0, token_ids=list(range(300)))          # TestBlockManagerAllocate: 300 tokens -> 2 blocks
    assert bm.can_allocate(seq)
    bm.allocate(seq)
    assert len(seq.block_table) == 2 and len(bm.free) == 98
    bm.deallocate(seq)                                            # TestBlockManagerDeallocate
    assert seq.block_table == [] and len(bm.free) == 100 and seq.num_cached_tokens == 0
    bm = BlockManager(100, 256)                                   # TestBlockManagerPrefixCaching: same 256 tokens twice
    s1, s2 = Sequence(seq_id=1, token_ids=list(range(256))), Sequence(seq_id=2, token_ids=list(range(256)))
    bm.allocate(s1)
    free_first = len(bm.free)
    bm.allocate(s2)
    assert s2.num_cached_tokens == 256 and s2.block_table == s1.block_table and len(bm.free) == free_first
    assert bm.blocks[s1.block_table[0]].ref_count == 2
    h1, h2 = bm.compute_hash([1, 2, 3, 4, 5], 0), bm.compute_hash([1, 2, 3, 4, 5], 0)   # TestBlockManagerComputeHash
    assert h1 == h2 and h1 != bm.compute_hash([1, 2, 3, 4, 6], 0)
    assert h1 != bm.compute_hash([1, 2, 3, 4, 5], 77)             # the prefix hash chains in (block_manager.go:74-78)
    seq = Sequence(seq_id=3, token_ids=list(range(600)))          # TestSequenceBlocks: 600 tokens -> 3 blocks, last of 88
    assert bm.num_blocks(seq) == 3 and 600 - 2 * 256 == 88
    # MayAppend (block_manager.go:231-263): a new block exactly when the appended token is the first of a block
    bm = BlockManager(8, 256)
    seq = Sequence(seq_id=4, token_ids=list(range(255)))
    bm.allocate(seq)
    seq.append_token(1); bm.may_append(seq)                       # 256 tokens: block full -> hashed, no new block
    assert len(seq.block_table) == 1 and bm.blocks[seq.block_table[0]].hash != 0
    seq.append_token(2); bm.may_append(seq)                       # 257 tokens: second block
    assert len(seq.block_table) == 2
