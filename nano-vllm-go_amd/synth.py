"""Synthetic checkpoints for tests and bench: seeded random weights at a model's shapes, produced in
the layout the reference holds AFTER loading (generic_loader.go:353-604): 2-D weights [in, out],
gate|up concatenated, Falcon K|V combined, MoE experts [E, out, in].  Values are rounded to bf16 so
the bf16 device copy is exact (as it is for a real bf16 checkpoint, generic_loader.go:802-805).

Not part of the hot path: an input generator shared by tests/, bench.py and __graft_entry__.smoke().
"""
from __future__ import annotations

import numpy as np


def round_bf16(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16 -> fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def make_weights(cfg: dict, seed: int = 42, scale: float = 0.02, peaked_head: float = 0.0) -> dict:
    """{(slot, layer): fp32 ndarray} in reference post-load layout.  SURVEY.md §8(d): weights
    N(0, scale^2) seeded per layer (seed+layer), norm weights 1+0.02n, biases 0.02n.
    peaked_head > 0 scales an UNTIED LM head (larger logits; used by the greedy fixtures)."""
    H, L, V = cfg["hidden"], cfg["num_layers"], cfg["vocab_size"]
    nH, hd, F = cfg["num_heads"], cfg["head_dim"], cfg["ffn_dim"]
    at = cfg["attention_type"]
    nKV = nH if at == "mha" else (1 if at == "mqa" else cfg["num_kv_heads"])
    ln = cfg["norm_type"] == "layernorm"
    t: dict = {}
    g = np.random.default_rng(seed)

    def w(rng, *shape, s=scale):
        return round_bf16(rng.standard_normal(shape, dtype=np.float32) * np.float32(s))

    def nw(rng, n):
        return round_bf16(1.0 + 0.02 * rng.standard_normal(n, dtype=np.float32))

    def nb(rng, n):
        return round_bf16(0.02 * rng.standard_normal(n, dtype=np.float32))

    emb_scale = scale * (peaked_head if peaked_head > 0 else 1.0)   # LM head only (untied fixtures)
    t[("tok_emb", 0)] = w(g, V, H)
    if cfg["position_type"] == "learned":
        t[("pos_emb", 0)] = w(g, cfg["max_seq_len"], H)
    if not cfg.get("tied_embedding", False):
        t[("lm_head", 0)] = w(g, H, V, s=emb_scale)
    t[("final_norm_w", 0)] = nw(g, H)
    if ln:
        t[("final_norm_b", 0)] = nb(g, H)
    hybrid = cfg.get("hybrid_layers") or []
    for li in range(L):
        r = np.random.default_rng(seed + 1 + li)
        t[("attn_norm_w", li)] = nw(r, H)
        if ln:
            t[("attn_norm_b", li)] = nb(r, H)
        if cfg["block_style"] == "sequential":
            t[("ffn_norm_w", li)] = nw(r, H)
            if ln:
                t[("ffn_norm_b", li)] = nb(r, H)
        if li < len(hybrid) and hybrid[li] in ("mamba", "mamba2"):
            # Mamba2Layer tensors as loadMamba2 leaves them (generic_loader.go:461-512): PyTorch layouts, NOT transposed;
            # the block keeps its shared MLP (loadFFN) and both norms
            EH, nh, ss, ng, K = cfg["mamba_expand"] * H, cfg["mamba_num_heads"], cfg["mamba_state_size"], cfg["mamba_n_groups"], cfg["mamba_conv_kernel"]
            conv_dim = EH + 2 * ng * ss
            t[("mamba_in_proj", li)] = w(r, EH + conv_dim + nh, H)
            t[("mamba_conv_w", li)] = w(r, conv_dim, K, s=0.3)
            t[("mamba_conv_b", li)] = nb(r, conv_dim)
            t[("mamba_a_log", li)] = round_bf16(np.log(r.uniform(1.0, 8.0, nh)).astype(np.float32))
            t[("mamba_d", li)] = nw(r, nh)
            t[("mamba_dt_bias", li)] = round_bf16(r.uniform(-3.0, 0.5, nh).astype(np.float32))
            t[("mamba_norm", li)] = nw(r, EH)
            t[("mamba_out_proj", li)] = w(r, H, EH)
            n1 = 2 * F if cfg["activation_type"] == "swiglu" else F
            t[("w1", li)] = w(r, H, n1)
            t[("w2", li)] = w(r, F, H)
            continue
        t[("wq", li)] = w(r, H, nH * hd)
        if at == "mqa":
            t[("wkv", li)] = w(r, H, 2 * hd)
        else:
            t[("wk", li)] = w(r, H, nKV * hd)
            t[("wv", li)] = w(r, H, nKV * hd)
        t[("wo", li)] = w(r, nH * hd, H)
        if at == "mha":   # GPT-2: attention biases ARE loaded (generic_loader.go:418-441)
            for b, n in (("bq", nH * hd), ("bk", nKV * hd), ("bv", nKV * hd), ("bo", H)):
                t[(b, li)] = nb(r, n)
        if cfg.get("use_moe", False):
            E = cfg["num_experts"]
            t[("router", li)] = w(r, H, E, s=0.5)
            t[("moe_in", li)] = w(r, E, 2 * F, H)
            t[("moe_out", li)] = w(r, E, H, F)
        else:
            n1 = 2 * F if cfg["activation_type"] == "swiglu" else F
            t[("w1", li)] = w(r, H, n1)
            t[("w2", li)] = w(r, F, H)
            # GPT-2 FFN biases exist on disk but the reference never loads them (key built as
            # "...c_fc.weight.bias", generic_loader.go:559-560) -> none here either.
    return t


def tiny_config(family: str, **over) -> dict:
    """Small models of each family for parity tests (shapes legal for the kernels: hd 64, H,F % 64)."""
    base = dict(vocab_size=1000, hidden=128, num_layers=2, num_heads=2, num_kv_heads=2, head_dim=64, ffn_dim=256,
                max_seq_len=256, rope_base=10000.0, norm_eps=1e-5, tied_embedding=False, use_moe=False,
                num_experts=0, num_experts_per_tok=0, embedding_multiplier=0.0, attention_multiplier=0.0,
                residual_multiplier=0.0, logits_scaling=0.0)
    if family == "gpt2":
        base.update(attention_type="mha", norm_type="layernorm", position_type="learned", activation_type="gelu",
                    block_style="sequential", tied_embedding=True, vocab_size=1003)
    elif family == "llama":
        base.update(attention_type="gqa", norm_type="rmsnorm", position_type="rope", activation_type="swiglu",
                    block_style="sequential", num_heads=4, num_kv_heads=2, hidden=256, ffn_dim=512,
                    rope_base=500000.0, tied_embedding=True)
    elif family == "falcon":
        base.update(attention_type="mqa", norm_type="layernorm", position_type="rope", activation_type="gelu",
                    block_style="parallel", num_heads=3, num_kv_heads=1, hidden=192, ffn_dim=768)
    elif family == "granite_moe":
        base.update(attention_type="gqa", norm_type="rmsnorm", position_type="rope", activation_type="swiglu",
                    block_style="sequential", num_heads=4, num_kv_heads=2, hidden=256, ffn_dim=64, use_moe=True,
                    num_experts=8, num_experts_per_tok=2, tied_embedding=True, vocab_size=1027,
                    embedding_multiplier=12.0, attention_multiplier=0.015625, residual_multiplier=0.22,
                    logits_scaling=6.0, norm_eps=1e-6)
    elif family == "granite_hybrid":
        # Granite-4 hybrid (config.go:241-330): GQA attention WITHOUT positional encoding + Mamba2 blocks, shared SwiGLU MLP,
        # muP multipliers; tiny sizes that keep the real ratios (heads x head_dim = expand x hidden)
        base.update(attention_type="gqa", norm_type="rmsnorm", position_type="nope", activation_type="swiglu",
                    block_style="sequential", num_heads=4, num_kv_heads=2, hidden=128, ffn_dim=256, num_layers=4,
                    tied_embedding=True, vocab_size=1019, embedding_multiplier=12.0, attention_multiplier=0.015625,
                    residual_multiplier=0.22, logits_scaling=6.0, norm_eps=1e-5,
                    mamba_expand=2, mamba_state_size=32, mamba_num_heads=8, mamba_head_dim=32, mamba_n_groups=2,
                    mamba_conv_kernel=4, hybrid_layers=["mamba", "attention", "mamba", "mamba"])
    else:
        raise ValueError(family)
    base.update(over)
    return base


FULL_CONFIGS = {
    # SURVEY.md §8 config shapes C1-C5
    "gpt2": dict(vocab_size=50257, hidden=768, num_layers=12, num_heads=12, num_kv_heads=12, head_dim=64,
                 ffn_dim=3072, max_seq_len=1024, attention_type="mha", norm_type="layernorm",
                 position_type="learned", activation_type="gelu", block_style="sequential", norm_eps=1e-5,
                 tied_embedding=True),
    "llama-3.2-1b": dict(vocab_size=128256, hidden=2048, num_layers=16, num_heads=32, num_kv_heads=8, head_dim=64,
                         ffn_dim=8192, max_seq_len=4096, attention_type="gqa", norm_type="rmsnorm",
                         position_type="rope", activation_type="swiglu", block_style="sequential",
                         rope_base=500000.0, norm_eps=1e-5, tied_embedding=True),
    "falcon-7b": dict(vocab_size=65024, hidden=4544, num_layers=32, num_heads=71, num_kv_heads=1, head_dim=64,
                      ffn_dim=18176, max_seq_len=2048, attention_type="mqa", norm_type="layernorm",
                      position_type="rope", activation_type="gelu", block_style="parallel", norm_eps=1e-5,
                      tied_embedding=False),
    "granite-3.0-1b-a400m": dict(vocab_size=49155, hidden=1024, num_layers=24, num_heads=16, num_kv_heads=8,
                                 head_dim=64, ffn_dim=512, max_seq_len=4096, attention_type="gqa",
                                 norm_type="rmsnorm", position_type="rope", activation_type="swiglu",
                                 block_style="sequential", rope_base=10000.0, norm_eps=1e-6, tied_embedding=True,
                                 use_moe=True, num_experts=32, num_experts_per_tok=8, embedding_multiplier=12.0,
                                 attention_multiplier=0.015625, residual_multiplier=0.22, logits_scaling=6.0),
    "llama-3-8b": dict(vocab_size=128256, hidden=4096, num_layers=32, num_heads=32, num_kv_heads=8, head_dim=128,
                       ffn_dim=14336, max_seq_len=4096, attention_type="gqa", norm_type="rmsnorm",
                       position_type="rope", activation_type="swiglu", block_style="sequential",
                       rope_base=500000.0, norm_eps=1e-5, tied_embedding=False),
}
for _c in FULL_CONFIGS.values():
    for _k, _v in dict(use_moe=False, num_experts=0, num_experts_per_tok=0, embedding_multiplier=0.0,
                       attention_multiplier=0.0, residual_multiplier=0.0, logits_scaling=0.0,
                       rope_base=10000.0).items():
        _c.setdefault(_k, _v)
