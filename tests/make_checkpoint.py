"""Test infrastructure: write synthetic weights (reference post-load layout, synth.make_weights) as the HuggingFace
checkpoint the reference's loader would read — the inverse of generic_loader.go's WeightMapping tables (:60-181),
its transposes (:398-403, :533-552) and its fused-projection splits (:674-748).  No reference file is involved."""
import json
import os

import numpy as np
import torch
from safetensors.torch import save_file


def to_hf_names(cfg: dict, t: dict, family: str) -> dict:
    """{(slot, layer): array} -> {hf_name: float32 array in the checkpoint's layout}."""
    L, F = cfg["num_layers"], cfg["ffn_dim"]
    out = {}
    T = lambda a: np.ascontiguousarray(a.T)
    if family == "gpt2":                         # weights stay [in, out] (Conv1D), c_attn fused by columns
        out["wte.weight"] = t[("tok_emb", 0)]
        out["wpe.weight"] = t[("pos_emb", 0)]
        out["ln_f.weight"], out["ln_f.bias"] = t[("final_norm_w", 0)], t[("final_norm_b", 0)]
        for l in range(L):
            p = f"h.{l}"
            out[p + ".attn.c_attn.weight"] = np.concatenate([t[("wq", l)], t[("wk", l)], t[("wv", l)]], axis=1)
            out[p + ".attn.c_attn.bias"] = np.concatenate([t[("bq", l)], t[("bk", l)], t[("bv", l)]])
            out[p + ".attn.c_proj.weight"], out[p + ".attn.c_proj.bias"] = t[("wo", l)], t[("bo", l)]
            out[p + ".mlp.c_fc.weight"], out[p + ".mlp.c_proj.weight"] = t[("w1", l)], t[("w2", l)]
            out[p + ".mlp.c_fc.bias"] = np.full(F, 0.5, np.float32)          # on disk, never loaded (:559-560)
            out[p + ".mlp.c_proj.bias"] = np.full(cfg["hidden"], 0.5, np.float32)
            out[p + ".ln_1.weight"], out[p + ".ln_1.bias"] = t[("attn_norm_w", l)], t[("attn_norm_b", l)]
            out[p + ".ln_2.weight"], out[p + ".ln_2.bias"] = t[("ffn_norm_w", l)], t[("ffn_norm_b", l)]
    elif family == "falcon":                     # PyTorch [out, in]; query_key_value = [Q heads | K | V] rows
        out["transformer.word_embeddings.weight"] = t[("tok_emb", 0)]
        out["lm_head.weight"] = T(t[("lm_head", 0)])
        out["transformer.ln_f.weight"], out["transformer.ln_f.bias"] = t[("final_norm_w", 0)], t[("final_norm_b", 0)]
        for l in range(L):
            p = f"transformer.h.{l}"
            out[p + ".self_attention.query_key_value.weight"] = T(np.concatenate([t[("wq", l)], t[("wkv", l)]], axis=1))
            out[p + ".self_attention.dense.weight"] = T(t[("wo", l)])
            out[p + ".mlp.dense_h_to_4h.weight"] = T(t[("w1", l)])
            out[p + ".mlp.dense_4h_to_h.weight"] = T(t[("w2", l)])
            out[p + ".input_layernorm.weight"], out[p + ".input_layernorm.bias"] = t[("attn_norm_w", l)], t[("attn_norm_b", l)]
    else:                                        # llama / granite_moe
        out["model.embed_tokens.weight"] = t[("tok_emb", 0)]
        if ("lm_head", 0) in t:
            out["lm_head.weight"] = T(t[("lm_head", 0)])
        out["model.norm.weight"] = t[("final_norm_w", 0)]
        for l in range(L):
            p = f"model.layers.{l}"
            for slot, name in (("wq", "q_proj"), ("wk", "k_proj"), ("wv", "v_proj"), ("wo", "o_proj")):
                out[f"{p}.self_attn.{name}.weight"] = T(t[(slot, l)])
            if cfg.get("use_moe"):
                out[p + ".block_sparse_moe.router.layer.weight"] = T(t[("router", l)])
                out[p + ".block_sparse_moe.input_linear.weight"] = t[("moe_in", l)]
                out[p + ".block_sparse_moe.output_linear.weight"] = t[("moe_out", l)]
            else:
                w1 = t[("w1", l)]
                out[p + ".mlp.gate_proj.weight"], out[p + ".mlp.up_proj.weight"] = T(w1[:, :F]), T(w1[:, F:])
                out[p + ".mlp.down_proj.weight"] = T(t[("w2", l)])
            out[p + ".input_layernorm.weight"] = t[("attn_norm_w", l)]
            out[p + ".post_attention_layernorm.weight"] = t[("ffn_norm_w", l)]
    return out


HF_CONFIG = {
    "llama": lambda c: dict(model_type="llama", vocab_size=c["vocab_size"], hidden_size=c["hidden"],
                            num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"],
                            num_key_value_heads=c["num_kv_heads"], head_dim=c["head_dim"], intermediate_size=c["ffn_dim"],
                            rope_theta=c["rope_base"], rms_norm_eps=c["norm_eps"], tie_word_embeddings=c["tied_embedding"],
                            max_position_embeddings=131072, rope_scaling={"factor": 32.0, "rope_type": "llama3"}),
    "gpt2": lambda c: dict(model_type="gpt2", vocab_size=c["vocab_size"], n_embd=c["hidden"], n_layer=c["num_layers"],
                           n_head=c["num_heads"], n_inner=None, layer_norm_epsilon=c["norm_eps"]),
    "falcon": lambda c: dict(model_type="falcon", vocab_size=c["vocab_size"], hidden_size=c["hidden"],
                             num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"], multi_query=True,
                             layer_norm_epsilon=c["norm_eps"], parallel_attn=True, bias=False),
    "granite_moe": lambda c: dict(model_type="granitemoe", vocab_size=c["vocab_size"], hidden_size=c["hidden"],
                                  num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"],
                                  num_key_value_heads=c["num_kv_heads"], intermediate_size=c["ffn_dim"],
                                  rope_theta=c["rope_base"], rms_norm_eps=c["norm_eps"], tie_word_embeddings=c["tied_embedding"],
                                  num_local_experts=c["num_experts"], num_experts_per_tok=c["num_experts_per_tok"],
                                  embedding_multiplier=c["embedding_multiplier"], attention_multiplier=c["attention_multiplier"],
                                  residual_multiplier=c["residual_multiplier"], logits_scaling=c["logits_scaling"]),
}


def write_checkpoint(dirname, cfg, tensors, family, dtype="bf16", shards=1, prefix=""):
    """-> directory with config.json + model.safetensors (or an index + `shards` files).  `prefix` prepends e.g.
    "transformer." to every name (newer GPT-2 exports; generic_loader.go:622-629 strips it by retrying)."""
    os.makedirs(dirname, exist_ok=True)
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    named = {prefix + k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(td).contiguous()
             for k, v in to_hf_names(cfg, tensors, family).items()}
    with open(os.path.join(dirname, "config.json"), "w") as f:
        json.dump(HF_CONFIG[family](cfg), f)
    if shards == 1:
        save_file(named, os.path.join(dirname, "model.safetensors"), metadata={"format": "pt"})
        return dirname
    keys = sorted(named)
    weight_map = {}
    for s in range(shards):
        part = {k: named[k] for k in keys[s::shards]}
        fn = f"model-{s + 1:05d}-of-{shards:05d}.safetensors"
        save_file(part, os.path.join(dirname, fn), metadata={"format": "pt"})
        weight_map.update({k: fn for k in part})
    with open(os.path.join(dirname, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": 0}, "weight_map": weight_map}, f)
    return dirname
