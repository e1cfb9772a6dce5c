"""Test helper: turn a HuggingFace state_dict into the tensors nano-vllm-go would hold after
LoadModel (purego/tensor/generic_loader.go:184-265, weight mappings :60-181), using the ORACLE's
restatements of the loader's layout functions (Transpose, splitGPT2QKV, splitFalconQKV, combineMQAKV,
ConcatenateLastDim) so the layout contract is exercised end to end."""
import numpy as np


def _np(t):
    return t.detach().to("cpu").float().numpy()


def convert(family, sd, cfg, O):
    """sd: HF state_dict; cfg: our config dict; O: oracle module.  Returns {(slot, layer): ndarray}."""
    t = {}
    L, H = cfg["num_layers"], cfg["hidden"]
    if family == "llama" or family == "granite_moe":
        t[("tok_emb", 0)] = _np(sd["model.embed_tokens.weight"])
        t[("final_norm_w", 0)] = _np(sd["model.norm.weight"])
        if not cfg["tied_embedding"]:
            t[("lm_head", 0)] = O.transpose(_np(sd["lm_head.weight"]))           # :246-251
        for i in range(L):
            p = f"model.layers.{i}"
            t[("attn_norm_w", i)] = _np(sd[p + ".input_layernorm.weight"])
            t[("ffn_norm_w", i)] = _np(sd[p + ".post_attention_layernorm.weight"])
            for slot, key in (("wq", "q_proj"), ("wk", "k_proj"), ("wv", "v_proj"), ("wo", "o_proj")):
                t[(slot, i)] = O.transpose(_np(sd[f"{p}.self_attn.{key}.weight"]))  # :398-403
            if family == "llama":
                gate = O.transpose(_np(sd[p + ".mlp.gate_proj.weight"]))
                up = O.transpose(_np(sd[p + ".mlp.up_proj.weight"]))
                t[("w1", i)] = O.concat_last_dim(gate, up)                        # :531-536
                t[("w2", i)] = O.transpose(_np(sd[p + ".mlp.down_proj.weight"]))
            else:
                # checkpoint names the reference maps (generic_loader.go:134-141); transformers >= 5 renamed the
                # parameters (router.weight / experts.gate_up_proj / experts.down_proj), same shapes and meaning
                def pick(old, new):
                    return _np(sd[p + old] if (p + old) in sd else sd[p + new])
                t[("router", i)] = O.transpose(pick(".block_sparse_moe.router.layer.weight",
                                                    ".block_sparse_moe.router.weight"))                 # :573-575
                t[("moe_in", i)] = pick(".block_sparse_moe.input_linear.weight",
                                        ".block_sparse_moe.experts.gate_up_proj")                        # as-is :578
                t[("moe_out", i)] = pick(".block_sparse_moe.output_linear.weight",
                                         ".block_sparse_moe.experts.down_proj")
    elif family == "granite_hybrid":
        # GetGraniteMapping (generic_loader.go:145-181) + loadMamba2 (:461-512, Mamba2 tensors kept in the checkpoint's
        # layouts) + the shared MLP every hybrid block has (shared_mlp.input_linear = gate | up fused)
        t[("tok_emb", 0)] = _np(sd["model.embed_tokens.weight"])
        t[("final_norm_w", 0)] = _np(sd["model.norm.weight"])
        for i in range(L):
            p = f"model.layers.{i}"
            t[("attn_norm_w", i)] = _np(sd[p + ".input_layernorm.weight"])
            t[("ffn_norm_w", i)] = _np(sd[p + ".post_attention_layernorm.weight"])
            if cfg["hybrid_layers"][i] in ("mamba", "mamba2"):
                t[("mamba_in_proj", i)] = _np(sd[p + ".mamba.in_proj.weight"])
                t[("mamba_conv_w", i)] = _np(sd[p + ".mamba.conv1d.weight"]).reshape(-1, cfg["mamba_conv_kernel"])
                t[("mamba_conv_b", i)] = _np(sd[p + ".mamba.conv1d.bias"])
                t[("mamba_a_log", i)] = _np(sd[p + ".mamba.A_log"])
                t[("mamba_d", i)] = _np(sd[p + ".mamba.D"])
                t[("mamba_dt_bias", i)] = _np(sd[p + ".mamba.dt_bias"])
                t[("mamba_norm", i)] = _np(sd[p + ".mamba.norm.weight"])
                t[("mamba_out_proj", i)] = _np(sd[p + ".mamba.out_proj.weight"])
            else:
                for slot, key in (("wq", "q_proj"), ("wk", "k_proj"), ("wv", "v_proj"), ("wo", "o_proj")):
                    t[(slot, i)] = O.transpose(_np(sd[f"{p}.self_attn.{key}.weight"]))
            t[("w1", i)] = O.transpose(_np(sd[p + ".shared_mlp.input_linear.weight"]))     # [2F, H] -> [H, gate | up]
            t[("w2", i)] = O.transpose(_np(sd[p + ".shared_mlp.output_linear.weight"]))
    elif family == "gpt2":
        t[("tok_emb", 0)] = _np(sd["transformer.wte.weight"])
        t[("pos_emb", 0)] = _np(sd["transformer.wpe.weight"])
        t[("final_norm_w", 0)] = _np(sd["transformer.ln_f.weight"])
        t[("final_norm_b", 0)] = _np(sd["transformer.ln_f.bias"])
        for i in range(L):
            p = f"transformer.h.{i}"
            t[("attn_norm_w", i)] = _np(sd[p + ".ln_1.weight"])
            t[("attn_norm_b", i)] = _np(sd[p + ".ln_1.bias"])
            t[("ffn_norm_w", i)] = _np(sd[p + ".ln_2.weight"])
            t[("ffn_norm_b", i)] = _np(sd[p + ".ln_2.bias"])
            q, k, v = O.split_gpt2_qkv(_np(sd[p + ".attn.c_attn.weight"]), H)      # Conv1D is already [in,out]
            t[("wq", i)], t[("wk", i)], t[("wv", i)] = q, k, v
            b = _np(sd[p + ".attn.c_attn.bias"])                                   # :418-427
            t[("bq", i)], t[("bk", i)], t[("bv", i)] = b[:H].copy(), b[H:2 * H].copy(), b[2 * H:].copy()
            t[("wo", i)] = _np(sd[p + ".attn.c_proj.weight"])
            t[("bo", i)] = _np(sd[p + ".attn.c_proj.bias"])
            t[("w1", i)] = _np(sd[p + ".mlp.c_fc.weight"])
            t[("w2", i)] = _np(sd[p + ".mlp.c_proj.weight"])
            # mlp biases: never loaded by the reference (key "...c_fc.weight.bias", :559-560)
    elif family == "falcon":
        nH, hd = cfg["num_heads"], cfg["head_dim"]
        t[("tok_emb", 0)] = _np(sd["transformer.word_embeddings.weight"])
        t[("final_norm_w", 0)] = _np(sd["transformer.ln_f.weight"])
        t[("final_norm_b", 0)] = _np(sd["transformer.ln_f.bias"])
        t[("lm_head", 0)] = O.transpose(_np(sd["lm_head.weight"]))
        for i in range(L):
            p = f"transformer.h.{i}"
            t[("attn_norm_w", i)] = _np(sd[p + ".input_layernorm.weight"])
            t[("attn_norm_b", i)] = _np(sd[p + ".input_layernorm.bias"])
            qkv = O.transpose(_np(sd[p + ".self_attention.query_key_value.weight"]))   # transpose BEFORE split :366-368
            q, k, v = O.split_falcon_qkv(qkv, H, nH, hd)
            t[("wq", i)] = q
            t[("wkv", i)] = O.combine_mqa_kv(k, v)
            t[("wo", i)] = O.transpose(_np(sd[p + ".self_attention.dense.weight"]))
            t[("w1", i)] = O.transpose(_np(sd[p + ".mlp.dense_h_to_4h.weight"]))
            t[("w2", i)] = O.transpose(_np(sd[p + ".mlp.dense_4h_to_h.weight"]))
    else:
        raise ValueError(family)
    return t
