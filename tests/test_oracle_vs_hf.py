"""The oracle is a restatement of Go code that cannot be run here (no Go toolchain), so it is
cross-checked against an independent implementation of the same architectures: HuggingFace's
modelling code, instantiated from LOCAL config objects with random weights (nothing is fetched).
The HF weights go through the reference's loader transforms (tests/hf_to_reference.py).

Reference behaviours that differ from HF and are patched on the HF side, not in the oracle:
GPT-2's MLP biases are zeroed (the reference never loads them, generic_loader.go:559-560); Falcon's
activation is set to tanh-GELU (the reference uses tensor.go:181-190 for every family)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from hf_to_reference import convert  # noqa: E402

TOL = 2e-4   # fp32 CPU vs fp32 CPU, different summation orders


def _run(family, pkg, oracle):
    torch.manual_seed(0)
    cfg = pkg.synth.tiny_config(family)
    if family == "granite_hybrid":      # HF derives head_dim = hidden / heads: use a geometry where the two agree (4 x 64 = 256)
        cfg = pkg.synth.tiny_config(family, hidden=256, ffn_dim=512, mamba_head_dim=64)
    if family == "llama":
        hc = transformers.LlamaConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"],
                                      intermediate_size=cfg["ffn_dim"], num_hidden_layers=cfg["num_layers"],
                                      num_attention_heads=cfg["num_heads"], num_key_value_heads=cfg["num_kv_heads"],
                                      max_position_embeddings=cfg["max_seq_len"], rope_theta=cfg["rope_base"],
                                      rms_norm_eps=cfg["norm_eps"], tie_word_embeddings=cfg["tied_embedding"])
        m = transformers.LlamaForCausalLM(hc)
    elif family == "gpt2":
        hc = transformers.GPT2Config(vocab_size=cfg["vocab_size"], n_embd=cfg["hidden"], n_layer=cfg["num_layers"],
                                     n_head=cfg["num_heads"], n_positions=cfg["max_seq_len"], n_inner=cfg["ffn_dim"],
                                     layer_norm_epsilon=cfg["norm_eps"], activation_function="gelu_new",
                                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
        m = transformers.GPT2LMHeadModel(hc)
        with torch.no_grad():
            for blk in m.transformer.h:
                blk.mlp.c_fc.bias.zero_()
                blk.mlp.c_proj.bias.zero_()
                blk.attn.c_attn.bias.normal_(0, 0.02)
                blk.attn.c_proj.bias.normal_(0, 0.02)
    elif family == "falcon":
        hc = transformers.FalconConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"],
                                       num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"],
                                       multi_query=True, parallel_attn=True, new_decoder_architecture=False,
                                       bias=False, alibi=False, layer_norm_epsilon=cfg["norm_eps"],
                                       ffn_hidden_size=cfg["ffn_dim"], activation="gelu_pytorch_tanh",
                                       max_position_embeddings=cfg["max_seq_len"], rope_theta=10000.0,
                                       hidden_dropout=0.0, attention_dropout=0.0, tie_word_embeddings=False)
        m = transformers.FalconForCausalLM(hc)
    elif family == "granite_hybrid":
        # Granite-4 hybrid: Mamba2 + attention (no positional encoding) + shared MLP, no experts
        hc = transformers.GraniteMoeHybridConfig(
            vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"], intermediate_size=cfg["ffn_dim"],
            shared_intermediate_size=cfg["ffn_dim"], num_hidden_layers=cfg["num_layers"],
            num_attention_heads=cfg["num_heads"], num_key_value_heads=cfg["num_kv_heads"],
            max_position_embeddings=cfg["max_seq_len"], rms_norm_eps=cfg["norm_eps"], tie_word_embeddings=True,
            num_local_experts=0, num_experts_per_tok=0, embedding_multiplier=cfg["embedding_multiplier"],
            attention_multiplier=cfg["attention_multiplier"], residual_multiplier=cfg["residual_multiplier"],
            logits_scaling=cfg["logits_scaling"], attention_dropout=0.0, position_embedding_type="nope",
            layer_types=["mamba" if t == "mamba" else "attention" for t in cfg["hybrid_layers"]],
            mamba_n_heads=cfg["mamba_num_heads"], mamba_n_groups=cfg["mamba_n_groups"], mamba_d_state=cfg["mamba_state_size"],
            mamba_d_head=cfg["mamba_head_dim"], mamba_d_conv=cfg["mamba_conv_kernel"], mamba_expand=cfg["mamba_expand"],
            mamba_chunk_size=8, mamba_conv_bias=True, mamba_proj_bias=False)
        m = transformers.GraniteMoeHybridForCausalLM(hc)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith("A_log"):
                    p.copy_(torch.log(torch.empty_like(p).uniform_(1.0, 8.0)))
                elif n.endswith("dt_bias"):
                    p.uniform_(-3.0, 0.5)
                elif n.endswith(".mamba.D"):
                    p.normal_(1.0, 0.1)
                elif n.endswith("conv1d.bias"):
                    p.normal_(0.0, 0.05)
    else:
        hc = transformers.GraniteMoeConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"],
                                           intermediate_size=cfg["ffn_dim"], num_hidden_layers=cfg["num_layers"],
                                           num_attention_heads=cfg["num_heads"], num_key_value_heads=cfg["num_kv_heads"],
                                           max_position_embeddings=cfg["max_seq_len"], rope_theta=cfg["rope_base"],
                                           rms_norm_eps=cfg["norm_eps"], tie_word_embeddings=True,
                                           num_local_experts=cfg["num_experts"],
                                           num_experts_per_tok=cfg["num_experts_per_tok"],
                                           embedding_multiplier=cfg["embedding_multiplier"],
                                           attention_multiplier=cfg["attention_multiplier"],
                                           residual_multiplier=cfg["residual_multiplier"],
                                           logits_scaling=cfg["logits_scaling"], attention_dropout=0.0)
        m = transformers.GraniteMoeForCausalLM(hc)
    m = m.float().eval()
    with torch.no_grad():   # HF inits norms to exactly 1 / biases to 0: perturb so they are exercised
        for n, p in m.named_parameters():
            if "norm" in n or "ln_" in n:
                p.add_(0.05 * torch.randn_like(p))
            elif "router" in n:
                p.mul_(8.0)     # decisive routing: top-k far from ties
    tensors = convert(family, m.state_dict(), cfg, oracle)
    om = oracle.OracleModel(cfg, tensors)
    toks = np.random.default_rng(1).integers(0, cfg["vocab_size"], 23)
    with torch.no_grad():
        want = m(torch.tensor(toks[None].astype(np.int64))).logits[0].numpy()
    got = om.forward_with_cache(toks.tolist(), om.new_cache(), 0)
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err <= TOL, (family, err)
    if family == "granite_hybrid":
        # full-sequence prefill only: the reference carries no convolution state across calls (mamba2.go: ConvCache unused),
        # so its token-by-token decode deliberately differs from HF's cached conv; that behaviour is mirrored, not "fixed"
        return
    # incremental decode through the KV cache reproduces the full-sequence logits
    kv = om.new_cache()
    om.forward_with_cache(toks[:15].tolist(), kv, 0)
    for i in range(15, 23):
        row = om.forward_with_cache([int(toks[i])], kv, i)[-1]
        assert np.abs(row - want[i]).max() / np.abs(want).max() <= TOL, (family, i)


@pytest.mark.parametrize("family", ["llama", "gpt2", "falcon", "granite_moe", "granite_hybrid"])
def test_oracle_matches_huggingface(family, pkg, oracle):
    _run(family, pkg, oracle)
