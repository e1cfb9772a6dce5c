"""Mirror of purego/tensor/config.go: ModelConfig (config.go:58-122), the per-architecture
templates (config.go:125-376) and LoadModelConfig (generic_loader.go:808-1007) — the plain-data
input contract of the forward path.  Field names are the Go names in snake_case.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field


@dataclass
class ModelConfig:
    architecture: str = "gpt2"
    model_name: str = ""
    vocab_size: int = 0
    hidden: int = 0
    num_layers: int = 0
    num_heads: int = 0
    num_kv_heads: int = 0
    head_dim: int = 0
    ffn_dim: int = 0
    max_seq_len: int = 0
    attention_type: str = "mha"       # mha | mqa | gqa            config.go:18-22
    norm_type: str = "layernorm"      # layernorm | rmsnorm        config.go:27-30
    position_type: str = "learned"    # learned | rope | nope      config.go:35-40
    activation_type: str = "gelu"     # gelu | swiglu              config.go:45-48
    block_style: str = "sequential"   # sequential | parallel      config.go:53-56
    eos_token_id: int = 0
    bos_token_id: int = 0
    pad_token_id: int = 0
    rope_base: float = 10000.0
    norm_eps: float = 1e-5
    tied_embedding: bool = False
    use_moe: bool = False
    num_experts: int = 0
    num_experts_per_tok: int = 0
    embedding_multiplier: float = 0.0
    attention_multiplier: float = 0.0
    residual_multiplier: float = 0.0
    logits_scaling: float = 0.0
    hybrid_layers: list = field(default_factory=list)     # config.go:113: "attention" | "mamba" | "mamba2" per layer
    mamba_expand: int = 0             # config.go:100-106
    mamba_state_size: int = 0
    mamba_num_heads: int = 0
    mamba_head_dim: int = 0
    mamba_n_groups: int = 0
    mamba_conv_kernel: int = 0

    @property
    def mamba_layer_mask(self):
        m = [0, 0]
        for i, t in enumerate(self.hybrid_layers[:128]):
            if t in ("mamba", "mamba2"):
                m[i >> 6] |= 1 << (i & 63)
        return m

    def to_dict(self) -> dict:
        return asdict(self)


def new_gpt2_config() -> ModelConfig:           # config.go:125-148
    return ModelConfig(architecture="gpt2", model_name="gpt2", vocab_size=50257, hidden=768, num_layers=12,
                       num_heads=12, num_kv_heads=12, head_dim=64, ffn_dim=3072, max_seq_len=1024,
                       attention_type="mha", norm_type="layernorm", position_type="learned",
                       activation_type="gelu", block_style="sequential", eos_token_id=50256,
                       bos_token_id=50256, pad_token_id=50256, norm_eps=1e-5, tied_embedding=True)


def new_falcon_config(size: str = "7b") -> ModelConfig:   # config.go:151-194
    c = ModelConfig(architecture="falcon", attention_type="mqa", norm_type="layernorm", position_type="rope",
                    activation_type="gelu", block_style="parallel", num_kv_heads=1, rope_base=10000.0,
                    norm_eps=1e-5, tied_embedding=False, eos_token_id=11, bos_token_id=11, pad_token_id=11)
    if size == "40b":
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.head_dim, c.ffn_dim, c.max_seq_len = \
            "falcon-40b", 65024, 8192, 60, 128, 64, 32768, 2048
    else:
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.head_dim, c.ffn_dim, c.max_seq_len = \
            "falcon-7b", 65024, 4544, 32, 71, 64, 18176, 2048
    return c


def new_llama_config(size: str = "7b") -> ModelConfig:    # config.go:197-242
    c = ModelConfig(architecture="llama", attention_type="gqa", norm_type="rmsnorm", position_type="rope",
                    activation_type="swiglu", block_style="sequential", rope_base=10000.0, norm_eps=1e-6,
                    tied_embedding=False, eos_token_id=2, bos_token_id=1, pad_token_id=0)
    if size == "13b":
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.ffn_dim, \
            c.max_seq_len = "llama-13b", 32000, 5120, 40, 40, 10, 128, 13824, 4096
    else:
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.ffn_dim, \
            c.max_seq_len = "llama-7b", 32000, 4096, 32, 32, 8, 128, 11008, 4096
    return c


def new_granite_moe_config(size: str = "350m") -> ModelConfig:   # config.go:333-376
    return ModelConfig(architecture="granite", model_name="granite-moe-350m", attention_type="gqa",
                       norm_type="rmsnorm", position_type="rope", activation_type="swiglu",
                       block_style="sequential", rope_base=10000.0, norm_eps=1e-6, tied_embedding=True,
                       use_moe=False, num_experts=32, num_experts_per_tok=8, vocab_size=49155, hidden=1024,
                       num_layers=24, num_heads=16, num_kv_heads=8, head_dim=64, ffn_dim=512, max_seq_len=4096)


def _infer_config_from_json(raw: dict) -> ModelConfig:    # generic_loader.go:975-1007
    mt = raw.get("model_type")
    if mt == "gpt2":
        return new_gpt2_config()
    if mt in ("falcon", "RefinedWeb", "RefinedWebModel"):
        return new_falcon_config("7b")
    if mt in ("llama", "LlamaForCausalLM"):
        return new_llama_config("7b")
    if mt == "granitemoe":
        return new_granite_moe_config("350m")
    if mt == "granitemoehybrid":                        # :993-1001
        hs = raw.get("hidden_size")
        return new_granite_config("350m" if not isinstance(hs, (int, float)) or hs <= 800 else "1b")
    return new_gpt2_config()


def new_granite_config(size: str = "350m") -> ModelConfig:   # config.go:241-330 (Granite-4 hybrid: attention + Mamba2)
    c = ModelConfig(architecture="granite", attention_type="gqa", norm_type="rmsnorm", position_type="nope",
                    activation_type="swiglu", block_style="sequential", rope_base=10000.0, norm_eps=1e-5, tied_embedding=True,
                    mamba_expand=2, mamba_state_size=128, mamba_conv_kernel=4, mamba_num_heads=48)
    if size == "1b":
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.ffn_dim, \
            c.max_seq_len = "granite-4.0-h-1b", 49152, 1536, 40, 12, 4, 128, 4096, 128000
        c.mamba_n_groups = 8
        c.hybrid_layers = ["mamba"] * 40
    else:
        c.model_name, c.vocab_size, c.hidden, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim, c.ffn_dim, \
            c.max_seq_len = "granite-4.0-h-350m", 49152, 768, 32, 12, 4, 64, 2048, 32768
        c.mamba_head_dim, c.mamba_n_groups = 32, 1
        c.hybrid_layers = ["attention" if i in (10, 13, 17, 27) else "mamba" for i in range(32)]
    return c


def load_model_config(raw: dict | str) -> ModelConfig:
    """LoadModelConfig (generic_loader.go:808-972) on an already-parsed HF config.json (or its text).
    Reference quirks kept: max_position_embeddings and rope_scaling are never read, so max_seq_len
    stays the template's constant and Llama-3 RoPE frequency scaling is not applied."""
    if isinstance(raw, str):
        raw = json.loads(raw)
    arch = raw.get("architecture")
    if arch == "gpt2":
        c = new_gpt2_config()
    elif arch == "falcon":
        c = new_falcon_config("7b")
    elif arch == "llama":
        c = new_llama_config("7b")
    else:
        c = _infer_config_from_json(raw)

    def num(key):
        v = raw.get(key)
        return v if isinstance(v, (int, float)) and not isinstance(v, bool) else None

    for key, attr in (("vocab_size", "vocab_size"), ("n_embd", "hidden"), ("hidden_size", "hidden"),
                      ("n_layer", "num_layers"), ("num_hidden_layers", "num_layers"), ("num_layers", "num_layers"),
                      ("n_head", "num_heads"), ("num_attention_heads", "num_heads"), ("num_heads", "num_heads"),
                      ("num_key_value_heads", "num_kv_heads"), ("num_kv_heads", "num_kv_heads")):
        v = num(key)
        if v is not None:
            setattr(c, attr, int(v))
    if raw.get("multi_query") is True:
        c.num_kv_heads = 1
    v = num("head_dim")
    if v is not None:
        c.head_dim = int(v)
    if c.head_dim == 0 and c.hidden > 0 and c.num_heads > 0:
        c.head_dim = c.hidden // c.num_heads
    for key, attr in (("eos_token_id", "eos_token_id"), ("bos_token_id", "bos_token_id"),
                      ("pad_token_id", "pad_token_id")):
        v = num(key)
        if v is not None:
            setattr(c, attr, int(v))
    v = num("rope_theta")
    if v is not None:
        c.rope_base = float(v)
    for key in ("rms_norm_eps", "layer_norm_epsilon"):
        v = num(key)
        if v is not None:
            c.norm_eps = float(v)
    for key in ("n_inner", "intermediate_size"):
        v = num(key)
        if v is not None:
            c.ffn_dim = int(v)
    if c.ffn_dim == 0 and c.hidden > 0:
        c.ffn_dim = 4 * c.hidden
    if isinstance(raw.get("tie_word_embeddings"), bool):
        c.tied_embedding = raw["tie_word_embeddings"]
    for key in ("embedding_multiplier", "attention_multiplier", "residual_multiplier", "logits_scaling"):
        v = num(key)
        if v is not None:
            setattr(c, key, float(v))
    if isinstance(raw.get("layer_types"), list):          # :917-925
        c.hybrid_layers = [t if isinstance(t, str) else "" for t in raw["layer_types"]]
    for key, attr in (("mamba_expand", "mamba_expand"), ("mamba_d_state", "mamba_state_size"), ("mamba_n_heads", "mamba_num_heads"),
                      ("mamba_d_head", "mamba_head_dim"), ("mamba_n_groups", "mamba_n_groups"), ("mamba_d_conv", "mamba_conv_kernel")):
        v = num(key)
        if v is not None:
            setattr(c, attr, int(v))
    v = num("num_local_experts")
    if v is not None:
        c.num_experts = int(v)
        # (reference: UseMoE = true for ANY value, 0 included; a hybrid checkpoint says 0 and runs the dense shared MLP —
        # the device library treats 0 experts as "no MoE", and so does this mirror)
        c.use_moe = v > 0
    v = num("num_experts_per_tok")
    if v is not None:
        c.num_experts_per_tok = int(v)
    return c
