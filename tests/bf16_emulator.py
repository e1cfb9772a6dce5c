"""A CPU emulation (numpy, float64) of the bf16 PRODUCT path's rounding points — test infrastructure, like the oracle.

The HIP path keeps the residual stream, norms, softmax statistics, RoPE and every accumulation in fp32 and rounds to bf16
exactly where a value becomes an MFMA operand:
    xn   normed input of the QKV projection        q    RoPE'd query             k, v  RoPE'd key / value in the KV cache
    p    softmax probabilities                      ao   attention output          xn2   normed input of the FFN-up projection
    h    silu(gate) * up                            xl   final-norm output (LM head operand)
This module runs a Llama-family stack (GQA + RoPE + RMSNorm + SwiGLU, sequential blocks: BASELINE configs[1]) in float64
with a bf16 rounding at each of those points, on the same weights and prompt as a device run.  If the kernels round there
and nowhere else, the device's error against the oracle equals this emulation's error against the oracle, layer by layer
— which is what tests/test_depth_parity_gpu.py asserts (and scripts/bf16_error_budget.py breaks down per rounding point).
"""
import numpy as np

ALL = ("xn", "q", "k", "v", "p", "ao", "xn2", "h", "xl")


def round_bf16(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def rb(a, on, dt=np.float64):
    return round_bf16(np.asarray(a, np.float32)).astype(dt) if on else np.asarray(a, dt)


def _rms(v, g, eps):
    return v / np.sqrt((v * v).mean(-1, keepdims=True) + eps) * g


def layer(x, w, li, cfg, sw):
    """one decoder layer on the [S, H] stream x; sw = the set of rounding points that are on.  Computes in x.dtype:
    float64 (the default of forward(): "exact" beside the roundings) or float32 (the 7-8 B-parameter configs: numpy sgemm,
    relative error ~1e-6, three orders below one bf16 rounding — and no float64 copy of every weight matrix)"""
    nH, nKV, hd, F = cfg["num_heads"], cfg["num_kv_heads"], cfg["head_dim"], cfg["ffn_dim"]
    S = x.shape[0]
    eps = cfg["norm_eps"]
    dt = x.dtype

    def W(slot):
        return w[(slot, li)].astype(dt, copy=False)

    def r(a, point):
        return rb(a, point in sw, dt)
    xn = r(_rms(x, W("attn_norm_w"), eps), "xn")
    q = xn @ W("wq")
    k = xn @ W("wk")
    v = xn @ W("wv")
    half = hd // 2
    inv = 1.0 / cfg["rope_base"] ** (np.arange(half) * 2.0 / hd)
    ang = np.arange(S)[:, None] * inv[None, :]
    cos = np.concatenate([np.cos(ang), np.cos(ang)], -1).astype(np.float32).astype(dt)   # fp32 tables (rope.go:18-50)
    sin = np.concatenate([np.sin(ang), np.sin(ang)], -1).astype(np.float32).astype(dt)

    def rope(t, heads):
        t = t.reshape(S, heads, hd)
        rot = np.concatenate([-t[..., half:], t[..., :half]], -1)
        return t * cos[:, None, :] + rot * sin[:, None, :]
    q = r(rope(q, nH), "q")
    k = r(rope(k, nKV), "k")
    v = r(v.reshape(S, nKV, hd), "v")
    g = nH // nKV
    ao = np.zeros((S, nH, hd), dt)
    mask = np.tril(np.ones((S, S), bool))
    scale = cfg.get("attention_multiplier") or 1.0 / np.sqrt(hd)
    for h in range(nH):
        s = np.where(mask, q[:, h, :] @ k[:, h // g, :].T * scale, -np.inf)
        p = r(np.exp(s - s.max(-1, keepdims=True)), "p")
        ao[:, h, :] = (p @ v[:, h // g, :]) / p.sum(-1, keepdims=True)      # row sums of the rounded p (ones . P^T on the MFMA pipe)
    ao = r(ao.reshape(S, nH * hd), "ao")
    x = x + ao @ W("wo")
    xn2 = r(_rms(x, W("ffn_norm_w"), eps), "xn2")
    gu = xn2 @ W("w1")
    gate, up = gu[:, :F], gu[:, F:]
    hh = r(gate / (1.0 + np.exp(-gate)) * up, "h")
    return x + hh @ W("w2")


def forward(cfg, w, tokens, sw=ALL, last_logits=True, dtype=np.float64):
    """-> (hidden [L, S, H], last-row logits [V] or None) in `dtype`"""
    assert cfg["attention_type"] == "gqa" and cfg["norm_type"] == "rmsnorm" and cfg["activation_type"] == "swiglu" and \
        cfg["block_style"] == "sequential" and cfg["position_type"] == "rope" and not cfg.get("use_moe")
    sw = set(sw)
    x = w[("tok_emb", 0)][np.asarray(tokens)].astype(dtype)
    hidden = []
    for li in range(cfg["num_layers"]):
        x = layer(x, w, li, cfg, sw)
        hidden.append(x)
    logits = None
    if last_logits:
        xl = rb(_rms(x[-1:], w[("final_norm_w", 0)].astype(dtype), cfg["norm_eps"]), "xl" in sw, dtype)
        head = w[("lm_head", 0)] if ("lm_head", 0) in w else w[("tok_emb", 0)].T      # tied: generic_loader.go:255-259
        logits = (xl @ head.astype(dtype, copy=False))[0]
    return np.stack(hidden), logits
