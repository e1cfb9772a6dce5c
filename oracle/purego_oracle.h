/*
 * purego_oracle.h — CPU restatement of nano-vllm-go's purego/tensor forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product path (nano-vllm-go_amd/): only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED (arithmetic): the reference is Go (go.mod wants go 1.25) and
 * no Go toolchain exists in this image, so the reference cannot be run to
 * produce vectors, and its own tests pin no arithmetic on this path.  The one
 * fixture the reference's tests hold for the path — the Falcon fused-QKV
 * de-interleave, purego/tensor/falcon_split_test.go:7-158 — is checked in
 * tests/test_oracle_layout.py.  Everything else here is a line-faithful
 * restatement (same loop order, fp32 storage, sequential fp32 sums, no FMA:
 * build with -ffp-contract=off; transcendental functions evaluated in double
 * and narrowed exactly where the Go code narrows) cross-checked against the
 * HuggingFace modelling code on random weights (tests/test_oracle_vs_hf.py).
 *
 * All citations are file:line under /root/reference/.
 */
#ifndef PUREGO_ORACLE_H
#define PUREGO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enums mirror purego/tensor/config.go:15-56 */
enum { PO_ATTN_MHA = 0, PO_ATTN_MQA = 1, PO_ATTN_GQA = 2 };
enum { PO_NORM_LAYER = 0, PO_NORM_RMS = 1 };
enum { PO_POS_LEARNED = 0, PO_POS_ROPE = 1, PO_POS_NONE = 2 };
enum { PO_ACT_GELU = 0, PO_ACT_SWIGLU = 1 };
enum { PO_BLOCK_SEQUENTIAL = 0, PO_BLOCK_PARALLEL = 1 };

/* Flat mirror of tensor.ModelConfig (purego/tensor/config.go:58-122), the
 * fields the attention/FFN/MoE path reads. */
typedef struct po_config {
    int32_t vocab_size, hidden, num_layers, num_heads, num_kv_heads, head_dim;
    int32_t ffn_dim, max_seq_len;
    int32_t attention_type, norm_type, position_type, activation_type, block_style;
    double  rope_base;
    float   norm_eps;
    int32_t tied_embedding;
    int32_t use_moe, num_experts, num_experts_per_tok;
    float   embedding_multiplier, attention_multiplier, residual_multiplier, logits_scaling;
    /* Mamba2 / hybrid layers (config.go:100-115): layer li is a Mamba2 block when layer_is_mamba[li] != 0
     * (HybridLayers[li] == "mamba" / "mamba2", generic_model.go:50-53, 74-76) */
    int32_t mamba_expand, mamba_state_size, mamba_num_heads, mamba_head_dim, mamba_n_groups, mamba_conv_kernel;
    uint8_t layer_is_mamba[128];
} po_config;

/* Tensor slots, in the layout the reference holds AFTER loading
 * (generic_loader.go:353-604): 2-D weights are [in, out] row-major fp32. */
enum {
    PO_T_TOK_EMB = 0,   /* [V, H]                     generic_loader.go:224 */
    PO_T_POS_EMB,       /* [max_seq, H]               generic_loader.go:229 */
    PO_T_LM_HEAD,       /* [H, V]                     generic_loader.go:246-259 */
    PO_T_FINAL_NORM_W,  /* [H]                                            */
    PO_T_FINAL_NORM_B,  /* [H] (nil => RMSNorm)       tensor.go:197        */
    /* per layer */
    PO_T_ATTN_NORM_W,   /* AttnLN (sequential) or InputLN (parallel)      */
    PO_T_ATTN_NORM_B,
    PO_T_FFN_NORM_W,
    PO_T_FFN_NORM_B,
    PO_T_WQ,            /* [H, nH*hd]                                      */
    PO_T_WK,            /* [H, nKV*hd]   (MHA/GQA)                         */
    PO_T_WV,
    PO_T_WKV,           /* [H, 2*hd]     (MQA, mqa.go:15)                  */
    PO_T_WO,            /* [nH*hd, H]                                      */
    PO_T_BQ, PO_T_BK, PO_T_BV, PO_T_BO,   /* MHA biases attention.go:19-23 */
    PO_T_W1,            /* [H, 2F] gate|up (SwiGLU) or [H, F] transformer.go:30 */
    PO_T_B1,
    PO_T_W2,            /* [F, H]                                          */
    PO_T_B2,
    PO_T_ROUTER,        /* [H, E]        moe.go:12                         */
    PO_T_MOE_IN,        /* [E, 2I, H]    moe.go:176 (NOT transposed)       */
    PO_T_MOE_OUT,       /* [E, H, I]                                       */
    /* Mamba2Layer (mamba2.go:9-27), as loadMamba2 leaves them (generic_loader.go:461-512: NO transpose) */
    PO_T_MAMBA_IN_PROJ, /* [gate + conv_dim + heads, H]   PyTorch [out, in]; used as MatMul(x, Transpose(InProj)) mamba2.go:90 */
    PO_T_MAMBA_CONV_W,  /* [conv_dim, 1, K]                                 */
    PO_T_MAMBA_CONV_B,  /* [conv_dim]                                       */
    PO_T_MAMBA_A_LOG,   /* [heads]                                          */
    PO_T_MAMBA_D,       /* [heads]                                          */
    PO_T_MAMBA_DT_BIAS, /* [heads]                                          */
    PO_T_MAMBA_NORM,    /* [expand * H]                                     */
    PO_T_MAMBA_OUT_PROJ,/* [H, expand * H]           PyTorch [out, in]; MatMul(y, Transpose(OutProj)) mamba2.go:175 */
    PO_T_COUNT
};

typedef struct po_model po_model;
typedef struct po_kvcache po_kvcache;

po_model*   po_model_new(const po_config* cfg);
void        po_model_free(po_model* m);
/* Copies n floats into the slot (layer ignored for model-level slots). */
int         po_model_set(po_model* m, int slot, int layer, const float* data, int64_t n);
/* Same, but borrows the caller's buffer (no copy; caller keeps it alive). */
int         po_model_set_borrowed(po_model* m, int slot, int layer, const float* data, int64_t n);

po_kvcache* po_kvcache_new(int num_layers);            /* kv_cache.go:10 */
void        po_kvcache_free(po_kvcache* kv);
int         po_kvcache_len(const po_kvcache* kv);      /* cached tokens in layer 0 */
/* copies layer K or V ([nKV, T, hd]) into out; returns T */
int         po_kvcache_get(const po_kvcache* kv, int layer, int which, float* out);

/* TransformerModel.ForwardWithCache (generic_model.go:276-480): logits_out is
 * [n_tokens, V] (ALL rows, as the reference computes).  If hidden_out != NULL
 * it receives the residual stream after every layer, [L, n_tokens, H].
 * Returns 0, or -1 on the conditions under which the reference panics. */
int po_forward_with_cache(po_model* m, const int32_t* tokens, int n_tokens,
                          po_kvcache* kv, int pos_offset,
                          float* logits_out, float* hidden_out);

/* Mamba2Layer.Forward (mamba2.go:74-181) on x [S, H] -> out [S, H], with the layer's persistent SSM state
 * (mamba2.go:29-30, 258-260); the model resets it like ForwardWithCache does (generic_model.go:285-292). */
void po_mamba2_reset(po_model* m);
int  po_mamba2_get_state(const po_model* m, int layer, float* out);   /* [heads, head_dim, state]; returns floats */

/* test-time knobs, not part of the restated algorithm (see purego_oracle.c): row-parallel MatMul (bit-identical per
 * row; default 1 thread like the reference) and "LM head on the last row only" (logits_out becomes [1, V]) */
void po_set_threads(int n);
void po_set_lm_head_last_only(int on);

/* cmd/ask/main.go:389-402 */
int po_argmax(const float* data, int n);

/* ---- op-level entry points (each mirrors one purego/tensor function) ---- */
void po_matmul(const float* a, const float* b, float* c, int m, int k, int n);      /* tensor.go:62-88 */
void po_layernorm(const float* x, const float* w, const float* bias /*NULL=RMS*/,
                  float eps, float* y, int rows, int hidden);                      /* tensor.go:193-250 */
void po_softmax_rows(const float* x, float* y, int rows, int cols);                /* tensor.go:128-160 */
void po_gelu(const float* x, float* y, int64_t n);                                 /* tensor.go:181-190 */
void po_silu(const float* x, float* y, int64_t n);                                 /* mamba2.go:360-367 */
void po_rope_tables(int head_dim, int max_seq, double base, float* cos_t, float* sin_t); /* rope.go:18-50 */
/* ApplyRoPESingleTensor (rope.go:153-205) on t = [heads, seq, hd]; -1 if pos overflows */
int  po_rope_apply(float* t, int heads, int seq, int hd, int start_pos,
                   const float* cos_t, const float* sin_t, int max_seq);
/* SwiGLU / GELU FeedForward.Forward (transformer.go:40-96) */
void po_ffn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
            int rows, int hidden, int ffn, int swiglu, float* y);
/* GQA scores->softmax->apply on already-projected heads (attention.go:354-470):
 * q [nH, S, hd], k/v [nKV, T, hd] -> out [nH, S, hd].  scale==0 => 1/sqrt(hd). */
void po_gqa_core(const float* q, const float* k, const float* v, int nH, int nKV,
                 int S, int T, int hd, float scale, float* out);
/* MoELayer.Forward with separate experts (moe.go:43-128,167-226) */
void po_moe(const float* x, const float* router, const float* w_in, const float* w_out,
            int rows, int hidden, int n_experts, int top_k, int inter, float* y);

/* SampleWithHistory (sampling.go:33-102) with the rand.Float32() draw of sampleMultinomial (:205) passed in as `u`.
 * Same sequential fp32 sums, exp in double.  sort.Slice is unstable in the reference; ties are ordered by index here.
 * probs_out (optional, [n]) receives the renormalised distribution; returns the sampled index. */
int po_sample_with_history(const float* logits, int n, const int32_t* prev, int n_prev, float temperature,
                           float top_p, int top_k, float rep_penalty, float u, float* probs_out);

/* ---- load-time layout contract (generic_loader.go) ---- */
void po_transpose(const float* t, float* out, int m, int n);                       /* tensor.go:112-125 */
void po_split_gpt2_qkv(const float* qkv, int hidden, float* q, float* k, float* v);/* generic_loader.go:674-702 */
void po_split_falcon_qkv(const float* qkv, int hidden, int num_heads, int head_dim,
                         float* q, float* k, float* v);                            /* generic_loader.go:705-748 */
void po_combine_mqa_kv(const float* k, const float* v, int hidden, int head_dim, float* kv); /* :751-765 */
void po_concat_last_dim(const float* a, const float* b, int rows, int c1, int c2, float* out); /* tensor.go:254-281 */
float po_f32_from_bf16(uint16_t bits);                                             /* generic_loader.go:802-805 */
float po_f32_from_f16(uint16_t bits);                                              /* generic_loader.go:774-800 */

#ifdef __cplusplus
}
#endif
#endif
