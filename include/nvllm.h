/*
 * nvllm.h — C ABI of libnvllm_hip.so: the MI355X (gfx950) forward path that
 * replaces nano-vllm-go's purego/tensor prefill + decode hot path.
 *
 * Drop-in boundary.  A Go maintainer binds these with cgo (stub in
 * INTEGRATION.md) and implements nanovllm.ModelRunner on top of them; nothing
 * else in nano-vllm-go changes.  Every entry point names the reference
 * interface it replaces (file:line under the reference repository).
 *
 *   - plain C, opaque handles, plain pointers + sizes, no C++/torch types;
 *   - every function returns 0 on success or a negative nvl_status; the text
 *     of the last failure is available from nvl_last_error();
 *   - the library never aborts where the reference panics (shape mismatch
 *     tensor.go:64-68, RoPE overflow rope.go:84-86): those are error codes;
 *   - host buffers are borrowed for the duration of the call only (cgo pointer
 *     rules); results are written into caller-provided host buffers;
 *   - a model handle owns one HIP stream and is NOT re-entrant (the reference's
 *     engine is single-goroutine: nanovllm/llm_engine.go:62-98).
 */
#ifndef NVLLM_H
#define NVLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVL_ABI_VERSION 3      /* 2: Mamba2 / hybrid layers (config fields, tensor kinds), nvl_stats.evictions, kernel stats;
                                  3: nvl_get_weight, nvl_get_stamps, nvl_tp_p2p_rearm, debug modes 2 / 4 (additions only) */

typedef enum nvl_status {
    NVL_OK = 0,
    NVL_ERR_INVALID = -1,      /* bad argument / shape mismatch (reference: panic tensor.go:64-68) */
    NVL_ERR_POSITION = -2,     /* pos >= max_seq_len           (reference: panic rope.go:84-86)   */
    NVL_ERR_UNKNOWN_SEQ = -3,  /* decode for a sequence with no cache slot                         */
    NVL_ERR_NO_SLOT = -4,      /* all KV slots in use                                              */
    NVL_ERR_OOM = -5,          /* hipMalloc failed                                                 */
    NVL_ERR_HIP = -6,          /* any other HIP runtime error                                      */
    NVL_ERR_STATE = -7,        /* call out of order (e.g. forward before finalize)                 */
    NVL_ERR_NO_DEVICE = -8     /* no gfx950 device visible: the library has NO CPU fallback        */
} nvl_status;

/* ---- model description: flat mirror of tensor.ModelConfig (purego/tensor/config.go:58-122) ---- */
enum { NVL_ATTN_MHA = 0, NVL_ATTN_MQA = 1, NVL_ATTN_GQA = 2 };        /* config.go:18-22 */
enum { NVL_NORM_LAYER = 0, NVL_NORM_RMS = 1 };                        /* config.go:27-30 */
enum { NVL_POS_LEARNED = 0, NVL_POS_ROPE = 1, NVL_POS_NONE = 2 };     /* config.go:35-40 */
enum { NVL_ACT_GELU = 0, NVL_ACT_SWIGLU = 1 };                        /* config.go:45-48 */
enum { NVL_BLOCK_SEQUENTIAL = 0, NVL_BLOCK_PARALLEL = 1 };            /* config.go:53-56 */

typedef struct nvl_model_config {
    int32_t vocab_size, hidden, num_layers, num_heads, num_kv_heads, head_dim;
    int32_t ffn_dim, max_seq_len;
    int32_t attention_type, norm_type, position_type, activation_type, block_style;
    double  rope_base;       /* config.go:89; ignored for MQA, which hard-wires 10000 (mqa.go:35) */
    float   norm_eps;
    int32_t tied_embedding;
    int32_t use_moe, num_experts, num_experts_per_tok;
    float   embedding_multiplier, attention_multiplier, residual_multiplier, logits_scaling;
    /* Mamba2 / hybrid layers (config.go:100-115; Granite-4: attention + Mamba2 blocks): layer li is a Mamba2 block when
     * bit li of mamba_layer_mask is set (HybridLayers[li] == "mamba" / "mamba2", generic_model.go:50-53, 74-76).
     * mamba_head_dim 0 = expand * hidden / mamba_num_heads (mamba2.go:58-61).  All zero: no Mamba2 layers. */
    int32_t  mamba_expand, mamba_state_size, mamba_num_heads, mamba_head_dim, mamba_n_groups, mamba_conv_kernel;
    uint64_t mamba_layer_mask[2];
} nvl_model_config;

/* Arithmetic the device path computes in. */
enum {
    NVL_PRECISION_BF16 = 0, /* bf16 weights + bf16 GEMM/attention operands on MFMA, fp32 accumulate,
                               fp32 residual stream / norms / softmax / RoPE (the product path)   */
    NVL_PRECISION_F32  = 1  /* fp32 weights, fp32 operands everywhere (slow; the tight-tolerance
                               parity mode used by the tests)                                     */
};

typedef struct nvl_runtime_opts {
    int32_t device;            /* HIP device ordinal                                               */
    int32_t precision;         /* NVL_PRECISION_*                                                  */
    int32_t max_seqs;          /* KV slots (concurrent sequences); replaces the unbounded
                                  map[int64]*KVCache of tensor_model_runner.go:13                  */
    int32_t max_batch_tokens;  /* largest sum(seq_lens) one nvl_forward call may carry             */
    int32_t tp_rank, tp_size;  /* tensor-parallel shard of this process (1 = none); consumer of the
                                  reference's inert Config.TensorParallelSize (nanovllm/config.go:61) */
    int32_t tp_force_single;   /* 1 with tp_size = 1: diagnostics — run the tensor-parallel projection +
                                  all-reduce path on a one-rank group (needs nvl_tp_init)             */
    int32_t kv_num_blocks;     /* > 0: PAGED KV — a pool of this many blocks addressed through the host's block
                                  tables (nanovllm/block_manager.go, Sequence.BlockTable): nvl_forward_paged /
                                  nvl_runner_run_paged; the slot calls (nvl_seq_*, nvl_forward) are refused.
                                  0: one contiguous slab per sequence slot                              */
    int32_t kv_block_size;     /* tokens per block in paged mode: 0 = 256 (Sequence.BlockSize, sequence.go:51);
                                  a multiple of 64                                                     */
    int32_t reserved;
} nvl_runtime_opts;

typedef struct nvl_model nvl_model;

/* ---- tensors: the layout contract of generic_loader.go:353-604 ---- */
typedef enum nvl_tensor_kind {
    NVL_T_TOK_EMB = 0,     /* [V, H]                 TransformerModel.TokenEmbedding generic_model.go:8 */
    NVL_T_POS_EMB,         /* [max_seq, H]           PosEmbedding                    generic_model.go:9 */
    NVL_T_LM_HEAD,         /* IN_OUT: [H, V]         LMHead (omit when tied)         generic_model.go:18 */
    NVL_T_FINAL_NORM_W,    /* [H]                    LNFinal.Weight                                  */
    NVL_T_FINAL_NORM_B,    /* [H]  absent => RMSNorm LNFinal.Bias           tensor.go:197            */
    /* per layer */
    NVL_T_ATTN_NORM_W,     /* AttnLN (sequential) / InputLN (parallel)      generic_model.go:26-28   */
    NVL_T_ATTN_NORM_B,
    NVL_T_FFN_NORM_W,
    NVL_T_FFN_NORM_B,
    NVL_T_WQ,              /* IN_OUT: [H, nH*hd]     attention.go:12,200  mqa.go:14                  */
    NVL_T_WK,              /* IN_OUT: [H, nKV*hd]    attention.go:13,201                             */
    NVL_T_WV,
    NVL_T_WKV,             /* IN_OUT: [H, 2*hd]      MQA fused K|V        mqa.go:15                  */
    NVL_T_WO,              /* IN_OUT: [nH*hd, H]                                                     */
    NVL_T_BQ, NVL_T_BK, NVL_T_BV, NVL_T_BO,   /* MHA biases               attention.go:19-23       */
    NVL_T_W1,              /* IN_OUT: [H, 2F] gate|up (SwiGLU) or [H, F]   transformer.go:30        */
    NVL_T_B1,
    NVL_T_W2,              /* IN_OUT: [F, H]                                                         */
    NVL_T_B2,
    NVL_T_ROUTER,          /* IN_OUT: [H, E]                               moe.go:12                */
    NVL_T_MOE_IN,          /* [E, 2I, H] exactly as the reference keeps it (un-transposed) moe.go:176 */
    NVL_T_MOE_OUT,         /* [E, H, I]                                                              */
    /* Mamba2Layer (mamba2.go:9-27), as loadMamba2 leaves them (generic_loader.go:461-512: PyTorch layouts, NO transpose) */
    NVL_T_MAMBA_IN_PROJ,   /* OUT_IN: [EH + conv_dim + heads, H]  InProj, used as MatMul(x, Transpose(InProj)) mamba2.go:90 */
    NVL_T_MAMBA_CONV_W,    /* [conv_dim * K]  ConvWeight [conv_dim, 1, K] flattened                  */
    NVL_T_MAMBA_CONV_B,    /* [conv_dim]                                                             */
    NVL_T_MAMBA_A_LOG,     /* [heads]                                                                */
    NVL_T_MAMBA_D,         /* [heads]                                                                */
    NVL_T_MAMBA_DT_BIAS,   /* [heads]                                                                */
    NVL_T_MAMBA_NORM,      /* [EH]                                                                   */
    NVL_T_MAMBA_OUT_PROJ,  /* OUT_IN: [H, EH]  OutProj, MatMul(y, Transpose(OutProj))           mamba2.go:175 */
    NVL_T_COUNT
} nvl_tensor_kind;

enum { NVL_DTYPE_F32 = 0, NVL_DTYPE_BF16 = 1, NVL_DTYPE_F16 = 2 };    /* generic_loader.go:645-663 */
enum {
    NVL_LAYOUT_IN_OUT = 0, /* [in, out]: what the reference holds after Transpose (generic_loader.go:398-403) */
    NVL_LAYOUT_OUT_IN = 1  /* [out, in]: the checkpoint's own PyTorch layout, uploaded without the host transpose */
};

/* ---- lifecycle ---- */

/* Replaces tensor.NewTransformerModel (generic_model.go:41-61) + the runner's model ownership
 * (tensor_model_runner.go:21-33).  Fails with NVL_ERR_NO_DEVICE when no GPU is present. */
int nvl_create(const nvl_model_config* cfg, const nvl_runtime_opts* opts, nvl_model** out);

/* Replaces loadTensorFromData + the per-kind placement in loadAttention/loadFFN/loadMoE/loadNorm
 * (generic_loader.go:353-604,619-671).  `data` may be a host or a device pointer.  2-D kinds
 * take (rows, cols) in the order of `layout`; 1-D kinds rows = n, cols = 1; MoE kinds pass
 * rows = E*out, cols = in. */
int nvl_upload_tensor(nvl_model* m, int kind, int layer, const void* data, int dtype,
                      int64_t rows, int64_t cols, int layout);

/* Replaces tensor.LoadModelFromDirectory / LoadModel / LoadShardedModel (generic_loader.go:184-265, 1016-1163): `path`
 * is a .safetensors file, or a directory holding model.safetensors or model.safetensors.index.json + shards.  The
 * file is mmap'ed and every weight is handed to the device in the checkpoint's dtype (F32/F16/BF16) and [out, in]
 * layout: no fp32 host copy of the model, no host transposes.  The WeightMapping (:60-181) is chosen from the
 * checkpoint's tensor names (GPT-2 / Falcon / Llama / Granite-MoE); names are also tried with a "transformer."
 * prefix (:622-629).  Call between nvl_create and nvl_finalize. */
int nvl_load_safetensors(nvl_model* m, const char* path);
/* LoadModelConfig (generic_loader.go:808-972) over the New*Config templates (config.go:125-376): HF config.json (or
 * the reference's model_info.json) -> nvl_model_config.  Reference quirks kept: max_position_embeddings and
 * rope_scaling are not read.  Errors are reported through nvl_last_error(NULL). */
int nvl_load_config_json(const char* path, nvl_model_config* cfg);

/* Helpers for checkpoints that keep fused projections (generic_loader.go:674-765):
 * GPT-2 c_attn [H, 3H] split by columns; Falcon query_key_value given in the reference's
 * post-transpose form [H, (nH+2)*hd]; both fp32, host pointers. */
int nvl_upload_gpt2_qkv(nvl_model* m, int layer, const float* c_attn_w, const float* c_attn_b);
int nvl_upload_falcon_qkv(nvl_model* m, int layer, const float* qkv_in_out);

/* Freezes weights into the kernels' layouts, builds RoPE tables (rope.go:18-50, in fp64 on the
 * host exactly as the reference), allocates KV slots and workspaces. */
int nvl_finalize(nvl_model* m);

/* Replaces TensorModelRunner.Close (tensor_model_runner.go:114-117). */
void nvl_destroy(nvl_model* m);

/* ---- tensor parallelism (the reference's Config.TensorParallelSize is inert: nanovllm/config.go:61) ----
 * A model created with tp_size = T > 1 is rank tp_rank's shard: column-parallel Q/K/V and gate/up, row-parallel O and
 * down, one all-reduce (sum, fp32) after the O projection and one after the down projection; embeddings, norms and the
 * LM head are replicated, so every rank produces the same logits and the same greedy token.  Upload the FULL tensors:
 * each rank keeps its slice.  One process per GPU: rank 0 calls nvl_tp_get_unique_id, ships the 128 bytes to the other
 * ranks by any channel, and every rank calls nvl_tp_init (collective; RCCL over xGMI) before its first nvl_forward.
 * nvl_tp_attach_local joins the T shard models of ONE process on ONE device into an emulated group (each driven
 * from its own host thread) — used to test the sharded arithmetic where only one GPU exists. */
int nvl_tp_get_unique_id(void* id_out, int bytes);
int nvl_tp_init(nvl_model* m, const void* id, int bytes);
int nvl_tp_attach_local(nvl_model** models, int n);
/* Hand-written all-reduce over the xGMI mesh (csrc/tp_p2p.h) in place of ncclAllReduce: one-shot direct peer stores for
 * decode-sized payloads, reduce-scatter + all-gather on the mesh for prefill-sized ones, bf16 payload with fp32
 * accumulation in the bf16 mode, the residual add fused.  One process per GPU: every rank exports the IPC handle of its
 * comm buffer (64 bytes, after nvl_finalize), the host gathers the tp_size handles in rank order by whatever channel it has,
 * and every rank attaches them.  Both calls are collective in that sense.  RCCL (nvl_tp_init) stays available and is used
 * when no peer buffers are attached. */
int nvl_tp_p2p_export(nvl_model* m, void* handle_out, int bytes);                 /* bytes >= 64 */
int nvl_tp_p2p_attach(nvl_model* m, const void* handles, int bytes_per_handle);   /* tp_size handles, rank order */
/* A rank waits a bounded time for its peers inside an all-reduce (nvl_set_tuning key 29, milliseconds, default 30 000); when
 * the wait gives up the forward call fails and the group stays unusable until EVERY rank has called nvl_tp_p2p_rearm while
 * no rank is inside a forward call (collective in that sense). */
int nvl_tp_p2p_rearm(nvl_model* m);

/* ---- sequences: replace map[int64]*KVCache (tensor_model_runner.go:11-18,59-68,100-112) ---- */
int nvl_seq_open(nvl_model* m, int64_t seq_id);     /* get-or-create a KV slot                     */
int nvl_seq_reset(nvl_model* m, int64_t seq_id);    /* NewKVCache on prefill (:63-66)              */
int nvl_seq_close(nvl_model* m, int64_t seq_id);    /* ClearCache (:100-104)                       */
int nvl_seq_close_all(nvl_model* m);                /* ClearAllCaches (:107-111)                   */
int nvl_seq_len(nvl_model* m, int64_t seq_id);      /* cached tokens, or <0                        */

/* ---- the hot path ---- */
enum {
    NVL_FWD_ALL_LOGITS = 1u   /* logits for every row (what generic_model.go:467-470 computes);
                                 default is the last row of each sequence only (what every caller
                                 keeps: GetLogitsForLastToken, generic_model.go:595-604)            */
};

/* Batched replacement of TransformerModel.ForwardWithCache (generic_model.go:276-480) +
 * GetLogitsForLastToken (:595-604) + greedy argmax (cmd/ask/main.go:389-402) for n_seqs
 * independent sequences.  `tokens` is the concatenation of each sequence's NEW tokens
 * (seq_lens[i] of them, appended at position pos_offsets[i], which must equal the sequence's
 * cached length).  logits_out (optional) receives [n_seqs, V] (or [sum(seq_lens), V] with
 * NVL_FWD_ALL_LOGITS); argmax_out (optional) receives one greedy token per sequence. */
int nvl_forward(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* tokens,
                const int32_t* seq_lens, const int32_t* pos_offsets, uint32_t flags,
                float* logits_out, int32_t* argmax_out);

/* ---- paged KV: the host's block manager owns the cache ---- */
/* One forward pass like nvl_forward, for a model created with kv_num_blocks > 0.  The KV cache of sequence i is the
 * block list block_tables[table_offsets[i] .. table_offsets[i+1]) (Sequence.BlockTable, sequence.go:23): position p
 * lives in its block p / block_size at row p % block_size.  pos_offsets[i] tokens are already cached — after a
 * prefix-cache hit that is Sequence.NumCachedTokens (block_manager.go:128-203), in decode len-1 — and only
 * seq_lens[i] new tokens are computed and written; blocks may be shared between sequences (same prefix), including
 * within one call.  The table must cover pos_offsets[i] + seq_lens[i] tokens.  The library keeps no per-sequence
 * state in this mode. */
int nvl_forward_paged(nvl_model* m, int n_seqs, const int32_t* tokens, const int32_t* seq_lens,
                      const int32_t* pos_offsets, const int32_t* block_tables, const int32_t* table_offsets,
                      uint32_t flags, float* logits_out, int32_t* argmax_out);
/* ModelRunner.Run for a block-table-aware runner: prefill computes tokens [num_cached_tokens[i], len) of each
 * sequence (all of them when num_cached_tokens is NULL; a fully cached prompt recomputes its last token), decode the
 * last token at len-1.  Greedy ids in next_tokens; sample with nvl_sample afterwards if wanted. */
int nvl_runner_run_paged(nvl_model* m, int n_seqs, const int32_t* const* token_ptrs, const int32_t* token_lens,
                         const int32_t* num_cached_tokens, const int32_t* const* block_table_ptrs,
                         const int32_t* block_table_lens, int is_prefill, int32_t* next_tokens, float* logits_out);
/* nvl_decode_greedy for a paged-KV model.  positions[i] = tokens already cached for sequence i; its block table must
 * cover positions[i] + n_steps tokens (the block manager allocates ahead).  out_tokens [n_steps][n_seqs]. */
int nvl_decode_greedy_paged(nvl_model* m, int n_seqs, const int32_t* first_tokens, const int32_t* positions, int n_steps,
                            const int32_t* block_tables, const int32_t* table_offsets, int32_t* out_tokens);
/* Debug/parity for paged mode: the K and V rows of `n_tokens` positions of one block list, as nvl_get_kv. */
int nvl_get_kv_paged(nvl_model* m, const int32_t* block_table, int n_blocks, int n_tokens, int layer, float* k_out,
                     float* v_out);

/* The greedy generation loop of cmd/ask (generateResponse, cmd/ask/main.go:315-360, without its EOS/"User" stop: the
 * caller truncates) for a batch, as ONE call: n_steps decode steps of one token per sequence starting from
 * first_tokens (the tokens sampled from the prefill), each step's device argmax fed back without a host round trip.
 * out_tokens receives [n_steps][n_seqs].  Same arithmetic, bit for bit, as n_steps calls of nvl_forward. */
int nvl_decode_greedy(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* first_tokens, int n_steps,
                      int32_t* out_tokens);

/* Debug/parity: copy the residual stream after layer `layer` of the LAST nvl_forward call,
 * [sum(seq_lens), H] fp32 (requires nvl_set_debug(m, 1 or 2) before the call).  Mode 1 completes every residual add in its
 * own launch (no split-K partials left to the next norm, no deferred RMSNorm, no graph replay): the layer outputs of a
 * simplified kernel sequence.  Mode 2 records the same values BESIDE the unmodified product path — kernel selection is
 * exactly what an untapped call runs; adds still pending for the next norm are included in the copy only. */
int nvl_set_debug(nvl_model* m, int keep_hidden);
int nvl_get_hidden(nvl_model* m, int layer, float* out, int64_t n_floats);
/* Measurement: nvl_set_debug(m, 4) makes the decode-sized projection and attention kernels of the following forward calls
 * (eager launches, no graph replay) record, per workgroup, the chip's 100 MHz constant clock at six points of their life —
 * entered / first loads issued / first data used / own stream done / workgroup's stream done / stores retired.
 * nvl_get_stamps returns the launches recorded so far: recs[3 i] = {site (nvl_kernel_site_name), phase, workgroups}, and the
 * stamps [workgroups][8] of the launches back to back.  scripts/decode_timeline.py prints the step's timeline from them.
 * Only in the diagnostic build of the library (make -C csrc diag -> libnvllm_hip_diag.so, -DNVL_STAMPS): the product library
 * has no stamp sites — compiled in and disabled they cost 3-7 % per decode projection — and answers mode 4 with
 * NVL_ERR_STATE.  The stamps perturb what they time (they pin the instruction schedule around them): use them to see where a
 * launch waits, and the kernel trace of the product build to decide whether a change paid. */
int nvl_get_stamps(nvl_model* m, int32_t* recs, int rec_cap, uint64_t* stamps, int64_t stamp_cap);
/* Debug/parity: copy a sequence's cache for one layer as the reference lays it out,
 * K and V each [nKV, T, hd] fp32 (kv_cache.go:5-6).  Returns T. */
int nvl_get_kv(nvl_model* m, int64_t seq_id, int layer, float* k_out, float* v_out);
/* Mamba2Layer.SSMState of a sequence's slot for a Mamba2 layer, [heads, head_dim, state] fp32 (mamba2.go:29-30; per
 * SEQUENCE here, per layer in the reference).  Returns the number of floats, or < 0. */
int nvl_get_mamba_state(nvl_model* m, int64_t seq_id, int layer, float* out);
/* Debug/parity: read a 2-D weight back from the device as the reference holds it after loading ([in, out] fp32,
 * generic_loader.go:398-403) — whatever layout the kernels keep it in.  kind = NVL_T_*; out is [in_features][out_features].
 * Available between the upload and nvl_finalize for the projections finalize fuses (Q, K, V, KV, W1, MoE-in), at any
 * time for the others; a tensor-parallel rank returns its own slice.  Returns 0, or NVL_ERR_STATE when the tensor is gone.
 * This is how the reference's own fixture for this path (purego/tensor/falcon_split_test.go:7-158: the fused Falcon QKV
 * split) is replayed against nvl_upload_falcon_qkv itself (tests/test_loader_gpu.py). */
int nvl_get_weight(nvl_model* m, int kind, int layer, float* out, int64_t in_features, int64_t out_features);

/* ---- runner: ModelRunner.Run semantics (nanovllm/model_runner.go:9-16) ---- */
/* Replaces TensorModelRunner.Run (tensor_model_runner.go:55-97) for a scheduler batch:
 * token_ptrs[i]/token_lens[i] is Sequence.TokenIDs (full history).  is_prefill != 0 discards the
 * sequence's cache and processes the whole history from position 0 (:63-66,75); otherwise only
 * the last token at position len-1 (:78-80) — and a sequence the library holds no cache for (or
 * a stale one) is transparently re-prefilled.  Returns greedy tokens in next_tokens[n_seqs]
 * and, when logits_out != NULL, the last-row logits [n_seqs, V] for host-side sampling
 * (tensor.SampleWithHistory stays in Go: sampling.go:33-102). */
int nvl_runner_run(nvl_model* m, int n_seqs, const int64_t* seq_ids,
                   const int32_t* const* token_ptrs, const int32_t* token_lens,
                   int is_prefill, int32_t* next_tokens, float* logits_out);

/* ---- sampling on the device (tensor.SampleWithHistory, purego/tensor/sampling.go:33-102) ---- */
/* tensor.SamplingParams (sampling.go:10-15).  Reference quirks kept: temperature <= 0 or == 1 leaves the logits
 * unscaled (:71; there is no greedy branch), top_k <= 0 or >= V is "off" (:78), top_p >= 1 is "off" (:83), the
 * repetition penalty is multiplied by the token's count, the last 10 history tokens counting 3 (:46-66). */
typedef struct nvl_sampling_params {
    float   temperature;
    float   top_p;
    int32_t top_k;
    float   repetition_penalty;
} nvl_sampling_params;
/* Sample one id per logits row of the LAST nvl_forward / nvl_runner_run group (rows in that call's sequence order), on
 * the device: only ids cross PCIe.  history_ptrs[i]/history_lens[i] is the row's previousTokens (may be NULL/0);
 * uniforms[i] replaces the rand.Float32() draw of sampleMultinomial (sampling.go:205), so the Go host keeps its
 * math/rand stream: call rand.Float32() once per sequence, in order, and pass the values.  Sums run in a fixed tree
 * order instead of the reference's sequential one: the id can differ from the Go result only when r lies within
 * float rounding of a CDF step (tests/test_sampling_gpu.py states the bound). */
int nvl_sample(nvl_model* m, int n_rows, const nvl_sampling_params* params, const int32_t* const* history_ptrs,
               const int32_t* history_lens, const float* uniforms, int32_t* out_tokens);
/* The decode loop of a sampling runner as ONE call (nvl_decode_greedy with SampleWithHistory instead of the argmax): each
 * of the n_steps steps samples one id per sequence on the device from the step's logits and the sequence's history
 * (Sequence.TokenIDs: history_ptrs/lens give it as the first step sees it, the sampled ids are appended on the device)
 * and feeds it back.  uniforms is [n_steps][n_seqs]: the rand.Float32() draws in the order the host's serial loop over
 * sequences (tensor_model_runner.go:58) would make them.  out_tokens [n_steps][n_seqs].  bf16 slab mode. */
int nvl_decode_sampled(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* first_tokens, int n_steps,
                       const nvl_sampling_params* params, const int32_t* const* history_ptrs, const int32_t* history_lens,
                       const float* uniforms, int32_t* out_tokens);
/* TensorModelRunner.Run including its sampling step (tensor_model_runner.go:55-97): nvl_runner_run, then
 * SampleWithHistory(lastTokenLogits, seq.TokenIDs, defaultSampling) per sequence on the device. */
int nvl_runner_run_sampled(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* const* token_ptrs,
                           const int32_t* token_lens, int is_prefill, const nvl_sampling_params* params,
                           const float* uniforms, int32_t* next_tokens);

/* ---- measurement (the reference only has wall-clock prints: cmd/ask/main.go:196-198) ---- */
typedef struct nvl_stats {
    uint64_t forward_calls, prefill_tokens, decode_tokens;
    double   prefill_ms, decode_ms;       /* device time of nvl_forward calls, by phase (HIP events) */
    /* per-kernel-class device time, only when nvl_set_profile(m, 1): HIP events around each launch */
    double   gemm_ms, gemm_flops;         /* all MFMA projection GEMMs (QKV, O, FFN, LM head)         */
    uint64_t gemm_launches;
    double   attn_ms, attn_flops;
    uint64_t attn_launches;
    double   other_ms;                    /* norms, RoPE/KV write, activation, embedding, argmax      */
    uint64_t other_launches;
    double   weight_bytes;                /* bytes of weights resident on the device                  */
    uint64_t evictions;                   /* KV slots reclaimed from the least-recently-forwarded sequence by
                                             nvl_runner_run / _sampled when every slot was taken            */
    uint64_t graph_replays;               /* decode passes issued as ONE hipGraphLaunch (a pass whose launch configuration was
                                             seen before is captured once and replayed; nvl_set_tuning key 21 = 0 turns it off) */
} nvl_stats;
/* Per-launch-site view of the same events (nvl_set_profile on): one entry per (phase, site) that launched — QKV / O /
 * FFN-up / FFN-down projections, attention, LM head, norms, MoE stages ... — with its launch count, summed device time
 * and summed ALGORITHMIC work (flops, bytes: every weight byte once + operands in / results out; attention: every
 * cached key and value once), so a caller can print achieved TFLOP/s or GB/s per kernel next to the roofline. */
typedef struct nvl_kernel_stat {
    int32_t  site;        /* name: nvl_kernel_site_name(site) */
    int32_t  phase;       /* 0 = prefill passes (some sequence carried more than one token), 1 = decode passes */
    uint64_t launches;
    double   ms, flops, bytes;
} nvl_kernel_stat;
int nvl_get_kernel_stats(nvl_model* m, nvl_kernel_stat* out, int cap);   /* returns the number of entries (may exceed cap) */
const char* nvl_kernel_site_name(int site);
int nvl_set_profile(nvl_model* m, int per_kernel_events);
int nvl_get_stats(nvl_model* m, nvl_stats* out);
int nvl_reset_stats(nvl_model* m);

const char* nvl_last_error(const nvl_model* m);   /* m may be NULL: last error of nvl_create */
int nvl_abi_version(void);
int nvl_device_count(void);                        /* number of visible HIP devices, 0 if none */
int nvl_sizeof(int which);                         /* 0 nvl_model_config, 1 nvl_runtime_opts, 2 nvl_stats: lets a
                                                      binding (cgo/ctypes) verify its struct mirrors */

/* ---- op-level entry points (host fp32 in / host fp32 out; used by the parity tests) ----
 * Each runs the SAME device kernel the model path uses, on one op, in `precision`. */
/* MatMul tensor.go:62-88: c[m,n] = a[m,k] x b[k,n] (b in the reference's [in,out] layout) */
int nvl_op_matmul(int device, int precision, const float* a, const float* b, float* c,
                  int m, int k, int n);
/* LayerNorm tensor.go:193-250 (bias == NULL => RMSNorm) */
int nvl_op_layernorm(int device, const float* x, const float* w, const float* bias, float eps,
                     float* y, int rows, int hidden);
/* Softmax tensor.go:128-160 over the last dim */
int nvl_op_softmax(int device, const float* x, float* y, int rows, int cols);
/* GELU tensor.go:181-190 / SiLU mamba2.go:360-367 */
int nvl_op_gelu(int device, const float* x, float* y, int64_t n);
int nvl_op_silu(int device, const float* x, float* y, int64_t n);
/* ApplyRoPESingleTensor rope.go:153-205 on t [heads, seq, hd] in place */
int nvl_op_rope(int device, float* t, int heads, int seq, int hd, int start_pos, double base,
                int max_seq);
/* GQA/MQA/MHA core on projected heads (attention.go:354-470, mqa.go:184-243):
 * q [nH,S,hd], k/v [nKV,T,hd] (keys 0..T-1, the last S are the new tokens) -> out [nH,S,hd] */
int nvl_op_attention(int device, int precision, const float* q, const float* k, const float* v,
                     int nH, int nKV, int S, int T, int hd, float scale, float* out);
/* FeedForward.Forward transformer.go:40-96 (w1 [H,2F] gate|up or [H,F]; w2 [F,H]) */
int nvl_op_ffn(int device, int precision, const float* x, const float* w1, const float* b1,
               const float* w2, const float* b2, int rows, int hidden, int ffn, int swiglu,
               float* y);
/* MoELayer.Forward moe.go:43-128 with separate experts (:167-226) */
int nvl_op_moe(int device, int precision, const float* x, const float* router, const float* w_in,
               const float* w_out, int rows, int hidden, int n_experts, int top_k, int inter,
               float* y);
/* argmax cmd/ask/main.go:389-402 (first strict maximum) */
int nvl_op_argmax(int device, const float* x, int rows, int cols, int32_t* out);

/* sampling.go:33-102 on host logits [rows, V]; probs_out (optional) receives the final renormalised distribution. */
int nvl_op_sample(int device, const float* logits, int rows, int V, const nvl_sampling_params* params,
                  const int32_t* const* history_ptrs, const int32_t* history_lens, const float* uniforms,
                  int32_t* out_tokens, float* probs_out);
/* ---- tuning / measurement (no reference counterpart) ----
 * Times one projection GEMM shape (random bf16 operands resident in HBM) with HIP events on the
 * library's stream: avg_us per launch over `iters` back-to-back launches.  epi: 0 store-fp32,
 * 1 residual-add, 2 SwiGLU, 3 GELU.  force_bnt / force_ksplit override the decode kernel's
 * automatic tile choice (0 = automatic). */
int nvl_bench_gemm(int device, int M, int N, int K, int epi, int force_bnt, int force_ksplit, int iters,
                   float* avg_us);
/* Process-wide tuning override used by tests and sweeps.  key 0: prefill GEMM kernel (0 automatic,
 * 1 = 128x128 two-stage, 2 = 256x128 three-stage, 3 = 256x256 two-stage, 5 = 256x256 ping-pong).
 * key 1: K slices of the decode residual projections (0 automatic, 1 never split, 2, 4).
 * key 2: decode (M <= 64) GEMM form: 0 automatic, 1/2/4 = narrow form with that many weight tiles per workgroup,
 * 8 = wide-N form (4 weight tiles per wave).
 * key 3: deferred RMSNorm in decode (0 off, 1 on, 2 on without the activation-tile split, 3 only the O-proj -> FFN-up seam).
 * keys 4-6 (sweeps): K split of the wide form, waves per narrow-form launch, K split of the deferred-norm residual
 * projections.  key 7: decode seam kernel, 8: MoE small-batch fusion, 9: 16-row norm kernel, 10: prefill attention
 * with 128 query rows per workgroup (0 never, 1 automatic, 2 always).  keys 11-14: decode batches of 65..512 rows (key 11, default 512; 64 = off) run the
 * projections whose 128x128 tile grid has fewer than key 12 (160) workgroups as ceil(M/64) groups of 64 rows of the
 * decode GEMM form, in one launch interleaved over the weight blocks (key 13 = 1) or one launch per group (0); up to
 * key 14 rows (64) every projection does.  key 15: waves per decode-attention workgroup (0 = from context length and
 * grid size, 2, 4, 8).  key 16: prefill MoE copies the token rows into expert order before the grouped GEMM (1, default)
 * or gathers them per lane inside it (0).  key 17: prefill MoE grouped GEMMs: 0 (default) = 256-row segments on the
 * ping-pong kernel, 128 / 256 = the lock-step tile kernels with that many rows per m-tile.
 * key 19: decode-sized MoE batches run the grouped GEMMs on the four-stage 128x128 instance (1, default) or two stages (0).
 * key 18: largest prefill batch (tokens) whose QKV projection runs as decode-form groups + rope_kv_kernel instead of the
 * fused-epilogue tile kernel (default 256: +5 % at 128 tokens, +1 % at 256, -4 % at 512).
 * key 21: hipGraph replay of decode passes (1, default; 0 = every launch eager).  key 22: decode MoE as two dense-masked
 * weight-streaming projections (1, default) or sort + grouped GEMMs (0).  key 23: largest payload (rows of the residual
 * stream) the tensor-parallel P2P all-reduce sends one-shot; above it reduce-scatter + all-gather (default 64).
 * key 25: a decode attention over >= 1024 keys in a tiny batch (<= 32 (sequence, kv head) pairs: Llama B <= 4) deals its key
 * tiles over up to 8 workgroups per pair (<= 128 in all), the last of which combines the partial results (1, default; 0 = never).
 * key 26: residual projections of 65..2048 rows split K over up to 8 workgroups per 128x128 tile, the slices summed by the
 * norm that follows (1, default; 0 = the 64-row groups of the decode form / plain tiles).
 * key 24: the decode GEMM kernels do not fetch the activation rows >= M of a padded 16-row tile (1, default).
 * key 27: deferred-norm residual projections of <= 16 rows run on 8-row half tiles, twice the workgroups (1, default).
 * key 29: milliseconds a tensor-parallel rank waits for its peers inside a P2P all-reduce before the call fails (default 30000).
 * key 30: Mamba2 prefill scan: 1 (default) the chunked SSD form on MFMA, chunks of a sequence in parallel when few (sequence, head)
 * pairs; 2 the chunked form, chunk after chunk only; 0 the sequential recurrence.
 * key 31: MoE decode routing (router logits, softmax, top-k, gate matrix) as one launch (1, default; 0 = router GEMM + gate kernel).
 * key 35: non-temporal K/V loads in the decode attention: 1 (default) when one workgroup serves a kv head's group and the launch
 * fills the chip (>= 160 workgroups) or the batch is one sequence; 0 never; 2 always.
 * key 36: weight tiles per wave of the wide decode projections (FFN-up, LM head): 0 (default) 2 for 17..32 rows and N < 65536, else 4;
 * 2 / 4 forced.
 * key 33: MoE decode: the experts' down projection carries the next RMSNorm (no K slices, no norm launch before the next QKV);
 * 0 (default): measured slower (rejected list (27)).
 * The settings are PROCESS-GLOBAL and unsynchronised (every model in the process sees them): set them from one thread
 * while no forward call is running.  Every call starts a new tuning epoch: captured decode graphs bake the tuning in and are re-captured.
 * Returns the previous value. */
int nvl_set_tuning(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* NVLLM_H */
