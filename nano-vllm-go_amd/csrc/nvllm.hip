// nvllm.hip — the C ABI of include/nvllm.h: model lifecycle, weight upload into kernel layouts,
// KV slot management, and the batched prefill/decode forward that replaces
// TransformerModel.ForwardWithCache (purego/tensor/generic_model.go:276-480).
//
// There is NO CPU fallback in this library: every entry point that computes needs a gfx950
// device and fails with NVL_ERR_NO_DEVICE / NVL_ERR_HIP otherwise.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstring>
#include <new>

#include <rccl/rccl.h>

#include "model.h"

using namespace nvl;

static thread_local std::string g_create_err;
constexpr int ATTN_SPLIT_MAX_PAIRS = 32, ATTN_SPLIT_MAX = 8;     // split decode attention: (sequence, kv head) pairs a launch may have; most workgroups per pair
constexpr int SK_TILE_MAX_M = 2048, SK_TILE_MAX_SLICES = 8;      // split-K of the tile kernel for mid-size residual projections (resid_gemm)
static int g_sk_tile = 1;            // nvl_set_tuning key 26: that split (0 = off)
constexpr int MOE_DOWN_SLICES = 4;    // at most this many K slices (workgroups per column tile) in the dense-masked MoE down projection
static inline int moe_down_slices(int E) { int s = MOE_DOWN_SLICES; while (s > 1 && E % s) s--; return s; }   // a whole number of experts per slice

namespace {

int fail(nvl_model* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_create_err = msg;
    return code;
}

// a member of an in-process tensor-parallel group (nvl_tp_attach_local) that fails must not leave its peers waiting in
// tp_allreduce forever: the failure poisons the group and wakes them
void note_failure(nvl_model* m) {
    if (!m || !m->tp_local) return;
    { std::lock_guard<std::mutex> lk(m->tp_local->mu); m->tp_local->aborted = true; }
    m->tp_local->cv.notify_all();
}

#define NVL_TRY(m) try {
#define NVL_CATCH(m)                                                                     \
    } catch (const HipError& e) {                                                        \
        note_failure(m);                                                                 \
        char buf[512];                                                                   \
        snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e.code,         \
                 hipGetErrorString(e.code), e.file, e.line, e.what);                     \
        return fail(m, e.code == hipErrorOutOfMemory ? NVL_ERR_OOM : NVL_ERR_HIP, buf);  \
    } catch (const std::bad_alloc&) {                                                    \
        note_failure(m);                                                                 \
        return fail(m, NVL_ERR_OOM, "out of host memory");                               \
    } catch (const std::exception& e) {                                                  \
        note_failure(m);                                                                 \
        return fail(m, NVL_ERR_INVALID, e.what());                                       \
    }

template <typename T>
T* dmalloc(int64_t n) {
    void* p = nullptr;
    if (n <= 0) n = 1;
    NVL_HIP(hipMalloc(&p, (size_t)n * sizeof(T)));
    return (T*)p;
}
void* dmalloc_bytes(int64_t n) {
    void* p = nullptr;
    if (n <= 0) n = 16;
    NVL_HIP(hipMalloc(&p, (size_t)n));
    return p;
}
void dfree(void* p) { if (p) (void)hipFree(p); }
// Captured decode passes (replay_or_capture).  A graph exec may still be executing on the stream (the fused decode loop
// enqueues its steps without a host sync), so nothing is destroyed while work can be in flight: retire_graphs() only moves
// the execs aside, reap_graphs() destroys them and is called right after a hipStreamSynchronize of the model's stream.
void reap_graphs(nvl_model* m) {
    for (auto ge : m->graphs_retired) (void)hipGraphExecDestroy(ge);
    m->graphs_retired.clear();
}
void retire_graphs(nvl_model* m) {
    for (auto& kv : m->graphs) m->graphs_retired.push_back(kv.second);
    m->graphs.clear();
    m->graph_seen.clear();
}
// every captured pass bakes in the addresses of the buffers it was captured with: call before any of them is re-allocated
void clear_graphs(nvl_model* m) {
    retire_graphs(m);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    reap_graphs(m);
}
void free_sample_bufs(SampleBufs& b) {
    dfree(b.work); dfree(b.cnt); dfree(b.hist); dfree(b.off); dfree(b.out); dfree(b.u); dfree(b.probs);
    b = SampleBufs{};
}

bool is_1d(int kind) {
    switch (kind) {
        case NVL_T_FINAL_NORM_W: case NVL_T_FINAL_NORM_B: case NVL_T_ATTN_NORM_W: case NVL_T_ATTN_NORM_B:
        case NVL_T_FFN_NORM_W: case NVL_T_FFN_NORM_B: case NVL_T_BQ: case NVL_T_BK: case NVL_T_BV:
        case NVL_T_BO: case NVL_T_B1: case NVL_T_B2:
        case NVL_T_MAMBA_CONV_W: case NVL_T_MAMBA_CONV_B: case NVL_T_MAMBA_A_LOG: case NVL_T_MAMBA_D: case NVL_T_MAMBA_DT_BIAS:
        case NVL_T_MAMBA_NORM: return true;
        default: return false;
    }
}
bool is_global(int kind) { return kind < NVL_T_ATTN_NORM_W; }

DevTensor* tensor_slot(nvl_model* m, int kind, int layer) {
    if (kind < 0 || kind >= NVL_T_COUNT) return nullptr;
    if (is_global(kind)) return &m->g[kind];
    if (layer < 0 || layer >= m->L) return nullptr;
    return &m->layers[layer].t[kind];
}

// ---- profiling helpers --------------------------------------------------------------------
hipEvent_t get_event(nvl_model* m) {
    if (!m->ev_pool.empty()) { hipEvent_t e = m->ev_pool.back(); m->ev_pool.pop_back(); return e; }
    hipEvent_t e; NVL_HIP(hipEventCreate(&e)); return e;
}
struct KScope {   // brackets ONE kernel launch with HIP events on the model's stream
    nvl_model* m; int cls; double flops; hipEvent_t a = nullptr, b = nullptr;
    int site, phase; double bytes;
    KScope(nvl_model* m_, int cls_, double flops_ = 0, int site_ = -1, double bytes_ = -1) : m(m_), cls(cls_), flops(flops_) {
        // the launch site and its algorithmic bytes: set by the caller (m->site / m->site_bytes) or passed here
        site = site_ >= 0 ? site_ : m->site; bytes = bytes_ >= 0 ? bytes_ : m->site_bytes; phase = m->phase;
        m->site = KS_OTHER; m->site_bytes = 0;
        if (m->profile) { a = get_event(m); b = get_event(m); NVL_HIP(hipEventRecord(a, m->stream)); }
    }
    ~KScope() {
        if (m->profile && a) { (void)hipEventRecord(b, m->stream); m->prof.push_back({a, b, cls, flops, site, phase, bytes}); }
    }
};
void drain_profile(nvl_model* m) {
    for (auto& r : m->prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        if (r.cls == KC_GEMM) { m->stats.gemm_ms += ms; m->stats.gemm_flops += r.flops; m->stats.gemm_launches++; }
        else if (r.cls == KC_ATTN) { m->stats.attn_ms += ms; m->stats.attn_flops += r.flops; m->stats.attn_launches++; }
        else { m->stats.other_ms += ms; m->stats.other_launches++; }
        SiteStat& ss = m->site_stats[r.phase & 1][r.site >= 0 && r.site < KS_COUNT ? r.site : 0];
        ss.ms += ms; ss.flops += r.flops; ss.bytes += r.bytes; ss.launches++;
        m->ev_pool.push_back(r.a); m->ev_pool.push_back(r.b);
    }
    m->prof.clear();
}

// ---- GEMM dispatch ---------------------------------------------------------------------------
// out_f32: output element type of STORE (fp32 vs activation type)
// diagnostic runs (nvl_set_debug mode 4): the stamp slab of the next launch, or NULL
unsigned long long* next_stamps(nvl_model* m, int site, int nwg) {
    if (!m->stamping || !m->stamp_buf || m->stamp_launches >= STAMP_MAX_LAUNCH || nwg > STAMP_MAX_WG) return nullptr;
    m->stamp_recs.push_back({site, m->phase, nwg});
    return m->stamp_buf + (size_t)(m->stamp_launches++) * STAMP_MAX_WG * 8;
}
void gemm(nvl_model* m, int epi, bool out_f32, GemmArgs a, double flops = -1.0) {
    a.stamps = (m->stamping && !m->f32 && a.M <= 64 && !a.tile_map) ? next_stamps(m, m->site, STAMP_MAX_WG) : nullptr;
    // algorithmic bytes of a projection: every weight byte once + the activation operand + the output rows
    const double gbytes = m->site_bytes > 0 ? m->site_bytes
        : ((double)a.N * a.K + (double)a.M * a.K) * (double)m->wsize + (double)a.M * a.N * (out_f32 || epi == EPI_RESID ? 4.0 : (double)m->wsize);
    KScope ks(m, KC_GEMM, flops >= 0 ? flops : 2.0 * (double)a.M * (double)a.N * (double)a.K, -1, gbytes);
    hipStream_t st = m->stream;
    if (m->f32) {
        switch (epi) {
            case EPI_STORE: launch_gemm_f32<EPI_STORE, float>(st, a); break;
            case EPI_RESID: launch_gemm_f32<EPI_RESID, float>(st, a); break;
            case EPI_GELU:  launch_gemm_f32<EPI_GELU, float>(st, a); break;
            default: throw std::runtime_error("gemm: epilogue not available in f32 mode");
        }
    } else {
        switch (epi) {
            case EPI_STORE:
                if (out_f32) launch_gemm_bf16<EPI_STORE, float>(st, a);
                else launch_gemm_bf16<EPI_STORE, bf16_t>(st, a);
                break;
            case EPI_RESID:  launch_gemm_bf16<EPI_RESID, float>(st, a); break;
            case EPI_SWIGLU: launch_gemm_bf16<EPI_SWIGLU, bf16_t>(st, a); break;
            case EPI_GELU:   launch_gemm_bf16<EPI_GELU, bf16_t>(st, a); break;
            case EPI_QKV:    launch_gemm_bf16<EPI_QKV, bf16_t>(st, a); break;
            default: throw std::runtime_error("gemm: bad epilogue");
        }
    }
    NVL_HIP(hipGetLastError());
}

}  // namespace

// =================================================================================================
// lifecycle
// =================================================================================================
extern "C" int nvl_abi_version(void) { return NVL_ABI_VERSION; }

extern "C" int nvl_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(nvl_model_config);
        case 1: return (int)sizeof(nvl_runtime_opts);
        case 2: return (int)sizeof(nvl_stats);
        case 3: return (int)sizeof(nvl_sampling_params);
        default: return -1;
    }
}

extern "C" int nvl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* nvl_last_error(const nvl_model* m) { return m ? m->err.c_str() : g_create_err.c_str(); }

extern "C" int nvl_create(const nvl_model_config* cfg, const nvl_runtime_opts* opts, nvl_model** out) {
    if (!cfg || !opts || !out) return fail(nullptr, NVL_ERR_INVALID, "nvl_create: null argument");
    *out = nullptr;
    const nvl_model_config& c = *cfg;
    if (c.hidden <= 0 || c.num_layers <= 0 || c.num_heads <= 0 || c.head_dim <= 0 || c.vocab_size <= 0 ||
        c.max_seq_len <= 0)
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: non-positive model dimension");
    if (c.head_dim != 64 && c.head_dim != 128)
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: head_dim must be 64 or 128");
    if (c.attention_multiplier < 0.f)      // the kernels take the row maximum before scaling
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: attention_multiplier must not be negative");
    if (c.hidden % 64 != 0 || (c.ffn_dim % 64) != 0)
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: hidden and ffn_dim must be multiples of 64");
    // runtime options are validated BEFORE any size is derived from them (a zero-valued host struct must not turn into
    // a 0-block KV cache): max_seqs >= 1 is required, max_batch_tokens 0 selects max_seq_len
    if (opts->max_seqs <= 0)
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: max_seqs must be >= 1 (KV slots / sequences per call)");
    if (opts->max_batch_tokens < 0 || opts->kv_num_blocks < 0 || opts->kv_block_size < 0)
        return fail(nullptr, NVL_ERR_INVALID, "nvl_create: negative max_batch_tokens / kv_num_blocks / kv_block_size");
    const int tp = opts->tp_size > 1 ? opts->tp_size : 1;
    if (tp > 1) {
        const int nkv_full = c.attention_type == NVL_ATTN_MHA ? c.num_heads : c.num_kv_heads;
        if (opts->tp_rank < 0 || opts->tp_rank >= tp || tp > 8)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: tp_rank/tp_size out of range (tp_size <= 8)");
        if (c.attention_type == NVL_ATTN_MQA || c.use_moe)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: tensor parallelism covers MHA/GQA dense models (MQA has one KV head; MoE shards by expert)");
        if (c.num_heads % tp || nkv_full % tp || c.ffn_dim % (tp * 64))
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: heads, kv heads and ffn_dim/64 must divide by tp_size");
    }
    if (opts->device < 0 || nvl_device_count() <= opts->device)
        return fail(nullptr, NVL_ERR_NO_DEVICE,
                    "nvl_create: no HIP device (this library has no CPU fallback)");
    const bool any_mamba = (c.mamba_layer_mask[0] | c.mamba_layer_mask[1]) != 0;
    if (any_mamba) {
        const int eh = c.mamba_expand * c.hidden;
        const int mhd = c.mamba_head_dim > 0 ? c.mamba_head_dim : (c.mamba_num_heads > 0 ? eh / c.mamba_num_heads : 0);
        if (c.num_layers > 128 || c.mamba_expand <= 0 || c.mamba_state_size <= 0 || c.mamba_num_heads <= 0 || c.mamba_n_groups <= 0 ||
            c.mamba_conv_kernel <= 0 || c.mamba_conv_kernel > 8 || mhd <= 0 || mhd * c.mamba_num_heads != eh)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: bad Mamba2 dimensions (heads x head_dim must equal expand x hidden)");
        // the scan kernel keeps state_size / (256 / head_dim) states per thread: head_dim | 256, at most 32 states per thread
        if (256 % mhd != 0 || 256 / mhd > 64 || c.mamba_state_size % (256 / mhd) != 0 || c.mamba_state_size / (256 / mhd) > 32 ||
            (c.mamba_state_size / (256 / mhd)) % 4 != 0 || eh % 64 != 0 || c.mamba_num_heads % c.mamba_n_groups != 0)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: Mamba2 head_dim / state_size not supported by the scan kernel");
        if (tp > 1) return fail(nullptr, NVL_ERR_INVALID, "nvl_create: tensor parallelism does not cover Mamba2 layers");
        if (opts->kv_num_blocks > 0)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: Mamba2 layers keep per-sequence state in KV slots: not available in paged-KV mode");
        if (c.use_moe || c.block_style != NVL_BLOCK_SEQUENTIAL)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_create: hybrid (Mamba2) models are sequential blocks with a dense MLP");
    }
    nvl_model* m = new (std::nothrow) nvl_model();
    if (!m) return fail(nullptr, NVL_ERR_OOM, "nvl_create: out of host memory");
    NVL_TRY(m)
    m->cfg = c; m->opts = *opts;
    m->f32 = (opts->precision == NVL_PRECISION_F32);
    m->wsize = m->f32 ? 4 : 2;
    m->device = opts->device;
    NVL_HIP(hipSetDevice(m->device));
    NVL_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    NVL_HIP(hipEventCreate(&m->ev0));
    NVL_HIP(hipEventCreate(&m->ev1));
    m->H = c.hidden; m->nH = c.num_heads; m->hd = c.head_dim; m->F = c.ffn_dim; m->V = c.vocab_size;
    m->L = c.num_layers;
    m->nKV = c.attention_type == NVL_ATTN_MHA ? c.num_heads : (c.attention_type == NVL_ATTN_MQA ? 1 : c.num_kv_heads);
    if (m->nKV <= 0 || m->nH % m->nKV != 0) throw std::runtime_error("num_heads must be a multiple of num_kv_heads");
    // tensor parallel shard: this rank owns nH/tp query heads, nKV/tp kv heads and F/tp FFN
    // columns; from here on nH / nKV / F are the LOCAL sizes (hidden H, vocab V stay full: residual stream,
    // norms, embeddings and the LM head are replicated)
    m->tp = tp; m->tp_rank = tp > 1 ? opts->tp_rank : 0;
    m->tp_force = tp == 1 && opts->tp_force_single == 1;   // diagnostics: run the row-parallel path + all-reduce with one rank
    m->nH_full = m->nH; m->nKV_full = m->nKV; m->F_full = m->F;
    m->nH /= tp; m->nKV /= tp; m->F /= tp;
    m->group = m->nH / m->nKV;
    m->Vpad = (int)round_up(m->V, 128);
    m->n_qkv = (m->nH + 2 * m->nKV) * m->hd;
    m->Tmax = (int)round_up(c.max_seq_len, 64);
    if (opts->kv_num_blocks > 0) {            // paged KV: the host's block manager names the blocks
        const int bs = opts->kv_block_size > 0 ? opts->kv_block_size : 256;
        if (bs % 64 != 0) throw std::runtime_error("kv_block_size must be a multiple of 64");
        m->paged = 1; m->Tmax = bs; m->num_blocks = opts->kv_num_blocks;
        m->blocks_per_seq = cdiv(c.max_seq_len, bs);
    } else {
        m->num_blocks = opts->max_seqs;
    }
    m->table_cap = opts->max_seqs * m->blocks_per_seq;
    // score scale: GQA uses AttentionMultiplier when set (attention.go:361-364); MHA/MQA 1/sqrt(hd)
    if (c.attention_type == NVL_ATTN_GQA && c.attention_multiplier != 0.f) m->attn_scale = c.attention_multiplier;
    else m->attn_scale = 1.0f / std::sqrt((float)m->hd);
    m->resid_alpha = c.residual_multiplier != 0.f ? c.residual_multiplier : 1.0f;
    m->layers.resize(m->L);
    for (int li = 0; li < m->L && li < 128; li++)
        if ((c.mamba_layer_mask[li >> 6] >> (li & 63)) & 1) { m->layers[li].mamba = true; m->layers[li].mamba_idx = m->n_mamba++; }
    if (m->n_mamba) {
        m->mEH = c.mamba_expand * c.hidden; m->m_nh = c.mamba_num_heads; m->m_ss = c.mamba_state_size; m->m_ng = c.mamba_n_groups;
        m->m_K = c.mamba_conv_kernel; m->m_hd = c.mamba_head_dim > 0 ? c.mamba_head_dim : m->mEH / m->m_nh;
        m->mConv = m->mEH + 2 * m->m_ng * m->m_ss; m->mP = m->mEH + m->mConv + m->m_nh;
    }
    if (m->opts.max_batch_tokens <= 0) m->opts.max_batch_tokens = c.max_seq_len;
    *out = m;
    return NVL_OK;
    } catch (const HipError& e) {
        std::string s = std::string("nvl_create: HIP error: ") + hipGetErrorString(e.code);
        delete m;
        return fail(nullptr, NVL_ERR_HIP, s);
    } catch (const std::exception& e) {
        std::string s = e.what();
        delete m;
        return fail(nullptr, NVL_ERR_INVALID, s);
    }
}

extern "C" void nvl_destroy(nvl_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (int k = 0; k < NVL_T_COUNT; k++) dfree(m->g[k].p);
    for (auto& l : m->layers) {
        for (int k = 0; k < NVL_T_COUNT; k++) dfree(l.t[k].p);
        dfree(l.w_qkv); dfree(l.b_qkv); dfree(l.w1); dfree(l.moe_in); dfree(l.moe_out_cat);
    }
    if (m->lm_head && m->lm_head != m->g[NVL_T_TOK_EMB].p && m->lm_head != m->g[NVL_T_LM_HEAD].p) dfree(m->lm_head);
    dfree(m->rope_cos); dfree(m->rope_sin); dfree(m->kcache); dfree(m->vcache);
    dfree(m->x); dfree(m->xn); dfree(m->qkv); dfree(m->q); dfree(m->attn_out); dfree(m->attn_part); dfree(m->attn_part_cnt); dfree(m->hbuf); dfree(m->h2);
    dfree(m->xn_last); dfree(m->logits); dfree(m->argmax_dev); dfree(m->argmax_pval); dfree(m->argmax_pidx); dfree(m->router_logits); dfree(m->expert_ids);
    dfree(m->expert_w); dfree(m->seg_start); dfree(m->moe_counts); dfree(m->moe_cursor); dfree(m->moe_tile_map); dfree(m->moe_n_mtiles); dfree(m->perm_token); dfree(m->slot_of); dfree(m->moe_eo); dfree(m->moe_xg);
    dfree(m->meta_dev); dfree(m->hidden); dfree(m->sk_part); dfree(m->tp_part); dfree(m->ring); dfree(m->rs_part);
    for (int r = 0; r < 8; r++) if (m->p2p_peer[r] && m->p2p_peer[r] != m->p2p_buf) (void)hipIpcCloseMemHandle(m->p2p_peer[r]);
    dfree(m->p2p_buf);
    dfree(m->moe_gate); dfree(m->moe_hall); dfree(m->moe_part);
    clear_graphs(m);
    if (m->am_host) (void)hipHostFree(m->am_host);
    dfree(m->ring_pos0);
    dfree(m->conv_tail); dfree(m->stamp_buf); dfree(m->ssd_z); dfree(m->ssd_decay);
    dfree(m->ssm_state); dfree(m->mproj); dfree(m->mxbc); dfree(m->mdelta); dfree(m->my); dfree(m->myn);
    free_sample_bufs(m->samp);
    dfree(m->samp_hist); dfree(m->samp_hist_len); dfree(m->samp_u_steps);
    if (m->tp_comm) (void)ncclCommDestroy((ncclComm_t)m->tp_comm);
    if (m->tp_local) {
        bool last;
        { std::lock_guard<std::mutex> lk(m->tp_local->mu); last = --m->tp_local->refs == 0; }
        if (last) { dfree(m->tp_local->scratch); delete m->tp_local; }
    }
    if (m->meta_host) (void)hipHostFree(m->meta_host);
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

// =================================================================================================
// weights
// =================================================================================================
namespace {

// canonical device copy: 2-D -> [rows_pad(W_ROW_PAD)][K] in weight dtype; 1-D -> fp32.
// `sl` selects the sub-block of the uploaded tensor that lands in rows [dn0, dn0+N) (the whole tensor for
// replicated kinds, this rank's shard for tensor-parallel kinds); total_rows sizes the destination.
void upload_2d(nvl_model* m, DevTensor* t, const void* data, int dtype, int64_t N, int64_t K, bool src_is_in_out,
               Slice2D sl, int64_t total_rows, bool first_piece) {
    const size_t esz = dtype == NVL_DTYPE_F32 ? 4 : 2;
    const int64_t src_elems = sl.SN * sl.SK;
    void* raw = dmalloc_bytes(src_elems * (int64_t)esz);
    NVL_HIP(hipMemcpyAsync(raw, data, (size_t)src_elems * esz, hipMemcpyDefault, m->stream));
    const int64_t Npad = round_up(total_rows, W_ROW_PAD);
    if (first_piece) {
        dfree(t->p);
        t->p = dmalloc_bytes(Npad * K * (int64_t)m->wsize);
        NVL_HIP(hipMemsetAsync(t->p, 0, (size_t)(Npad * K) * m->wsize, m->stream));
        m->stats.weight_bytes += (double)Npad * K * m->wsize;
    }
    dim3 grid((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32));
    if (m->f32)
        hipLaunchKernelGGL((convert_2d_kernel<float, false>), grid, dim3(256), 0, m->stream, raw, dtype,
                           src_is_in_out ? 1 : 0, (float*)t->p, N, K, sl);
    else   // bf16 weights (and the embedding tables) live in the fragment-major layout
        hipLaunchKernelGGL((convert_2d_kernel<bf16_t, true>), grid, dim3(256), 0, m->stream, raw, dtype,
                           src_is_in_out ? 1 : 0, (bf16_t*)t->p, N, K, sl);
    NVL_HIP(hipGetLastError());
    NVL_HIP(hipStreamSynchronize(m->stream));
    dfree(raw);
    t->rows = total_rows; t->cols = K; t->rows_pad = Npad;
}

void upload_1d(nvl_model* m, DevTensor* t, const void* data, int dtype, int64_t n_src, int64_t off, int64_t n) {
    const size_t esz = dtype == NVL_DTYPE_F32 ? 4 : 2;
    void* raw = dmalloc_bytes(n_src * (int64_t)esz);
    NVL_HIP(hipMemcpyAsync(raw, data, (size_t)n_src * esz, hipMemcpyDefault, m->stream));
    dfree(t->p);
    t->p = dmalloc<float>(n);
    hipLaunchKernelGGL(convert_1d_kernel, dim3(cdiv(n, 256)), dim3(256), 0, m->stream, raw, dtype, (float*)t->p, n, off);
    NVL_HIP(hipGetLastError());
    NVL_HIP(hipStreamSynchronize(m->stream));
    dfree(raw);
    t->rows = n; t->cols = 1; t->rows_pad = n;
    m->stats.weight_bytes += (double)n * 4;
}

// FULL (unsharded) logical sizes of a 2-D weight kind, as the host holds it
int64_t full_out(const nvl_model* m, int kind) {
    switch (kind) {
        case NVL_T_LM_HEAD: return m->V;
        case NVL_T_WQ: return (int64_t)m->nH_full * m->hd;
        case NVL_T_WK: case NVL_T_WV: return (int64_t)m->nKV_full * m->hd;
        case NVL_T_WKV: return 2 * m->hd;
        case NVL_T_WO: return m->H;
        case NVL_T_W1: return m->cfg.activation_type == NVL_ACT_SWIGLU ? 2 * (int64_t)m->F_full : m->F_full;
        case NVL_T_W2: return m->H;
        case NVL_T_ROUTER: return m->cfg.num_experts;
        case NVL_T_MAMBA_IN_PROJ: return m->mP;
        case NVL_T_MAMBA_OUT_PROJ: return m->H;
        default: return -1;
    }
}
int64_t full_in(const nvl_model* m, int kind) {
    switch (kind) {
        case NVL_T_WO: return (int64_t)m->nH_full * m->hd;
        case NVL_T_W2: return m->F_full;
        case NVL_T_MAMBA_OUT_PROJ: return m->mEH;
        default: return m->H;
    }
}

}  // namespace

extern "C" int nvl_upload_tensor(nvl_model* m, int kind, int layer, const void* data, int dtype,
                                 int64_t rows, int64_t cols, int layout) {
    if (!m || !data) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: null argument");
    if (m->finalized) return fail(m, NVL_ERR_STATE, "nvl_upload_tensor: model already finalized");
    if (dtype < 0 || dtype > 2) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: unsupported dtype");
    DevTensor* t = tensor_slot(m, kind, layer);
    if (!t) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: bad kind or layer");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    char buf[256];
    const int r = m->tp_rank;
    if (is_1d(kind)) {
        const int64_t n = rows * (cols > 0 ? cols : 1);
        int64_t off = 0, cnt = n;
        {   // the kernels read exactly this many elements: a short norm / bias tensor would be an out-of-bounds read
            int64_t want = m->H;
            if (kind == NVL_T_BQ) want = (int64_t)m->nH_full * m->hd;
            else if (kind == NVL_T_BK || kind == NVL_T_BV) want = (int64_t)m->nKV_full * m->hd;
            else if (kind == NVL_T_B1) want = m->F_full;
            else if (kind == NVL_T_MAMBA_CONV_W) want = (int64_t)m->mConv * m->m_K;
            else if (kind == NVL_T_MAMBA_CONV_B) want = m->mConv;
            else if (kind == NVL_T_MAMBA_A_LOG || kind == NVL_T_MAMBA_D || kind == NVL_T_MAMBA_DT_BIAS) want = m->m_nh;
            else if (kind == NVL_T_MAMBA_NORM) want = m->mEH;
            if (kind >= NVL_T_MAMBA_IN_PROJ && !m->n_mamba) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: the model has no Mamba2 layers");
            if (rows <= 0 || n != want) {
                snprintf(buf, sizeof buf, "1-D tensor kind %d has %lld elements, the model needs %lld", kind, (long long)n,
                         (long long)want);
                return fail(m, NVL_ERR_INVALID, buf);
            }
        }
        if (m->tp > 1) {   // column-parallel biases follow their projection's shard; row-parallel biases stay on rank 0
            if (kind == NVL_T_BQ) { cnt = (int64_t)m->nH * m->hd; off = r * cnt; }
            else if (kind == NVL_T_BK || kind == NVL_T_BV) { cnt = (int64_t)m->nKV * m->hd; off = r * cnt; }
            else if (kind == NVL_T_B1) { cnt = m->F; off = r * cnt; }
            else if ((kind == NVL_T_BO || kind == NVL_T_B2) && r != 0) return NVL_OK;
            if (off + cnt > n) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: bias shorter than the full (unsharded) size");
        }
        upload_1d(m, t, data, dtype, n, off, cnt);
        return NVL_OK;
    }
    if (kind == NVL_T_TOK_EMB || kind == NVL_T_POS_EMB) {
        const int64_t want_rows = kind == NVL_T_TOK_EMB ? m->V : rows;
        if (cols != m->H || rows != want_rows) {
            snprintf(buf, sizeof buf, "embedding shape [%lld,%lld] does not match [%lld,%d]", (long long)rows,
                     (long long)cols, (long long)want_rows, m->H);
            return fail(m, NVL_ERR_INVALID, buf);
        }
        upload_2d(m, t, data, dtype, rows, cols, false, Slice2D{rows, cols, 0, 0, 0}, rows, true);
        return NVL_OK;
    }
    if (kind == NVL_T_MOE_IN || kind == NVL_T_MOE_OUT) {
        const int64_t E = m->cfg.num_experts;
        const int64_t out = kind == NVL_T_MOE_IN ? 2 * (int64_t)m->F : m->H;
        const int64_t in = kind == NVL_T_MOE_IN ? m->H : m->F;
        if (rows != E * out || cols != in) {
            snprintf(buf, sizeof buf, "MoE tensor shape [%lld,%lld] does not match [E*%lld,%lld]", (long long)rows,
                     (long long)cols, (long long)out, (long long)in);
            return fail(m, NVL_ERR_INVALID, buf);
        }
        upload_2d(m, t, data, dtype, rows, cols, false, Slice2D{rows, cols, 0, 0, 0}, rows, true);
        return NVL_OK;
    }
    if (kind >= NVL_T_MAMBA_IN_PROJ && !m->n_mamba) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: the model has no Mamba2 layers");
    const int64_t SN = full_out(m, kind), SK = full_in(m, kind);
    if (SN < 0) return fail(m, NVL_ERR_INVALID, "nvl_upload_tensor: kind is not a 2-D weight");
    const int64_t got_in = layout == NVL_LAYOUT_IN_OUT ? rows : cols;
    const int64_t got_out = layout == NVL_LAYOUT_IN_OUT ? cols : rows;
    if (got_in != SK || got_out != SN) {   // reference: panic "incompatible shapes" tensor.go:67
        snprintf(buf, sizeof buf, "weight kind %d: got [in=%lld,out=%lld], model needs [in=%lld,out=%lld]", kind,
                 (long long)got_in, (long long)got_out, (long long)SK, (long long)SN);
        return fail(m, NVL_ERR_INVALID, buf);
    }
    const bool io = layout == NVL_LAYOUT_IN_OUT;
    if (m->tp == 1) {
        upload_2d(m, t, data, dtype, SN, SK, io, Slice2D{SN, SK, 0, 0, 0}, SN, true);
        return NVL_OK;
    }
    // ---- tensor-parallel shard of the uploaded FULL tensor ----
    const int64_t hd = m->hd;
    switch (kind) {
        case NVL_T_WQ: { const int64_t nl = m->nH * hd;     // column parallel: this rank's query heads
            upload_2d(m, t, data, dtype, nl, SK, io, Slice2D{SN, SK, r * nl, 0, 0}, nl, true); break; }
        case NVL_T_WK: case NVL_T_WV: { const int64_t nl = m->nKV * hd;
            upload_2d(m, t, data, dtype, nl, SK, io, Slice2D{SN, SK, r * nl, 0, 0}, nl, true); break; }
        case NVL_T_WO: { const int64_t kl = m->nH * hd;     // row parallel: K slice = this rank's heads
            upload_2d(m, t, data, dtype, SN, kl, io, Slice2D{SN, SK, 0, r * kl, 0}, SN, true); break; }
        case NVL_T_W1: { const int64_t fl = m->F;
            if (m->cfg.activation_type == NVL_ACT_SWIGLU) {   // [gate F | up F] -> [gate_loc | up_loc]
                upload_2d(m, t, data, dtype, fl, SK, io, Slice2D{SN, SK, r * fl, 0, 0}, 2 * fl, true);
                upload_2d(m, t, data, dtype, fl, SK, io, Slice2D{SN, SK, m->F_full + r * fl, 0, fl}, 2 * fl, false);
            } else {
                upload_2d(m, t, data, dtype, fl, SK, io, Slice2D{SN, SK, r * fl, 0, 0}, fl, true);
            }
            break; }
        case NVL_T_W2: { const int64_t kl = m->F;           // row parallel
            upload_2d(m, t, data, dtype, SN, kl, io, Slice2D{SN, SK, 0, r * kl, 0}, SN, true); break; }
        default:                                            // LM head etc.: replicated
            upload_2d(m, t, data, dtype, SN, SK, io, Slice2D{SN, SK, 0, 0, 0}, SN, true);
    }
    return NVL_OK;
    NVL_CATCH(m)
}

// splitGPT2QKV (generic_loader.go:674-702): c_attn [H, 3H] is [in,out]; columns split Q|K|V.
extern "C" int nvl_upload_gpt2_qkv(nvl_model* m, int layer, const float* w, const float* b) {
    if (!m || !w) return fail(m, NVL_ERR_INVALID, "nvl_upload_gpt2_qkv: null argument");
    NVL_TRY(m)
    const int64_t H = m->H;
    std::vector<float> part((size_t)H * H);
    const int kinds[3] = {NVL_T_WQ, NVL_T_WK, NVL_T_WV};
    const int bk[3] = {NVL_T_BQ, NVL_T_BK, NVL_T_BV};
    for (int s = 0; s < 3; s++) {
        for (int64_t r = 0; r < H; r++)
            memcpy(&part[(size_t)(r * H)], w + r * 3 * H + s * H, (size_t)H * sizeof(float));
        int rc = nvl_upload_tensor(m, kinds[s], layer, part.data(), NVL_DTYPE_F32, H, H, NVL_LAYOUT_IN_OUT);
        if (rc) return rc;
        if (b) {   // generic_loader.go:418-427
            rc = nvl_upload_tensor(m, bk[s], layer, b + s * H, NVL_DTYPE_F32, H, 1, 0);
            if (rc) return rc;
        }
    }
    return NVL_OK;
    NVL_CATCH(m)
}

// splitFalconQKV + combineMQAKV (generic_loader.go:705-765): rows of [H, (nH+2)*hd] hold
// [Q_0 .. Q_{nH-1} | K | V] chunks of hd.
extern "C" int nvl_upload_falcon_qkv(nvl_model* m, int layer, const float* qkv) {
    if (!m || !qkv) return fail(m, NVL_ERR_INVALID, "nvl_upload_falcon_qkv: null argument");
    NVL_TRY(m)
    const int64_t H = m->H, nH = m->nH, hd = m->hd, roww = (nH + 2) * hd;
    std::vector<float> q((size_t)(H * nH * hd)), kv((size_t)(H * 2 * hd));
    for (int64_t r = 0; r < H; r++) {
        memcpy(&q[(size_t)(r * nH * hd)], qkv + r * roww, (size_t)(nH * hd) * sizeof(float));
        memcpy(&kv[(size_t)(r * 2 * hd)], qkv + r * roww + nH * hd, (size_t)(2 * hd) * sizeof(float));
    }
    int rc = nvl_upload_tensor(m, NVL_T_WQ, layer, q.data(), NVL_DTYPE_F32, H, nH * hd, NVL_LAYOUT_IN_OUT);
    if (rc) return rc;
    return nvl_upload_tensor(m, NVL_T_WKV, layer, kv.data(), NVL_DTYPE_F32, H, 2 * hd, NVL_LAYOUT_IN_OUT);
    NVL_CATCH(m)
}

namespace {

// dst row r = src row idx[r] (or zeros).  idx must move whole 16-row groups together (it does: the
// SwiGLU interleave works in blocks of 16), because in the fragment-major bf16 layout the unit that
// can be relocated is a 16-row n-tile (16*K contiguous elements).
void gather_rows(nvl_model* m, const void* src, const std::vector<int32_t>& idx, void* dst, int64_t K) {
    std::vector<int32_t> use = idx;
    int64_t unit = K;
    if (!m->f32) {
        if (idx.size() % 16) throw std::runtime_error("gather_rows: row count must be a multiple of 16");
        use.resize(idx.size() / 16);
        for (size_t t = 0; t < use.size(); t++) {
            const int32_t r0 = idx[t * 16];
            for (int c = 1; c < 16; c++)
                if ((r0 < 0) != (idx[t * 16 + c] < 0) || (r0 >= 0 && (idx[t * 16 + c] != r0 + c || (r0 & 15))))
                    throw std::runtime_error("gather_rows: permutation does not preserve 16-row tiles");
            use[t] = r0 < 0 ? -1 : r0 / 16;
        }
        unit = 16 * K;
    }
    int32_t* didx = dmalloc<int32_t>((int64_t)use.size());
    NVL_HIP(hipMemcpyAsync(didx, use.data(), use.size() * 4, hipMemcpyHostToDevice, m->stream));
    if (m->f32)
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3((unsigned)use.size()), dim3(256), 0, m->stream,
                           (const float*)src, didx, (float*)dst, unit);
    else
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3((unsigned)use.size()), dim3(256), 0, m->stream,
                           (const bf16_t*)src, didx, (bf16_t*)dst, unit);
    NVL_HIP(hipGetLastError());
    NVL_HIP(hipStreamSynchronize(m->stream));
    dfree(didx);
}

// interleave [gate F rows | up F rows] into blocks of 16: [g0..15 | u0..15 | g16..31 | ...]
std::vector<int32_t> swiglu_interleave(int64_t F, int64_t src_base) {
    std::vector<int32_t> idx((size_t)(2 * F));
    for (int64_t b = 0; b < F / 16; b++)
        for (int c = 0; c < 16; c++) {
            idx[(size_t)(32 * b + c)] = (int32_t)(src_base + 16 * b + c);
            idx[(size_t)(32 * b + 16 + c)] = (int32_t)(src_base + F + 16 * b + c);
        }
    return idx;
}

}  // namespace

extern "C" int nvl_finalize(nvl_model* m) {
    if (!m) return NVL_ERR_INVALID;
    if (m->finalized) return fail(m, NVL_ERR_STATE, "nvl_finalize: already finalized");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const nvl_model_config& c = m->cfg;
    const int64_t H = m->H, hd = m->hd;
    auto need = [&](bool ok, const char* what) { if (!ok) throw std::runtime_error(std::string("nvl_finalize: missing tensor: ") + what); };
    need(m->g[NVL_T_TOK_EMB].present(), "token embedding");
    need(m->g[NVL_T_FINAL_NORM_W].present(), "final norm weight");
    if (c.norm_type == NVL_NORM_LAYER) need(m->g[NVL_T_FINAL_NORM_B].present(), "final norm bias (LayerNorm)");

    for (int li = 0; li < m->L; li++) {
        LayerW& l = m->layers[li];
        need(l.t[NVL_T_ATTN_NORM_W].present(), "attention/input norm weight");
        if (l.mamba) {       // Mamba2 block: in/out projections + SSM parameters, then the shared MLP (loadMamba2 + loadFFN)
            need(l.t[NVL_T_MAMBA_IN_PROJ].present() && l.t[NVL_T_MAMBA_OUT_PROJ].present() && l.t[NVL_T_MAMBA_CONV_W].present(),
                 "Mamba2 in_proj / out_proj / conv1d weight");
            need(l.t[NVL_T_FFN_NORM_W].present() && l.t[NVL_T_W1].present() && l.t[NVL_T_W2].present(), "Mamba2 block: FFN norm / shared MLP");
            const bool swiglu = c.activation_type == NVL_ACT_SWIGLU;
            l.n1 = swiglu ? 2 * m->F : m->F;
            if (swiglu && !m->f32) {
                auto idx = swiglu_interleave(m->F, 0);
                const int64_t np = round_up(l.n1, W_ROW_PAD);
                idx.resize((size_t)np, -1);
                l.w1 = dmalloc_bytes(np * H * (int64_t)m->wsize);
                gather_rows(m, l.t[NVL_T_W1].p, idx, l.w1, H);
                dfree(l.t[NVL_T_W1].p); l.t[NVL_T_W1].p = nullptr;
            } else {
                l.w1 = l.t[NVL_T_W1].p; l.t[NVL_T_W1].p = nullptr;
            }
            continue;
        }
        need(l.t[NVL_T_WQ].present() && l.t[NVL_T_WO].present(), "attention Q/O projection");
        const bool mqa = c.attention_type == NVL_ATTN_MQA;
        if (mqa) need(l.t[NVL_T_WKV].present(), "MQA fused KV projection");
        else need(l.t[NVL_T_WK].present() && l.t[NVL_T_WV].present(), "K/V projection");
        if (c.block_style == NVL_BLOCK_SEQUENTIAL) need(l.t[NVL_T_FFN_NORM_W].present(), "FFN norm weight");
        // ---- fused QKV [n_qkv_pad][H]: rows Q | K | V (canonical rows are contiguous: plain copies)
        const int64_t nq = (int64_t)m->nH * hd, nkv = (int64_t)m->nKV * hd;
        const int64_t npad = round_up(m->n_qkv, W_ROW_PAD);
        l.w_qkv = dmalloc_bytes(npad * H * (int64_t)m->wsize);
        NVL_HIP(hipMemsetAsync(l.w_qkv, 0, (size_t)(npad * H) * m->wsize, m->stream));
        char* dst = (char*)l.w_qkv;
        const size_t rowb = (size_t)H * m->wsize;
        NVL_HIP(hipMemcpyAsync(dst, l.t[NVL_T_WQ].p, nq * rowb, hipMemcpyDeviceToDevice, m->stream));
        if (mqa) {
            NVL_HIP(hipMemcpyAsync(dst + nq * rowb, l.t[NVL_T_WKV].p, 2 * hd * rowb, hipMemcpyDeviceToDevice, m->stream));
        } else {
            NVL_HIP(hipMemcpyAsync(dst + nq * rowb, l.t[NVL_T_WK].p, nkv * rowb, hipMemcpyDeviceToDevice, m->stream));
            NVL_HIP(hipMemcpyAsync(dst + (nq + nkv) * rowb, l.t[NVL_T_WV].p, nkv * rowb, hipMemcpyDeviceToDevice, m->stream));
        }
        l.n_qkv = m->n_qkv;
        if (l.t[NVL_T_BQ].present() || l.t[NVL_T_BK].present() || l.t[NVL_T_BV].present()) {
            l.b_qkv = dmalloc<float>(m->n_qkv);
            NVL_HIP(hipMemsetAsync(l.b_qkv, 0, (size_t)m->n_qkv * 4, m->stream));
            if (l.t[NVL_T_BQ].present()) NVL_HIP(hipMemcpyAsync(l.b_qkv, l.t[NVL_T_BQ].p, nq * 4, hipMemcpyDeviceToDevice, m->stream));
            if (l.t[NVL_T_BK].present()) NVL_HIP(hipMemcpyAsync(l.b_qkv + nq, l.t[NVL_T_BK].p, nkv * 4, hipMemcpyDeviceToDevice, m->stream));
            if (l.t[NVL_T_BV].present()) NVL_HIP(hipMemcpyAsync(l.b_qkv + nq + nkv, l.t[NVL_T_BV].p, nkv * 4, hipMemcpyDeviceToDevice, m->stream));
        }
        NVL_HIP(hipStreamSynchronize(m->stream));
        for (int k : {NVL_T_WQ, NVL_T_WK, NVL_T_WV, NVL_T_WKV}) { dfree(l.t[k].p); l.t[k].p = nullptr; }

        // ---- FFN / MoE
        if (c.use_moe) {
            need(l.t[NVL_T_ROUTER].present() && l.t[NVL_T_MOE_IN].present() && l.t[NVL_T_MOE_OUT].present(), "MoE tensors");
            if (!m->f32) {   // interleave gate/up rows per expert for the fused SwiGLU epilogue
                const int64_t E = c.num_experts, I = m->F;
                std::vector<int32_t> idx;
                idx.reserve((size_t)(E * 2 * I));
                for (int64_t e = 0; e < E; e++) {
                    auto one = swiglu_interleave(I, e * 2 * I);
                    idx.insert(idx.end(), one.begin(), one.end());
                }
                l.moe_in = dmalloc_bytes(round_up(E * 2 * I, W_ROW_PAD) * H * (int64_t)m->wsize);   // (+ zero pad rows: the decode kernels read whole tiles)
                NVL_HIP(hipMemsetAsync(l.moe_in, 0, (size_t)(round_up(E * 2 * I, W_ROW_PAD) * H) * m->wsize, m->stream));
                gather_rows(m, l.t[NVL_T_MOE_IN].p, idx, l.moe_in, H);
                dfree(l.t[NVL_T_MOE_IN].p); l.t[NVL_T_MOE_IN].p = nullptr;
                if (I % 32 == 0 && H % 16 == 0 && E <= 64 && E / moe_down_slices((int)E) <= 16) {
                    // decode form: the experts' down matrices side by side along K
                    const int64_t hp = round_up(H, W_ROW_PAD);
                    l.moe_out_cat = dmalloc_bytes(hp * E * I * 2);
                    NVL_HIP(hipMemsetAsync(l.moe_out_cat, 0, (size_t)(hp * E * I) * 2, m->stream));
                    hipLaunchKernelGGL(moe_cat_down_kernel, dim3((unsigned)(H / 16), (unsigned)E), dim3(256), 0, m->stream,
                                       (const bf16_t*)l.t[NVL_T_MOE_OUT].p, (bf16_t*)l.moe_out_cat, (int)H, (int)I, (int)E);
                    NVL_HIP(hipGetLastError());
                    NVL_HIP(hipStreamSynchronize(m->stream));
                }
            } else {
                l.moe_in = l.t[NVL_T_MOE_IN].p; l.t[NVL_T_MOE_IN].p = nullptr;
            }
        } else {
            need(l.t[NVL_T_W1].present() && l.t[NVL_T_W2].present(), "FFN weights");
            const bool swiglu = c.activation_type == NVL_ACT_SWIGLU;
            l.n1 = swiglu ? 2 * m->F : m->F;
            if (swiglu && !m->f32) {
                auto idx = swiglu_interleave(m->F, 0);
                const int64_t np = round_up(l.n1, W_ROW_PAD);
                idx.resize((size_t)np, -1);
                l.w1 = dmalloc_bytes(np * H * (int64_t)m->wsize);
                gather_rows(m, l.t[NVL_T_W1].p, idx, l.w1, H);
                dfree(l.t[NVL_T_W1].p); l.t[NVL_T_W1].p = nullptr;
            } else {
                l.w1 = l.t[NVL_T_W1].p; l.t[NVL_T_W1].p = nullptr;
            }
        }
    }
    // ---- LM head: tied (or absent) => the token embedding itself, already [Vpad][H] K-contiguous
    if (!c.tied_embedding && m->g[NVL_T_LM_HEAD].present()) m->lm_head = m->g[NVL_T_LM_HEAD].p;
    else m->lm_head = m->g[NVL_T_TOK_EMB].p;   // generic_loader.go:255-259

    // ---- RoPE tables in fp64 on the host, exactly as NewRoPECache (rope.go:18-50)
    bool rope = false; double base = c.rope_base;
    if (c.attention_type == NVL_ATTN_MQA) { rope = true; base = 10000.0; }           // mqa.go:35
    else if (c.attention_type == NVL_ATTN_GQA && c.position_type == NVL_POS_ROPE) rope = true;
    if (rope) {
        const int half = (int)hd / 2;
        std::vector<float> ct((size_t)c.max_seq_len * hd), stt((size_t)c.max_seq_len * hd);
        for (int pos = 0; pos < c.max_seq_len; pos++)
            for (int i = 0; i < half; i++) {
                const double freq = 1.0 / std::pow(base, (double)(2 * i) / (double)hd);
                const double ang = (double)pos * freq;
                const float cv = (float)std::cos(ang), sv = (float)std::sin(ang);
                ct[(size_t)pos * hd + i] = cv; ct[(size_t)pos * hd + half + i] = cv;
                stt[(size_t)pos * hd + i] = sv; stt[(size_t)pos * hd + half + i] = sv;
            }
        m->rope_cos = dmalloc<float>((int64_t)ct.size());
        m->rope_sin = dmalloc<float>((int64_t)stt.size());
        NVL_HIP(hipMemcpy(m->rope_cos, ct.data(), ct.size() * 4, hipMemcpyHostToDevice));
        NVL_HIP(hipMemcpy(m->rope_sin, stt.data(), stt.size() * 4, hipMemcpyHostToDevice));
    }

    // ---- KV slabs: [slot][layer][kv head][Tmax][hd]; zero-filled so stale tiles hold finite values
    m->layer_stride = (int64_t)m->nKV * m->Tmax * hd;
    m->slot_stride = m->layer_stride * m->L;
    const int64_t kv_elems = m->slot_stride * m->num_blocks;
    m->kcache = dmalloc_bytes(kv_elems * (int64_t)m->wsize);
    m->vcache = dmalloc_bytes(kv_elems * (int64_t)m->wsize);
    NVL_HIP(hipMemsetAsync(m->kcache, 0, (size_t)kv_elems * m->wsize, m->stream));
    NVL_HIP(hipMemsetAsync(m->vcache, 0, (size_t)kv_elems * m->wsize, m->stream));
    m->slot_len.assign((size_t)m->opts.max_seqs, 0);
    m->slot_tick.assign((size_t)m->opts.max_seqs, 0);
    m->slot_pin.assign((size_t)m->opts.max_seqs, 0);
    m->free_slots.clear();
    for (int s = m->opts.max_seqs - 1; s >= 0; s--) m->free_slots.push_back(s);

    // ---- workspaces
    const int64_t Mmax = m->opts.max_batch_tokens, S = m->opts.max_seqs;
    const int64_t Mp = round_up(Mmax, 64);   // GEMM operands are whole 16-row tiles; decode reads up to 64 rows
    const int64_t qw = (int64_t)m->nH * hd;
    m->x = dmalloc<float>(Mmax * H);
    m->xn = dmalloc_bytes(Mp * H * (int64_t)m->wsize);
    m->qkv = dmalloc<float>(Mmax * m->n_qkv);
    m->q = dmalloc_bytes(Mmax * qw * (int64_t)m->wsize);
    m->attn_out = dmalloc_bytes(Mp * qw * (int64_t)m->wsize);
    if (!m->f32 && (m->hd == 64 || m->hd == 128)) {     // partial results of a split decode attention (attn.h)
        m->attn_part = dmalloc<float>((int64_t)ATTN_SPLIT_MAX_PAIRS * ATTN_SPLIT_MAX * (m->hd == 64 ? attn_part_floats<64>() : attn_part_floats<128>()));
        m->attn_part_cnt = dmalloc<int32_t>(ATTN_SPLIT_MAX_PAIRS);
        NVL_HIP(hipMemset(m->attn_part_cnt, 0, ATTN_SPLIT_MAX_PAIRS * sizeof(int32_t)));
    }
    const int64_t k = c.use_moe ? c.num_experts_per_tok : 1;
    m->hbuf = dmalloc_bytes(round_up(Mmax * k, 64) * m->F * (int64_t)m->wsize);
    NVL_HIP(hipMemsetAsync(m->xn, 0, (size_t)(Mp * H) * m->wsize, m->stream));
    NVL_HIP(hipMemsetAsync(m->attn_out, 0, (size_t)(Mp * qw) * m->wsize, m->stream));
    NVL_HIP(hipMemsetAsync(m->hbuf, 0, (size_t)(round_up(Mmax * k, 64) * m->F) * m->wsize, m->stream));
    if (m->f32 && (c.activation_type == NVL_ACT_SWIGLU || c.use_moe)) m->h2 = dmalloc<float>(Mmax * k * 2 * m->F);
    m->xn_last = dmalloc_bytes(round_up(S, 64) * H * (int64_t)m->wsize);
    NVL_HIP(hipMemsetAsync(m->xn_last, 0, (size_t)(round_up(S, 64) * H) * m->wsize, m->stream));
    m->logit_rows = S;
    m->logits = dmalloc<float>(S * (int64_t)m->Vpad);
    m->argmax_dev = dmalloc<int32_t>(std::max<int64_t>(S, Mmax));
    m->argmax_pval = dmalloc<float>(std::max<int64_t>(S, Mmax) * cdiv(m->V, ARGMAX_CHUNK));
    m->argmax_pidx = dmalloc<int32_t>(std::max<int64_t>(S, Mmax) * cdiv(m->V, ARGMAX_CHUNK));
    if (c.use_moe) {
        m->router_logits = dmalloc<float>(Mmax * 128);
        m->expert_ids = dmalloc<int32_t>(Mmax * k);
        m->expert_w = dmalloc<float>(Mmax * k);
        m->seg_start = dmalloc<int32_t>(c.num_experts + 1);
        m->moe_counts = dmalloc<int32_t>(c.num_experts);
        m->moe_cursor = dmalloc<int32_t>(c.num_experts);
        m->moe_tile_map = dmalloc<int32_t>(4 * (cdiv(Mmax * k, 128) + c.num_experts));
        m->moe_n_mtiles = dmalloc<int32_t>(4);
        m->perm_token = dmalloc<int32_t>(Mmax * k);
        m->slot_of = dmalloc<int32_t>(Mmax * k);
        m->moe_eo = dmalloc<float>(Mmax * k * H);
        if (!m->f32) m->moe_xg = dmalloc_bytes(round_up(Mmax * k, 64) * H * 2);
        if (!m->f32) {
            m->moe_gate = dmalloc<float>(64 * (int64_t)c.num_experts);
            m->moe_hall = dmalloc_bytes(64 * (int64_t)c.num_experts * m->F * 2);
            NVL_HIP(hipMemsetAsync(m->moe_hall, 0, (size_t)(64 * (int64_t)c.num_experts * m->F) * 2, m->stream));
            m->moe_part = dmalloc<float>((int64_t)MOE_DOWN_SLICES * 64 * H);
        }
    }
    if (m->n_mamba) {
        m->ssm_layer_stride = (int64_t)m->m_nh * m->m_hd * m->m_ss;
        m->ssm_slot_stride = m->ssm_layer_stride * m->n_mamba;
        m->ssm_state = dmalloc<float>(m->ssm_slot_stride * S);
        NVL_HIP(hipMemsetAsync(m->ssm_state, 0, (size_t)(m->ssm_slot_stride * S) * 4, m->stream));
        if (m->m_K > 1) {     // raw xBC rows of a chunked prefill's last K-1 tokens, per (slot, Mamba2 layer)
            m->conv_tail_layer_stride = (int64_t)(m->m_K - 1) * m->mConv;
            m->conv_tail_slot_stride = m->conv_tail_layer_stride * m->n_mamba;
            m->conv_tail = dmalloc<float>(m->conv_tail_slot_stride * S);
            NVL_HIP(hipMemsetAsync(m->conv_tail, 0, (size_t)(m->conv_tail_slot_stride * S) * 4, m->stream));
        }
        m->mproj = dmalloc<float>(Mmax * m->mP);
        m->mxbc = dmalloc<float>(Mmax * m->mConv);
        m->mdelta = dmalloc<float>(Mmax * m->m_nh);
        m->my = dmalloc<float>(Mmax * m->mEH);
        m->myn = dmalloc_bytes(Mp * m->mEH * (int64_t)m->wsize);
        NVL_HIP(hipMemsetAsync(m->myn, 0, (size_t)(Mp * m->mEH) * m->wsize, m->stream));
    }
    // split-K partial slices of the residual projections: decode (<= 64 rows) and mid-size batches (<= SK_TILE_MAX_M rows)
    m->sk_rows = (int)std::min<int64_t>(std::max<int64_t>(Mmax, 64), SK_TILE_MAX_M);
    if (!m->f32) m->sk_part = dmalloc<float>((int64_t)std::max(m->sk_max_slices, SK_TILE_MAX_SLICES) * m->sk_rows * H);
    if (!m->f32) m->rs_part = dmalloc<float>((int64_t)cdiv(H, 8) * DEFER_MAX_M);      // x^2 partials per 16 (half tiles: 8) columns
    if (m->tp > 1 || m->tp_force) m->tp_part = dmalloc<float>(Mmax * H);
    m->meta_ints = 3 * Mmax + 5 * S + m->table_cap + 16;
    NVL_HIP(hipHostMalloc((void**)&m->meta_host, (size_t)m->meta_ints * 4, hipHostMallocDefault));
    m->meta_dev = dmalloc<int32_t>(m->meta_ints);
    NVL_HIP(hipHostMalloc((void**)&m->am_host, (size_t)std::max<int64_t>(S, Mmax) * 4, hipHostMallocDefault));
    m->ring_pos0 = dmalloc<int32_t>(S);
    NVL_HIP(hipStreamSynchronize(m->stream));   // every memset/copy above ran on the model's own (non-blocking) stream
    m->finalized = true;
    return NVL_OK;
    NVL_CATCH(m)
}

// =================================================================================================
// tensor parallel group
// =================================================================================================
extern "C" int nvl_tp_get_unique_id(void* id_out, int bytes) {
    if (!id_out || bytes < (int)sizeof(ncclUniqueId)) return fail(nullptr, NVL_ERR_INVALID, "nvl_tp_get_unique_id: need a 128-byte buffer");
    ncclUniqueId id;
    const ncclResult_t rc = ncclGetUniqueId(&id);
    if (rc != ncclSuccess) return fail(nullptr, NVL_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(rc));
    memcpy(id_out, &id, sizeof id);
    return NVL_OK;
}
extern "C" int nvl_tp_init(nvl_model* m, const void* id_in, int bytes) {
    if (!m || !id_in || bytes < (int)sizeof(ncclUniqueId)) return fail(m, NVL_ERR_INVALID, "nvl_tp_init: bad arguments");
    if (m->tp_comm || m->tp_local) return fail(m, NVL_ERR_STATE, "nvl_tp_init: a group is already attached");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    ncclComm_t comm;
    const ncclResult_t rc = ncclCommInitRank(&comm, m->tp, id, m->tp_rank);
    if (rc != ncclSuccess) return fail(m, NVL_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(rc));
    m->tp_comm = comm;
    return NVL_OK;
    NVL_CATCH(m)
}
namespace {
constexpr int P2P_ONESHOT_CAP_ROWS = 64;
void p2p_alloc(nvl_model* m) {
    if (m->p2p_buf) return;
    const int64_t esz = m->f32 ? 4 : 2, T = m->tp;
    const int64_t cap = (int64_t)m->opts.max_batch_tokens * m->H;
    m->p2p_in1_stride = (int64_t)P2P_ONESHOT_CAP_ROWS * m->H;
    m->p2p_in2_stride = round_up(cdiv(cap, T), 8);       // (16-byte peer stores: 8 bf16)
    m->p2p_res_stride = round_up(cap, 8);
    m->p2p_off_in1 = 1024;
    m->p2p_off_in2 = round_up(m->p2p_off_in1 + 2 * T * m->p2p_in1_stride * esz, 256);
    m->p2p_off_res = round_up(m->p2p_off_in2 + 2 * T * m->p2p_in2_stride * esz, 256);
    m->p2p_bytes = (size_t)round_up(m->p2p_off_res + 2 * m->p2p_res_stride * esz, 256);
    void* p = nullptr;
    NVL_HIP(hipExtMallocWithFlags(&p, m->p2p_bytes, hipDeviceMallocUncached));     // coherent across GPUs: peers write it, we poll it
    m->p2p_buf = (char*)p;
    NVL_HIP(hipMemset(m->p2p_buf, 0, m->p2p_bytes));
}
}  // namespace
extern "C" int nvl_tp_p2p_export(nvl_model* m, void* handle_out, int bytes) {
    if (!m || !handle_out || bytes < (int)sizeof(hipIpcMemHandle_t)) return fail(m, NVL_ERR_INVALID, "nvl_tp_p2p_export: need a 64-byte buffer");
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_tp_p2p_export: finalize the model first");
    if (m->tp < 2) return fail(m, NVL_ERR_STATE, "nvl_tp_p2p_export: the model is not tensor-parallel");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    p2p_alloc(m);
    hipIpcMemHandle_t h;
    NVL_HIP(hipIpcGetMemHandle(&h, m->p2p_buf));
    memcpy(handle_out, &h, sizeof h);
    return NVL_OK;
    NVL_CATCH(m)
}
extern "C" int nvl_tp_p2p_attach(nvl_model* m, const void* handles, int bytes_per_handle) {
    if (!m || !handles || bytes_per_handle < (int)sizeof(hipIpcMemHandle_t)) return fail(m, NVL_ERR_INVALID, "nvl_tp_p2p_attach: bad arguments");
    if (!m->p2p_buf) return fail(m, NVL_ERR_STATE, "nvl_tp_p2p_attach: call nvl_tp_p2p_export first");
    if (m->p2p_ready) return fail(m, NVL_ERR_STATE, "nvl_tp_p2p_attach: already attached");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    for (int r = 0; r < m->tp; r++) {
        if (r == m->tp_rank) { m->p2p_peer[r] = m->p2p_buf; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * bytes_per_handle, sizeof h);
        void* p = nullptr;
        NVL_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        m->p2p_peer[r] = (char*)p;
    }
    m->p2p_ready = true;
    return NVL_OK;
    NVL_CATCH(m)
}

// After a timed-out all-reduce (nvl_forward failed with "timed out waiting for a peer rank") the group's arrival counters are
// out of step.  Every rank calls this while NO rank is inside a forward call (e.g. behind a host barrier): it zeroes the
// rank's own header — arrival counters, call counters, error word — and drops the captured graphs.
extern "C" int nvl_tp_p2p_rearm(nvl_model* m) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->p2p_buf) return fail(m, NVL_ERR_STATE, "nvl_tp_p2p_rearm: no P2P group (nvl_tp_p2p_export / _attach first)");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    NVL_HIP(hipStreamSynchronize(m->stream));
    NVL_HIP(hipMemset(m->p2p_buf, 0, 1024));
    NVL_HIP(hipDeviceSynchronize());
    return NVL_OK;
    NVL_CATCH(m)
}

extern "C" int nvl_tp_attach_local(nvl_model** models, int n) {
    if (!models || n < 1 || n > 8) return fail(nullptr, NVL_ERR_INVALID, "nvl_tp_attach_local: bad arguments");
    bool seen[8] = {false};
    int64_t need = 0;
    for (int i = 0; i < n; i++) {
        nvl_model* m = models[i];
        if (!m || m->tp != n || m->tp_rank < 0 || m->tp_rank >= n || seen[m->tp_rank] || m->tp_comm || m->tp_local ||
            m->device != models[0]->device)
            return fail(nullptr, NVL_ERR_INVALID, "nvl_tp_attach_local: members must be the n distinct ranks of one tp_size = n group on one device");
        seen[m->tp_rank] = true;
        need = std::max<int64_t>(need, (int64_t)m->opts.max_batch_tokens * m->H);
    }
    try {
        nvl_local_group* g = new nvl_local_group();
        g->n = n; g->refs = n;
        NVL_HIP(hipSetDevice(models[0]->device));
        g->scratch = dmalloc<float>(need);
        g->scratch_floats = need;
        for (int i = 0; i < n; i++) models[i]->tp_local = g;
        return NVL_OK;
    } catch (const HipError& e) {
        return fail(nullptr, NVL_ERR_HIP, hipGetErrorString(e.code));
    }
}

// =================================================================================================
// sequences
// =================================================================================================
namespace {
// Open a KV slot for seq_id.  evict: when every slot is taken, drop the least-recently-forwarded sequence that is not
// pinned (= not part of the forward call being assembled) and reuse its slot.  The reference keeps map[int64]*KVCache
// forever and the engine never calls ClearCache (tensor_model_runner.go:11-18,59-68; llm_engine.go:35-37), so the runner
// entry points evict; a later decode of an evicted sequence finds no slot and is re-prefilled from its full TokenIDs
// (runner_impl), which is what makes eviction safe.  The explicit nvl_seq_open keeps its NVL_ERR_NO_SLOT contract.
int seq_open_impl(nvl_model* m, int64_t seq_id, bool evict) {
    if (!m || !m->finalized) return fail(m, NVL_ERR_STATE, "nvl_seq_open: model not finalized");
    if (m->paged) return fail(m, NVL_ERR_STATE, "nvl_seq_open: the model is in paged-KV mode (the host's block manager owns the cache)");
    auto it = m->seq_slot.find(seq_id);
    if (it != m->seq_slot.end()) { m->slot_tick[(size_t)it->second] = ++m->tick; return NVL_OK; }
    if (m->free_slots.empty()) {
        if (!evict) return fail(m, NVL_ERR_NO_SLOT, "nvl_seq_open: all KV slots in use");
        auto victim = m->seq_slot.end();
        for (auto jt = m->seq_slot.begin(); jt != m->seq_slot.end(); ++jt) {
            if (m->slot_pin[(size_t)jt->second]) continue;
            if (victim == m->seq_slot.end() || m->slot_tick[(size_t)jt->second] < m->slot_tick[(size_t)victim->second]) victim = jt;
        }
        if (victim == m->seq_slot.end()) return fail(m, NVL_ERR_NO_SLOT, "nvl_runner_run: every KV slot belongs to the batch being assembled");
        m->free_slots.push_back(victim->second);
        m->seq_slot.erase(victim);
        m->stats.evictions++;
    }
    const int s = m->free_slots.back(); m->free_slots.pop_back();
    m->seq_slot[seq_id] = s; m->slot_len[(size_t)s] = 0; m->slot_tick[(size_t)s] = ++m->tick; m->slot_pin[(size_t)s] = 0;
    return NVL_OK;
}
}  // namespace
extern "C" int nvl_seq_open(nvl_model* m, int64_t seq_id) {
    NVL_TRY(m)
    return seq_open_impl(m, seq_id, false);
    NVL_CATCH(m)
}
extern "C" int nvl_seq_reset(nvl_model* m, int64_t seq_id) {
    const int rc = nvl_seq_open(m, seq_id);
    if (rc) return rc;
    m->slot_len[(size_t)m->seq_slot[seq_id]] = 0;
    return NVL_OK;
}
extern "C" int nvl_seq_close(nvl_model* m, int64_t seq_id) {
    if (!m) return NVL_ERR_INVALID;
    auto it = m->seq_slot.find(seq_id);
    if (it == m->seq_slot.end()) return NVL_OK;   // delete() of a missing key is a no-op in Go
    m->free_slots.push_back(it->second);
    m->seq_slot.erase(it);
    return NVL_OK;
}
extern "C" int nvl_seq_close_all(nvl_model* m) {
    if (!m) return NVL_ERR_INVALID;
    for (auto& kv : m->seq_slot) m->free_slots.push_back(kv.second);
    m->seq_slot.clear();
    return NVL_OK;
}
extern "C" int nvl_seq_len(nvl_model* m, int64_t seq_id) {
    if (!m) return NVL_ERR_INVALID;
    auto it = m->seq_slot.find(seq_id);
    if (it == m->seq_slot.end()) return NVL_ERR_UNKNOWN_SEQ;
    return m->slot_len[(size_t)it->second];
}

// =================================================================================================
// forward
// =================================================================================================
namespace {

struct Meta {   // device pointers into meta_dev
    int32_t *tokens, *tok_pos, *tok_tbl, *seq_tok_start, *seq_len, *seq_pos, *seq_tbl, *last_rows, *blk_table;
};
// layout of meta_host / meta_dev for a call carrying M tokens:
//   [seq_tok_start S | seq_len S | seq_pos S | seq_tbl S | last_rows S | blk_table table_cap | tokens M | tok_pos M | tok_tbl M]
Meta bind_meta(const nvl_model* m, int32_t* base, int M) {
    const int64_t S = m->opts.max_seqs;
    Meta md;
    md.seq_tok_start = base; md.seq_len = base + S; md.seq_pos = base + 2 * S; md.seq_tbl = base + 3 * S;
    md.last_rows = base + 4 * S; md.blk_table = base + 5 * S;
    md.tokens = md.blk_table + m->table_cap; md.tok_pos = md.tokens + M; md.tok_tbl = md.tok_pos + M;
    return md;
}
size_t meta_bytes(const nvl_model* m, int M) { return (size_t)(5 * (int64_t)m->opts.max_seqs + m->table_cap + 3 * (int64_t)M) * 4; }

static int g_norm_t16 = 1;     // nvl_set_tuning key 9: norm_tile16_kernel for prefill-sized bf16 norms (0 = norm_kernel)
template <typename ActT>
void launch_norm(nvl_model* m, float* x, const int32_t* rows_idx, const float* w, const float* b,
                 void* y, int rows) {
    KScope ks(m, KC_OTHER, 0, KS_NORM, (double)rows * m->H * (4.0 + (double)m->wsize));
    PendingResid pr{nullptr, 0, 0, 0.f, nullptr, nullptr};
    if (m->pending_slices > 0) {            // complete the residual add the previous decode GEMM left as split-K slices
        pr.part = m->pending_part; pr.slices = m->pending_slices; pr.rows_total = m->pending_rows; pr.alpha = m->pending_alpha;
        pr.slot_of = m->pending_slot_of; pr.gate_w = m->pending_gate_w;
        m->pending_slices = 0; m->pending_slot_of = nullptr; m->pending_gate_w = nullptr;
        const bool row_kernel = rows <= 512 && m->H <= 1024 * NORM_ROW_MAXCH;
        const bool tile_kernel = sizeof(ActT) == 2 && m->H % 32 == 0 && m->H <= 2048 && g_norm_t16 && !pr.slot_of;   // (the 16-chunk instance would spill)
        if (!row_kernel && !tile_kernel) throw std::runtime_error("norm: no kernel for this pending residual");
    }
    if (rows <= 512 && m->H <= 1024 * NORM_ROW_MAXCH)
        hipLaunchKernelGGL((norm_row_kernel<ActT>), dim3(rows), dim3(256), 0, m->stream, x, rows_idx, w, b,
                           m->cfg.norm_eps, (ActT*)y, m->H, pr);
    else if (sizeof(ActT) == 2 && m->H % 32 == 0 && m->H <= 4096 && g_norm_t16) {   // (wider rows would spill: norm_kernel)
        // 16 rows x H bf16 of LDS (up to 160 KiB); the row sits in registers: instance by chunks per lane
#define NVL_NT16(CH, PEND)                                                                                             \
        do {                                                                                                           \
            NVL_LDS_ATTR((norm_tile16_kernel<CH, PEND>), 160 * 1024);                                                  \
            hipLaunchKernelGGL((norm_tile16_kernel<CH, PEND>), dim3(cdiv(rows, 16)), dim3(1024), (size_t)16 * m->H * 2, m->stream, x, \
                               rows_idx, w, b, m->cfg.norm_eps, (bf16_t*)y, rows, m->H, pr);                            \
        } while (0)
        if (pr.part) NVL_NT16(8, true);        // (H <= 2048: checked above)
        else { if (m->H <= 2048) NVL_NT16(8, false); else NVL_NT16(16, false); }
#undef NVL_NT16
    } else
        hipLaunchKernelGGL((norm_kernel<ActT>), dim3(cdiv(rows, 4)), dim3(256), 0, m->stream, x, rows_idx, w, b,
                           m->cfg.norm_eps, (ActT*)y, rows, m->H);
}
void norm(nvl_model* m, float* x, const int32_t* rows_idx, const DevTensor& w, const DevTensor& b, void* y, int rows) {
    if (m->f32) launch_norm<float>(m, x, rows_idx, (const float*)w.p, (const float*)b.p, y, rows);
    else launch_norm<bf16_t>(m, x, rows_idx, (const float*)w.p, (const float*)b.p, y, rows);
    NVL_HIP(hipGetLastError());
}

static int g_p2p_oneshot_rows = 64;   // nvl_set_tuning key 23: payloads of up to this many rows take the one-shot all-reduce
static int g_use_graphs = 1;   // nvl_set_tuning key 21: hipGraph replay of decode passes (0 = every launch eager)
static int g_tune_epoch = 0;   // bumped by every nvl_set_tuning: captured graphs bake the tuning in (part of their key)
static int g_attn_nw = 0;      // nvl_set_tuning key 15: waves per decode-attention workgroup (0 = from the context length, 2, 4, 8)
static int g_attn_tq2 = 1;     // nvl_set_tuning key 10: unused since round 2 (the prefill kernel always keeps two sub-tiles per wave)
static int g_attn_kv_nt = 1;   // nvl_set_tuning key 35: non-temporal K/V loads in the decode attention (0 never, 1 automatic, 2 always)
static int g_attn_split = 1;   // nvl_set_tuning key 25: split a decode attention's keys over up to 8 workgroups per (sequence, kv head) (0 = never)
// Launch shape of the decode attention for the batch being enqueued (also part of a captured graph's key): waves per
// workgroup | workgroups per (sequence, kv head) << 8.
// Split: a workgroup pulls all K/V of its (sequence, kv head) through ONE CU; with a few sequences and a long context
// (B = 1 at 2048 keys: 8 workgroups x 512 KB) that, not HBM, is the time.  Then the key tiles are dealt over up to 8
// workgroups of 2 waves each; the last of them to finish combines the partial results (attn.h).  Every workgroup of a
// split launch pays a device-scope release fence, so the launch stays within 128 workgroups.  Measured (decode ms/step,
// Llama-3.2-1B, 2048 / 1024 keys): B=1 0.891 -> 0.811 / 0.808 -> 0.780, B=2 0.893 -> 0.834 / 0.869 -> 0.792, B=4 (4
// workgroups per pair) 0.924 -> 0.883 / 0.812 -> 0.817 (hence from 2048 keys on above 16 pairs); B=8 loses either way
// (2 workgroups per pair 0.96 -> 0.99, 8 per pair -> 1.13).
int decode_attn_cfg(const nvl_model* m, int n_seqs) {
    const int n_kt = m->ctx_hint > 0 ? (m->ctx_hint - 1) / 64 + 1 : 1 << 20;
    const int per_wave = m->hd == 64 ? 2 : 1;
    const int wgs = m->nKV * cdiv(m->group, 16) * n_seqs;
    if (g_attn_split && !g_attn_nw && m->attn_part && m->ctx_hint > 0 && wgs <= ATTN_SPLIT_MAX_PAIRS && n_kt >= 16) {
        const int want = cdiv(n_kt, 2 * per_wave);                       // workgroups of 2 waves that get a full round each
        const int ns = (wgs > 16 && n_kt < 32) ? 1 : std::min(std::min(want, ATTN_SPLIT_MAX), 128 / wgs);
        if (ns >= 2) return 2 | (ns << 8);
    }
    const int cap = wgs >= 512 ? 2 : (wgs >= 320 ? 4 : 8);
    const int nw = n_kt <= 2 * per_wave ? 2 : (n_kt <= 4 * per_wave ? 4 : 8);
    return (g_attn_nw ? g_attn_nw : std::min(nw, cap)) | (1 << 8);
}
int decode_attn_waves(const nvl_model* m, int n_seqs) { return decode_attn_cfg(m, n_seqs); }   // (graph keys: the whole shape)
void attention(nvl_model* m, int li, const Meta& md, int n_seqs, int max_len, double flops, bool fused_qkv = false) {
    AttnArgs a{};
    a.q = m->q; a.q_stride = m->nH * m->hd;
    a.out = m->attn_out; a.out_stride = m->nH * m->hd;
    a.kcache = (char*)m->kcache + (size_t)li * m->layer_stride * m->wsize;
    a.vcache = (char*)m->vcache + (size_t)li * m->layer_stride * m->wsize;
    a.slot_stride = m->slot_stride; a.Tmax = m->Tmax;
    a.seq_tok_start = md.seq_tok_start; a.seq_len = md.seq_len; a.seq_pos = md.seq_pos; a.blk_table = md.blk_table; a.tbl_stride = m->blocks_per_seq;
    a.bs_shift = (m->Tmax & (m->Tmax - 1)) == 0 ? __builtin_ctz((unsigned)m->Tmax) : -1;
    a.nH = m->nH; a.nKV = m->nKV; a.group = m->group; a.scale = m->attn_scale;
    // algorithmic bytes: every cached key/value of the batch once + Q in + O out (+ the fp32 QKV row of a fused decode step)
    const double abytes = m->kv_tok * 2.0 * m->nKV * m->hd * (double)m->wsize;
    KScope ks(m, KC_ATTN, flops, KS_ATTN, abytes);
    if (m->f32) {
        const size_t lds = (size_t)m->Tmax * 4;
        hipLaunchKernelGGL(attn_f32_kernel, dim3(max_len, m->nH, n_seqs), dim3(256), lds, m->stream, a, m->hd);
    } else if (max_len == 1) {
        const int cfg = decode_attn_cfg(m, n_seqs), nsplit = cfg >> 8;
        dim3 grid(m->nKV * cdiv(m->group, 16), n_seqs, nsplit);      // (kv head x 16-head query tile, sequence, key split)
        a.part = m->attn_part; a.part_cnt = m->attn_part_cnt;
        // waves per workgroup: each wave takes NT2 key tiles per round trip (2 for hd 64, 1 for hd 128).  Short contexts
        // need fewer than 8 waves (idle waves still cost LDS and a slot in the final merge), and once the grid alone
        // fills the chip (>= 2 workgroups per CU) fewer, longer-running waves stream better than many short ones
        // (profiles/r01g_decode_attention_waves.txt).  ctx_hint = the batch's longest context when the caller knows it.
        const int nw = cfg & 255;
        // non-temporal K/V loads when one workgroup serves a (sequence, kv head)'s whole group (the K/V are then read once) and
        // the launch fills the chip — by batch size on Llama-3.2-1B (8 kv heads, 520 keys; profiles/r03_kv_nontemporal.txt):
        // B = 1 +1.7 %, 2 +-0, 4 -2.3 %, 8 -2.4 %, 16 +-0, 24 +2.7 %, 32 +5 %, 48 +1 %; GPT-2 B = 128 (1536 workgroups) +3.2 %
        a.kv_nt = g_attn_kv_nt == 2 || (g_attn_kv_nt == 1 && cdiv(m->group, 16) == 1 && (m->nKV * n_seqs >= 160 || n_seqs == 1));
        a.stamps = (fused_qkv && m->hd == 64) ? next_stamps(m, KS_ATTN, (int)(grid.x * grid.y * grid.z)) : nullptr;
        if (fused_qkv) {   // RoPE + KV append + attention in one launch, straight from the QKV projection's fp32 output
            a.qkv = m->qkv; a.qkv_stride = m->n_qkv; a.cos_t = m->rope_cos; a.sin_t = m->rope_sin;
        }
#define NVL_DEC_LEAD a.seq_pos, a.blk_table, a.tbl_stride, a.bs_shift, a.Tmax, a.group      /* the preloaded leading arguments (attn.h) */
#define NVL_DEC(HDv, FUSEDv)                                                                                           \
        do {                                                                                                           \
            constexpr int tile_bytes = 64 * HDv * 2;                                                                   \
            if (nw == 2) hipLaunchKernelGGL((attn_decode_bf16_kernel<HDv, 2, FUSEDv>), grid, dim3(128), 2 * tile_bytes, m->stream, NVL_DEC_LEAD, a);      \
            else if (nw == 4) hipLaunchKernelGGL((attn_decode_bf16_kernel<HDv, 4, FUSEDv>), grid, dim3(256), 4 * tile_bytes, m->stream, NVL_DEC_LEAD, a); \
            else {                                                                                                     \
                NVL_LDS_ATTR((attn_decode_bf16_kernel<HDv, 8, FUSEDv>), 8 * tile_bytes);                               \
                hipLaunchKernelGGL((attn_decode_bf16_kernel<HDv, 8, FUSEDv>), grid, dim3(512), 8 * tile_bytes, m->stream, NVL_DEC_LEAD, a);               \
            }                                                                                                          \
        } while (0)
        if (a.stamps && fused_qkv && m->hd == 64) {     // diagnostic instantiation with the in-kernel stamps compiled in
            constexpr int tb = 64 * 64 * 2;
            if (nw == 2) hipLaunchKernelGGL((attn_decode_bf16_kernel<64, 2, true, true>), grid, dim3(128), 2 * tb, m->stream, NVL_DEC_LEAD, a);
            else if (nw == 4) hipLaunchKernelGGL((attn_decode_bf16_kernel<64, 4, true, true>), grid, dim3(256), 4 * tb, m->stream, NVL_DEC_LEAD, a);
            else {
                NVL_LDS_ATTR((attn_decode_bf16_kernel<64, 8, true, true>), 8 * tb);
                hipLaunchKernelGGL((attn_decode_bf16_kernel<64, 8, true, true>), grid, dim3(512), 8 * tb, m->stream, NVL_DEC_LEAD, a);
            }
        } else if (fused_qkv) { if (m->hd == 64) NVL_DEC(64, true); else NVL_DEC(128, true); }
        else { if (m->hd == 64) NVL_DEC(64, false); else NVL_DEC(128, false); }
#undef NVL_DEC
#undef NVL_DEC_LEAD
    } else {
        // 256 query rows (position x head-in-group, position-major) per workgroup of 8 waves; K/V tiles by LDS-DMA
        const int64_t qrows = (int64_t)max_len * m->group;
        if (m->blocks_per_seq > 128) throw std::runtime_error("attention: more than 128 KV blocks per sequence");
        dim3 grid(m->nKV, n_seqs, cdiv(qrows, 256));      // (kv head fastest, query tile slowest: attn.h)
        if (m->hd == 64) {
            hipLaunchKernelGGL((attn_prefill_bf16_kernel<64>), grid, dim3(512), attn_prefill_lds_bytes<64>(), m->stream, a);
        } else {
            NVL_LDS_ATTR(attn_prefill_bf16_kernel<128>, attn_prefill_lds_bytes<128>());
            hipLaunchKernelGGL((attn_prefill_bf16_kernel<128>), grid, dim3(512), attn_prefill_lds_bytes<128>(), m->stream, a);
        }
    }
    NVL_HIP(hipGetLastError());
}

void rope_kv(nvl_model* m, int li, const Meta& md, int M) {
    KScope ks(m, KC_OTHER, 0, KS_ROPE, (double)M * m->n_qkv * (4.0 + (double)m->wsize));
    dim3 grid(M, m->nH + 2 * m->nKV);
    const int thr = m->hd / 2;
    void* kc = (char*)m->kcache + (size_t)li * m->layer_stride * m->wsize;
    void* vc = (char*)m->vcache + (size_t)li * m->layer_stride * m->wsize;
    if (m->f32)
        hipLaunchKernelGGL((rope_kv_kernel<float, false>), grid, dim3(thr), 0, m->stream, m->qkv, m->n_qkv,
                           md.tok_pos, md.tok_tbl, md.blk_table, m->rope_cos, m->rope_sin, (float*)m->q, m->nH * m->hd,
                           (float*)kc, (float*)vc, m->slot_stride, m->Tmax, m->nH, m->nKV, m->hd);
    else
        hipLaunchKernelGGL((rope_kv_kernel<bf16_t, false>), grid, dim3(thr), 0, m->stream, m->qkv, m->n_qkv,
                           md.tok_pos, md.tok_tbl, md.blk_table, m->rope_cos, m->rope_sin, (bf16_t*)m->q, m->nH * m->hd,
                           (bf16_t*)kc, (bf16_t*)vc, m->slot_stride, m->Tmax, m->nH, m->nKV, m->hd);
    NVL_HIP(hipGetLastError());
}

void launch_argmax(hipStream_t st, float* logits, int ld, int V, int rows, float scaling, float* pval, int32_t* pidx,
                   int32_t* out) {
    const int chunks = cdiv(V, ARGMAX_CHUNK);
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(chunks, rows), dim3(256), 0, st, logits, ld, V, scaling, pval, pidx);
    if (out) hipLaunchKernelGGL(argmax_final_kernel, dim3(rows), dim3(64), 0, st, pval, pidx, chunks, out);
}

GemmArgs mk(const void* A, int lda, const void* W, void* C, int ldc, const float* bias, float alpha, int M, int N, int K) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.a_rows = nullptr; a.W = W; a.C = C; a.ldc = ldc; a.bias = bias; a.alpha = alpha;
    a.M = M; a.N = N; a.K = K; a.seg = nullptr; a.c_row0 = 0;
    a.qkv = QkvEpi{};
    a.tile_map = nullptr; a.n_mtiles = nullptr; a.w_expert_stride = 0;
    a.sk_part = nullptr; a.sk_slices = 1; a.m_split = 0; a.rs_half = 0; a.grid_y = 1; a.nrm_w = nullptr; a.nrm_xn = nullptr; a.rs_out = nullptr; a.rs_in = nullptr; a.rs_tiles = 0; a.rs_inv_h = 0.f; a.rs_eps = 0.f;
    return a;
}

// ---- tensor-parallel all-reduce (sum) of a row-parallel projection's fp32 partial -----------------------
// Real runs: one process per GPU, RCCL over xGMI (nvl_tp_init).  Tests on one GPU: the shard models of a
// process form a local group (nvl_tp_attach_local) and are driven from one host thread each; the last
// thread to arrive sums every member's buffer on the device and hands the result back to all of them.
void tp_allreduce(nvl_model* m, float* buf, int64_t count) {
    KScope ks(m, KC_OTHER, 0, KS_ALLREDUCE, (double)count * 4.0);
    if (m->tp_comm) {
        const ncclResult_t rc = ncclAllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)m->tp_comm, m->stream);
        if (rc != ncclSuccess) throw std::runtime_error(std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
        return;
    }
    nvl_local_group* g = m->tp_local;
    if (!g) throw std::runtime_error("tp_size > 1 but no communicator: call nvl_tp_init (RCCL) or nvl_tp_attach_local first");
    NVL_HIP(hipStreamSynchronize(m->stream));                 // this rank's partial is complete
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->aborted) throw std::runtime_error("tp local group: another member failed (group aborted)");
    if (count > g->scratch_floats) throw std::runtime_error("tp local group: scratch too small");
    g->bufs[m->tp_rank] = buf;
    const uint64_t my_gen = g->gen;
    if (++g->arrived == g->n) {
        PtrList8 pl{};
        for (int r = 0; r < g->n; r++) pl.p[r] = g->bufs[r];
        hipLaunchKernelGGL(sum_bufs_kernel, dim3(cdiv(count, 1024)), dim3(256), 0, m->stream, g->scratch, pl, g->n, count);
        for (int r = 0; r < g->n; r++)
            NVL_HIP(hipMemcpyAsync(g->bufs[r], g->scratch, (size_t)count * 4, hipMemcpyDeviceToDevice, m->stream));
        NVL_HIP(hipStreamSynchronize(m->stream));
        g->arrived = 0;
        g->gen++;
        g->cv.notify_all();
    } else {
        // a peer that fails sets `aborted` (note_failure); the timeout covers a peer that never calls at all
        const bool ok = g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->gen != my_gen || g->aborted; });
        if (!ok || g->gen == my_gen) {
            g->aborted = true;
            g->cv.notify_all();
            throw std::runtime_error(ok ? "tp local group: another member failed (group aborted)"
                                        : "tp local group: timed out waiting for the other members");
        }
    }
}

static int g_half_tiles = 1;   // nvl_set_tuning key 27: deferred-norm residual projections of <= 16 rows on 8-row half tiles (gemm.h HALF)
// (producer and consumer agree through M alone: every projection of a pass has the same row count)
// (only while the 16-row tiles leave CUs idle: at H = 4096 — 256 tiles — the halves doubled the x^2 partials every consumer
// sums and the workgroups that each fetch the activations: Llama-3-8B B = 16 FFN-down 23.7 -> 34.2 us, O 9.3 -> 12.4 us)
bool defer_half(const nvl_model* m, int M) { return g_half_tiles && M <= 16 && m->H % 128 == 0 && m->H / 16 < 256; }
// after a forward's stream sync: did a P2P wait give up?  (a peer died or never called: the residual stream is garbage)
void p2p_check(nvl_model* m) {
    if (!m->p2p_ready) return;
    int err = 0;
    NVL_HIP(hipMemcpy(&err, m->p2p_buf + P2P_OFF_ERR, 4, hipMemcpyDeviceToHost));
    if (err) throw std::runtime_error("tensor-parallel all-reduce: timed out waiting for a peer rank (nvl_tp_p2p_rearm on every rank clears the group)");
}
static int g_p2p_spin_ms = 30000;     // nvl_set_tuning key 29: how long a rank waits for its peers inside an all-reduce before it gives up
// x += alpha * sum over the tensor-parallel ranks of `part` [M][N], by direct peer stores over the xGMI mesh (tp_p2p.h): ONE
// launch.  nrm_w != NULL (decode, RMSNorm + SwiGLU): the same launch emits the deferred-RMSNorm operand of the next projection.
void tp_p2p_allreduce_resid(nvl_model* m, const float* part, int M, int N, float alpha, const float* nrm_w = nullptr) {
    const int64_t count = (int64_t)M * N;
    if (count % 8 || N % 16) throw std::runtime_error("tp p2p: the row length must be a multiple of 16");
    P2PArgs a{};
    a.T = m->tp; a.rank = m->tp_rank;
    a.count = count; a.part = part; a.x = m->x; a.alpha = alpha;
    for (int r = 0; r < m->tp; r++) a.peer[r] = m->p2p_peer[r];
    a.off_in1 = m->p2p_off_in1; a.off_in2 = m->p2p_off_in2; a.off_res = m->p2p_off_res;
    a.in1_stride = m->p2p_in1_stride; a.in2_stride = m->p2p_in2_stride; a.res_stride = m->p2p_res_stride;
    a.spin_limit = (int)std::min<int64_t>((int64_t)g_p2p_spin_ms * 1000, 2000000000);      // x ~1 us of s_sleep per poll
    a.oneshot = (M <= g_p2p_oneshot_rows && M <= P2P_ONESHOT_CAP_ROWS) ? 1 : 0;
    if (!a.oneshot) a.chunk = round_up(cdiv(count, a.T), 8);
    if (nrm_w) {
        if (!a.oneshot) throw std::runtime_error("tp p2p: the fused norm rides on the one-shot form");
        a.H = N; a.nrm_w = nrm_w; a.nrm_xn = (bf16_t*)m->xn; a.rs_out = m->rs_part; a.rs_cols = defer_half(m, M) ? 8 : 16;
    }
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(count, 256 * 8), 512));    // all resident: the kernel spins
    KScope ks(m, KC_OTHER, 0, KS_ALLREDUCE, (double)count * (m->f32 ? 4.0 : 2.0) * 2.0);
    if (m->f32) hipLaunchKernelGGL((p2p_allreduce_kernel<float, false>), dim3(nwg), dim3(256), 0, m->stream, a);
    else if (nrm_w) hipLaunchKernelGGL((p2p_allreduce_kernel<bf16_t, true>), dim3(nwg), dim3(256), 0, m->stream, a);
    else hipLaunchKernelGGL((p2p_allreduce_kernel<bf16_t, false>), dim3(nwg), dim3(256), 0, m->stream, a);
    NVL_HIP(hipGetLastError());
}

// Residual projection (O projection / W2): x += alpha * (A·W^T + bias)   (generic_model.go:320-326,383-389).
// Prefill and the fp32 mode fuse the add into the GEMM epilogue.  Decode-sized batches split K over
// `slices` workgroups per column block so that all 256 CUs stream weights (a 2048-column projection has
// only 128 column blocks), and leave the add to the norm kernel that always follows (PendingResid).
static int g_qkv_store_max_m = 256;   // nvl_set_tuning key 18: largest prefill batch whose QKV projection runs as decode-form groups + rope_kv_kernel
static int g_moe_dense = 1;    // nvl_set_tuning key 22: decode MoE as two dense-masked projections (gemm.h moe_gate); 0 = sort + grouped GEMMs
static int g_moe_deep = 1;     // nvl_set_tuning key 19: four-stage grouped GEMM for decode-sized MoE batches (0 = two stages)
static int g_moe_bm = 0;       // nvl_set_tuning key 17: prefill MoE grouped GEMMs: 0 = 256-row tiles on the ping-pong kernel, 128 / 256 = the lock-step tile kernels
static int g_moe_gather = 1;   // nvl_set_tuning key 16: prefill MoE gathers the token rows into expert order before the grouped GEMM (0 = per-lane gather inside it)
static int g_moe_defer_down = 0;    // nvl_set_tuning key 33: MoE decode: the down projection carries the next norm (deferred RMSNorm producer: all of
                                    // K in one workgroup, one launch less per layer) instead of K slices + a norm launch.  Off: with 128 workgroups of
                                    // 16 waves each wave walks 8 blocks (two experts) instead of 4 — Granite-1B B=8 down projection 8.9 -> 14.5 us, more
                                    // than the norm launch it saves: 6.48 K -> 6.26 K decode tok/s (profiles/r03_moe_decode_experiments.txt)
static int g_moe_fused_route = 1;   // nvl_set_tuning key 31: MoE decode routing (router GEMM + softmax / top-k / gate matrix) as one launch (0 = two)
static int g_moe_small = 1;    // nvl_set_tuning key 8: fused MoE planning launch + combine folded into the next norm (0 = off)
static int g_mamba_ssd = 1;    // nvl_set_tuning key 30: chunked (SSD) Mamba2 scan on MFMA for prefill-sized bf16 calls (0 = the sequential scan,
                               // 2 = chunk after chunk only, never the chunk-parallel three-launch form)
static int g_decode_seam = 1;  // nvl_set_tuning key 7: decode_seam_kernel in nvl_decode_greedy (0 = separate kernels)
static int g_defer_norm = 1;   // nvl_set_tuning key 3: deferred RMSNorm between O-proj and FFN-up in decode (0 = off)
static int g_sk_slices = 0;    // tuning override (nvl_set_tuning key 1): 0 automatic, 1 = never split
// consumer side of the deferred RMSNorm: the projection reads xn_raw and scales its accumulators (gemm.h)
void set_deferred_in(nvl_model* m, GemmArgs& a) {
    a.rs_in = m->rs_part; a.rs_tiles = defer_half(m, a.M) ? m->H / 8 : m->H / 16; a.rs_inv_h = 1.0f / (float)m->H; a.rs_eps = m->cfg.norm_eps;
}
void resid_gemm(nvl_model* m, const void* A, int lda, const void* W, const float* bias, float alpha, int M, int N, int K,
                const float* defer_norm_w = nullptr) {
    if (m->tp > 1 || m->tp_force) {
        // row-parallel projection: this rank holds a K slice -> fp32 partial [M][N] (the bias lives on rank 0 only),
        // all-reduce over the tensor-parallel group, then the residual add (folded into the next norm for decode)
        gemm(m, EPI_STORE, true, mk(A, lda, W, m->tp_part, N, bias, 1.f, M, N, K));
        if (m->p2p_ready) { tp_p2p_allreduce_resid(m, m->tp_part, M, N, alpha, defer_norm_w); return; }     // sum + residual add (+ the deferred norm) in one go
        if (defer_norm_w) throw std::runtime_error("resid_gemm: the deferred norm of a tensor-parallel projection needs the P2P all-reduce");
        tp_allreduce(m, m->tp_part, (int64_t)M * N);
        if (!m->f32 && M <= 64 && !m->keep_hidden && m->pending_slices == 0) {
            m->pending_part = m->tp_part; m->pending_slices = 1; m->pending_rows = M; m->pending_alpha = alpha;
        } else {
            KScope ks(m, KC_OTHER);
            hipLaunchKernelGGL(axpy_rows_kernel, dim3(cdiv((int64_t)M * N / 4, 256)), dim3(256), 0, m->stream, m->x,
                               m->tp_part, (const float*)nullptr, alpha, (int64_t)M, N);
            NVL_HIP(hipGetLastError());
        }
        return;
    }
    GemmArgs a = mk(A, lda, W, m->x, N, bias, alpha, M, N, K);
    if (defer_norm_w) {     // deferred RMSNorm: x complete in this launch (no K split over workgroups) + xn_raw + x^2 partials
        a.nrm_w = defer_norm_w; a.nrm_xn = (bf16_t*)m->xn; a.rs_out = m->rs_part; a.m_split = g_defer_norm >= 2 ? 0 : 1;
        a.rs_half = a.m_split && N == m->H && defer_half(m, M) ? 1 : 0;
        gemm(m, EPI_RESID, true, a);
        return;
    }
    int slices = 1;
    if (!m->f32 && M <= 64 && !m->keep_hidden && m->sk_part && m->pending_slices == 0 && N % 16 == 0) {
        const int nblocks = N / 16;
        if (g_sk_slices > 0) slices = g_sk_slices;
        else while (slices < m->sk_max_slices && nblocks * slices < 256 && (K >> 5) / (slices * 2) >= 8) slices *= 2;
    }
    // 65..512 rows (a short prompt, a few hundred decode rows that do not take the decode form): N / 128 x M / 128 tiles of the
    // lock-step kernel are a handful of workgroups (FFN-down of Llama-1B at 512 rows: 64), and the 64-row groups of the decode
    // form pull every activation row through every CU (1.3 GB of ingest for that projection: 85 us).  K split over
    // gridDim.y workgroups per tile instead, the slices summed by the norm that follows (which must be the row kernel).
    const bool norm_ok = (M <= 512 && m->H <= 1024 * NORM_ROW_MAXCH) || (m->H % 32 == 0 && m->H <= 2048 && g_norm_t16);   // launch_norm
    if (g_sk_tile && !m->f32 && M > 64 && M <= m->sk_rows && !m->keep_hidden && m->sk_part && m->pending_slices == 0 &&
        N % 128 == 0 && K % 64 == 0 && norm_ok && N == m->H && g_force_tile == 0 && m->n_mamba == 0) {
        const int tiles = cdiv(M, 128) * (N / 128);
        while (slices < SK_TILE_MAX_SLICES && tiles * slices < 512 && K / (slices * 2) >= 512) slices *= 2;
        // (fewer than 64 workgroups even so — GPT-2's 768 columns below 512 rows —: the decode form's 64-row groups stream
        // better, profiles/r03_gpt2_large_decode_batches.txt: B = 128 147.5 K -> 159.0 K, B = 256 224.8 K -> 237.2 K tok/s)
        if (tiles * slices < 64) slices = 1;
    }
    if (slices > 1) {
        a.sk_part = m->sk_part; a.sk_slices = slices;
        gemm(m, EPI_RESID, true, a);
        m->pending_part = m->sk_part; m->pending_slices = slices; m->pending_rows = M; m->pending_alpha = alpha;
    } else {
        gemm(m, EPI_RESID, true, a);
    }
}

// FeedForward.Forward (transformer.go:40-96) up to (not including) the W2 projection:
// leaves act(x·W1) in m->hbuf [M][F]
void ffn_up(nvl_model* m, const LayerW& l, int M, bool deferred_norm = false) {
    const bool swiglu = m->cfg.activation_type == NVL_ACT_SWIGLU;
    const float* b1 = (const float*)l.t[NVL_T_B1].p;
    if (swiglu) {
        if (m->f32) {
            m->site = KS_FFN_UP;
            gemm(m, EPI_STORE, true, mk(m->xn, m->H, l.w1, m->h2, 2 * m->F, nullptr, 1.f, M, 2 * m->F, m->H));
            KScope ks(m, KC_OTHER);
            const int64_t n = (int64_t)M * m->F;
            hipLaunchKernelGGL((swiglu_kernel<float>), dim3(cdiv(n, 256)), dim3(256), 0, m->stream, m->h2,
                               (float*)m->hbuf, (int64_t)M, m->F);
            NVL_HIP(hipGetLastError());
        } else {
            GemmArgs a = mk(m->xn, m->H, l.w1, m->hbuf, m->F, nullptr, 1.f, M, 2 * m->F, m->H);
            if (deferred_norm) set_deferred_in(m, a);
            m->site = KS_FFN_UP;
            gemm(m, EPI_SWIGLU, false, a);
        }
    } else {
        m->site = KS_FFN_UP;
        gemm(m, EPI_GELU, false, mk(m->xn, m->H, l.w1, m->hbuf, m->F, b1, 1.f, M, m->F, m->H));
    }
}

// MoELayer.Forward (moe.go:43-128): router GEMM -> route (softmax, top-k, renormalise) -> counting sort of
// the (token, rank) pairs by expert -> ONE grouped gate/up GEMM and ONE grouped down GEMM over all experts
// (m-tiles mapped to (expert, row segment) by a device table; token rows gathered by per-lane source address)
// -> combine in rank order into x (with the residual multiplier).  The fp32 parity mode launches per expert.
// can this call take the decode ("dense-masked") MoE form?  (also decides whether the O projection may carry the FFN norm)
bool moe_dense_ok(const nvl_model* m, const LayerW& l, int M) {
    return !m->f32 && M <= 64 && g_moe_dense && l.moe_out_cat && m->moe_part && !m->keep_hidden && m->H % 4 == 0 &&
           m->H <= 1024 * NORM_ROW_MAXCH && g_force_tile == 0;
}
// deferred_norm: m->xn holds xn_raw = bf16(x * w_norm) and m->rs_part the x^2 partials (the O projection carried the FFN
// norm): the router and the expert-up projection scale their accumulators by rstd[m] instead of reading a normed operand
// next_norm_w (decode, dense-masked form only): the down projection carries the NEXT norm (the following layer's attention
// norm / the final norm) as the deferred-RMSNorm producer — all of K in one workgroup per (weight half-tile, row tile), no
// slices left for a norm launch to sum; returns true when it did (m->xn / m->rs_part then hold that norm's operand).
bool moe(nvl_model* m, const LayerW& l, int M, bool deferred_norm = false, const float* next_norm_w = nullptr) {
    const nvl_model_config& c = m->cfg;
    const int E = c.num_experts, k = c.num_experts_per_tok, I = m->F, H = m->H;
    const int pairs = M * k;
    // algorithmic weight bytes of the expert GEMMs: the experts a batch of `pairs` (token, rank) pairs can touch, once
    const double e_touch = (double)std::min(E, pairs);
    m->site = KS_MOE_ROUTER;
    const bool dense = moe_dense_ok(m, l, M) && m->pending_slices == 0;
    // decode: router logits, softmax, top-k and the dense gate matrix in one launch (gemm.h moe_router_gate_kernel)
    const bool fused_route = dense && g_moe_fused_route && E <= 64 && H % 32 == 0;
    {
        GemmArgs ar = mk(m->xn, H, l.t[NVL_T_ROUTER].p, m->router_logits, 128, nullptr, 1.f, M, E, H);
        if (deferred_norm) set_deferred_in(m, ar);
        if (fused_route) {
            KScope ks(m, KC_GEMM, 2.0 * (double)M * E * H, -1, ((double)E * H + (double)M * H) * (double)m->wsize + (double)M * E * 8.0);
            hipLaunchKernelGGL(moe_router_gate_kernel, dim3(cdiv(M, 16)), dim3(1024), 0, m->stream, ar, k, m->moe_gate, m->router_logits, 128);
            NVL_HIP(hipGetLastError());
        } else {
            gemm(m, EPI_STORE, true, ar);
        }
    }
    // Decode: the dense-masked form (gemm.h GemmArgs::moe_gate).  Routing weights as a dense [M][E] matrix, then all experts
    // as ONE weight-streaming projection each way; untouched experts are skipped inside the kernels, the gate weight in the up
    // epilogue turns the down projection's K reduction into the weighted combine, its K slices are summed by the next norm.
    if (deferred_norm && !dense)
        throw std::runtime_error("moe: a deferred FFN norm needs the dense-masked decode form");
    if (dense) {
        if (!fused_route) {
            KScope ks(m, KC_OTHER, 0, KS_MOE_PLAN, (double)M * (128 + E) * 4.0);
            hipLaunchKernelGGL(moe_gate_kernel, dim3(cdiv(M, 4)), dim3(256), 0, m->stream, m->router_logits, 128, M, E, k, m->moe_gate);
            NVL_HIP(hipGetLastError());
        }
        GemmArgs a = mk(m->xn, H, l.moe_in, m->moe_hall, E * I, nullptr, 1.f, M, E * 2 * I, H);
        a.moe_gate = m->moe_gate; a.moe_E = E; a.moe_I = I;
        if (deferred_norm) set_deferred_in(m, a);
        m->site = KS_MOE_UP; m->site_bytes = (e_touch * 2 * I * H + (double)M * H + e_touch * M * I) * (double)m->wsize;
        gemm(m, EPI_SWIGLU, false, a, 2.0 * pairs * 2 * I * H);
        const int nks = (E * I) >> 5, spe = I >> 5;
        if (next_norm_w && g_moe_defer_down && nks % 64 == 0 && spe % 4 == 0 && nks / 16 <= 4 * spe && nks / 64 <= 32) {
            // (the launcher's conditions: 16 waves, whole blocks of 4 k-steps inside one expert, <= 4 experts per wave)
            GemmArgs d = mk(m->moe_hall, E * I, l.moe_out_cat, m->x, H, nullptr, m->resid_alpha, M, H, E * I);
            d.moe_gate = m->moe_gate; d.moe_E = E; d.moe_I = I;
            d.nrm_w = next_norm_w; d.nrm_xn = (bf16_t*)m->xn; d.rs_out = m->rs_part; d.m_split = 1;
            d.rs_half = defer_half(m, M) ? 1 : 0;
            m->site = KS_MOE_DOWN; m->site_bytes = (e_touch * H * I + e_touch * M * I) * (double)m->wsize + (double)M * H * 10.0;
            gemm(m, EPI_RESID, true, d, 2.0 * pairs * H * I);
            return true;
        }
        GemmArgs d = mk(m->moe_hall, E * I, l.moe_out_cat, m->x, H, nullptr, 1.f, M, H, E * I);
        const int nsl = moe_down_slices(E);
        d.moe_gate = m->moe_gate; d.moe_E = E; d.moe_I = I; d.sk_part = m->moe_part; d.sk_slices = nsl;
        m->site = KS_MOE_DOWN; m->site_bytes = (e_touch * H * I + e_touch * M * I) * (double)m->wsize + (double)nsl * M * H * 4.0;
        gemm(m, EPI_RESID, true, d, 2.0 * pairs * H * I);
        m->pending_part = m->moe_part; m->pending_slices = nsl; m->pending_rows = M; m->pending_alpha = m->resid_alpha;
        return false;
    }
    // decode-sized batches: one planning launch, and the weighted combine rides on the norm that follows
    const bool small = !m->f32 && M <= 64 && pairs <= MOE_PLAN_MAX_PAIRS && E <= MOE_PLAN_MAX_E && g_moe_small;
    // m-tile height of the grouped GEMMs: 128 rows; the 256x128 three-stage instance stays a tuning option (Granite-1B,
    // 4096 rows per expert: 441 K vs 450 K prefill tok/s)
    // prefill default: 256-row m-tiles on the ping-pong kernel (needs the rows copied into expert order and K >= 160)
    const bool pp = !m->f32 && !small && g_moe_bm == 0 && m->moe_xg && g_moe_gather && H >= 160 && I >= 160 && (2 * I) % 256 == 0 && H % 256 == 0;
    const int BM = pp ? 256 : ((!m->f32 && !small && g_moe_bm == 256) ? 256 : 128);
    if (small) {
        KScope ks(m, KC_OTHER, 0, KS_MOE_PLAN);
        hipLaunchKernelGGL(moe_plan_kernel, dim3(1), dim3(M <= 4 ? 256 : (M <= 8 ? 512 : 1024)), 0, m->stream, m->router_logits, 128, M, E, k, 128, m->expert_ids,
                           m->expert_w, m->seg_start, m->moe_tile_map, m->moe_n_mtiles, m->perm_token, m->slot_of);
        NVL_HIP(hipGetLastError());
    } else {
        KScope ks(m, KC_OTHER, 0, KS_MOE_PLAN);
        NVL_HIP(hipMemsetAsync(m->moe_counts, 0, (size_t)E * 4, m->stream));
        hipLaunchKernelGGL(moe_route_kernel, dim3(cdiv(M, 4)), dim3(256), 0, m->stream, m->router_logits, 128, M, E, k,
                           m->expert_ids, m->expert_w);
        hipLaunchKernelGGL(moe_hist_kernel, dim3(cdiv(pairs, MOE_PAIRS_PER_WG)), dim3(256), 0, m->stream, m->expert_ids, pairs, m->moe_counts);
        hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(64), 0, m->stream, m->moe_counts, E, BM, m->seg_start, m->moe_cursor,
                           m->moe_tile_map, m->moe_n_mtiles);
        hipLaunchKernelGGL(moe_scatter_kernel, dim3(cdiv(pairs, MOE_PAIRS_PER_WG)), dim3(256), 0, m->stream, m->expert_ids, pairs, k,
                           m->moe_cursor, m->perm_token, m->slot_of);
        NVL_HIP(hipGetLastError());
    }
    const int max_mtiles = cdiv(pairs, BM) + E;      // every expert may end on a partial tile
    if (m->f32) {
        for (int e = 0; e < E; e++) {   // grid bounded by M: a token picks an expert at most once; surplus blocks exit
            const char* win = (const char*)l.moe_in + (size_t)e * 2 * I * H * m->wsize;
            GemmArgs a = mk(m->xn, H, win, m->h2, 2 * I, nullptr, 1.f, M, 2 * I, H);
            a.a_rows = m->perm_token; a.seg = m->seg_start + e;
            gemm(m, EPI_STORE, true, a);
        }
        {
            KScope ks(m, KC_OTHER);
            const int64_t n = (int64_t)pairs * I;
            hipLaunchKernelGGL((swiglu_kernel<float>), dim3(cdiv(n, 256)), dim3(256), 0, m->stream, m->h2, (float*)m->hbuf,
                               (int64_t)pairs, I);
            NVL_HIP(hipGetLastError());
        }
        for (int e = 0; e < E; e++) {
            const char* wout = (const char*)l.t[NVL_T_MOE_OUT].p + (size_t)e * H * I * m->wsize;
            GemmArgs a = mk(m->hbuf, I, wout, m->moe_eo, H, nullptr, 1.f, M, H, I);
            a.seg = m->seg_start + e;
            gemm(m, EPI_STORE, true, a);
        }
    } else {
        GemmArgs a = mk(m->xn, H, l.moe_in, m->hbuf, I, nullptr, 1.f, max_mtiles, 2 * I, H);
        a.a_rows = m->perm_token; a.tile_map = m->moe_tile_map; a.n_mtiles = m->moe_n_mtiles;
        if (!small && m->moe_xg && g_moe_gather) {
            // prefill: every m-tile's rows would be re-gathered 16 bytes at a time by each of its 2I/128 column tiles
            // (64 cache lines per load instruction); copy them into expert order once, then the GEMM streams whole
            // 1-KiB operand blocks (Granite-1B, 16384 tokens: expert-up 1119 -> ? us per layer)
            KScope ks(m, KC_OTHER);
            hipLaunchKernelGGL(moe_gather_fm_kernel, dim3(cdiv(pairs, 16)), dim3(256), 0, m->stream, (const bf16_t*)m->xn,
                               m->perm_token, (bf16_t*)m->moe_xg, pairs, H);
            NVL_HIP(hipGetLastError());
            a.A = m->moe_xg; a.a_rows = nullptr;
        }
        a.w_expert_stride = (int64_t)2 * I * H; a.grp_bm = pp ? 512 : BM; a.grp_deep = small && g_moe_deep;
        m->site = KS_MOE_UP; m->site_bytes = (e_touch * 2 * I * H + (double)pairs * H + (double)pairs * I) * (double)m->wsize;
        gemm(m, EPI_SWIGLU, false, a, 2.0 * pairs * 2 * I * H);
        GemmArgs d = mk(m->hbuf, I, l.t[NVL_T_MOE_OUT].p, m->moe_eo, H, nullptr, 1.f, max_mtiles, H, I);
        d.tile_map = m->moe_tile_map; d.n_mtiles = m->moe_n_mtiles; d.w_expert_stride = (int64_t)H * I; d.grp_bm = pp ? 512 : BM; d.grp_deep = small && g_moe_deep;
        m->site = KS_MOE_DOWN; m->site_bytes = (e_touch * H * I + (double)pairs * I) * (double)m->wsize + (double)pairs * H * 4.0;
        gemm(m, EPI_STORE, true, d, 2.0 * pairs * H * I);
    }
    if (small && !m->keep_hidden && m->pending_slices == 0 && H % 4 == 0 && H <= 1024 * NORM_ROW_MAXCH) {
        // x += rm * sum_k w[t][k] * eo[slot[t][k]] is completed by the next norm's read of x (PendingResid)
        m->pending_part = m->moe_eo; m->pending_slices = k; m->pending_rows = M; m->pending_alpha = c.residual_multiplier;
        m->pending_slot_of = m->slot_of; m->pending_gate_w = m->expert_w;
    } else {
        KScope ks(m, KC_OTHER, 0, KS_MOE_COMBINE, (double)pairs * H * 4.0 + (double)M * H * 8.0);
        hipLaunchKernelGGL(moe_combine_kernel, dim3(M), dim3(256), 0, m->stream, m->moe_eo, m->slot_of, m->expert_w, k,
                           c.residual_multiplier, m->x, H, 1);
        NVL_HIP(hipGetLastError());
    }
    return false;
}

}  // namespace

namespace {
// ---- hipGraph replay of decode passes ----------------------------------------------------------------------------------
// A decode pass is ~85 dependent launches of a few microseconds each.  A pass whose launch configuration (the key) has been
// seen before is captured once from the model's stream and replayed with hipGraphLaunch: the arguments of every launch are
// device buffers whose CONTENTS change (token ids, positions, block tables, KV slabs), never their addresses.  First sight
// of a key runs eagerly (it also sets the per-function attributes, which must not happen during capture).
template <typename F>
bool replay_or_capture(nvl_model* m, const std::array<int, 5>& key, F&& enqueue) {
    // (tensor parallel: only with the P2P all-reduce, whose call state lives on the device; RCCL / the in-process emulation
    //  are host-driven)
    if (!g_use_graphs || !m->graphs_ok || m->profile || m->keep_hidden || m->tap || m->stamping || ((m->tp > 1 || m->tp_force) && !m->p2p_ready)) return false;
    auto it = m->graphs.find(key);
    if (it != m->graphs.end()) { NVL_HIP(hipGraphLaunch(it->second, m->stream)); m->stats.graph_replays++; return true; }
    if (!m->graph_seen.count(key)) { m->graph_seen.insert(key); return false; }
    if (m->graphs.size() >= 64) retire_graphs(m);        // (keys come and go with the context length: keep the table small;
                                                         //  earlier steps' replays may still be running: destroyed at the next sync)
    if (hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); m->graphs_ok = false; return false; }
    hipGraph_t g = nullptr;
    try {
        enqueue();
    } catch (...) {
        (void)hipStreamEndCapture(m->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        m->graphs_ok = false;
        throw;
    }
    if (hipStreamEndCapture(m->stream, &g) != hipSuccess || !g) { (void)hipGetLastError(); m->graphs_ok = false; enqueue(); return true; }
    hipGraphExec_t ge = nullptr;
    const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || !ge) { (void)hipGetLastError(); m->graphs_ok = false; enqueue(); return true; }
    m->graphs[key] = ge;
    NVL_HIP(hipGraphLaunch(ge, m->stream));
    m->stats.graph_replays++;
    return true;
}

// debug taps: the residual stream after layer li.  Mode 1 (keep_hidden) runs every residual add to completion in its own
// launch, so x is the layer's output; mode 2 (tap) leaves the product path alone and adds what is still pending.
void tap_layer(nvl_model* m, int li, int M) {
    if (!m->keep_hidden && !m->tap) return;
    float* dst = m->hidden + (int64_t)li * M * m->H;
    if (m->pending_slices == 0) { NVL_HIP(hipMemcpyAsync(dst, m->x, (size_t)M * m->H * 4, hipMemcpyDeviceToDevice, m->stream)); return; }
    PendingResid pr{m->pending_part, m->pending_slices, m->pending_rows, m->pending_alpha, m->pending_slot_of, m->pending_gate_w};
    hipLaunchKernelGGL(tap_hidden_kernel, dim3(M), dim3(256), 0, m->stream, m->x, dst, m->H, pr);
    NVL_HIP(hipGetLastError());
}

// Enqueue ONE forward pass (embedding ... argmax) on the model's stream for the batch described by the device
// metadata `md` — no host synchronisation.  Returns the number of logits rows produced.
// seam: 0 = whole pass; bit 0 = x and layer 0's normed operand are already in place (skip embed + first norm);
// bit 1 = stop after the argmax partials (the caller's decode_seam_kernel finishes the step); bit 2 = no argmax at
// all (the caller samples from the logits)
int enqueue_forward(nvl_model* m, const Meta& md, int n_seqs, int M, int max_len, double attn_flops, uint32_t flags,
                    int seam = 0) {
    const nvl_model_config& c = m->cfg;
    const int H = m->H;
    m->pending_slices = 0;
    m->phase = max_len > 1 ? 0 : 1;         // per-site profile: prefill / decode
    if ((m->keep_hidden || m->tap) && m->hidden_tokens < M) {
        dfree(m->hidden);
        m->hidden = dmalloc<float>((int64_t)m->L * M * H);
        m->hidden_tokens = M;
    }
    m->hidden_last_M = M;

    // ---- embed (generic_model.go:295-302) -------------------------------------------------
    if (!(seam & 1)) {
        KScope ks(m, KC_OTHER, 0, KS_EMBED, (double)M * H * (4.0 + (double)m->wsize));
        const void* pe = (c.position_type == NVL_POS_LEARNED) ? m->g[NVL_T_POS_EMB].p : nullptr;
        const int pe_rows = pe ? (int)std::min<int64_t>(m->g[NVL_T_POS_EMB].rows, c.max_seq_len) : 0;
        if (m->f32)
            hipLaunchKernelGGL((embed_kernel<float, false>), dim3(M), dim3(256), 0, m->stream, md.tokens, md.tok_pos,
                               (const float*)m->g[NVL_T_TOK_EMB].p, (const float*)pe, pe_rows, c.embedding_multiplier, m->x, H);
        else
            hipLaunchKernelGGL((embed_kernel<bf16_t, true>), dim3(M), dim3(256), 0, m->stream, md.tokens, md.tok_pos,
                               (const bf16_t*)m->g[NVL_T_TOK_EMB].p, (const bf16_t*)pe, pe_rows, c.embedding_multiplier, m->x, H);
        NVL_HIP(hipGetLastError());
    }

    const int qw = m->nH * m->hd;
    const bool parallel = c.block_style == NVL_BLOCK_PARALLEL;

    // decode, RMSNorm + SwiGLU, sequential block: every residual projection carries the norm that follows it (deferred
    // RMSNorm, gemm.h), so a step keeps ONE norm launch (layer 0's, after the embedding) instead of 2L + 1
    const bool big_decode = max_len == 1 && M <= g_chunk_max_m && g_force_tile == 0;
    // (tensor parallel: the all-reduce launch carries the norm — one-shot payloads only, i.e. M <= 64 — when the P2P group is attached)
    const bool tp_on = m->tp > 1 || m->tp_force;
    const bool tp_defer = tp_on && m->p2p_ready && M <= g_p2p_oneshot_rows && M <= 64;
    const bool defer_ok = g_defer_norm && !m->f32 && (M <= 64 || (M <= DEFER_MAX_M && big_decode && g_defer_norm == 1 && !tp_on)) && !c.use_moe && (!tp_on || tp_defer) && !m->keep_hidden &&
                          c.block_style == NVL_BLOCK_SEQUENTIAL && c.norm_type == NVL_NORM_RMS &&
                          c.activation_type == NVL_ACT_SWIGLU && H % 256 == 0 && m->rs_part && m->n_mamba == 0;
    // MoE layers (decode form): only the O projection carries a norm (the FFN norm); the expert-down projection leaves
    // K-slice partials that the next attention norm sums
    const bool defer_moe_ok = g_defer_norm && c.use_moe && M <= 64 && m->tp == 1 && !m->tp_force && c.block_style == NVL_BLOCK_SEQUENTIAL &&
                              c.norm_type == NVL_NORM_RMS && H % 256 == 0 && m->rs_part && m->n_mamba == 0;
    const bool all_rows = (flags & NVL_FWD_ALL_LOGITS) != 0;
    bool xn_deferred = false;       // m->xn holds xn_raw of the upcoming norm, m->rs_part its x^2 partials
    for (int li = 0; li < m->L; li++) {
        const LayerW& l = m->layers[li];
        if (l.mamba) {
            // forwardMamba2 (generic_model.go:160-202): norm, Mamba2Layer.Forward (mamba2.go:74-181), residual; norm, shared MLP, residual
            norm(m, m->x, nullptr, l.t[NVL_T_ATTN_NORM_W], l.t[NVL_T_ATTN_NORM_B], m->xn, M);
            m->site = KS_MAMBA_IN;
            gemm(m, EPI_STORE, true, mk(m->xn, H, l.t[NVL_T_MAMBA_IN_PROJ].p, m->mproj, m->mP, nullptr, 1.f, M, m->mP, H));
            MambaArgs a{};
            a.proj = m->mproj; a.P = m->mP; a.EH = m->mEH; a.conv_dim = m->mConv; a.nh = m->m_nh; a.hd = m->m_hd; a.ss = m->m_ss;
            a.ng = m->m_ng; a.K = m->m_K;
            a.conv_w = (const float*)l.t[NVL_T_MAMBA_CONV_W].p; a.conv_b = (const float*)l.t[NVL_T_MAMBA_CONV_B].p;
            a.a_log = (const float*)l.t[NVL_T_MAMBA_A_LOG].p; a.Dskip = (const float*)l.t[NVL_T_MAMBA_D].p;
            a.dt_bias = (const float*)l.t[NVL_T_MAMBA_DT_BIAS].p; a.norm_w = (const float*)l.t[NVL_T_MAMBA_NORM].p;
            a.xbc = m->mxbc; a.delta = m->mdelta; a.y = m->my;
            a.state = m->ssm_state + (int64_t)l.mamba_idx * m->ssm_layer_stride; a.state_slot_stride = m->ssm_slot_stride;
            a.tok_pos = md.tok_pos; a.tok_seq = md.tok_tbl; a.seq_tok_start = md.seq_tok_start; a.seq_len = md.seq_len;
            a.seq_pos = md.seq_pos; a.seq_slot = md.blk_table;     // slab mode: token -> sequence index i, blk_table[i] = slot
            a.chain = m->conv_tail ? m->conv_chain : 0;
            a.tail = m->conv_tail ? m->conv_tail + (int64_t)l.mamba_idx * m->conv_tail_layer_stride : nullptr;
            a.tail_slot_stride = m->conv_tail_slot_stride;
            {
                KScope ks(m, KC_OTHER, 0, KS_MAMBA_CONV, (double)M * (m->mP + m->mConv + m->m_nh) * 4.0);
                hipLaunchKernelGGL(mamba_conv_kernel, dim3(M), dim3(256), 0, m->stream, a);
                if (a.chain) hipLaunchKernelGGL(mamba_tail_kernel, dim3(n_seqs), dim3(256), 0, m->stream, a);
                NVL_HIP(hipGetLastError());
            }
            {
                KScope ks(m, KC_OTHER, 2.0 * 3.0 * (double)M * m->mEH * m->m_ss, KS_MAMBA_SCAN,
                          (double)M * (m->mConv + m->m_nh + m->mEH) * 4.0 + 2.0 * n_seqs * (double)m->ssm_layer_stride * 4.0);
                const int per = m->m_ss / (256 / m->m_hd);
                dim3 grid(m->m_nh, n_seqs);
                // prefill-sized calls, bf16 mode: the chunked (SSD) form on MFMA — 64 tokens per step instead of one
                const bool ssd = g_mamba_ssd && !m->f32 && max_len >= 16 && (m->m_hd == 32 || m->m_hd == 64) &&
                                 (m->m_ss == 32 || m->m_ss == 64 || m->m_ss == 128) && m->mConv % 4 == 0 && m->mEH % 4 == 0;
                // (sequence, head) pairs for fewer than half the CUs and several chunks: the chunks of a sequence in parallel (state
                // contributions, a short recurrence over the chunks, outputs: three launches; 1 x 2048: 318 -> 90 us per layer) —
                // with a workgroup per CU already (8 x 512) the extra state traffic loses (86 -> 174 us); scratch capped at 256 MB
                const int max_chunks = cdiv(max_len, 64);
                const int64_t z_floats = (int64_t)n_seqs * max_chunks * m->m_nh * m->m_hd * m->m_ss;
                const bool par = ssd && g_mamba_ssd >= 1 && g_mamba_ssd != 2 && max_chunks >= 4 && m->m_nh * n_seqs < 128 && z_floats * 4 <= (256ll << 20);
                if (par && z_floats > m->ssd_z_floats) {
                    NVL_HIP(hipStreamSynchronize(m->stream));
                    dfree(m->ssd_z); dfree(m->ssd_decay);
                    m->ssd_z = dmalloc<float>(z_floats); m->ssd_z_floats = z_floats;
                    m->ssd_decay = dmalloc<float>((int64_t)n_seqs * max_chunks * m->m_nh + 1024);
                    m->ssd_decay_floats = (int64_t)n_seqs * max_chunks * m->m_nh + 1024;
                }
                if (par && (int64_t)n_seqs * max_chunks * m->m_nh > m->ssd_decay_floats) {
                    NVL_HIP(hipStreamSynchronize(m->stream));
                    dfree(m->ssd_decay);
                    m->ssd_decay_floats = (int64_t)n_seqs * max_chunks * m->m_nh + 1024;
                    m->ssd_decay = dmalloc<float>(m->ssd_decay_floats);
                }
                a.ssd_z = m->ssd_z; a.ssd_decay = m->ssd_decay;
#define NVL_SSD(HDv, SSv)                                                                                              \
                do {                                                                                                   \
                    constexpr int lds_ = mamba_ssd_lds_bytes<HDv, SSv>();                                              \
                    if (par) {                                                                                         \
                        dim3 g3(m->m_nh, n_seqs, max_chunks);                                                          \
                        NVL_LDS_ATTR((mamba_ssd_kernel<HDv, SSv, 1>), lds_);                                           \
                        NVL_LDS_ATTR((mamba_ssd_kernel<HDv, SSv, 2>), lds_);                                           \
                        hipLaunchKernelGGL((mamba_ssd_kernel<HDv, SSv, 1>), g3, dim3(256), lds_, m->stream, a);        \
                        hipLaunchKernelGGL(mamba_ssd_prefix_kernel, dim3(m->m_nh, n_seqs, cdiv(HDv * SSv, 1024)), dim3(256), 0, m->stream, a, max_chunks, HDv * SSv); \
                        hipLaunchKernelGGL((mamba_ssd_kernel<HDv, SSv, 2>), g3, dim3(256), lds_, m->stream, a);        \
                    } else {                                                                                           \
                        NVL_LDS_ATTR((mamba_ssd_kernel<HDv, SSv, 0>), lds_);                                           \
                        hipLaunchKernelGGL((mamba_ssd_kernel<HDv, SSv, 0>), grid, dim3(256), lds_, m->stream, a);      \
                    }                                                                                                  \
                } while (0)
                if (ssd) {
                    if (m->m_hd == 64) { if (m->m_ss == 128) NVL_SSD(64, 128); else if (m->m_ss == 64) NVL_SSD(64, 64); else NVL_SSD(64, 32); }
                    else { if (m->m_ss == 128) NVL_SSD(32, 128); else if (m->m_ss == 64) NVL_SSD(32, 64); else NVL_SSD(32, 32); }
                } else
#undef NVL_SSD
                if (per <= 4) hipLaunchKernelGGL(mamba_scan_kernel<4>, grid, dim3(256), 0, m->stream, a);
                else if (per <= 8) hipLaunchKernelGGL(mamba_scan_kernel<8>, grid, dim3(256), 0, m->stream, a);
                else if (per <= 16) hipLaunchKernelGGL(mamba_scan_kernel<16>, grid, dim3(256), 0, m->stream, a);
                else hipLaunchKernelGGL(mamba_scan_kernel<32>, grid, dim3(256), 0, m->stream, a);
                NVL_HIP(hipGetLastError());
            }
            {
                KScope ks(m, KC_OTHER, 0, KS_MAMBA_GATE, (double)M * m->mEH * (8.0 + (double)m->wsize));
                if (m->f32) hipLaunchKernelGGL((mamba_gate_norm_kernel<float>), dim3(M), dim3(256), 0, m->stream, a, (float*)m->myn);
                else hipLaunchKernelGGL((mamba_gate_norm_kernel<bf16_t>), dim3(M), dim3(256), 0, m->stream, a, (bf16_t*)m->myn);
                NVL_HIP(hipGetLastError());
            }
            m->site = KS_MAMBA_OUT;
            resid_gemm(m, m->myn, m->mEH, l.t[NVL_T_MAMBA_OUT_PROJ].p, nullptr, m->resid_alpha, M, H, m->mEH);
            norm(m, m->x, nullptr, l.t[NVL_T_FFN_NORM_W], l.t[NVL_T_FFN_NORM_B], m->xn, M);
            ffn_up(m, l, M);
            m->site = KS_FFN_DOWN;
            resid_gemm(m, m->hbuf, m->F, l.t[NVL_T_W2].p, (const float*)l.t[NVL_T_B2].p, m->resid_alpha, M, H, m->F);
            tap_layer(m, li, M);
            continue;
        }
        const bool qkv_deferred = xn_deferred;
        if (!xn_deferred && !(li == 0 && (seam & 1)))
            norm(m, m->x, nullptr, l.t[NVL_T_ATTN_NORM_W], l.t[NVL_T_ATTN_NORM_B], m->xn, M);
        xn_deferred = false;
        bool fused_dec = false;
        // a decode batch of 64 < M <= g_chunk_max_m rows keeps the decode form (64-row passes, gemm.h) and the fused
        // decode attention
        // short prefill batches (65..g_qkv_store_max_m tokens): the fused-epilogue tile kernel would run on M/128 * 24
        // workgroups; the decode form + rope_kv_kernel is faster there
        const bool short_prefill = max_len > 1 && M <= g_qkv_store_max_m && M <= g_chunk_max_m && g_force_tile == 0;
        if (!m->f32 && M > 64 && !big_decode && !short_prefill) {
            // prefill: RoPE + Q/K/V placement fused into the projection's epilogue (no fp32 qkv round trip)
            GemmArgs a = mk(m->xn, H, l.w_qkv, nullptr, 0, l.b_qkv, 1.f, M, m->n_qkv, H);
            a.qkv.tok_pos = md.tok_pos; a.qkv.tok_tbl = md.tok_tbl; a.qkv.blk_table = md.blk_table; a.qkv.cos_t = m->rope_cos; a.qkv.sin_t = m->rope_sin;
            a.qkv.q_out = (bf16_t*)m->q;
            a.qkv.kcache = (bf16_t*)m->kcache + (int64_t)li * m->layer_stride;
            a.qkv.vcache = (bf16_t*)m->vcache + (int64_t)li * m->layer_stride;
            a.qkv.slot_stride = m->slot_stride; a.qkv.Tmax = m->Tmax; a.qkv.nH = m->nH; a.qkv.nKV = m->nKV; a.qkv.hd = m->hd;
            m->site = KS_QKV;
            gemm(m, EPI_QKV, false, a);
        } else {
            GemmArgs aq = mk(m->xn, H, l.w_qkv, m->qkv, m->n_qkv, l.b_qkv, 1.f, M, m->n_qkv, H);
            if (qkv_deferred) set_deferred_in(m, aq);
            m->site = KS_QKV;
            gemm(m, EPI_STORE, true, aq);
            fused_dec = !m->f32 && max_len == 1;     // decode: RoPE + KV append live in the attention kernel
            if (!fused_dec) rope_kv(m, li, md, M);
        }
        attention(m, li, md, n_seqs, max_len, attn_flops, fused_dec);
        const float* bo = (const float*)l.t[NVL_T_BO].p;
        if (parallel) {
            // generic_model.go:395-418: r + [am*]attn + [rm*]ffn, both branches read the same normed x
            const bool mults = c.attention_multiplier != 0.f && c.residual_multiplier != 0.f;
            ffn_up(m, l, M);
            m->site = KS_OPROJ;
            resid_gemm(m, m->attn_out, qw, l.t[NVL_T_WO].p, bo, mults ? c.attention_multiplier : 1.f, M, H, qw);
            m->site = KS_FFN_DOWN;
            resid_gemm(m, m->hbuf, m->F, l.t[NVL_T_W2].p, (const float*)l.t[NVL_T_B2].p, mults ? c.residual_multiplier : 1.f,
                       M, H, m->F);   // (only one of the two may be pending: the second one adds into x directly)
        } else {
            // decode, RMSNorm + SwiGLU: the O-projection carries the FFN norm (deferred RMSNorm, gemm.h) — one launch less
            const bool defer = (defer_ok || (defer_moe_ok && moe_dense_ok(m, l, M))) && m->pending_slices == 0 && !l.t[NVL_T_FFN_NORM_B].present();
            m->site = KS_OPROJ;
            if (defer) {
                resid_gemm(m, m->attn_out, qw, l.t[NVL_T_WO].p, bo, m->resid_alpha, M, H, qw, (const float*)l.t[NVL_T_FFN_NORM_W].p);
            } else {
                resid_gemm(m, m->attn_out, qw, l.t[NVL_T_WO].p, bo, m->resid_alpha, M, H, qw);
                norm(m, m->x, nullptr, l.t[NVL_T_FFN_NORM_W], l.t[NVL_T_FFN_NORM_B], m->xn, M);
            }
            // the FFN-down projection carries the NEXT layer's attention norm, or the final norm when every row is
            // a last row (decode)
            const DevTensor* nxt_w = li + 1 < m->L ? &m->layers[li + 1].t[NVL_T_ATTN_NORM_W] : &m->g[NVL_T_FINAL_NORM_W];
            const DevTensor* nxt_b = li + 1 < m->L ? &m->layers[li + 1].t[NVL_T_ATTN_NORM_B] : &m->g[NVL_T_FINAL_NORM_B];
            if (c.use_moe) {
                const bool defer2 = defer && defer_moe_ok && g_defer_norm != 3 && !nxt_b->present() &&
                                    (li + 1 < m->L || (!all_rows && M == n_seqs && M <= 64));
                xn_deferred = moe(m, l, M, defer, defer2 ? (const float*)nxt_w->p : nullptr);
            } else {
                ffn_up(m, l, M, defer);
                const bool defer2 = defer_ok && g_defer_norm != 3 && m->pending_slices == 0 && !nxt_b->present() &&
                                    (li + 1 < m->L || (!all_rows && M == n_seqs && M <= 64));   // (the LM head of a larger batch runs on the tile kernels)
                m->site = KS_FFN_DOWN;
                resid_gemm(m, m->hbuf, m->F, l.t[NVL_T_W2].p, (const float*)l.t[NVL_T_B2].p, m->resid_alpha, M, H, m->F,
                           defer2 ? (const float*)nxt_w->p : nullptr);
                xn_deferred = defer2;
            }
        }
        tap_layer(m, li, M);
    }

    // ---- final norm + LM head on the rows the caller keeps (generic_model.go:464-477, :595-604) ----
    const bool all = (flags & NVL_FWD_ALL_LOGITS) != 0;
    const int rows = all ? M : n_seqs;
    if (rows > m->logit_rows) {
        clear_graphs(m);               // (captured passes hold the old addresses)
        dfree(m->logits); dfree(m->xn_last);
        m->logits = dmalloc<float>((int64_t)rows * m->Vpad);
        m->xn_last = dmalloc_bytes(round_up(rows, 64) * H * (int64_t)m->wsize);
        NVL_HIP(hipMemsetAsync(m->xn_last, 0, (size_t)(round_up(rows, 64) * H) * m->wsize, m->stream));
        m->logit_rows = rows;
    }
    if (xn_deferred) {       // the last FFN-down projection already produced the final norm's operand for every row
        GemmArgs al = mk(m->xn, H, m->lm_head, m->logits, m->Vpad, nullptr, 1.f, rows, m->V, H);
        set_deferred_in(m, al);
        m->site = KS_LM_HEAD;
        gemm(m, EPI_STORE, true, al);
    } else {
        norm(m, m->x, all ? nullptr : md.last_rows, m->g[NVL_T_FINAL_NORM_W], m->g[NVL_T_FINAL_NORM_B], m->xn_last, rows);
        if (m->pending_slices != 0) throw std::runtime_error("forward: a split-K residual was left unconsumed");
        m->site = KS_LM_HEAD;
        gemm(m, EPI_STORE, true, mk(m->xn_last, H, m->lm_head, m->logits, m->Vpad, nullptr, 1.f, rows, m->V, H));
    }
    if (seam & 4) {      // sampling reads the logits: apply LogitsScaling here (argmax_partial_kernel does it otherwise)
        if (c.logits_scaling != 0.f) {
            KScope ks(m, KC_OTHER);
            hipLaunchKernelGGL(scale_rows_kernel, dim3(cdiv(m->V, 256), rows), dim3(256), 0, m->stream, m->logits, m->Vpad, m->V,
                               c.logits_scaling);
            NVL_HIP(hipGetLastError());
        }
    } else {
        KScope ks(m, KC_OTHER, 0, KS_ARGMAX, (double)rows * m->V * 4.0);
        launch_argmax(m->stream, m->logits, m->Vpad, m->V, rows, c.logits_scaling, m->argmax_pval, m->argmax_pidx,
                      (seam & 2) ? nullptr : m->argmax_dev);
        NVL_HIP(hipGetLastError());
    }
    m->last_rows = rows;
    m->ctx_hint = 0;
    return rows;
}

// after a decode step: feed the greedy tokens back as the next step's inputs and advance every position by one
// (cmd/ask/main.go:315-320: allTokens = append(allTokens, nextToken); ForwardWithCache([last], kv, len-1))
__global__ void advance_decode_kernel(const int32_t* __restrict__ argmax, int32_t* __restrict__ tokens,
                                      int32_t* __restrict__ tok_pos, int32_t* __restrict__ seq_pos,
                                      int32_t* __restrict__ out_step, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = argmax[i];
    out_step[i] = t;
    tokens[i] = t;
    tok_pos[i] += 1;
    seq_pos[i] += 1;
}
}  // namespace

extern "C" int nvl_forward(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* tokens,
                           const int32_t* seq_lens, const int32_t* pos_offsets, uint32_t flags,
                           float* logits_out, int32_t* argmax_out) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_forward: model not finalized");
    if (n_seqs <= 0 || !seq_ids || !tokens || !seq_lens || !pos_offsets)
        return fail(m, NVL_ERR_INVALID, "nvl_forward: null/empty arguments");
    if (n_seqs > m->opts.max_seqs) return fail(m, NVL_ERR_INVALID, "nvl_forward: n_seqs exceeds max_seqs");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const nvl_model_config& c = m->cfg;
    const int H = m->H;
    // ---- validate + build metadata -------------------------------------------------------
    int64_t M64 = 0; int max_len = 0; bool prefill = false;
    for (int i = 0; i < n_seqs; i++) {
        if (seq_lens[i] <= 0) return fail(m, NVL_ERR_INVALID, "nvl_forward: empty sequence");
        M64 += seq_lens[i]; max_len = std::max(max_len, seq_lens[i]);
        if (seq_lens[i] > 1) prefill = true;
    }
    if (M64 > m->opts.max_batch_tokens) return fail(m, NVL_ERR_INVALID, "nvl_forward: batch exceeds max_batch_tokens");
    const int M = (int)M64;
    if (m->paged) return fail(m, NVL_ERR_STATE, "nvl_forward: the model is in paged-KV mode (use nvl_forward_paged)");
    const Meta hm = bind_meta(m, m->meta_host, M);
    int32_t *h_sts = hm.seq_tok_start, *h_len = hm.seq_len, *h_pos = hm.seq_pos, *h_tbl = hm.seq_tbl, *h_last = hm.last_rows,
            *h_tokens = hm.tokens, *h_tok_pos = hm.tok_pos, *h_tok_tbl = hm.tok_tbl;
    std::vector<int> h_slot((size_t)n_seqs);
    int t = 0;
    for (int i = 0; i < n_seqs; i++) {
        auto it = m->seq_slot.find(seq_ids[i]);
        if (it == m->seq_slot.end()) return fail(m, NVL_ERR_UNKNOWN_SEQ, "nvl_forward: sequence has no KV slot (nvl_seq_open first)");
        const int slot = it->second;
        for (int j = 0; j < i; j++) if (seq_ids[j] == seq_ids[i]) return fail(m, NVL_ERR_INVALID, "nvl_forward: duplicate sequence in batch");
        if (pos_offsets[i] != m->slot_len[(size_t)slot]) return fail(m, NVL_ERR_INVALID, "nvl_forward: pos_offset does not equal the cached length");
        if (pos_offsets[i] > c.max_seq_len - seq_lens[i])   // rope.go:84-86 / :176-178 panic
            return fail(m, NVL_ERR_POSITION, "nvl_forward: position exceeds max_seq_len");
        // slab mode: the sequence's block table is the one entry blk_table[i] = its slot
        h_sts[i] = t; h_len[i] = seq_lens[i]; h_pos[i] = pos_offsets[i]; h_slot[(size_t)i] = slot; h_tbl[i] = i; hm.blk_table[i] = slot;     // (blocks_per_seq == 1 here)
        for (int j = 0; j < seq_lens[i]; j++, t++) {
            const int tok = tokens[t];
            if (tok < 0 || tok >= m->V) return fail(m, NVL_ERR_INVALID, "nvl_forward: token id out of range");
            h_tokens[t] = tok; h_tok_pos[t] = pos_offsets[i] + j; h_tok_tbl[t] = i;
        }
        h_last[i] = t - 1;
    }
    const Meta md = bind_meta(m, m->meta_dev, M);
    NVL_HIP(hipEventRecord(m->ev0, m->stream));

    double attn_flops = 0, kv_tok = 0;   // 4*hd per (query, visible key) pair per head; cached + new keys of the batch
    for (int i = 0; i < n_seqs; i++) {
        const double s = seq_lens[i], p0 = pos_offsets[i];
        attn_flops += 4.0 * m->hd * m->nH * (s * p0 + s * (s + 1) / 2);
        m->ctx_hint = std::max(m->ctx_hint, (int)(pos_offsets[i] + seq_lens[i]));
        kv_tok += p0 + s;
    }
    m->kv_tok = kv_tok;
    const bool all = (flags & NVL_FWD_ALL_LOGITS) != 0;
    int rows = all ? M : n_seqs;
    std::vector<int32_t> am;
    const int32_t* am_p = nullptr;
    bool replayed = false;
    if (!prefill && rows <= m->logit_rows) {
        // a decode pass (what TensorModelRunner.Run issues every step): metadata upload + all launches + the argmax
        // download as ONE graph launch once this configuration has been seen
        const std::array<int, 5> key{0, n_seqs, decode_attn_waves(m, n_seqs), (int)flags, g_tune_epoch};
        replayed = replay_or_capture(m, key, [&] {
            NVL_HIP(hipMemcpyAsync(m->meta_dev, m->meta_host, meta_bytes(m, M), hipMemcpyHostToDevice, m->stream));
            enqueue_forward(m, md, n_seqs, M, max_len, attn_flops, flags);
            NVL_HIP(hipMemcpyAsync(m->am_host, m->argmax_dev, (size_t)rows * 4, hipMemcpyDeviceToHost, m->stream));
        });
        if (replayed) { am_p = m->am_host; m->ctx_hint = 0; }
    }
    if (!replayed) {
        NVL_HIP(hipMemcpyAsync(m->meta_dev, m->meta_host, meta_bytes(m, M), hipMemcpyHostToDevice, m->stream));
        rows = enqueue_forward(m, md, n_seqs, M, max_len, attn_flops, flags);
        am.resize((size_t)rows);
        NVL_HIP(hipMemcpyAsync(am.data(), m->argmax_dev, (size_t)rows * 4, hipMemcpyDeviceToHost, m->stream));
        am_p = am.data();
    }
    NVL_HIP(hipEventRecord(m->ev1, m->stream));
    if (logits_out)
        NVL_HIP(hipMemcpy2DAsync(logits_out, (size_t)m->V * 4, m->logits, (size_t)m->Vpad * 4, (size_t)m->V * 4,
                                 (size_t)rows, hipMemcpyDeviceToHost, m->stream));
    NVL_HIP(hipStreamSynchronize(m->stream));
    reap_graphs(m);
    p2p_check(m);
    m->last_rows = rows;
    if (argmax_out) {
        if (all) for (int i = 0; i < n_seqs; i++) argmax_out[i] = am_p[(size_t)h_last[i]];
        else for (int i = 0; i < n_seqs; i++) argmax_out[i] = am_p[(size_t)i];
    }
    for (int i = 0; i < n_seqs; i++) { m->slot_len[(size_t)h_slot[(size_t)i]] += seq_lens[i]; m->slot_tick[(size_t)h_slot[(size_t)i]] = ++m->tick; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, m->ev0, m->ev1);
    m->stats.forward_calls++;
    if (prefill) { m->stats.prefill_tokens += (uint64_t)M; m->stats.prefill_ms += ms; }
    else { m->stats.decode_tokens += (uint64_t)M; m->stats.decode_ms += ms; }
    if (m->profile) drain_profile(m);
    return NVL_OK;
    NVL_CATCH(m)
}

// nvl_forward for a model whose KV cache is the host block manager's pool (block_manager.go:128-263,
// sequence.go:22-23): no per-sequence state here, the block tables say where every position lives.
extern "C" int nvl_forward_paged(nvl_model* m, int n_seqs, const int32_t* tokens, const int32_t* seq_lens,
                                 const int32_t* pos_offsets, const int32_t* block_tables, const int32_t* table_offsets,
                                 uint32_t flags, float* logits_out, int32_t* argmax_out) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_forward_paged: model not finalized");
    if (!m->paged) return fail(m, NVL_ERR_STATE, "nvl_forward_paged: the model was created without kv_num_blocks");
    if (n_seqs <= 0 || !tokens || !seq_lens || !pos_offsets || !block_tables || !table_offsets)
        return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: null/empty arguments");
    if (n_seqs > m->opts.max_seqs) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: n_seqs exceeds max_seqs");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const nvl_model_config& c = m->cfg;
    const int BS = m->Tmax;
    int64_t M64 = 0; int max_len = 0; bool prefill = false;
    for (int i = 0; i < n_seqs; i++) {
        if (seq_lens[i] <= 0) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: empty sequence");
        if (pos_offsets[i] < 0) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: negative pos_offset");
        M64 += seq_lens[i]; max_len = std::max(max_len, seq_lens[i]);
        if (seq_lens[i] > 1) prefill = true;
    }
    if (M64 > m->opts.max_batch_tokens) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: batch exceeds max_batch_tokens");
    const int M = (int)M64;
    const Meta hm = bind_meta(m, m->meta_host, M);
    int t = 0, tb = 0;
    if (table_offsets[0] < 0) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: negative table offset");
    for (int i = 0; i < n_seqs; i++) {
        if (pos_offsets[i] > c.max_seq_len - seq_lens[i])      // (written so that the sum cannot overflow)
            return fail(m, NVL_ERR_POSITION, "nvl_forward_paged: position exceeds max_seq_len");
        const int end = pos_offsets[i] + seq_lens[i];
        if (table_offsets[i + 1] < table_offsets[i]) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: table_offsets must be non-decreasing");
        const int nb = table_offsets[i + 1] - table_offsets[i];
        if (nb < cdiv(end, BS)) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: block table shorter than the sequence");
        if (nb > m->blocks_per_seq) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: block table too long");
        tb = i * m->blocks_per_seq;                      // fixed stride: the kernels index blk_table[seq][block]
        hm.seq_tok_start[i] = t; hm.seq_len[i] = seq_lens[i]; hm.seq_pos[i] = pos_offsets[i]; hm.seq_tbl[i] = tb;
        for (int j = 0; j < nb; j++) {
            const int b = block_tables[table_offsets[i] + j];
            if (b < 0 || b >= m->num_blocks) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: block id out of range");
            hm.blk_table[tb + j] = b;
        }
        for (int j = 0; j < seq_lens[i]; j++, t++) {
            const int tok = tokens[t];
            if (tok < 0 || tok >= m->V) return fail(m, NVL_ERR_INVALID, "nvl_forward_paged: token id out of range");
            hm.tokens[t] = tok; hm.tok_pos[t] = pos_offsets[i] + j; hm.tok_tbl[t] = tb;
        }
        hm.last_rows[i] = t - 1;
    }
    const Meta md = bind_meta(m, m->meta_dev, M);
    NVL_HIP(hipEventRecord(m->ev0, m->stream));
    NVL_HIP(hipMemcpyAsync(m->meta_dev, m->meta_host, meta_bytes(m, M), hipMemcpyHostToDevice, m->stream));
    double attn_flops = 0, kv_tok = 0;
    for (int i = 0; i < n_seqs; i++) {
        const double s = seq_lens[i], p0 = pos_offsets[i];
        attn_flops += 4.0 * m->hd * m->nH * (s * p0 + s * (s + 1) / 2);
        m->ctx_hint = std::max(m->ctx_hint, (int)(pos_offsets[i] + seq_lens[i]));
        kv_tok += p0 + s;
    }
    m->kv_tok = kv_tok;
    const bool all = (flags & NVL_FWD_ALL_LOGITS) != 0;
    const int rows = enqueue_forward(m, md, n_seqs, M, max_len, attn_flops, flags);
    NVL_HIP(hipEventRecord(m->ev1, m->stream));
    std::vector<int32_t> am((size_t)rows);
    NVL_HIP(hipMemcpyAsync(am.data(), m->argmax_dev, (size_t)rows * 4, hipMemcpyDeviceToHost, m->stream));
    if (logits_out)
        NVL_HIP(hipMemcpy2DAsync(logits_out, (size_t)m->V * 4, m->logits, (size_t)m->Vpad * 4, (size_t)m->V * 4,
                                 (size_t)rows, hipMemcpyDeviceToHost, m->stream));
    NVL_HIP(hipStreamSynchronize(m->stream));
    if (argmax_out) {
        if (all) for (int i = 0; i < n_seqs; i++) argmax_out[i] = am[(size_t)hm.last_rows[i]];
        else for (int i = 0; i < n_seqs; i++) argmax_out[i] = am[(size_t)i];
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, m->ev0, m->ev1);
    m->stats.forward_calls++;
    if (prefill) { m->stats.prefill_tokens += (uint64_t)M; m->stats.prefill_ms += ms; }
    else { m->stats.decode_tokens += (uint64_t)M; m->stats.decode_ms += ms; }
    if (m->profile) drain_profile(m);
    return NVL_OK;
    NVL_CATCH(m)
}

// ModelRunner.Run over block tables: what a runner that honours Sequence.BlockTable / NumCachedTokens does
// (the reference's runners ignore both: SURVEY §2 row 17).
extern "C" int nvl_runner_run_paged(nvl_model* m, int n_seqs, const int32_t* const* token_ptrs, const int32_t* token_lens,
                                    const int32_t* num_cached_tokens, const int32_t* const* block_table_ptrs,
                                    const int32_t* block_table_lens, int is_prefill, int32_t* next_tokens,
                                    float* logits_out) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->paged) return fail(m, NVL_ERR_STATE, "nvl_runner_run_paged: the model was created without kv_num_blocks");
    if (n_seqs <= 0 || !token_ptrs || !token_lens || !block_table_ptrs || !block_table_lens || !next_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_runner_run_paged: null/empty arguments");
    NVL_TRY(m)
    const int V = m->V;
    int k = 0;
    while (k < n_seqs) {
        // pack sequences into one forward call while they fit max_seqs / max_batch_tokens
        std::vector<int32_t> toks, lens, pos, tbl, off{0};
        std::vector<int> who;
        int64_t budget = m->opts.max_batch_tokens;
        while (k < n_seqs && (int)who.size() < m->opts.max_seqs) {
            const int len = token_lens[k];
            if (len <= 0) return fail(m, NVL_ERR_INVALID, "nvl_runner_run_paged: sequence without tokens");
            int p0;
            if (is_prefill) {
                p0 = num_cached_tokens ? num_cached_tokens[k] : 0;
                if (p0 < 0 || p0 > len) return fail(m, NVL_ERR_INVALID, "nvl_runner_run_paged: num_cached_tokens outside [0, len]");
                if (p0 == len) p0 = len - 1;                 // whole prompt cached: the last token still has to produce logits
            } else {
                p0 = len - 1;                                // tensor_model_runner.go:78-80
            }
            const int n_new = len - p0;
            if (n_new > m->opts.max_batch_tokens) return fail(m, NVL_ERR_INVALID, "nvl_runner_run_paged: uncached suffix exceeds max_batch_tokens");
            if (n_new > budget) break;
            toks.insert(toks.end(), token_ptrs[k] + p0, token_ptrs[k] + len);
            lens.push_back(n_new); pos.push_back(p0);
            if (block_table_lens[k] < 0) return fail(m, NVL_ERR_INVALID, "nvl_runner_run_paged: negative block table length");
            tbl.insert(tbl.end(), block_table_ptrs[k], block_table_ptrs[k] + block_table_lens[k]);
            off.push_back((int32_t)tbl.size());
            who.push_back(k);
            budget -= n_new;
            k++;
        }
        std::vector<int32_t> out(who.size());
        std::vector<float> lg(logits_out ? who.size() * (size_t)V : 0);
        const int rc = nvl_forward_paged(m, (int)who.size(), toks.data(), lens.data(), pos.data(), tbl.data(), off.data(), 0,
                                         logits_out ? lg.data() : nullptr, out.data());
        if (rc) return rc;
        for (size_t j = 0; j < who.size(); j++) {
            next_tokens[who[j]] = out[j];
            if (logits_out) memcpy(logits_out + (size_t)who[j] * V, &lg[j * (size_t)V], (size_t)V * 4);
        }
    }
    return NVL_OK;
    NVL_CATCH(m)
}

namespace {
// the n_steps decode passes of nvl_decode_greedy[_paged]: metadata for step 0 is in meta_host (one token per sequence),
// every later step's tokens / positions are produced on the device
// `sp` != NULL: every step samples (tensor.SampleWithHistory, sample.h) instead of taking the argmax; the histories live
// on the device (m->samp_hist, one row of max_seq_len ids per sequence) and grow by the sampled token each step;
// `uniforms_dev` is [n_steps][n_seqs]
void decode_loop(nvl_model* m, const Meta& hm, int n_seqs, int n_steps, int32_t* out_tokens,
                 const nvl_sampling_params* sp = nullptr, const float* uniforms_dev = nullptr) {
    const int M = n_seqs;
    const int32_t* h_pos = hm.seq_pos;
    const Meta md = bind_meta(m, m->meta_dev, M);
    if ((int64_t)n_steps * n_seqs > m->ring_ints) {
        clear_graphs(m);      // the fused loop's captured steps hold the old ring address as a kernel argument
        dfree(m->ring);
        m->ring_ints = (int64_t)n_steps * n_seqs;
        m->ring = dmalloc<int32_t>(m->ring_ints);
    }
    NVL_HIP(hipEventRecord(m->ev0, m->stream));
    NVL_HIP(hipMemcpyAsync(m->meta_dev, m->meta_host, meta_bytes(m, M), hipMemcpyHostToDevice, m->stream));
    NVL_HIP(hipMemcpyAsync(m->ring_pos0, h_pos, (size_t)n_seqs * 4, hipMemcpyHostToDevice, m->stream));   // (pinned meta_host)
    const bool dbg = m->keep_hidden, dbg_tap = m->tap;
    m->keep_hidden = false; m->tap = false;      // the per-layer taps belong to single nvl_forward calls
    for (int s = 0; s < n_steps; s++) {
        double attn_flops = 0, kv_tok = 0;
        for (int i = 0; i < n_seqs; i++) {
            attn_flops += 4.0 * m->hd * m->nH * (double)(h_pos[i] + s + 1);
            m->ctx_hint = std::max(m->ctx_hint, h_pos[i] + s + 1);
            kv_tok += h_pos[i] + s + 1;
        }
        m->kv_tok = kv_tok;
        // bf16 path: the step's tail (argmax, token feedback) and the next step's head (embedding gather + layer 0's
        // norm) are one launch (decode_seam_kernel); fp32 parity mode keeps the separate kernels
        const bool seam_ok = (g_decode_seam || sp) && !m->f32 && m->H <= 1024 * NORM_ROW_MAXCH && m->H % 4 == 0;
        if (sp && !seam_ok) throw std::runtime_error("sampled decode loop: bf16 models only");
        auto one_step = [&] {
        enqueue_forward(m, md, n_seqs, M, 1, attn_flops, 0, seam_ok ? ((s > 0 ? 1 : 0) | (sp ? 4 : 2)) : 0);
        if (sp) {
            SampleArgs a{};
            a.logits = m->logits; a.ld = m->Vpad; a.work = m->samp.work; a.cnt = m->samp.cnt; a.hist = m->samp_hist;
            a.hist_off = nullptr; a.hist_stride = m->cfg.max_seq_len; a.hist_len = m->samp_hist_len;
            a.uniforms = uniforms_dev + (int64_t)s * n_seqs; a.out = m->samp.out; a.probs_out = nullptr; a.ldp = m->V;
            a.V = m->V; a.top_k = sp->top_k; a.temperature = sp->temperature; a.top_p = sp->top_p; a.rep_penalty = sp->repetition_penalty;
            hipLaunchKernelGGL(sample_row_kernel, dim3(n_seqs), dim3(SAMPLE_THREADS), 0, m->stream, a);
            NVL_HIP(hipGetLastError());
        }
        if (seam_ok) {
            const void* pe = (m->cfg.position_type == NVL_POS_LEARNED) ? m->g[NVL_T_POS_EMB].p : nullptr;
            const int pe_rows = pe ? (int)std::min<int64_t>(m->g[NVL_T_POS_EMB].rows, m->cfg.max_seq_len) : 0;
            const LayerW& l0 = m->layers[0];
            hipLaunchKernelGGL(decode_seam_kernel, dim3(n_seqs), dim3(256), 0, m->stream, m->argmax_pval, m->argmax_pidx,
                               cdiv(m->V, ARGMAX_CHUNK), sp ? (const int32_t*)m->samp.out : (const int32_t*)nullptr, m->samp_hist,
                               (int64_t)m->cfg.max_seq_len, m->samp_hist_len, m->argmax_dev, m->ring, (const int32_t*)m->ring_pos0, n_seqs,
                               md.tokens, md.tok_pos,
                               md.seq_pos, (const bf16_t*)m->g[NVL_T_TOK_EMB].p, (const bf16_t*)pe, pe_rows,
                               m->cfg.embedding_multiplier, m->x, (const float*)l0.t[NVL_T_ATTN_NORM_W].p,
                               (const float*)l0.t[NVL_T_ATTN_NORM_B].p, m->cfg.norm_eps, (bf16_t*)m->xn, m->H);
        } else {
            hipLaunchKernelGGL(advance_decode_kernel, dim3(cdiv(n_seqs, 256)), dim3(256), 0, m->stream, m->argmax_dev, md.tokens,
                               md.tok_pos, md.seq_pos, m->ring + (int64_t)s * n_seqs, n_seqs);
        }
        NVL_HIP(hipGetLastError());
        };
        // steps 1.. of the greedy seam loop launch the SAME kernels with the SAME arguments (the ring row comes from the
        // positions): one captured graph per (batch, attention wave count), replayed
        bool replayed = false;
        if (s > 0 && seam_ok && !sp) {
            const std::array<int, 5> key{1, n_seqs, decode_attn_waves(m, n_seqs), 0, g_tune_epoch};
            replayed = replay_or_capture(m, key, one_step);
            if (replayed) m->ctx_hint = 0;
        }
        if (!replayed) one_step();
    }
    m->keep_hidden = dbg; m->tap = dbg_tap;
    NVL_HIP(hipEventRecord(m->ev1, m->stream));
    NVL_HIP(hipMemcpyAsync(out_tokens, m->ring, (size_t)n_steps * n_seqs * 4, hipMemcpyDeviceToHost, m->stream));
    NVL_HIP(hipStreamSynchronize(m->stream));
    reap_graphs(m);
    p2p_check(m);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, m->ev0, m->ev1);
    m->stats.forward_calls += (uint64_t)n_steps;
    m->stats.decode_tokens += (uint64_t)n_steps * n_seqs; m->stats.decode_ms += ms;
    if (m->profile) drain_profile(m);
}
}  // namespace

// The greedy decode loop of cmd/ask (main.go:315-360, without the EOS stop) as ONE call: `n_steps` forward passes of one
// token per sequence, each step's argmax fed back on the device — no host round trip between steps.  Identical
// arithmetic to calling nvl_forward n_steps times with the returned tokens.
extern "C" int nvl_decode_greedy(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* first_tokens,
                                 int n_steps, int32_t* out_tokens) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_decode_greedy: model not finalized");
    if (n_seqs <= 0 || n_steps <= 0 || !seq_ids || !first_tokens || !out_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy: null/empty arguments");
    if (n_seqs > m->opts.max_seqs || n_seqs > m->opts.max_batch_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy: n_seqs exceeds max_seqs / max_batch_tokens");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    if (m->paged) return fail(m, NVL_ERR_STATE, "nvl_decode_greedy: not available in paged-KV mode");
    const int M = n_seqs;
    const Meta hm = bind_meta(m, m->meta_host, M);
    int32_t *h_sts = hm.seq_tok_start, *h_len = hm.seq_len, *h_pos = hm.seq_pos, *h_tbl = hm.seq_tbl, *h_last = hm.last_rows,
            *h_tokens = hm.tokens, *h_tok_pos = hm.tok_pos, *h_tok_tbl = hm.tok_tbl;
    std::vector<int> h_slot((size_t)n_seqs);
    for (int i = 0; i < n_seqs; i++) {
        auto it = m->seq_slot.find(seq_ids[i]);
        if (it == m->seq_slot.end()) return fail(m, NVL_ERR_UNKNOWN_SEQ, "nvl_decode_greedy: sequence has no KV slot");
        for (int j = 0; j < i; j++) if (seq_ids[j] == seq_ids[i]) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy: duplicate sequence");
        const int slot = it->second, pos = m->slot_len[(size_t)slot];
        if (pos + n_steps > m->cfg.max_seq_len) return fail(m, NVL_ERR_POSITION, "nvl_decode_greedy: position exceeds max_seq_len");
        if (first_tokens[i] < 0 || first_tokens[i] >= m->V) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy: token id out of range");
        h_sts[i] = i; h_len[i] = 1; h_pos[i] = pos; h_slot[(size_t)i] = slot; h_last[i] = i; h_tbl[i] = i; hm.blk_table[i] = slot;
        h_tokens[i] = first_tokens[i]; h_tok_pos[i] = pos; h_tok_tbl[i] = i;
    }
    decode_loop(m, hm, n_seqs, n_steps, out_tokens);
    for (int i = 0; i < n_seqs; i++) { m->slot_len[(size_t)h_slot[(size_t)i]] += n_steps; m->slot_tick[(size_t)h_slot[(size_t)i]] = ++m->tick; }
    return NVL_OK;
    NVL_CATCH(m)
}

// nvl_decode_greedy for a paged-KV model: positions[i] tokens are cached (the sequence length before the first
// generated token is appended); the block tables must already cover positions[i] + n_steps tokens (the host's block
// manager allocates ahead: BlockManager.MayAppend per step, block_manager.go:231-263).
extern "C" int nvl_decode_greedy_paged(nvl_model* m, int n_seqs, const int32_t* first_tokens, const int32_t* positions,
                                       int n_steps, const int32_t* block_tables, const int32_t* table_offsets,
                                       int32_t* out_tokens) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_decode_greedy_paged: model not finalized");
    if (!m->paged) return fail(m, NVL_ERR_STATE, "nvl_decode_greedy_paged: the model was created without kv_num_blocks");
    if (n_seqs <= 0 || n_steps <= 0 || !first_tokens || !positions || !block_tables || !table_offsets || !out_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: null/empty arguments");
    if (n_seqs > m->opts.max_seqs || n_seqs > m->opts.max_batch_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: n_seqs exceeds max_seqs / max_batch_tokens");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const int M = n_seqs, BS = m->Tmax;
    const Meta hm = bind_meta(m, m->meta_host, M);
    for (int i = 0; i < n_seqs; i++) {
        const int pos = positions[i];
        if (pos < 0) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: negative position");
        if (pos > m->cfg.max_seq_len - n_steps) return fail(m, NVL_ERR_POSITION, "nvl_decode_greedy_paged: position exceeds max_seq_len");
        const int end = pos + n_steps;
        if ((i == 0 && table_offsets[0] < 0) || table_offsets[i + 1] < table_offsets[i])
            return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: table_offsets must be non-negative and non-decreasing");
        if (first_tokens[i] < 0 || first_tokens[i] >= m->V) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: token id out of range");
        const int nb = table_offsets[i + 1] - table_offsets[i];
        if (nb < cdiv(end, BS) || nb > m->blocks_per_seq) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: block table does not cover the generated positions");
        const int tb = i * m->blocks_per_seq;
        for (int j = 0; j < nb; j++) {
            const int b = block_tables[table_offsets[i] + j];
            if (b < 0 || b >= m->num_blocks) return fail(m, NVL_ERR_INVALID, "nvl_decode_greedy_paged: block id out of range");
            hm.blk_table[tb + j] = b;
        }
        hm.seq_tok_start[i] = i; hm.seq_len[i] = 1; hm.seq_pos[i] = pos; hm.seq_tbl[i] = tb; hm.last_rows[i] = i;
        hm.tokens[i] = first_tokens[i]; hm.tok_pos[i] = pos; hm.tok_tbl[i] = tb;
    }
    decode_loop(m, hm, n_seqs, n_steps, out_tokens);
    return NVL_OK;
    NVL_CATCH(m)
}

// nvl_decode_greedy with the reference runner's sampling step instead of the argmax (tensor_model_runner.go:89-93):
// every step draws SampleWithHistory(logits, seq.TokenIDs, params) on the device and feeds the token back.
// history_ptrs[i]/history_lens[i] = Sequence.TokenIDs as the first decode step sees it (prompt + the token sampled from
// the prefill = first_tokens[i]); uniforms is [n_steps][n_seqs] (one rand.Float32() per sequence per step, in the
// order the host's serial loop would draw them).
extern "C" int nvl_decode_sampled(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* first_tokens, int n_steps,
                                  const nvl_sampling_params* params, const int32_t* const* history_ptrs,
                                  const int32_t* history_lens, const float* uniforms, int32_t* out_tokens) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_decode_sampled: model not finalized");
    if (m->paged) return fail(m, NVL_ERR_STATE, "nvl_decode_sampled: not available in paged-KV mode");
    if (m->f32) return fail(m, NVL_ERR_STATE, "nvl_decode_sampled: bf16 models only");
    if (n_seqs <= 0 || n_steps <= 0 || !seq_ids || !first_tokens || !params || !history_ptrs || !history_lens || !uniforms || !out_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: null/empty arguments");
    if (n_seqs > m->opts.max_seqs || n_seqs > m->opts.max_batch_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: n_seqs exceeds max_seqs / max_batch_tokens");
    if (params->repetition_penalty == 0.f || !(params->repetition_penalty == params->repetition_penalty) ||
        !(params->temperature == params->temperature) || !(params->top_p == params->top_p))
        return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: bad sampling parameters");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const int M = n_seqs, T = m->cfg.max_seq_len;
    const Meta hm = bind_meta(m, m->meta_host, M);
    std::vector<int> h_slot((size_t)n_seqs);
    std::vector<int32_t> hist((size_t)n_seqs * T, 0), hlen((size_t)n_seqs);
    for (int i = 0; i < n_seqs; i++) {
        auto it = m->seq_slot.find(seq_ids[i]);
        if (it == m->seq_slot.end()) return fail(m, NVL_ERR_UNKNOWN_SEQ, "nvl_decode_sampled: sequence has no KV slot");
        for (int j = 0; j < i; j++) if (seq_ids[j] == seq_ids[i]) return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: duplicate sequence");
        const int slot = it->second, pos = m->slot_len[(size_t)slot];
        if (pos + n_steps > T) return fail(m, NVL_ERR_POSITION, "nvl_decode_sampled: position exceeds max_seq_len");
        if (first_tokens[i] < 0 || first_tokens[i] >= m->V) return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: token id out of range");
        const int hn = history_lens[i];
        if (hn < 0 || hn + n_steps > T || (hn > 0 && !history_ptrs[i])) return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: history longer than max_seq_len");
        for (int j = 0; j < hn; j++) {
            if (history_ptrs[i][j] < 0) return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: negative token id in the history");
            hist[(size_t)i * T + j] = history_ptrs[i][j];
        }
        hlen[(size_t)i] = hn;
        hm.seq_tok_start[i] = i; hm.seq_len[i] = 1; hm.seq_pos[i] = pos; h_slot[(size_t)i] = slot; hm.last_rows[i] = i;
        hm.seq_tbl[i] = i; hm.blk_table[i] = slot;
        hm.tokens[i] = first_tokens[i]; hm.tok_pos[i] = pos; hm.tok_tbl[i] = i;
    }
    for (int64_t i = 0; i < (int64_t)n_steps * n_seqs; i++)
        if (!(uniforms[i] >= 0.f && uniforms[i] <= 1.f)) return fail(m, NVL_ERR_INVALID, "nvl_decode_sampled: uniform draw outside [0, 1]");
    // device state of the sampler
    SampleBufs& b = m->samp;
    const int64_t elems = (int64_t)n_seqs * round_up(m->V, 4);
    if (elems > b.elems) {
        dfree(b.work); dfree(b.cnt);
        b.work = dmalloc<float>(elems); b.cnt = dmalloc<int32_t>(elems); b.elems = elems;
        NVL_HIP(hipMemsetAsync(b.cnt, 0, (size_t)elems * 4, m->stream));
    }
    if (n_seqs > b.rows_cap) {
        dfree(b.off); dfree(b.out); dfree(b.u);
        b.rows_cap = (int)round_up(n_seqs, 64);
        b.off = dmalloc<int32_t>(b.rows_cap + 1); b.out = dmalloc<int32_t>(b.rows_cap); b.u = dmalloc<float>(b.rows_cap);
    }
    if ((int64_t)n_seqs * T > m->samp_hist_cap) {
        dfree(m->samp_hist); dfree(m->samp_hist_len);
        m->samp_hist_cap = (int64_t)m->opts.max_seqs * T;
        m->samp_hist = dmalloc<int32_t>(m->samp_hist_cap); m->samp_hist_len = dmalloc<int32_t>(m->opts.max_seqs);
    }
    if ((int64_t)n_steps * n_seqs > m->samp_u_cap) {
        dfree(m->samp_u_steps);
        m->samp_u_cap = (int64_t)n_steps * n_seqs;
        m->samp_u_steps = dmalloc<float>(m->samp_u_cap);
    }
    NVL_HIP(hipMemcpyAsync(m->samp_hist, hist.data(), hist.size() * 4, hipMemcpyHostToDevice, m->stream));
    NVL_HIP(hipMemcpyAsync(m->samp_hist_len, hlen.data(), hlen.size() * 4, hipMemcpyHostToDevice, m->stream));
    NVL_HIP(hipMemcpyAsync(m->samp_u_steps, uniforms, (size_t)n_steps * n_seqs * 4, hipMemcpyHostToDevice, m->stream));
    decode_loop(m, hm, n_seqs, n_steps, out_tokens, params, m->samp_u_steps);   // (synchronises: the host vectors stay alive)
    for (int i = 0; i < n_seqs; i++) { m->slot_len[(size_t)h_slot[(size_t)i]] += n_steps; m->slot_tick[(size_t)h_slot[(size_t)i]] = ++m->tick; }
    return NVL_OK;
    NVL_CATCH(m)
}

// =================================================================================================
// debug / parity taps, stats
// =================================================================================================
extern "C" int nvl_set_debug(nvl_model* m, int keep_hidden) {
    if (!m) return NVL_ERR_INVALID;
    m->keep_hidden = keep_hidden == 1;      // 1: per-layer residual stream, every residual add completed in its own launch
    m->tap = keep_hidden == 2;              // 2: the same record taken beside the unmodified product path (pending adds included)
    m->stamping = keep_hidden == 4;         // 4: in-kernel time stamps of the decode kernels (nvl_get_stamps), eager launches
#ifndef NVL_STAMPS
    if (m->stamping) {      // the product library carries no stamp sites (common.h NvlStamps): the diagnostic build does
        m->stamping = false;
        return fail(m, NVL_ERR_STATE, "nvl_set_debug: mode 4 needs the diagnostic build (make -C csrc diag -> libnvllm_hip_diag.so, NVLLM_LIB)");
    }
#endif
    if (m->stamping) {
        NVL_TRY(m)
        NVL_HIP(hipSetDevice(m->device));
        if (!m->stamp_buf) m->stamp_buf = (unsigned long long*)dmalloc_bytes((int64_t)STAMP_MAX_LAUNCH * STAMP_MAX_WG * 8 * 8);
        NVL_HIP(hipStreamSynchronize(m->stream));
        NVL_HIP(hipMemset(m->stamp_buf, 0, (size_t)STAMP_MAX_LAUNCH * STAMP_MAX_WG * 8 * 8));
        m->stamp_launches = 0; m->stamp_recs.clear();
        NVL_CATCH(m)
    }
    return NVL_OK;
}
// nvl_set_debug(m, 4) ... forward passes ... nvl_get_stamps: per stamped launch {site, phase, workgroups} in `recs`
// (3 ints each) and its stamps [workgroups][8] of the 100 MHz clock back to back in `stamps`.  Returns the launch count.
extern "C" int nvl_get_stamps(nvl_model* m, int32_t* recs, int rec_cap, uint64_t* stamps, int64_t stamp_cap) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->stamp_buf) return fail(m, NVL_ERR_STATE, "nvl_get_stamps: nvl_set_debug(m, 4) first");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    NVL_HIP(hipStreamSynchronize(m->stream));
    const int n = (int)m->stamp_recs.size();
    int64_t off = 0;
    for (int i = 0; i < n && i < rec_cap; i++) {
        const auto& r = m->stamp_recs[(size_t)i];
        recs[3 * i] = r.site; recs[3 * i + 1] = r.phase; recs[3 * i + 2] = r.nwg;
        if (off + (int64_t)r.nwg * 8 > stamp_cap) return fail(m, NVL_ERR_INVALID, "nvl_get_stamps: stamp buffer too small");
        NVL_HIP(hipMemcpy(stamps + off, m->stamp_buf + (size_t)i * STAMP_MAX_WG * 8, (size_t)r.nwg * 64, hipMemcpyDeviceToHost));
        off += (int64_t)r.nwg * 8;
    }
    return n;
    NVL_CATCH(m)
}
extern "C" int nvl_get_hidden(nvl_model* m, int layer, float* out, int64_t n_floats) {
    if (!m || !out) return NVL_ERR_INVALID;
    if (!m->hidden || layer < 0 || layer >= m->L) return fail(m, NVL_ERR_STATE, "nvl_get_hidden: nothing recorded");
    NVL_TRY(m)
    const int64_t per = (int64_t)m->hidden_last_M * m->H;   // [L][M][H] of the recording call
    if (n_floats > per) n_floats = per;
    NVL_HIP(hipMemcpy(out, m->hidden + (int64_t)layer * per, (size_t)n_floats * 4, hipMemcpyDeviceToHost));
    return NVL_OK;
    NVL_CATCH(m)
}

namespace {
// K and V of positions 0..T-1 of one block list, laid out as the reference does ([nKV, T, hd] fp32, kv_cache.go:5-6)
void read_kv(nvl_model* m, const int32_t* blocks, int n_blocks, int T, int layer, float* k_out, float* v_out) {
    const int hd = m->hd, BS = m->Tmax;
    const int64_t n = m->layer_stride;
    std::vector<char> kb((size_t)n * m->wsize), vb((size_t)n * m->wsize);
    auto rd = [&](const std::vector<char>& b, int64_t i) -> float {
        if (m->f32) return ((const float*)b.data())[i];
        uint32_t u = ((uint32_t)((const uint16_t*)b.data())[i]) << 16;
        float f; memcpy(&f, &u, 4); return f;
    };
    for (int bi = 0; bi < n_blocks && bi * BS < T; bi++) {
        const size_t off = ((size_t)blocks[bi] * m->slot_stride + (size_t)layer * m->layer_stride) * m->wsize;
        NVL_HIP(hipMemcpy(kb.data(), (char*)m->kcache + off, kb.size(), hipMemcpyDeviceToHost));
        NVL_HIP(hipMemcpy(vb.data(), (char*)m->vcache + off, vb.size(), hipMemcpyDeviceToHost));
        const int t1 = std::min(T, (bi + 1) * BS);
        for (int h = 0; h < m->nKV; h++)
            for (int t = bi * BS; t < t1; t++)
                for (int d = 0; d < hd; d++) {
                    const int r = t - bi * BS;
                    const int64_t o = ((int64_t)h * T + t) * hd + d;
                    if (k_out) k_out[o] = rd(kb, ((int64_t)h * BS + r) * hd + d);
                    if (v_out) v_out[o] = rd(vb, ((int64_t)h * BS + r) * hd + d);
                }
    }
}
}  // namespace

extern "C" int nvl_get_kv(nvl_model* m, int64_t seq_id, int layer, float* k_out, float* v_out) {
    if (!m) return NVL_ERR_INVALID;
    auto it = m->seq_slot.find(seq_id);
    if (it == m->seq_slot.end()) return fail(m, NVL_ERR_UNKNOWN_SEQ, "nvl_get_kv: unknown sequence");
    if (layer < 0 || layer >= m->L) return fail(m, NVL_ERR_INVALID, "nvl_get_kv: bad layer");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    const int32_t slot = it->second;
    const int T = m->slot_len[(size_t)slot];
    read_kv(m, &slot, 1, T, layer, k_out, v_out);
    return T;
    NVL_CATCH(m)
}

extern "C" int nvl_get_mamba_state(nvl_model* m, int64_t seq_id, int layer, float* out) {
    if (!m || !out) return NVL_ERR_INVALID;
    auto it = m->seq_slot.find(seq_id);
    if (it == m->seq_slot.end()) return fail(m, NVL_ERR_UNKNOWN_SEQ, "nvl_get_mamba_state: unknown sequence");
    if (layer < 0 || layer >= m->L || !m->layers[(size_t)layer].mamba) return fail(m, NVL_ERR_INVALID, "nvl_get_mamba_state: not a Mamba2 layer");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    NVL_HIP(hipStreamSynchronize(m->stream));
    const float* src = m->ssm_state + (int64_t)it->second * m->ssm_slot_stride + (int64_t)m->layers[(size_t)layer].mamba_idx * m->ssm_layer_stride;
    NVL_HIP(hipMemcpy(out, src, (size_t)m->ssm_layer_stride * 4, hipMemcpyDeviceToHost));
    return (int)m->ssm_layer_stride;
    NVL_CATCH(m)
}

extern "C" int nvl_get_weight(nvl_model* m, int kind, int layer, float* out, int64_t in_features, int64_t out_features) {
    if (!m || !out) return NVL_ERR_INVALID;
    DevTensor* t = tensor_slot(m, kind, layer);
    if (!t || is_1d(kind)) return fail(m, NVL_ERR_INVALID, "nvl_get_weight: not a 2-D weight kind");
    if (!t->present()) return fail(m, NVL_ERR_STATE, "nvl_get_weight: the tensor is not on the device (never uploaded, or fused away by nvl_finalize)");
    if (t->rows != out_features || t->cols != in_features) return fail(m, NVL_ERR_INVALID, "nvl_get_weight: shape mismatch");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    NVL_HIP(hipStreamSynchronize(m->stream));
    const int64_t N = t->rows, K = t->cols, n_el = t->rows_pad * K;
    std::vector<char> raw((size_t)n_el * m->wsize);
    NVL_HIP(hipMemcpy(raw.data(), t->p, raw.size(), hipMemcpyDeviceToHost));
    for (int64_t n = 0; n < N; n++)
        for (int64_t k = 0; k < K; k++) {
            float v;
            if (m->f32) v = ((const float*)raw.data())[n * K + k];
            else { const uint32_t u = ((uint32_t)((const uint16_t*)raw.data())[fm_index(n, k, K)]) << 16; memcpy(&v, &u, 4); }
            out[k * N + n] = v;      // the reference's [in, out]
        }
    return NVL_OK;
    NVL_CATCH(m)
}

extern "C" int nvl_get_kv_paged(nvl_model* m, const int32_t* block_table, int n_blocks, int n_tokens, int layer,
                                float* k_out, float* v_out) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->paged) return fail(m, NVL_ERR_STATE, "nvl_get_kv_paged: the model is not in paged-KV mode");
    if (!block_table || n_blocks <= 0 || n_tokens < 0 || n_tokens > n_blocks * m->Tmax || layer < 0 || layer >= m->L)
        return fail(m, NVL_ERR_INVALID, "nvl_get_kv_paged: bad arguments");
    for (int i = 0; i < n_blocks; i++)
        if (block_table[i] < 0 || block_table[i] >= m->num_blocks) return fail(m, NVL_ERR_INVALID, "nvl_get_kv_paged: block id out of range");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    read_kv(m, block_table, n_blocks, n_tokens, layer, k_out, v_out);
    return n_tokens;
    NVL_CATCH(m)
}

extern "C" int nvl_set_profile(nvl_model* m, int on) {
    if (!m) return NVL_ERR_INVALID;
    m->profile = on != 0;
    return NVL_OK;
}
extern "C" int nvl_get_stats(nvl_model* m, nvl_stats* out) {
    if (!m || !out) return NVL_ERR_INVALID;
    *out = m->stats;
    return NVL_OK;
}
extern "C" int nvl_reset_stats(nvl_model* m) {
    if (!m) return NVL_ERR_INVALID;
    const double wb = m->stats.weight_bytes;
    m->stats = nvl_stats{};
    m->stats.weight_bytes = wb;
    for (auto& ph : m->site_stats) for (auto& st : ph) st = SiteStat{};
    return NVL_OK;
}
static const char* const k_site_names[KS_COUNT] = {
    "other", "qkv_proj", "attention", "o_proj", "ffn_up", "ffn_down", "lm_head", "norm", "moe_router", "moe_plan", "moe_up",
    "moe_down", "moe_combine", "embed", "argmax", "rope_kv", "mamba_in_proj", "mamba_conv", "mamba_scan", "mamba_gate_norm",
    "mamba_out_proj", "tp_allreduce"};
extern "C" const char* nvl_kernel_site_name(int site) { return site >= 0 && site < KS_COUNT ? k_site_names[site] : ""; }
extern "C" int nvl_get_kernel_stats(nvl_model* m, nvl_kernel_stat* out, int cap) {
    if (!m || (cap > 0 && !out)) return NVL_ERR_INVALID;
    int n = 0;
    for (int ph = 0; ph < 2; ph++)
        for (int st = 0; st < KS_COUNT; st++) {
            const SiteStat& ss = m->site_stats[ph][st];
            if (!ss.launches) continue;
            if (n < cap) out[n] = nvl_kernel_stat{st, ph, ss.launches, ss.ms, ss.flops, ss.bytes};
            n++;
        }
    return n;
}

// =================================================================================================
// runner: ModelRunner.Run semantics (tensor_model_runner.go:55-97)
// =================================================================================================
// =================================================================================================
// on-device sampling: tensor.SampleWithHistory, sampling.go:33-102
// =================================================================================================
namespace {
// rows of device logits -> sampled ids on the host.  `err` receives the reason when the arguments are refused.
int sample_rows(hipStream_t st, SampleBufs& b, const float* logits_dev, int64_t ld, int rows, int V,
                const nvl_sampling_params* sp, const int32_t* const* hist_ptrs, const int32_t* hist_lens,
                const float* uniforms, int32_t* out_host, float* probs_host, std::string& err) {
    if (rows <= 0 || V <= 0 || !sp || !uniforms || !out_host) { err = "null/empty arguments"; return NVL_ERR_INVALID; }
    if (!(sp->repetition_penalty == sp->repetition_penalty) || !(sp->temperature == sp->temperature) || !(sp->top_p == sp->top_p)) {
        err = "NaN sampling parameter"; return NVL_ERR_INVALID;
    }
    if (sp->repetition_penalty == 0.f) { err = "repetition_penalty 0 (the reference would divide by zero)"; return NVL_ERR_INVALID; }
    std::vector<int32_t> off((size_t)rows + 1, 0), hist;
    for (int i = 0; i < rows; i++) {
        const int n = (hist_ptrs && hist_lens && hist_ptrs[i]) ? hist_lens[i] : 0;
        if (n < 0) { err = "negative history length"; return NVL_ERR_INVALID; }
        for (int j = 0; j < n; j++) {
            if (hist_ptrs[i][j] < 0) { err = "negative token id in the history (the reference would panic)"; return NVL_ERR_INVALID; }
            hist.push_back(hist_ptrs[i][j]);
        }
        off[(size_t)i + 1] = (int32_t)hist.size();
        if (!(uniforms[i] >= 0.f && uniforms[i] <= 1.f)) { err = "uniform draw outside [0, 1]"; return NVL_ERR_INVALID; }
    }
    const int64_t elems = (int64_t)rows * round_up(V, 4);
    if (elems > b.elems) {
        dfree(b.work); dfree(b.cnt);
        b.work = dmalloc<float>(elems); b.cnt = dmalloc<int32_t>(elems); b.elems = elems;
        NVL_HIP(hipMemsetAsync(b.cnt, 0, (size_t)elems * 4, st));
    }
    if ((int64_t)hist.size() > b.hist_cap) {
        dfree(b.hist); b.hist_cap = round_up((int64_t)hist.size(), 4096); b.hist = dmalloc<int32_t>(b.hist_cap);
    }
    if (rows > b.rows_cap) {
        dfree(b.off); dfree(b.out); dfree(b.u);
        b.rows_cap = (int)round_up(rows, 64);
        b.off = dmalloc<int32_t>(b.rows_cap + 1); b.out = dmalloc<int32_t>(b.rows_cap); b.u = dmalloc<float>(b.rows_cap);
    }
    if (probs_host && elems > b.probs_elems) { dfree(b.probs); b.probs = dmalloc<float>(elems); b.probs_elems = elems; }
    if (!hist.empty()) NVL_HIP(hipMemcpyAsync(b.hist, hist.data(), hist.size() * 4, hipMemcpyHostToDevice, st));
    NVL_HIP(hipMemcpyAsync(b.off, off.data(), off.size() * 4, hipMemcpyHostToDevice, st));
    NVL_HIP(hipMemcpyAsync(b.u, uniforms, (size_t)rows * 4, hipMemcpyHostToDevice, st));
    SampleArgs a{};
    a.logits = logits_dev; a.ld = ld; a.work = b.work; a.cnt = b.cnt; a.hist = b.hist; a.hist_off = b.off;
    a.uniforms = b.u; a.out = b.out; a.probs_out = probs_host ? b.probs : nullptr; a.ldp = V;
    a.V = V; a.top_k = sp->top_k; a.temperature = sp->temperature; a.top_p = sp->top_p; a.rep_penalty = sp->repetition_penalty;
    hipLaunchKernelGGL(sample_row_kernel, dim3(rows), dim3(SAMPLE_THREADS), 0, st, a);
    NVL_HIP(hipGetLastError());
    NVL_HIP(hipMemcpyAsync(out_host, b.out, (size_t)rows * 4, hipMemcpyDeviceToHost, st));
    if (probs_host) NVL_HIP(hipMemcpyAsync(probs_host, b.probs, (size_t)rows * V * 4, hipMemcpyDeviceToHost, st));
    NVL_HIP(hipStreamSynchronize(st));   // (also keeps the host vectors alive until the copies are done)
    return NVL_OK;
}
}  // namespace

extern "C" int nvl_sample(nvl_model* m, int n_rows, const nvl_sampling_params* params, const int32_t* const* history_ptrs,
                          const int32_t* history_lens, const float* uniforms, int32_t* out_tokens) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_sample: model not finalized");
    if (n_rows <= 0 || n_rows > m->last_rows) return fail(m, NVL_ERR_INVALID, "nvl_sample: n_rows exceeds the logits rows of the last forward");
    NVL_TRY(m)
    NVL_HIP(hipSetDevice(m->device));
    std::string err;
    const int rc = sample_rows(m->stream, m->samp, m->logits, m->Vpad, n_rows, m->V, params, history_ptrs, history_lens,
                               uniforms, out_tokens, nullptr, err);
    return rc ? fail(m, rc, "nvl_sample: " + err) : NVL_OK;
    NVL_CATCH(m)
}

namespace {
int runner_impl(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* const* token_ptrs,
                const int32_t* token_lens, int is_prefill, int32_t* next_tokens, float* logits_out,
                const nvl_sampling_params* sp, const float* uniforms);
}
extern "C" int nvl_runner_run(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* const* token_ptrs,
                              const int32_t* token_lens, int is_prefill, int32_t* next_tokens, float* logits_out) {
    return runner_impl(m, n_seqs, seq_ids, token_ptrs, token_lens, is_prefill, next_tokens, logits_out, nullptr, nullptr);
}
extern "C" int nvl_runner_run_sampled(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* const* token_ptrs,
                                      const int32_t* token_lens, int is_prefill, const nvl_sampling_params* params,
                                      const float* uniforms, int32_t* next_tokens) {
    if (!m) return NVL_ERR_INVALID;
    if (!params || !uniforms) return fail(m, NVL_ERR_INVALID, "nvl_runner_run_sampled: null sampling arguments");
    return runner_impl(m, n_seqs, seq_ids, token_ptrs, token_lens, is_prefill, next_tokens, nullptr, params, uniforms);
}
namespace {
// after a forward over the sequences `who` (indices into the caller's arrays; logits rows in that order): replace the
// greedy ids by SampleWithHistory(logits, seq.TokenIDs, params) (tensor_model_runner.go:93)
int runner_sample(nvl_model* m, const std::vector<int>& who, const int32_t* const* token_ptrs, const int32_t* token_lens,
                  const nvl_sampling_params* sp, const float* uniforms, int32_t* next_tokens) {
    std::vector<const int32_t*> hp(who.size()); std::vector<int32_t> hl(who.size()), out(who.size()); std::vector<float> u(who.size());
    for (size_t j = 0; j < who.size(); j++) { hp[j] = token_ptrs[who[j]]; hl[j] = token_lens[who[j]]; u[j] = uniforms[who[j]]; }
    const int rc = nvl_sample(m, (int)who.size(), sp, hp.data(), hl.data(), u.data(), out.data());
    if (rc) return rc;
    for (size_t j = 0; j < who.size(); j++) next_tokens[who[j]] = out[j];
    return NVL_OK;
}
// pins the slots of the sequences of the forward call being assembled, so that opening a slot for one of them never
// evicts another (released on scope exit, also on errors)
struct PinScope {
    nvl_model* m; std::vector<int> slots;
    explicit PinScope(nvl_model* m_) : m(m_) {}
    void pin(int64_t seq_id) {
        auto it = m->seq_slot.find(seq_id);
        if (it != m->seq_slot.end() && !m->slot_pin[(size_t)it->second]) { m->slot_pin[(size_t)it->second] = 1; slots.push_back(it->second); }
    }
    ~PinScope() { for (int s : slots) m->slot_pin[(size_t)s] = 0; }
};
int runner_impl(nvl_model* m, int n_seqs, const int64_t* seq_ids, const int32_t* const* token_ptrs,
                const int32_t* token_lens, int is_prefill, int32_t* next_tokens, float* logits_out,
                const nvl_sampling_params* sp, const float* uniforms) {
    if (!m) return NVL_ERR_INVALID;
    if (!m->finalized) return fail(m, NVL_ERR_STATE, "nvl_runner_run: model not finalized");
    if (m->paged) return fail(m, NVL_ERR_STATE, "nvl_runner_run: the model is in paged-KV mode (use nvl_runner_run_paged)");
    if (n_seqs <= 0 || !seq_ids || !token_ptrs || !token_lens || !next_tokens)
        return fail(m, NVL_ERR_INVALID, "nvl_runner_run: null/empty arguments");
    NVL_TRY(m)      // (host containers below may throw bad_alloc: nothing may cross the C ABI)
    // Partition: sequences that can take the 1-token decode path vs sequences that must be (re)prefilled: an unknown id
    // (never seen, closed, or EVICTED since) or a cache whose length does not match is re-prefilled from the full history.
    std::vector<int> dec, pre;
    for (int i = 0; i < n_seqs; i++) {
        if (!token_ptrs[i] || token_lens[i] <= 0) return fail(m, NVL_ERR_INVALID, "nvl_runner_run: sequence without tokens");
        if (token_lens[i] > m->cfg.max_seq_len) return fail(m, NVL_ERR_POSITION, "nvl_runner_run: sequence longer than max_seq_len");
        for (int j = 0; j < i; j++) if (seq_ids[j] == seq_ids[i]) return fail(m, NVL_ERR_INVALID, "nvl_runner_run: duplicate sequence in batch");
        bool can_decode = false;
        if (!is_prefill) {
            auto it = m->seq_slot.find(seq_ids[i]);
            can_decode = it != m->seq_slot.end() && m->slot_len[(size_t)it->second] == token_lens[i] - 1;
        }
        (can_decode ? dec : pre).push_back(i);
    }
    const int V = m->V;
    // ---- decode group: one token per sequence at position len-1 (:78-80).  Runs before any slot is (re)opened below,
    // so no decodable sequence of this batch can be evicted before its step.
    for (size_t base = 0; base < dec.size(); base += (size_t)m->opts.max_seqs) {
        const size_t n = std::min(dec.size() - base, (size_t)m->opts.max_seqs);
        std::vector<int64_t> ids(n); std::vector<int32_t> toks(n), lens(n, 1), pos(n), out(n);
        for (size_t j = 0; j < n; j++) {
            const int i = dec[base + j];
            ids[j] = seq_ids[i]; toks[j] = token_ptrs[i][token_lens[i] - 1]; pos[j] = token_lens[i] - 1;
        }
        std::vector<float> lg(logits_out ? n * (size_t)V : 0);
        const int rc = nvl_forward(m, (int)n, ids.data(), toks.data(), lens.data(), pos.data(), 0,
                                   logits_out ? lg.data() : nullptr, out.data());
        if (rc) return rc;
        for (size_t j = 0; j < n; j++) {
            next_tokens[dec[base + j]] = out[j];
            if (logits_out) memcpy(logits_out + (size_t)dec[base + j] * V, &lg[j * (size_t)V], (size_t)V * 4);
        }
        if (sp) {
            const std::vector<int> who(dec.begin() + (long)base, dec.begin() + (long)(base + n));
            const int rs = runner_sample(m, who, token_ptrs, token_lens, sp, uniforms, next_tokens);
            if (rs) return rs;
        }
    }
    // ---- prefill group: discard the cache, run the whole history from position 0 (:63-66,75);
    // packed into forward calls of at most max_batch_tokens; a longer history is fed in chunks.  A sequence without a
    // slot takes a free one, or — the engine never calls ClearCache — the least-recently-forwarded slot outside the
    // call being assembled (seq_open_impl).
    auto reset_evict = [&](int64_t id) -> int {
        const int rc = seq_open_impl(m, id, true);
        if (rc) return rc;
        m->slot_len[(size_t)m->seq_slot[id]] = 0;
        return NVL_OK;
    };
    for (int i : pre) {      // this call's own sequences are the most recently used: victims come from outside the batch first
        auto it = m->seq_slot.find(seq_ids[i]);
        if (it != m->seq_slot.end()) m->slot_tick[(size_t)it->second] = ++m->tick;
    }
    size_t k = 0;
    while (k < pre.size()) {
        std::vector<int64_t> ids; std::vector<int32_t> toks, lens, pos; std::vector<int> who;
        int64_t budget = m->opts.max_batch_tokens;
        PinScope pins(m);
        while (k < pre.size() && (int)ids.size() < m->opts.max_seqs) {
            const int i = pre[k];
            if (token_lens[i] > budget) break;
            int rc = reset_evict(seq_ids[i]);
            if (rc) return rc;
            pins.pin(seq_ids[i]);
            ids.push_back(seq_ids[i]); lens.push_back(token_lens[i]); pos.push_back(0); who.push_back(i);
            toks.insert(toks.end(), token_ptrs[i], token_ptrs[i] + token_lens[i]);
            budget -= token_lens[i];
            k++;
        }
        if (ids.empty()) {   // one history longer than max_batch_tokens: chunked prefill of that sequence
            const int i = pre[k];
            int rc = reset_evict(seq_ids[i]);
            if (rc) return rc;
            int done = 0; int32_t out = 0;
            std::vector<float> lg(logits_out ? (size_t)V : 0);
            while (done < token_lens[i]) {
                const int32_t n = (int32_t)std::min<int64_t>(m->opts.max_batch_tokens, token_lens[i] - done);
                const int32_t p0 = done;
                // hybrid models: the reference prefills the history in ONE Forward, whose Mamba2 convolution window spans
                // the chunk seams (mamba2.go:183-254): chunks after the first take the previous chunk's tail (mamba.h)
                m->conv_chain = done == 0 ? 1 : 2;
                rc = nvl_forward(m, 1, &seq_ids[i], token_ptrs[i] + done, &n, &p0, 0, logits_out ? lg.data() : nullptr, &out);
                m->conv_chain = 0;
                if (rc) return rc;
                done += n;
            }
            next_tokens[i] = out;
            if (logits_out) memcpy(logits_out + (size_t)i * V, lg.data(), (size_t)V * 4);
            if (sp) {
                const int rs = runner_sample(m, std::vector<int>{i}, token_ptrs, token_lens, sp, uniforms, next_tokens);
                if (rs) return rs;
            }
            k++;
            continue;
        }
        std::vector<int32_t> out(ids.size());
        std::vector<float> lg(logits_out ? ids.size() * (size_t)V : 0);
        const int rc = nvl_forward(m, (int)ids.size(), ids.data(), toks.data(), lens.data(), pos.data(), 0,
                                   logits_out ? lg.data() : nullptr, out.data());
        if (rc) return rc;
        for (size_t j = 0; j < who.size(); j++) {
            next_tokens[who[j]] = out[j];
            if (logits_out) memcpy(logits_out + (size_t)who[j] * V, &lg[j * (size_t)V], (size_t)V * 4);
        }
        if (sp) {
            const int rs = runner_sample(m, who, token_ptrs, token_lens, sp, uniforms, next_tokens);
            if (rs) return rs;
        }
    }
    return NVL_OK;
    NVL_CATCH(m)
}
}  // namespace

#include "ops_impl.h"
#include "loader.h"
