// model.h — host-side state of one nvl_model handle (device weights in kernel layout, KV slabs,
// workspaces, the HIP stream) and the kernel launch helpers shared by nvllm.hip and ops.hip.
#pragma once
#include <condition_variable>
#include <array>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nvllm.h"
#include "attn.h"
#include "common.h"
#include "elem.h"
#include "sample.h"
#include "gemm.h"
#include "mamba.h"
#include "tp_p2p.h"

namespace nvl {

struct DevTensor {
    void* p = nullptr;      // canonical device copy: 2-D [rows_pad][cols] in weight dtype, 1-D fp32
    int64_t rows = 0, cols = 0, rows_pad = 0;
    bool present() const { return p != nullptr; }
};

struct LayerW {
    DevTensor t[NVL_T_COUNT];
    // frozen by finalize()
    void* w_qkv = nullptr; int n_qkv = 0; float* b_qkv = nullptr;
    void* w1 = nullptr; int n1 = 0;      // fused rows (2F for SwiGLU, F for GELU)
    void* moe_in = nullptr;              // [E][2I (interleaved in bf16 mode)][H]
    void* moe_out_cat = nullptr;         // bf16: W_down of all experts along K, [H_pad][E*I] (decode "dense-masked" form)
    bool mamba = false;                  // a Mamba2 block (attention tensors absent); index among the Mamba2 layers
    int mamba_idx = -1;
};

enum KClass { KC_GEMM = 0, KC_ATTN = 1, KC_OTHER = 2 };
// where in the layer a launch sits (per-kernel roofline entries of nvl_get_kernel_stats; names in nvllm.hip)
enum KSite { KS_OTHER = 0, KS_QKV, KS_ATTN, KS_OPROJ, KS_FFN_UP, KS_FFN_DOWN, KS_LM_HEAD, KS_NORM, KS_MOE_ROUTER, KS_MOE_PLAN,
             KS_MOE_UP, KS_MOE_DOWN, KS_MOE_COMBINE, KS_EMBED, KS_ARGMAX, KS_ROPE, KS_MAMBA_IN, KS_MAMBA_CONV, KS_MAMBA_SCAN,
             KS_MAMBA_GATE, KS_MAMBA_OUT, KS_ALLREDUCE, KS_COUNT };

struct ProfRec { hipEvent_t a, b; int cls; double flops; int site, phase; double bytes; };
struct SiteStat { double ms = 0, flops = 0, bytes = 0; uint64_t launches = 0; };

}  // namespace nvl

// in-process tensor-parallel group (tests on one GPU): see tp_allreduce in nvllm.hip
// device scratch of the sampler (sample.h); grows on demand
struct SampleBufs {
    float* work = nullptr; int32_t* cnt = nullptr; int64_t elems = 0;     // [rows][V] each; cnt is kept zero-filled
    int32_t* hist = nullptr; int64_t hist_cap = 0;
    int32_t* off = nullptr; int32_t* out = nullptr; float* u = nullptr; int rows_cap = 0;
    float* probs = nullptr; int64_t probs_elems = 0;
};

struct nvl_local_group {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0, arrived = 0, refs = 0;
    bool aborted = false;        // a member failed: every waiter and every later all-reduce of the group errors out
    uint64_t gen = 0;
    float* bufs[8] = {nullptr};
    float* scratch = nullptr;
    int64_t scratch_floats = 0;
};

struct nvl_model {
    nvl_model_config cfg{};
    nvl_runtime_opts opts{};
    bool f32 = false;            // NVL_PRECISION_F32
    bool finalized = false;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // derived (nH, nKV, F are the LOCAL sizes of this tensor-parallel rank; *_full the model's)
    int H = 0, nH = 0, nKV = 0, hd = 0, F = 0, V = 0, Vpad = 0, L = 0, group = 1;
    int tp = 1, tp_rank = 0, nH_full = 0, nKV_full = 0, F_full = 0;
    bool tp_force = false;
    void* tp_comm = nullptr;          // ncclComm_t
    nvl_local_group* tp_local = nullptr;   // in-process emulation (tests on one GPU)
    float* tp_part = nullptr;         // [Mmax][H] fp32 partial of a row-parallel projection
    // hand-written all-reduce over peer-mapped buffers (tp_p2p.h): own comm buffer + the peers' (IPC-opened) ones
    char* p2p_buf = nullptr; size_t p2p_bytes = 0; char* p2p_peer[8] = {nullptr}; bool p2p_ready = false;
    int64_t p2p_off_in1 = 0, p2p_off_in2 = 0, p2p_off_res = 0, p2p_in1_stride = 0, p2p_in2_stride = 0, p2p_res_stride = 0;
    int n_qkv = 0, Tmax = 0;     // Tmax: tokens per KV block (slab mode: the whole slot; paged: kv_block_size)
    int paged = 0, num_blocks = 0, blocks_per_seq = 1, table_cap = 0;   // KV addressing (see kv_locate)
    float attn_scale = 0.f, resid_alpha = 1.f;
    size_t wsize = 2;            // bytes per weight/activation element

    nvl::DevTensor g[NVL_T_COUNT];       // model-level tensors
    std::vector<nvl::LayerW> layers;
    void* lm_head = nullptr;             // [Vpad][H] (aliases tok_emb when tied)
    float *rope_cos = nullptr, *rope_sin = nullptr;

    // KV slabs
    void *kcache = nullptr, *vcache = nullptr;
    int64_t slot_stride = 0, layer_stride = 0;    // elements
    std::map<int64_t, int> seq_slot;
    std::vector<int> free_slots;
    std::vector<int> slot_len;
    std::vector<uint64_t> slot_tick;   // last forward that touched the slot (LRU eviction in the runner entry points)
    std::vector<char> slot_pin;        // 1 while the slot's sequence is part of the forward call being assembled
    uint64_t tick = 0;

    // workspaces
    float* x = nullptr;          // residual stream fp32 [Mmax][H]
    void* xn = nullptr;          // normed activations (ActT)
    float* qkv = nullptr;        // fp32 [Mmax][n_qkv]
    void* q = nullptr;           // ActT [Mmax][nH*hd]
    void* attn_out = nullptr;    // ActT [Mmax][nH*hd]
    float* attn_part = nullptr;  // bf16: partial results of a decode attention whose keys are split over workgroups (attn.h AttnArgs::part)
    int32_t* attn_part_cnt = nullptr;   // ... and their arrival tickets per (sequence, kv head)
    void* hbuf = nullptr;        // ActT [Mmax][F]   (MoE: [Mmax*k][I])
    float* h2 = nullptr;         // fp32 [Mmax][2F]  (f32 mode un-fused gate|up; MoE f32 too)
    void* xn_last = nullptr;     // ActT [rows][H]
    float* logits = nullptr;     // fp32 [logit_rows][Vpad]
    int64_t logit_rows = 0;
    int32_t* argmax_dev = nullptr; float* argmax_pval = nullptr; int32_t* argmax_pidx = nullptr;
    // MoE
    float* router_logits = nullptr;   // [Mmax][128]
    int32_t* expert_ids = nullptr; float* expert_w = nullptr;
    int32_t *seg_start = nullptr, *perm_token = nullptr, *slot_of = nullptr;
    float* moe_eo = nullptr;          // [Mmax*k][H]
    float* moe_gate = nullptr;        // decode: [64][E] routing weights (0 = not routed)
    void* moe_hall = nullptr;         // decode: bf16 [64][E*I] gate-weighted SwiGLU outputs of every touched expert
    float* moe_part = nullptr;        // decode: [MOE_DOWN_SLICES][64][H] K-slice partials of the down projection
    void* moe_xg = nullptr;           // bf16: the normed rows in expert order, fragment-major [round_up(Mmax*k, 64)][H] (prefill)
    int32_t *moe_counts = nullptr, *moe_cursor = nullptr, *moe_tile_map = nullptr, *moe_n_mtiles = nullptr;
    // decode split-K: partial slices of the last residual GEMM, consumed by the next norm launch
    float* rs_part = nullptr;    // deferred RMSNorm: [H/16][64] partial sums of x^2 (gemm.h)
    int ctx_hint = 0;    // longest context (keys) of the batch being enqueued; 0 = unknown (decode attention sizing)
    float* sk_part = nullptr; int sk_max_slices = 4; int pending_slices = 0, pending_rows = 0; float pending_alpha = 1.f; const float* pending_part = nullptr;
    int sk_rows = 64;             // rows the sk_part slices are allocated for
    const int32_t* pending_slot_of = nullptr; const float* pending_gate_w = nullptr;   // MoE combine folded into the next norm
    // per-call metadata (one pinned host block mirrored on the device)
    int32_t* meta_host = nullptr; int32_t* meta_dev = nullptr; int64_t meta_ints = 0;
    // Mamba2 (hybrid models): sizes, per-slot SSM state [slot][mamba layer][heads][hd][ss] fp32, workspaces
    int n_mamba = 0, mEH = 0, mConv = 0, mP = 0, m_nh = 0, m_hd = 0, m_ss = 0, m_ng = 0, m_K = 0;
    float* ssm_state = nullptr; int64_t ssm_layer_stride = 0, ssm_slot_stride = 0;
    float *mproj = nullptr, *mxbc = nullptr, *mdelta = nullptr, *my = nullptr; void* myn = nullptr;
    float* conv_tail = nullptr; int64_t conv_tail_layer_stride = 0, conv_tail_slot_stride = 0;   // chunked prefill only (mamba.h MambaArgs::chain)
    int conv_chain = 0;          // set by runner_impl around the chunks of one long history
    float *ssd_z = nullptr, *ssd_decay = nullptr; int64_t ssd_z_floats = 0, ssd_decay_floats = 0;   // chunk-parallel SSD scan scratch (mamba.h)
    SampleBufs samp;             // nvl_sample scratch
    int32_t* samp_hist = nullptr; int32_t* samp_hist_len = nullptr; int64_t samp_hist_cap = 0;   // nvl_decode_sampled: device-kept histories
    float* samp_u_steps = nullptr; int64_t samp_u_cap = 0;
    int last_rows = 0;           // logits rows the last forward left in `logits`
    int32_t* ring = nullptr; int64_t ring_ints = 0;   // nvl_decode_greedy: [steps][seqs] tokens on the device
    int32_t* ring_pos0 = nullptr;                     // position of every sequence when the fused loop started (ring row index)
    // hipGraph replay of decode passes (one graph per launch configuration; first sight eager, second captured, then replayed)
    std::map<std::array<int, 5>, hipGraphExec_t> graphs;
    std::set<std::array<int, 5>> graph_seen;
    std::vector<hipGraphExec_t> graphs_retired;       // evicted execs, destroyed after the next stream sync (reap_graphs)
    bool graphs_ok = true;
    int32_t* am_host = nullptr;                       // pinned: argmax ids of a replayed decode pass
    // debug
    // nvl_set_debug mode 4: in-kernel time stamps of the decode kernels (common.h nvl_stamp), one record per launch
    bool stamping = false; unsigned long long* stamp_buf = nullptr; int stamp_launches = 0;
    struct StampRec { int site, phase, nwg; };
    std::vector<StampRec> stamp_recs;
    bool tap = false;            // nvl_set_debug mode 2: record every layer's residual stream without leaving the product path
    bool keep_hidden = false; float* hidden = nullptr; int64_t hidden_tokens = 0; int hidden_last_M = 0;
    // stats
    nvl_stats stats{};
    nvl::SiteStat site_stats[2][nvl::KS_COUNT];     // [phase: 0 prefill, 1 decode][site], filled when profiling
    int site = 0, phase = 0;     // set by the forward pass just before a launch (consumed by KScope)
    double kv_tok = 0;           // cached + new keys of the batch being enqueued (attention's algorithmic KV bytes)
    double site_bytes = 0;       // algorithmic bytes of the next launch (0 = let the helper compute them)
    bool profile = false;
    std::vector<nvl::ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};
