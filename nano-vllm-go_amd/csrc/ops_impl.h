// ops_impl.h — op-level C-ABI entry points (nvl_op_*), included at the end of nvllm.hip.
// Host fp32 in / host fp32 out; every one runs the SAME device kernels as the model path, on
// one op, so tests/ can compare each reference function with the oracle in isolation.
// (Test/diagnostic surface: allocates and frees device memory on every call.)
#pragma once

namespace {

struct OpCtx {   // a model-less nvl_model carrying just a stream and the fields the helpers read
    nvl_model m;
    std::vector<void*> bufs;
    OpCtx(int device, int precision) {
        if (nvl_device_count() <= device) throw std::runtime_error("no HIP device (this library has no CPU fallback)");
        NVL_HIP(hipSetDevice(device));
        m.device = device;
        m.f32 = precision == NVL_PRECISION_F32;
        m.wsize = m.f32 ? 4 : 2;
        NVL_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    }
    ~OpCtx() {
        if (m.stream) { (void)hipStreamSynchronize(m.stream); (void)hipStreamDestroy(m.stream); }
        for (void* p : bufs) dfree(p);
    }
    void* alloc(int64_t bytes) { void* p = dmalloc_bytes(bytes); bufs.push_back(p); return p; }
    float* up_f32(const float* h, int64_t n) {
        float* d = (float*)alloc(n * 4);
        NVL_HIP(hipMemcpyAsync(d, h, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
        return d;
    }
    // host fp32 [N][K] (or [K][N] when in_out) -> device [Npad][K] in the kernels' operand layout
    // (bf16: fragment-major, rows padded to whole tiles; f32: row-major).  row_major=true keeps a bf16
    // matrix row-major (attention's q and KV slabs are not GEMM operands).
    void* up_mat(const float* h, int64_t N, int64_t K, bool in_out, int64_t pad_rows_to = 64, bool row_major = false) {
        const bool weight = !row_major;
        float* raw = up_f32(h, N * K);
        const int64_t Np = round_up(N, pad_rows_to);
        void* d = alloc(Np * K * (int64_t)m.wsize);
        NVL_HIP(hipMemsetAsync(d, 0, (size_t)(Np * K) * m.wsize, m.stream));
        dim3 grid((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32));
        if (m.f32) hipLaunchKernelGGL((convert_2d_kernel<float, false>), grid, dim3(256), 0, m.stream, raw, 0, in_out ? 1 : 0, (float*)d, N, K, Slice2D{N, K, 0, 0, 0});
        else if (weight) hipLaunchKernelGGL((convert_2d_kernel<bf16_t, true>), grid, dim3(256), 0, m.stream, raw, 0, in_out ? 1 : 0, (bf16_t*)d, N, K, Slice2D{N, K, 0, 0, 0});
        else hipLaunchKernelGGL((convert_2d_kernel<bf16_t, false>), grid, dim3(256), 0, m.stream, raw, 0, in_out ? 1 : 0, (bf16_t*)d, N, K, Slice2D{N, K, 0, 0, 0});
        NVL_HIP(hipGetLastError());
        return d;
    }
    void down(float* h, const void* d, int64_t n) {
        NVL_HIP(hipMemcpyAsync(h, d, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
        NVL_HIP(hipStreamSynchronize(m.stream));
    }
};

int op_fail(const std::string& s, int code = NVL_ERR_INVALID) { g_create_err = s; return code; }

#define OP_TRY try {
#define OP_CATCH                                                                              \
    } catch (const HipError& e) {                                                             \
        return op_fail(std::string("HIP error: ") + hipGetErrorString(e.code) + " (" + e.what + ")", NVL_ERR_HIP); \
    } catch (const std::exception& e) {                                                       \
        const std::string w = e.what();                                                       \
        return op_fail(w, w.find("no HIP device") != std::string::npos ? NVL_ERR_NO_DEVICE : NVL_ERR_INVALID); \
    }

}  // namespace

extern "C" int nvl_op_matmul(int device, int precision, const float* a, const float* b, float* c, int M, int K, int N) {
    if (!a || !b || !c || M <= 0 || K <= 0 || N <= 0) return op_fail("nvl_op_matmul: bad arguments");
    if (K % 64 != 0) return op_fail("nvl_op_matmul: K must be a multiple of 64");
    OP_TRY
    OpCtx cx(device, precision);
    void* A = cx.up_mat(a, M, K, false);
    void* W = cx.up_mat(b, N, K, true, W_ROW_PAD);
    const int ldc = (int)round_up(N, 4);
    float* C = (float*)cx.alloc((int64_t)M * ldc * 4);
    gemm(&cx.m, EPI_STORE, true, mk(A, K, W, C, ldc, nullptr, 1.f, M, N, K));
    std::vector<float> tmp((size_t)M * ldc);
    cx.down(tmp.data(), C, (int64_t)M * ldc);
    for (int i = 0; i < M; i++) memcpy(c + (size_t)i * N, &tmp[(size_t)i * ldc], (size_t)N * 4);
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_layernorm(int device, const float* x, const float* w, const float* bias, float eps, float* y,
                                int rows, int hidden) {
    if (!x || !w || !y || rows <= 0 || hidden <= 0 || hidden % 4) return op_fail("nvl_op_layernorm: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    float* dx = cx.up_f32(x, (int64_t)rows * hidden);
    float* dw = cx.up_f32(w, hidden);
    float* db = bias ? cx.up_f32(bias, hidden) : nullptr;
    float* dy = (float*)cx.alloc((int64_t)rows * hidden * 4);
    hipLaunchKernelGGL((norm_kernel<float>), dim3(cdiv(rows, 4)), dim3(256), 0, cx.m.stream, dx, (const int32_t*)nullptr, dw,
                       db, eps, dy, rows, hidden);
    NVL_HIP(hipGetLastError());
    cx.down(y, dy, (int64_t)rows * hidden);
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_softmax(int device, const float* x, float* y, int rows, int cols) {
    if (!x || !y || rows <= 0 || cols <= 0) return op_fail("nvl_op_softmax: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    float* dx = cx.up_f32(x, (int64_t)rows * cols);
    float* dy = (float*)cx.alloc((int64_t)rows * cols * 4);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, cx.m.stream, dx, dy, rows, cols);
    NVL_HIP(hipGetLastError());
    cx.down(y, dy, (int64_t)rows * cols);
    return NVL_OK;
    OP_CATCH
}

static int op_unary(int device, const float* x, float* y, int64_t n, int which) {
    if (!x || !y || n <= 0) return op_fail("nvl_op_gelu/silu: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    float* dx = cx.up_f32(x, n);
    float* dy = (float*)cx.alloc(n * 4);
    if (which == 0) hipLaunchKernelGGL(gelu_kernel, dim3(cdiv(n, 256)), dim3(256), 0, cx.m.stream, dx, dy, n);
    else hipLaunchKernelGGL(silu_kernel, dim3(cdiv(n, 256)), dim3(256), 0, cx.m.stream, dx, dy, n);
    NVL_HIP(hipGetLastError());
    cx.down(y, dy, n);
    return NVL_OK;
    OP_CATCH
}
extern "C" int nvl_op_gelu(int device, const float* x, float* y, int64_t n) { return op_unary(device, x, y, n, 0); }
extern "C" int nvl_op_silu(int device, const float* x, float* y, int64_t n) { return op_unary(device, x, y, n, 1); }

// ApplyRoPESingleTensor on t [heads, seq, hd]: runs rope_kv_kernel with every head treated as a
// query head of a one-sequence batch (the K/V branches are exercised by the model-level tests).
extern "C" int nvl_op_rope(int device, float* t, int heads, int seq, int hd, int start_pos, double base, int max_seq) {
    if (!t || heads <= 0 || seq <= 0 || hd <= 0 || hd % 2) return op_fail("nvl_op_rope: bad arguments");
    if (start_pos < 0 || start_pos + seq > max_seq) return op_fail("nvl_op_rope: position exceeds max_seq", NVL_ERR_POSITION);
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    const int half = hd / 2;
    std::vector<float> ct((size_t)max_seq * hd), st((size_t)max_seq * hd);
    for (int pos = 0; pos < max_seq; pos++)
        for (int i = 0; i < half; i++) {
            const double freq = 1.0 / std::pow(base, (double)(2 * i) / (double)hd);
            const double ang = (double)pos * freq;
            ct[(size_t)pos * hd + i] = ct[(size_t)pos * hd + half + i] = (float)std::cos(ang);
            st[(size_t)pos * hd + i] = st[(size_t)pos * hd + half + i] = (float)std::sin(ang);
        }
    // [heads, seq, hd] -> token-major [seq][heads*hd]
    std::vector<float> tm((size_t)heads * seq * hd);
    for (int h = 0; h < heads; h++)
        for (int s = 0; s < seq; s++)
            memcpy(&tm[((size_t)s * heads + h) * hd], t + ((size_t)h * seq + s) * hd, (size_t)hd * 4);
    std::vector<int32_t> pos((size_t)seq), slot((size_t)seq, 0);
    for (int s = 0; s < seq; s++) pos[(size_t)s] = start_pos + s;
    float* dq = cx.up_f32(tm.data(), (int64_t)tm.size());
    float* dc = cx.up_f32(ct.data(), (int64_t)ct.size());
    float* ds = cx.up_f32(st.data(), (int64_t)st.size());
    int32_t* dpos = (int32_t*)cx.alloc((int64_t)seq * 4);
    int32_t* dslot = (int32_t*)cx.alloc((int64_t)seq * 4);
    NVL_HIP(hipMemcpyAsync(dpos, pos.data(), (size_t)seq * 4, hipMemcpyHostToDevice, cx.m.stream));
    NVL_HIP(hipMemcpyAsync(dslot, slot.data(), (size_t)seq * 4, hipMemcpyHostToDevice, cx.m.stream));
    float* dout = (float*)cx.alloc((int64_t)tm.size() * 4);
    hipLaunchKernelGGL((rope_kv_kernel<float, false>), dim3(seq, heads), dim3(half), 0, cx.m.stream, dq, heads * hd, dpos, dslot, dslot,
                       dc, ds, dout, heads * hd, (float*)nullptr, (float*)nullptr, (int64_t)0, max_seq, heads, 0, hd);
    NVL_HIP(hipGetLastError());
    cx.down(tm.data(), dout, (int64_t)tm.size());
    for (int h = 0; h < heads; h++)
        for (int s = 0; s < seq; s++)
            memcpy(t + ((size_t)h * seq + s) * hd, &tm[((size_t)s * heads + h) * hd], (size_t)hd * 4);
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_attention(int device, int precision, const float* q, const float* k, const float* v, int nH, int nKV,
                                int S, int T, int hd, float scale, float* out) {
    if (!q || !k || !v || !out || nH <= 0 || nKV <= 0 || nH % nKV || S <= 0 || T < S || (hd != 64 && hd != 128))
        return op_fail("nvl_op_attention: bad arguments");
    OP_TRY
    OpCtx cx(device, precision);
    nvl_model& m = cx.m;
    const int Tmax = (int)round_up(T, 64);
    if (m.f32 && Tmax > 12000) return op_fail("nvl_op_attention: T too long for the f32 kernel");
    // token-major Q, slab-layout K and V
    std::vector<float> qt((size_t)S * nH * hd), kc((size_t)nKV * Tmax * hd, 0.f), vc((size_t)nKV * Tmax * hd, 0.f);
    for (int h = 0; h < nH; h++)
        for (int s = 0; s < S; s++)
            memcpy(&qt[((size_t)s * nH + h) * hd], q + ((size_t)h * S + s) * hd, (size_t)hd * 4);
    for (int h = 0; h < nKV; h++)
        for (int t = 0; t < T; t++)
            for (int d = 0; d < hd; d++) {
                kc[((size_t)h * Tmax + t) * hd + d] = k[((size_t)h * T + t) * hd + d];
                vc[((size_t)h * Tmax + t) * hd + d] = v[((size_t)h * T + t) * hd + d];
            }
    m.q = cx.up_mat(qt.data(), S, (int64_t)nH * hd, false, 1, true);
    m.kcache = cx.up_mat(kc.data(), 1, (int64_t)kc.size(), false, 1, true);
    m.vcache = cx.up_mat(vc.data(), 1, (int64_t)vc.size(), false, 1, true);
    m.attn_out = cx.alloc(round_up(S, 64) * nH * hd * (int64_t)m.wsize);
    m.nH = nH; m.nKV = nKV; m.group = nH / nKV; m.hd = hd; m.Tmax = Tmax; m.L = 1;
    m.layer_stride = (int64_t)nKV * Tmax * hd; m.slot_stride = m.layer_stride;
    m.attn_scale = scale != 0.f ? scale : 1.0f / std::sqrt((float)hd);
    const int32_t meta[5] = {0, S, T - S, 0, 0};      // one sequence whose block table is the single block 0
    int32_t* dmeta = (int32_t*)cx.alloc(20);
    NVL_HIP(hipMemcpyAsync(dmeta, meta, 20, hipMemcpyHostToDevice, m.stream));
    Meta md{};
    md.seq_tok_start = dmeta; md.seq_len = dmeta + 1; md.seq_pos = dmeta + 2; md.seq_tbl = dmeta + 3; md.blk_table = dmeta + 4;
    m.blocks_per_seq = 1;
    attention(&m, 0, md, 1, S, 0.0);
    // back to fp32 [nH, S, hd]
    std::vector<float> ot((size_t)S * nH * hd);
    if (m.f32) {
        cx.down(ot.data(), m.attn_out, (int64_t)ot.size());
    } else {
        std::vector<uint16_t> ob((size_t)(round_up(S, 64) * nH * hd));
        NVL_HIP(hipMemcpyAsync(ob.data(), m.attn_out, ob.size() * 2, hipMemcpyDeviceToHost, m.stream));
        NVL_HIP(hipStreamSynchronize(m.stream));
        const int64_t W = (int64_t)nH * hd;     // the kernel wrote the next GEMM's fragment-major operand
        for (int64_t r = 0; r < S; r++)
            for (int64_t c = 0; c < W; c++) {
                uint32_t u = ((uint32_t)ob[(size_t)fm_index(r, c, W)]) << 16;
                memcpy(&ot[(size_t)(r * W + c)], &u, 4);
            }
    }
    for (int h = 0; h < nH; h++)
        for (int s = 0; s < S; s++)
            memcpy(out + ((size_t)h * S + s) * hd, &ot[((size_t)s * nH + h) * hd], (size_t)hd * 4);
    m.q = m.kcache = m.vcache = m.attn_out = nullptr;   // owned by cx.bufs
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_ffn(int device, int precision, const float* x, const float* w1, const float* b1, const float* w2,
                          const float* b2, int rows, int hidden, int ffn, int swiglu, float* y) {
    if (!x || !w1 || !w2 || !y || rows <= 0 || hidden % 64 || ffn % 64) return op_fail("nvl_op_ffn: bad arguments");
    OP_TRY
    OpCtx cx(device, precision);
    nvl_model& m = cx.m;
    m.H = hidden; m.F = ffn;
    m.cfg.activation_type = swiglu ? NVL_ACT_SWIGLU : NVL_ACT_GELU;
    m.xn = cx.up_mat(x, rows, hidden, false);
    LayerW l;
    const int n1 = swiglu ? 2 * ffn : ffn;
    void* w1c = cx.up_mat(w1, n1, hidden, true, W_ROW_PAD);   // canonical [n1][H] (gate rows | up rows)
    if (swiglu && !m.f32) {
        auto idx = swiglu_interleave(ffn, 0);
        idx.resize((size_t)round_up(n1, W_ROW_PAD), -1);
        l.w1 = cx.alloc((int64_t)idx.size() * hidden * 2);
        gather_rows(&m, w1c, idx, l.w1, hidden);
    } else {
        l.w1 = w1c;
    }
    if (b1) l.t[NVL_T_B1].p = cx.up_f32(b1, ffn);
    void* w2d = cx.up_mat(w2, hidden, ffn, true, W_ROW_PAD);
    float* b2d = b2 ? cx.up_f32(b2, hidden) : nullptr;
    m.hbuf = cx.alloc(round_up(rows, 64) * ffn * (int64_t)m.wsize);
    if (m.f32 && swiglu) m.h2 = (float*)cx.alloc((int64_t)rows * 2 * ffn * 4);
    ffn_up(&m, l, rows);
    float* yd = (float*)cx.alloc((int64_t)rows * hidden * 4);
    gemm(&m, EPI_STORE, true, mk(m.hbuf, ffn, w2d, yd, hidden, b2d, 1.f, rows, hidden, ffn));
    cx.down(y, yd, (int64_t)rows * hidden);
    l.w1 = nullptr; l.t[NVL_T_B1].p = nullptr;
    m.xn = m.hbuf = nullptr; m.h2 = nullptr;
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_moe(int device, int precision, const float* x, const float* router, const float* w_in,
                          const float* w_out, int rows, int hidden, int n_experts, int top_k, int inter, float* y) {
    if (!x || !router || !w_in || !w_out || !y || rows <= 0 || hidden % 64 || inter % 64 || n_experts > 64 || top_k > n_experts)
        return op_fail("nvl_op_moe: bad arguments");
    OP_TRY
    OpCtx cx(device, precision);
    nvl_model& m = cx.m;
    m.H = hidden; m.F = inter;
    m.cfg.num_experts = n_experts; m.cfg.num_experts_per_tok = top_k; m.cfg.use_moe = 1; m.cfg.residual_multiplier = 0.f;
    m.xn = cx.up_mat(x, rows, hidden, false);
    LayerW l;
    l.t[NVL_T_ROUTER].p = cx.up_mat(router, n_experts, hidden, true, W_ROW_PAD);
    void* inc = cx.up_mat(w_in, (int64_t)n_experts * 2 * inter, hidden, false, W_ROW_PAD);
    if (!m.f32) {
        std::vector<int32_t> idx;
        for (int e = 0; e < n_experts; e++) {
            auto one = swiglu_interleave(inter, (int64_t)e * 2 * inter);
            idx.insert(idx.end(), one.begin(), one.end());
        }
        l.moe_in = cx.alloc((int64_t)idx.size() * hidden * 2);
        gather_rows(&m, inc, idx, l.moe_in, hidden);
    } else {
        l.moe_in = inc;
    }
    l.t[NVL_T_MOE_OUT].p = cx.up_mat(w_out, (int64_t)n_experts * hidden, inter, false, W_ROW_PAD);
    const int64_t pairs = (int64_t)rows * top_k;
    m.router_logits = (float*)cx.alloc((int64_t)rows * 128 * 4);
    m.expert_ids = (int32_t*)cx.alloc(pairs * 4); m.expert_w = (float*)cx.alloc(pairs * 4);
    m.seg_start = (int32_t*)cx.alloc((n_experts + 1) * 4);
    m.moe_counts = (int32_t*)cx.alloc(n_experts * 4); m.moe_cursor = (int32_t*)cx.alloc(n_experts * 4);
    m.moe_tile_map = (int32_t*)cx.alloc(16 * (cdiv(pairs, 128) + n_experts)); m.moe_n_mtiles = (int32_t*)cx.alloc(16);
    m.perm_token = (int32_t*)cx.alloc(pairs * 4); m.slot_of = (int32_t*)cx.alloc(pairs * 4);
    m.moe_eo = (float*)cx.alloc(pairs * hidden * 4);
    if (!m.f32) m.moe_xg = cx.alloc(round_up(pairs, 64) * hidden * 2);
    m.hbuf = cx.alloc(round_up(pairs, 64) * inter * (int64_t)m.wsize);
    if (m.f32) m.h2 = (float*)cx.alloc(pairs * 2 * inter * 4);
    m.x = (float*)cx.alloc((int64_t)rows * hidden * 4);
    NVL_HIP(hipMemsetAsync(m.x, 0, (size_t)rows * hidden * 4, m.stream));
    m.keep_hidden = true;   // (no norm follows here: keep the stand-alone combine instead of leaving it pending)
    moe(&m, l, rows);    // accumulates into x (zeroed) with multiplier 1
    cx.down(y, m.x, (int64_t)rows * hidden);
    l.moe_in = nullptr; l.t[NVL_T_ROUTER].p = nullptr; l.t[NVL_T_MOE_OUT].p = nullptr;
    m.xn = m.hbuf = nullptr; m.h2 = nullptr; m.x = nullptr; m.router_logits = nullptr; m.expert_ids = nullptr;
    m.expert_w = nullptr; m.seg_start = m.perm_token = m.slot_of = nullptr; m.moe_eo = nullptr; m.moe_xg = nullptr;
    m.moe_counts = m.moe_cursor = m.moe_tile_map = m.moe_n_mtiles = nullptr;
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_op_argmax(int device, const float* x, int rows, int cols, int32_t* out) {
    if (!x || !out || rows <= 0 || cols <= 0) return op_fail("nvl_op_argmax: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    float* dx = cx.up_f32(x, (int64_t)rows * cols);
    int32_t* d = (int32_t*)cx.alloc((int64_t)rows * 4);
    const int chunks = cdiv(cols, ARGMAX_CHUNK);
    float* pv = (float*)cx.alloc((int64_t)rows * chunks * 4);
    int32_t* pi = (int32_t*)cx.alloc((int64_t)rows * chunks * 4);
    launch_argmax(cx.m.stream, dx, cols, cols, rows, 0.f, pv, pi, d);
    NVL_HIP(hipGetLastError());
    NVL_HIP(hipMemcpyAsync(out, d, (size_t)rows * 4, hipMemcpyDeviceToHost, cx.m.stream));
    NVL_HIP(hipStreamSynchronize(cx.m.stream));
    return NVL_OK;
    OP_CATCH
}


// Tuning/measurement entry: time one projection GEMM shape on the device with HIP events.
// epi: 0 STORE(fp32 out) 1 RESID 2 SWIGLU 3 GELU; force_bnt/force_ksplit = 0 -> automatic.
extern "C" int nvl_op_sample(int device, const float* logits, int rows, int V, const nvl_sampling_params* params,
                             const int32_t* const* history_ptrs, const int32_t* history_lens, const float* uniforms,
                             int32_t* out_tokens, float* probs_out) {
    if (!logits || rows <= 0 || V <= 0) return op_fail("nvl_op_sample: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_F32);
    float* L = cx.up_f32(logits, (int64_t)rows * V);
    SampleBufs b;
    std::string err;
    int rc;
    try { rc = sample_rows(cx.m.stream, b, L, V, rows, V, params, history_ptrs, history_lens, uniforms, out_tokens, probs_out, err); }
    catch (...) { free_sample_bufs(b); throw; }
    free_sample_bufs(b);
    return rc ? op_fail("nvl_op_sample: " + err, rc) : NVL_OK;
    OP_CATCH
}

extern "C" int nvl_bench_gemm(int device, int M, int N, int K, int epi, int force_bnt, int force_ksplit, int iters,
                              float* avg_us) {
    if (!avg_us || M <= 0 || N <= 0 || K % 64 || iters <= 0) return op_fail("nvl_bench_gemm: bad arguments");
    OP_TRY
    OpCtx cx(device, NVL_PRECISION_BF16);
    const int64_t Np = round_up(N, W_ROW_PAD);
    std::vector<uint16_t> ha((size_t)round_up(M, 64) * K), hw((size_t)Np * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (uint16_t)(0x3c00u + ((s >> 9) & 0x3ffu) - ((s >> 20) & 1u) * 0x8000u * 0); };
    for (auto& v : ha) { v = rnd(); if (s & 0x80000000u) v |= 0x8000u; }
    for (auto& v : hw) { v = rnd(); if (s & 0x80000000u) v |= 0x8000u; }
    void* A = cx.alloc((int64_t)ha.size() * 2);
    NVL_HIP(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    // decode shapes (M <= 64) are weight-streaming: rotate through enough copies of W (> 512 MiB in all) that every
    // launch reads its weights cold from HBM, as in the model, not from the 256 MiB Infinity Cache
    const int64_t wbytes = (int64_t)hw.size() * 2;
    const int copies = M <= 64 ? (int)std::min<int64_t>(64, std::max<int64_t>(1, (640ll << 20) / wbytes + 1)) : 1;
    std::vector<void*> Ws((size_t)copies);
    for (int cidx = 0; cidx < copies; cidx++) {
        Ws[(size_t)cidx] = cx.alloc(wbytes);
        if (cidx == 0) NVL_HIP(hipMemcpy(Ws[0], hw.data(), (size_t)wbytes, hipMemcpyHostToDevice));
        else NVL_HIP(hipMemcpy(Ws[(size_t)cidx], Ws[0], (size_t)wbytes, hipMemcpyDeviceToDevice));
    }
    const int ldc = epi == EPI_SWIGLU ? N / 2 : (int)round_up(N, 4);
    void* C = cx.alloc(round_up(M, 64) * ldc * 4);
    NVL_HIP(hipMemsetAsync(C, 0, (size_t)round_up(M, 64) * ldc * 4, cx.m.stream));
    float* bias = nullptr;
    if (epi == EPI_GELU) { bias = (float*)cx.alloc((int64_t)N * 4); NVL_HIP(hipMemsetAsync(bias, 0, (size_t)N * 4, cx.m.stream)); }
    GemmArgs a = mk(A, K, Ws[0], C, ldc, bias, 1e-3f, M, N, K);
    g_force_ntw = force_bnt; g_force_ksplit = force_ksplit;
    if (M > 64) { g_force_tile = force_bnt; g_force_ntw = 0; }   // prefill shapes: force_bnt selects the tile kernel (1/2)
    for (int i = 0; i < 3; i++) gemm(&cx.m, epi, epi == EPI_STORE, a);
    hipEvent_t e0, e1;
    NVL_HIP(hipEventCreate(&e0)); NVL_HIP(hipEventCreate(&e1));
    NVL_HIP(hipEventRecord(e0, cx.m.stream));
    for (int i = 0; i < iters; i++) { a.W = Ws[(size_t)(i % copies)]; gemm(&cx.m, epi, epi == EPI_STORE, a); }
    NVL_HIP(hipEventRecord(e1, cx.m.stream));
    NVL_HIP(hipEventSynchronize(e1));
    g_force_ntw = 0; g_force_ksplit = 0; g_force_tile = 0;
    float ms = 0.f;
    NVL_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms * 1e3f / (float)iters;
    return NVL_OK;
    OP_CATCH
}

extern "C" int nvl_set_tuning(int key, int value) {
    g_tune_epoch++;        // captured decode graphs bake the tuning in: a new epoch makes them miss
    if (key == 21) { const int old = g_use_graphs; g_use_graphs = value; return old; }
    if (key == 22) { const int old = g_moe_dense; g_moe_dense = value; return old; }
    if (key == 26) { const int old = g_sk_tile; g_sk_tile = value; return old; }
    if (key == 25) { const int old = g_attn_split; g_attn_split = value; return old; }
    if (key == 24) { const int old = g_x_mask; g_x_mask = value; return old; }
    if (key == 23) { const int old = g_p2p_oneshot_rows; g_p2p_oneshot_rows = value; return old; }
    if (key == 0) { const int old = g_force_tile; g_force_tile = value; return old; }
    if (key == 1) { const int old = g_sk_slices; g_sk_slices = value; return old; }
    if (key == 19) { const int old = g_moe_deep; g_moe_deep = value; return old; }
    if (key == 18) { const int old = g_qkv_store_max_m; g_qkv_store_max_m = value; return old; }
    if (key == 17) { const int old = g_moe_bm; g_moe_bm = value; return old; }
    if (key == 16) { const int old = g_moe_gather; g_moe_gather = value; return old; }
    if (key == 15) { const int old = g_attn_nw; g_attn_nw = value; return old; }
    if (key == 14) { const int old = g_chunk_all_m; g_chunk_all_m = value; return old; }
    if (key == 13) { const int old = g_pass_interleave; g_pass_interleave = value; return old; }
    if (key == 12) { const int old = g_chunk_min_tiles; g_chunk_min_tiles = value; return old; }
    if (key == 11) { const int old = g_chunk_max_m; g_chunk_max_m = value; return old; }
    if (key == 10) { const int old = g_attn_tq2; g_attn_tq2 = value; return old; }
    if (key == 9) { const int old = g_norm_t16; g_norm_t16 = value; return old; }
    if (key == 8) { const int old = g_moe_small; g_moe_small = value; return old; }
    if (key == 7) { const int old = g_decode_seam; g_decode_seam = value; return old; }
    if (key == 6) { const int old = g_msplit_ks; g_msplit_ks = value; return old; }
    if (key == 5) { const int old = g_narrow_waves; g_narrow_waves = value; return old; }
    if (key == 4) { const int old = g_wide_ksplit; g_wide_ksplit = value; return old; }
    if (key == 3) { const int old = g_defer_norm; g_defer_norm = value; return old; }
    if (key == 2) { const int old = g_force_ntw; g_force_ntw = value; return old; }
    if (key == 27) { const int old = g_half_tiles; g_half_tiles = value; return old; }
    if (key == 29) { const int old = g_p2p_spin_ms; g_p2p_spin_ms = value; return old; }
    if (key == 30) { const int old = g_mamba_ssd; g_mamba_ssd = value; return old; }
    if (key == 31) { const int old = g_moe_fused_route; g_moe_fused_route = value; return old; }
    if (key == 33) { const int old = g_moe_defer_down; g_moe_defer_down = value; return old; }
    if (key == 35) { const int old = g_attn_kv_nt; g_attn_kv_nt = value; return old; }
    if (key == 36) { const int old = g_wide_ntb; g_wide_ntb = value; return old; }
    return -1;
}
