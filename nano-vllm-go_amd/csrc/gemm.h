// gemm.h — projection GEMMs of the forward path (replaces tensor.MatMul,
// purego/tensor/tensor.go:62-88, at every call site listed in SURVEY.md §8 a3).
//
//   C[M,N] = A[M,K] · W[N,K]^T   (+ fused epilogue)
//
// bf16 path (NVL_PRECISION_BF16).  Both operands live in HBM in the fragment-major layout of
// common.h (fm_index): 16-row x 32-k MFMA operand blocks, 1 KiB contiguous each.  The reference
// keeps weights [in,out] fp32 (generic_loader.go:398-403); nvl_upload_tensor converts once at load,
// and every kernel that produces a GEMM input (norm, attention, SwiGLU/GELU epilogue) writes it
// directly in this layout.
//   * gemm_bf16_pp_kernel (prefill, >= 192 tiles of 256x256): the ping-pong form — two groups of four waves one phase
//     apart, K in 32-wide stages through a 4-slot LDS ring filled by global_load_lds_dwordx4, so one group's fragment
//     reads run beside the other's MFMAs.  MFMA-bound.  Also the grouped launch of the MoE prefill (GemmArgs::tile_map:
//     an m-tile = an expert's 256-row segment of the expert-ordered row buffer).
//   * gemm_bf16_kernel (prefill, fewer tiles; decode-sized MoE expert groups): lock-step tiles, by default 128x128x64 with 4 waves
//     (2x2), each wave a 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators; operand blocks go HBM->LDS with
//     global_load_lds_dwordx4 (one contiguous KiB per wave-instruction, no VGPR round trip), double-buffered; the LDS
//     image is lane-linear so the fragment ds_read_b128s are bank-conflict-free with no swizzle.
//   * gemm_skinny_bf16_kernel / gemm_skinny_wide_bf16_kernel (decode, M <= 64): weight-streaming, HBM-bound (see below);
//     the residual projections also carry the deferred RMSNorm (GemmArgs::rs_out / rs_in); their MOE instantiations run
//     a decode MoE block as two dense projections over all experts with the routing weights as a mask (GemmArgs::moe_gate).
// The MFMA is issued "swapped" (weights as the A operand) so each lane ends up with 4 consecutive N
// elements of one output row: 8/16-byte epilogue accesses, and the bf16 epilogues can write the
// next GEMM's fragment-major operand directly.
//
// f32 path (NVL_PRECISION_F32): plain LDS-tiled fp32 FMA kernel on row-major operands, k ascending —
// the tight-tolerance parity mode, not a performance path.
#pragma once
#include <stdexcept>
#include "common.h"

namespace nvl {

enum GemmEpi {
    EPI_STORE = 0,   // C = acc (+bias)
    EPI_RESID = 1,   // X += alpha * (acc (+bias))      X fp32, in place   (generic_model.go:320-326)
    EPI_SWIGLU = 2,  // C[m, f] = silu(gate) * up        W rows interleaved (transformer.go:50-66)
    EPI_GELU = 3,    // C = gelu_tanh(acc + bias)                           (transformer.go:67-78)
    EPI_QKV = 4,     // fused QKV projection epilogue (head_dim 64 / 128): bias, RoPE on Q and K (rope.go:153-205),
                     // Q -> q buffer, K and V -> their KV slabs [pos][hd]  (replaces the fp32 qkv
                     // round trip + rope_kv_kernel in prefill; Concatenate tensor.go:283-321)
};

struct QkvEpi {             // EPI_QKV only
    const int32_t* tok_pos;
    const int32_t* tok_tbl;     // per token: where its sequence's block table starts in blk_table
    const int32_t* blk_table;   // KV block ids (slab mode: the slot); Tmax tokens per block
    const float* cos_t;     // [max_seq][64] or NULL (no RoPE)
    const float* sin_t;
    bf16_t* q_out;          // [tokens][nH*64] row-major
    bf16_t* kcache;         // layer base
    bf16_t* vcache;
    int64_t slot_stride;
    int Tmax, nH, nKV, hd;
};

struct GemmArgs {
    const void* A;          // bf16: fragment-major [M_pad16][K]; f32: row-major [M][lda]
    int lda;
    const int32_t* a_rows;  // optional gather: logical row r reads A row a_rows[r] (MoE), else NULL
    const void* W;          // bf16: fragment-major [N_pad128][K]; f32: row-major [N_pad128][K]
    void* C;                // fp32 outputs row-major [.][ldc]; bf16 outputs fragment-major [.][ldc]
    int ldc;
    const float* bias;      // [N] or NULL
    float alpha;            // EPI_RESID multiplier
    int M, N, K;            // logical sizes
    const int32_t* seg;     // optional device {start,end}: rows [start,end) of A (via a_rows if set)
                            // and of C; M is then only the grid bound (MoE expert segments)
    int c_row0;             // row offset added to every output row (set from seg inside the kernel)
    // grouped (MoE) launch: m-tile i of the grid is rows [row0, row0+nrows) of expert e, from the device table
    // tile_map[i] = {e, row0, nrows, 0}; blocks with i >= *n_mtiles exit; expert e's weights start at
    // W + e * w_expert_stride.  One launch covers every expert (moe.go:99-120 loops experts per token).
    const int32_t* tile_map;
    const int32_t* n_mtiles;
    int64_t w_expert_stride;
    int x_mask;             // decode kernels: rows >= M of a 16-row activation tile are not fetched (set when M % 16 != 0)
    int grp_bm;             // rows per m-tile of the grouped launch (128, or 256: the 256x128 three-stage instance)
    int grp_deep;           // grouped launch on the four-stage 128x128 instance (decode-sized MoE batches)
    QkvEpi qkv;             // EPI_QKV
    // decode kernel, EPI_RESID only: K split over gridDim.y = sk_slices workgroups per column block.  Each
    // writes its fp32 partial tile to sk_part[slice][m][n] (ld = N) with plain stores; the NEXT kernel in the
    // stream — the norm that follows every residual add — folds x += alpha * sum(slices) into its read of x.
    // The kernel boundary is the synchronisation: no fences, no atomics, fixed summation order.
    float* sk_part;
    int sk_slices;
    // Deferred RMSNorm (decode; RMSNorm + SwiGLU, sequential block): the residual projection that completes x also
    // emits the NEXT projection's operand un-normalised, xn_raw = bf16(x * w_norm) (fragment-major), plus the sums of
    // x^2 over its 16 columns, rs_out[tile][m]; the consumer scales its accumulators by
    // rstd[m] = 1 / sqrt(sum_tiles(rs_in[tile][m]) / H + eps) — (x*w/rms)·W == (1/rms)·((x*w)·W).  The norm launch
    // between the two projections disappears; all sums keep a fixed order.
    int m_split;            // narrow decode kernel: gridDim.y indexes 16-row activation tiles (not K slices): the
                            // workgroups of a weight tile stream the same weights (one XCD: L2 serves the second)
    int m_passes;           // decode kernels, 64 < M: the launch covers m_passes = ceil(M/64) groups of 64 activation rows;
                            // the workgroups of one weight tile sit 8 block ids apart — same XCD, dispatched together —
                            // so the weights cross HBM once and the later groups hit L2 (see launch_gemm_bf16)
    int grid_y;             // decode kernels: gridDim.y of the launch (set by the launcher)
    unsigned long long* stamps;   // diagnostic runs: per-workgroup time stamps (common.h nvl_stamp), else NULL
    int rs_half;            // producer, M <= 16: 8 weight rows per workgroup, x^2 partials per 8 columns (rs_tiles = N / 8)
    const float* nrm_w;     // producer: weight of the norm that follows, [N]
    bf16_t* nrm_xn;         // producer: xn_raw [M_pad16][N] fragment-major
    float* rs_out;          // producer: [64 rows][N/16]
    const float* rs_in;     // consumer: the same buffer
    int rs_tiles;
    float rs_inv_h, rs_eps;
    // MoE decode, "dense-masked" form (decode kernels only, M <= 64).  All experts are ONE projection each way:
    //   up   W = [E * 2I][H] (the experts' interleaved gate|up rows back to back), output column f belongs to expert f / I;
    //   down W = [H][E * I] (the experts concatenated along K), a wave's K range lies inside ONE expert.
    // moe_gate [M][E] holds the renormalised routing weight of (token, expert), 0 when the token did not pick the expert.
    // A workgroup (up) / wave (down) whose expert no token of the batch picked returns before it loads a weight byte, so
    // only the touched experts are streamed; the up epilogue multiplies row m by moe_gate[m][e], which makes the down
    // projection's K reduction the weighted combine itself (moe.go:105-120) — no sort, no plan, no combine launch.
    const float* moe_gate;
    int moe_E, moe_I;
};
// does any row of the batch (M <= 64: one per lane) route to expert e?  Call with all 64 lanes active.
__device__ __forceinline__ bool moe_expert_live(const GemmArgs& p, int e) {
    const int lane = threadIdx.x & 63;
    const float g = lane < p.M ? p.moe_gate[lane * p.moe_E + e] : 0.f;
    return __any(g != 0.f);
}

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }
__device__ __forceinline__ float deferred_rstd(const GemmArgs& p, int m);

// ------------------------------------------------------------------------------------------
// epilogue shared by all kernels: 4 consecutive n for one row m
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT>
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v, float rstd = 1.0f) {
    if (m >= p.M || n >= p.N) return;
    v *= rstd;                       // deferred RMSNorm of the input row (1 otherwise), before the bias
    const bool full = (n + 3 < p.N);
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (n + r < p.N) v[r] += p.bias[n + r];
    }
    if (EPI == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gelu_tanh_f(v[r]);
    }
    const int64_t row = (int64_t)p.c_row0 + m;
    if (EPI == EPI_RESID) {
        float* x = (float*)p.C + row * p.ldc + n;
        if (full) {
            f32x4 o = *(f32x4*)x;
            o += p.alpha * v;
            *(f32x4*)x = o;
        } else {
            for (int r = 0; r < 4; r++)
                if (n + r < p.N) x[r] += p.alpha * v[r];
        }
        return;
    }
    if (sizeof(OutT) == 4) {
        float* c = (float*)p.C + row * p.ldc + n;
        if (full) *(f32x4*)c = v;
        else
            for (int r = 0; r < 4; r++)
                if (n + r < p.N) c[r] = v[r];
    } else {
        // bf16 activation for the next GEMM: widths are multiples of 64, so the group is always full
        act_store4<bf16_t>((bf16_t*)p.C, row, n, p.ldc, v);
    }
}

// deferred RMSNorm scale of input row m (< 64): the 4 lanes that share a row (same lane & 15) split the tile partials.
// Call with all 64 lanes active; 1 when the projection's operand is already normalised.
__device__ __forceinline__ float deferred_rstd(const GemmArgs& p, int m) {
    if (!p.rs_in) return 1.0f;
    // rs_in is [64 rows][rs_tiles] (rs_tiles % 16 == 0): a lane sums a quarter of its row with independent 16-byte loads
    const int per = p.rs_tiles >> 2;
    const f32x4* src = (const f32x4*)(p.rs_in + (int64_t)m * p.rs_tiles + ((threadIdx.x & 63) >> 4) * per);
    f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
    // only the lanes of live rows fetch: each of these loads touches one cache line per row, every epilogue wave of every
    // workgroup issues them, and at batch 1 the 15 padding rows were 94 % of the requests queued at the L2 channels that
    // hold these few KiB (batch 1: QKV 5.1 -> 4.6 us, FFN-up 13.3 -> 12.2 us, LM head 45.6 -> 42.8 us per launch,
    // profiles/r03_timeline_b1_rstd.txt; the non-deferred forms take 4.1 / 12.0 / 42.2).  Rows >= M get a finite scale
    // nobody stores with.  (Tried on top: the sums once per workgroup, coalesced, issued ahead of the weight stream — the
    // first weight data came 0.9 us earlier and the stream took 0.9 us longer, same file.)
    if (m < p.M) {
#pragma unroll 8
        for (int q = 0; q < (per >> 2); q++) a4 += src[q];
    }
    float s = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return 1.0f / sqrtf(s * p.rs_inv_h + p.rs_eps);
}

template <typename OutT, bool MOE = false>
__device__ __forceinline__ void epilogue_swiglu4(const GemmArgs& p, int m, int f, f32x4 g, f32x4 u, float rstd = 1.0f) {
    // p.N counts fused rows (2F); the output has F = N/2 columns, ldc = F
    if (m >= p.M || f >= (p.N >> 1)) return;
    g *= rstd; u *= rstd;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = silu_f(g[r]) * u[r];
    if constexpr (MOE) v *= p.moe_gate[m * p.moe_E + f / p.moe_I];      // routing weight of (token, expert); 0 = not routed
    act_store4<OutT>((OutT*)p.C, (int64_t)p.c_row0 + m, f, p.ldc, v);
}

// EPI_QKV: one wave sub-tile of TH*16 columns = one head (TH = 4: hd 64, TH = 8: hd 128).  The lane holds, for
// token m, d = 16j + 4fg + r (j = 0..TH-1): the RoPE partner d + hd/2 is tile j + TH/2 of the SAME lane.
template <int TH>
__device__ __forceinline__ void epilogue_qkv_head(const GemmArgs& p, int m, int head, int fg, f32x4 (&t)[TH]) {
    constexpr int HD = TH * 16;
    if (m >= p.M) return;
    const QkvEpi& q = p.qkv;
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < TH; j++) t[j] += *(const f32x4*)(p.bias + head * HD + 16 * j + 4 * fg);
    }
    const int pos = q.tok_pos[m];
    if (head < q.nH + q.nKV && q.cos_t) {
#pragma unroll
        for (int j = 0; j < TH / 2; j++) {
            const f32x4 c = *(const f32x4*)(q.cos_t + (int64_t)pos * HD + 16 * j + 4 * fg);
            const f32x4 s = *(const f32x4*)(q.sin_t + (int64_t)pos * HD + 16 * j + 4 * fg);
            const f32x4 x1 = t[j], x2 = t[j + TH / 2];
            t[j] = x1 * c + (-x2) * s;          // rope.go:196-202 (table halves are duplicates: cos[d+hd/2] == cos[d])
            t[j + TH / 2] = x2 * c + x1 * s;
        }
    }
    int blk = 0, row = 0;
    if (head >= q.nH) kv_locate(q.blk_table, q.tok_tbl[m], pos, q.Tmax, blk, row);
    // Q -> q buffer; K and V -> their slabs, both row-major [pos][hd] (kv_cache.go:5-6)
    bf16_t* dst;
    if (head < q.nH) dst = q.q_out + (int64_t)m * (q.nH * HD) + head * HD + 4 * fg;
    else if (head < q.nH + q.nKV) dst = q.kcache + (int64_t)blk * q.slot_stride + ((int64_t)(head - q.nH) * q.Tmax + row) * HD + 4 * fg;
    else dst = q.vcache + (int64_t)blk * q.slot_stride + ((int64_t)(head - q.nH - q.nKV) * q.Tmax + row) * HD + 4 * fg;
#pragma unroll
    for (int j = 0; j < TH; j++) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = (bf16_t)t[j][r];
        *(bf16x4*)(dst + 16 * j) = o;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel (prefill), one template, three instances chosen per shape by the launcher:
//   <128,128, 2,2, 2>  4 waves, wave tile  64x64, 2 LDS stages (64 KiB, 2 workgroups/CU): small M, MoE segments
//   <256,128, 4,2, 3>  8 waves, wave tile  64x64, 3 LDS stages (144 KiB): two K tiles in flight
//   <256,256, 2,4, 2>  8 waves, wave tile 128x64, 2 LDS stages (128 KiB): least LDS traffic per MFMA
//                      ((8+4) fragment reads per 32 MFMAs instead of (4+4) per 16) — the large-M default
// BK = 64 (two MFMA k-steps).  Every operand tile is a run of 1-KiB fragment-major blocks, so one
// global_load_lds_dwordx4 per block fills LDS lane-linearly and the fragment ds_read_b128s are
// conflict-free.  Raw s_barrier + counted s_waitcnt vmcnt keep (STAGES-2) later tiles in flight
// across the barrier (never vmcnt(0) in the steady state of the 3-stage instance).
// ------------------------------------------------------------------------------------------
constexpr int G_BK = 64;

// XCD-aware block remap (bijective for any grid size): blocks b and b+8 share an XCD, so give
// each XCD a contiguous run of tiles -> neighbouring tiles (same A row panel) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// tile id -> (tm, tn), "grouped" order: ids run down GM row-tiles before moving to the next column
// tile, so the ~32 workgroups an XCD runs at once (a contiguous id range after xcd_remap) cover a
// compact GM x (32/GM) patch: every A row-panel chunk pulled into that XCD's L2 is reused by 32/GM
// column tiles and every W panel chunk by GM row tiles, instead of each tile streaming its own W
// panel from MALL/HBM (tiles advance through K in near lock-step, so the live window is small).
__device__ __forceinline__ void tile_coords(int pid, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
    const int per_group = GM * tiles_n;
    const int group = pid / per_group, first = group * GM;
    const int gsz = (tiles_m - first) < GM ? (tiles_m - first) : GM;
    const int r = pid - group * per_group;
    tm = first + r % gsz;
    tn = r / gsz;
}

template <int BM, int BN, int STAGES>
constexpr int gemm_lds_bytes() { return STAGES * (BM + BN) * G_BK * 2; }

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, typename OutT>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 4 && STAGES == 2) ? 2 : 1)
void gemm_bf16_kernel(GemmArgs p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;   // 16x16 accumulator tiles per wave
    constexpr int NA = BM / 16 * 2, NB = BN / 16 * 2;               // 1-KiB blocks per stage (A, W)
    constexpr int PW = (NA + NB) / NW;                               // blocks staged per wave per K tile
    constexpr int STAGE_BYTES = (BM + BN) * G_BK * 2;
    static_assert((NA + NB) % NW == 0, "blocks must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    int row_base = 0;
    if (p.tile_map) {                      // grouped launch: the m-tile index selects (expert, row segment)
        tn = blockIdx.x % tiles_n;
        const int ti = blockIdx.x / tiles_n;
        if (ti >= *p.n_mtiles) return;
        const int e = p.tile_map[4 * ti];
        row_base = p.tile_map[4 * ti + 1];
        p.M = p.tile_map[4 * ti + 2];
        p.c_row0 = row_base;
        p.W = (const bf16_t*)p.W + (int64_t)e * p.w_expert_stride;
        tm = 0;
    } else {
        tile_coords(xcd_remap(blockIdx.x, gridDim.x), gridDim.x / tiles_n, tiles_n, NW == 4 ? 8 : 4, tm, tn);
        if (p.seg) {
            row_base = p.seg[0];
            p.M = p.seg[1] - row_base;
            p.c_row0 = row_base;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fg = lane >> 4;
    if (m0 >= p.M) return;

    // ---- staging sources: wave w moves blocks w*PW .. w*PW+PW-1 of the stage's [A blocks | W blocks] list
    const bf16_t* src[PW];
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int blk = wave * PW + i;
        if (blk < NA) {
            const int rt = blk >> 1, ksb = blk & 1;
            int am = m0 + rt * 16 + fr;
            if (am > p.M - 1) am = p.M - 1;                 // clamp: rows >= M are never stored
            am += row_base;
            if (p.a_rows) am = p.a_rows[am];
            src[i] = (const bf16_t*)p.A + ((((int64_t)(am >> 4) * (p.K >> 5) + ksb) * 64) + (am & 15) + 16 * fg) * 8;
        } else {
            const int wb = blk - NA, rt = wb >> 1, ksb = wb & 1;
            src[i] = (const bf16_t*)p.W + (((int64_t)((n0 >> 4) + rt) * (p.K >> 5) + ksb) * 64 + lane) * 8;
        }
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * (PW * 1024);
        const int64_t koff = (int64_t)kt * 1024;            // 2 k-steps x 512 elements per K tile
#pragma unroll
        for (int i = 0; i < PW; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + koff),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
    };
    auto wait_tiles_in_flight = [&](int tiles) {   // s_waitcnt needs an immediate
        if (tiles <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (PW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (tiles == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");       // four stages: two younger tiles of 8 loads each
    };
    static_assert(STAGES == 2 || PW == 6 || PW == 8, "add the immediate for this PW");
    static_assert(STAGES <= 3 || PW == 8, "four stages: only the PW = 8 immediates exist");

    // ---- fragment reads: stage image [A row-tile][k-step][lane][16 B] then [W row-tile][k-step][lane][16 B]
    const int a_blk_off = wm * TM * 2048 + lane * 16;                 // activation rows -> MFMA B operand (output col m)
    const int w_blk_off = NA * 1024 + wn * TN * 2048 + lane * 16;     // weight rows     -> MFMA A operand (output row n)

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K split over gridDim.y workgroups per tile (residual projections of 65..512 rows: N / 128 x M / 128 tiles alone are
    // far too few workgroups): slice y takes K tiles [kt0, kt0 + nt) and leaves an fp32 partial tile in sk_part[y]; the
    // norm that follows sums the slices into x (PendingResid: the kernel boundary is the synchronisation)
    const int nt_all = p.K / G_BK;
    int nt = nt_all, kt0 = 0;
    if (EPI == EPI_RESID && p.sk_part) {
        const int per = nt_all / (int)gridDim.y, extra = nt_all - per * (int)gridDim.y;
        kt0 = (int)blockIdx.y * per + ((int)blockIdx.y < extra ? (int)blockIdx.y : extra);
        nt = per + ((int)blockIdx.y < extra ? 1 : 0);
    }
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++)
        if (s < nt) stage(s, kt0 + s);
    wait_tiles_in_flight((nt < STAGES - 1 ? nt : STAGES - 1) - 1);
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int t = 0; t < nt; t++) {
        if (t + STAGES - 1 < nt) {
            int nb = cur + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            stage(nb, kt0 + t + STAGES - 1);
        }
        const char* sbuf = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TN; j++) wf[j] = *(const bf16x8*)(sbuf + w_blk_off + j * 2048 + ks * 1024);
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = *(const bf16x8*)(sbuf + a_blk_off + i * 2048 + ks * 1024);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        // tile t+1 must have landed; up to STAGES-2 younger tiles may stay in flight
        int younger = nt - 2 - t;
        if (younger > STAGES - 2) younger = STAGES - 2;
        wait_tiles_in_flight(younger);
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }

    // ---- epilogue: lane holds rows n = 4*fg + r of column m = fr of each 16x16 tile ----
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + wm * (TM * 16) + i * 16 + fr;
        if (EPI == EPI_QKV) {
            // the wave's column tile is exactly one head: TN = 4 <-> head_dim 64, TN = 8 <-> head_dim 128
            const int head = (n0 + wn * (TN * 16)) / (TN * 16);
            if (head * (TN * 16) < p.N) epilogue_qkv_head<TN>(p, m, head, fg, acc[i]);
        } else if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const int ntile = (n0 + wn * (TN * 16) + j * 16) >> 4;   // even: gate block, +1: up block
                const int f = (ntile >> 1) * 16 + 4 * fg;
                epilogue_swiglu4<OutT>(p, m, f, acc[i][j], acc[i][j + 1]);
            }
        } else if (EPI == EPI_RESID && p.sk_part) {
            GemmArgs q = p;                                   // this slice's partial tile: plain fp32 store, bias with slice 0
            q.C = p.sk_part + (int64_t)blockIdx.y * p.M * p.N; q.ldc = p.N; q.c_row0 = 0;
            if (blockIdx.y != 0) q.bias = nullptr;
#pragma unroll
            for (int j = 0; j < TN; j++)
                epilogue4<EPI_STORE, float>(q, m, n0 + wn * (TN * 16) + j * 16 + 4 * fg, acc[i][j]);
        } else {
#pragma unroll
            for (int j = 0; j < TN; j++)
                epilogue4<EPI, OutT>(p, m, n0 + wn * (TN * 16) + j * 16 + 4 * fg, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel, ping-pong form (prefill, tuning option 5): 256x256 tile, 8 waves as two groups of four (one wave
// of each group per SIMD), K in 32-wide stages through a STAGES-slot LDS ring (4 slots = 128 KiB by default).  A stage is one PHASE for a wave:
//   L: 12 fragment ds_reads of stage s, LDS-DMA issue of stage s+STAGES-1, lgkmcnt(0), counted vmcnt (stage s+1 landed)
//   -- barrier --   C: 32 MFMAs at priority 1   -- barrier --
// Group 1 enters the loop one barrier late, so while one wave of a SIMD issues its MFMAs the other does its LDS reads
// and DMA issue: the fragment reads (384 of every 896 cycles in the lock-step kernel above) move under the MFMAs.
// Hazards (LDS-DMA is ordered only by the issuer's vmcnt + a barrier the reader passes): a slot is re-filled in L(s)
// after its last readers retired their ds_reads before the barrier that ended L(s-1) of the LATE group (WAR);
// stage s+1 is waited for at the end of L(s) by every wave, one barrier before the early group reads it (RAW).
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT, int STAGES = 4, int PRIO = 0, int WAVES_N = 4, int BN = 256>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_kernel(GemmArgs p) {
    // BN 256: WAVES_N = 4 -> wave tile 128x64 (a head of 64 per wave), WAVES_N = 2 -> 64x128 (a head of 128 per wave);
    // BN 128 (mid-size M: twice the tiles): WAVES_N = 2 -> wave tile 64x64
    constexpr int BM = 256, WAVES_M = 8 / WAVES_N, TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
    constexpr int NA = BM / 16, NB = BN / 16;            // 1-KiB blocks per stage (one k-step)
    constexpr int PW = (NA + NB) / 8;                    // blocks per wave per stage: 4 (BN 256) or 3 (BN 128)
    constexpr int STAGE_BYTES = (BM + BN) * 32 * 2;      // 32 KiB / 24 KiB
    static_assert((NA + NB) % 8 == 0, "stage blocks must divide over the 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + BN - 1) / BN;
    int tm, tn;
    int row_base = 0;
    if (p.tile_map) {
        // grouped launch (MoE prefill): m-tile ti = (expert, 256-row segment of its rows in the expert-ordered buffer).
        // After xcd_remap an XCD owns a contiguous range of (ti, tn): whole experts, so an expert's weights and each
        // row segment are re-read from that XCD's L2 by the tiles that share them.
        const int pid = xcd_remap(blockIdx.x, gridDim.x);
        tn = pid % tiles_n;
        const int ti = pid / tiles_n;
        if (ti >= *p.n_mtiles) return;                    // (whole workgroup: before any barrier)
        row_base = p.tile_map[4 * ti + 1];
        p.M = p.tile_map[4 * ti + 2];
        p.c_row0 = row_base;
        p.W = (const bf16_t*)p.W + (int64_t)p.tile_map[4 * ti] * p.w_expert_stride;
        tm = 0;
    } else {
        tile_coords(xcd_remap(blockIdx.x, gridDim.x), gridDim.x / tiles_n, tiles_n, 4, tm, tn);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = wave >> 2;                            // the ping-pong group: first / second dispatched half
    const int fr = lane & 15, fg = lane >> 4;
    if (m0 >= p.M) return;

    const bf16_t* src[PW];
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int blk = wave * PW + i;
        if (blk < NA) {
            int am = m0 + blk * 16 + fr;
            if (am > p.M - 1) am = p.M - 1;
            am += row_base;                               // (a segment may start on any row: per-lane 16-byte sources)
            src[i] = (const bf16_t*)p.A + (((int64_t)(am >> 4) * (p.K >> 5) * 64) + (am & 15) + 16 * fg) * 8;
        } else {
            src[i] = (const bf16_t*)p.W + ((int64_t)((n0 >> 4) + blk - NA) * (p.K >> 5) * 64 + lane) * 8;
        }
    }
    auto stage = [&](int buf, int ks) {
        char* base = smem + buf * STAGE_BYTES + wave * (PW * 1024);
#pragma unroll
        for (int i = 0; i < PW; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)ks * 512),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
    };
    const int a_off = wm * TM * 1024 + lane * 16;
    const int w_off = NA * 1024 + wn * TN * 1024 + lane * 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K >> 5;            // stages
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++)
        if (s < nt) stage(s, s);
    // stage 0 landed (my part); up to STAGES-2 younger stages stay in flight
    auto wait_younger = [&](int younger) {
        if (younger > STAGES - 2) younger = STAGES - 2;
        if (PW == 4) {
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    static_assert(STAGES >= 3 && STAGES <= 5 && (PW == 4 || PW == 3), "vmcnt immediates above cover up to three stages in flight");
    wait_younger((nt < STAGES - 1 ? nt : STAGES - 1) - 1);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();         // the late group: one phase behind from here on
    if (PRIO == 1 && grp == 1) __builtin_amdgcn_s_setprio(1);  // static priority for the second-dispatched half
    int buf = 0;
    for (int s = 0; s < nt; s++) {
        // ---- L phase ----
        bf16x8 af[TM], wf[TN];
        const char* sbuf = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < TN; j++) wf[j] = *(const bf16x8*)(sbuf + w_off + j * 1024);
#pragma unroll
        for (int i = 0; i < TM; i++) af[i] = *(const bf16x8*)(sbuf + a_off + i * 1024);
        if (s + STAGES - 1 < nt) {
            int nb = buf + STAGES - 1;                   // the slot read in phase s-1
            if (nb >= STAGES) nb -= STAGES;
            stage(nb, s + STAGES - 1);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): my fragment reads are retired
        // stage s+1 landed; stages s+2 .. s+4 (those that exist) may stay in flight
        wait_younger(nt - 2 - s);                        // (stages beyond s+1 that exist)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- C phase ----
        if (PRIO == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();         // pairs with the late group's last barrier

#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + wm * (TM * 16) + i * 16 + fr;
        if (EPI == EPI_QKV) {
            const int head = (n0 + wn * (TN * 16)) / (TN * 16);      // the wave's column block is exactly one head
            if (head * (TN * 16) < p.N) epilogue_qkv_head<TN>(p, m, head, fg, acc[i]);
        } else if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const int ntile = (n0 + wn * (TN * 16) + j * 16) >> 4;
                epilogue_swiglu4<OutT>(p, m, (ntile >> 1) * 16 + 4 * fg, acc[i][j], acc[i][j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; j++)
                epilogue4<EPI, OutT>(p, m, n0 + wn * (TN * 16) + j * 16 + 4 * fg, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 skinny kernel (decode: M <= 64 rows).  The projection is then a weight-streaming problem
// (HBM-bound: every weight byte is read once, activations are a few hundred KB in L2), so the
// shape of the kernel is set by memory-level parallelism, not by MFMA:
//   * one workgroup owns NTW 16-row weight tiles (16*NTW output columns) over ALL of K;
//   * it runs NTW*KSPLIT waves; wave w streams tile w/KSPLIT over K slice w%KSPLIT straight
//     HBM -> VGPR with non-temporal 16-byte loads — one contiguous KiB per instruction thanks to the
//     fragment-major layout; no LDS round trip, nothing is shared between waves; two register sets
//     keep the next block of k-steps in flight behind the current block's MFMAs;
//   * activations (<= 64 x K bf16, fragment-major too) are read as MFMA fragments from L2;
//   * the K slices are reduced through LDS in fixed wave order (deterministic, no atomics) and the
//     fused epilogue (bias / residual / SwiGLU / GELU) runs once per output element.  For SwiGLU
//     NTW = 2: tile 0 is the gate block, tile 1 the up block of the same 16 features.
// ------------------------------------------------------------------------------------------
// weight fragments are read once per launch: non-temporal.  In the 64-row-group launches the groups of a weight block
// read them side by side, so there they take the regular path and stay in L2 for the other groups.
template <bool KEEP>
__device__ __forceinline__ bf16x8 weight_load(const bf16x8* p) {
    if constexpr (KEEP) return *p;
    else return __builtin_nontemporal_load(p);
}

// m_passes > 1: block id -> (weight block, 64-row activation group).  Ids that differ by 8 share an XCD and are
// dispatched back to back: the groups of one weight block stream the same weights at the same time.
__device__ __forceinline__ void skinny_pass_remap(GemmArgs& p, int& bx) {
    const int slot = bx >> 3, z = slot % p.m_passes;
    bx = (slot / p.m_passes) * 8 + (bx & 7);
    p.A = (const bf16_t*)p.A + (int64_t)z * 64 * p.K;      // fragment-major: 16-row tiles are contiguous
    p.c_row0 += z * 64;
    p.M = min(64, p.M - z * 64);                           // the last group may be ragged (operands are padded to 64 rows)
    if (p.rs_in) p.rs_in += (int64_t)z * 64 * p.rs_tiles;  // deferred RMSNorm: this group's rows of the x^2 partials
    if (p.rs_out) {                                        // ... and on the producer side (its epilogue indexes rows locally)
        p.C = (float*)p.C + (int64_t)z * 64 * p.ldc;
        p.nrm_xn += (int64_t)z * 64 * p.N;
        p.rs_out += (int64_t)z * 64 * (p.N >> 4);
    }
}

// HALF (deferred-norm residual projections of <= 16 rows: O and FFN-down at decode batches 1..16): a workgroup owns 8 of
// a weight tile's 16 rows, so a 2048-column projection runs on 256 workgroups instead of 128 (x must be complete when the
// launch ends — no K split over workgroups —, and with one workgroup per 16-row tile half of the CUs stayed idle while the
// others pulled 256 KB each through one CU).  The other 8 lanes of every 16-lane row group feed zeros (their loads are
// predicated off: 4 x 128-byte lines per k-step instead of 8); only the owned rows are stored; the x^2 partials are per
// 8 columns (rs_tiles = N / 8).
template <int MT, int NTW, int U, int EPI, typename OutT, bool PASSES = false, bool MOE = false, bool HALF = false>
__global__ __launch_bounds__(1024) void gemm_skinny_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = (f32x4*)smem;                       // [NTW][ksplit][MT][64]
    const int swg = blockIdx.y * gridDim.x + blockIdx.x;
    NvlStamps stamps(p.stamps, swg);                 // (diagnostic runs; slot 0: workgroup entered)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksplit = (blockDim.x >> 6) / NTW;
    const int tile = wave / ksplit, kw = wave - tile * ksplit;
    const int fr = lane & 15, fg = lane >> 4;
    int bx = blockIdx.x;
    if constexpr (PASSES) skinny_pass_remap(p, bx);     // (its own instantiation: rewriting the arguments costs the
                                                        // single-group kernels a few scratch dwords otherwise)
    const int half = HALF ? (bx & 1) : 0;
    if constexpr (HALF) bx >>= 1;
    const bool w_live = !HALF || (fr >> 3) == half;  // HALF: this lane's weight row belongs to the workgroup
    const int nt0 = bx * NTW;                        // first 16-row weight tile of this workgroup
    // K/32 k-steps dealt over (gridDim.y slices) x (ksplit waves) as evenly as possible (K need not divide:
    // Falcon's 4544 = 142 steps)
    const int i_off = p.m_split ? blockIdx.y : 0;                  // first 16-row activation tile of this workgroup
    // deferred RMSNorm (consumer side): the scale of the rows of this wave's first epilogue element, fetched under the weight
    // stream instead of after it.  Issued HERE, ahead of the first weight block.  (Round 3 moved it behind the first block on
    // the evidence of in-kernel time stamps — first weight data 0.5 us earlier — and lost 0.5 us per QKV launch and 0.25 us
    // per FFN-up launch at B = 32 by the kernel trace of the unstamped build: the stamp sites themselves had changed the
    // schedule they measured.  profiles/r03_ab_vs_r02.txt)
    const int i_pre = wave % MT;
    // (MT = 4 — batches above 32 and the 64-row groups of larger ones — has no registers to hold it across the stream:
    // 34-54 spilled there, QKV at 128 rows 7.2 -> 10.8 us; those launches are long enough to fetch it in the epilogue)
    constexpr bool RSTD_PRE = MT <= 2;
    const float rstd_pre = (RSTD_PRE && (EPI == EPI_STORE || EPI == EPI_SWIGLU) && wave < NTW * MT) ? deferred_rstd(p, 16 * i_pre + fr) : 1.0f;   // (only the waves that run an epilogue element)
    // (grid_y: gridDim.y as an explicit argument — the built-in lives in the hidden part of the kernarg block and costs a
    // dependent scalar round trip of its own before the first weight load)
    const int nslices = p.m_split ? 1 : p.grid_y, slice = p.m_split ? 0 : blockIdx.y;
    const int nparts = ksplit * nslices, part = slice * ksplit + kw;
    const int nks = p.K >> 5, q = nks / nparts, rr = nks - q * nparts;
    int my_steps = q + (part < rr ? 1 : 0);
    const int ks0 = part * q + (part < rr ? part : rr);   // first k-step of this wave
    // MoE decode (its own instantiation): stream only the experts some token of the batch picked.  Up: a workgroup's rows
    // belong to one expert.  Down: a wave's K range covers up to four experts (one with the K split over workgroups, two
    // to four in the deferred-norm form, which keeps all of K in one workgroup); blk_live bit b = block b (U k-steps,
    // inside one expert: the launcher checks the divisibility) of this wave's range is to be streamed.
    uint32_t blk_live = 0xffffffffu;
    bool tail_live = true;                 // the ragged tail (fewer than U k-steps; it lies inside the range's last expert)
    if constexpr (MOE) {
        if (EPI == EPI_SWIGLU) { if (!moe_expert_live(p, (nt0 * 16) / (2 * p.moe_I))) return; }          // (whole workgroup: uniform)
        else {
            const int spe = p.moe_I >> 5, e0 = ks0 / spe, e1 = (ks0 + my_steps - 1) / spe;      // k-steps per expert; this wave's experts
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; j++) g[j] = (e0 + j <= e1 && lane < p.M) ? p.moe_gate[lane * p.moe_E + e0 + j] : 0.f;   // (one round trip for all four)
            blk_live = 0;
            tail_live = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!__any(g[j] != 0.f)) continue;
                if (e0 + j == e1) tail_live = true;
                const int b0 = max(((e0 + j) * spe - ks0) / U, 0), b1 = min(((e0 + j + 1) * spe - ks0) / U, 32);
                if (b1 > b0) blk_live |= (b1 - b0 >= 32 ? 0xffffffffu : ((1u << (b1 - b0)) - 1u)) << b0;
            }
        }
    }

    const bf16_t* wp = (const bf16_t*)p.W + (((int64_t)(nt0 + tile) * (p.K >> 5) + ks0) * 64 + lane) * 8;
    const bf16_t* xp[MT];
#pragma unroll
    for (int i = 0; i < MT; i++)   // rows >= M of the last 16-row tile exist (padded allocation) and are never stored
        xp[i] = (const bf16_t*)p.A + (((int64_t)(i + i_off) * (p.K >> 5) + ks0) * 64 + lane) * 8;
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Only the lanes of rows < M fetch their activation fragment (the rest feed zeros: their output columns are never
    // stored).  The CU's ingest rate, not HBM, bounds these kernels, and at M = 1 a padded 16-row fragment would be as
    // many bytes as the weight fragment beside it (M <= 4: a quarter of it with the mask, four 64-byte sectors).
    bool x_live[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) x_live[i] = !p.x_mask || 16 * (i + i_off) + fr < p.M;   // (x_mask: the host sets it when M % 16 != 0)
    const bf16x8 x_zero = {};

    // k loop: blocks of U k-steps (U*32 of K), two register sets -> the next block's loads are in
    // flight while the current block's MFMAs issue (the compiler emits counted vmcnt for these).
    const int nblk = my_steps / U;
    bf16x8 wA[U], xA[U][MT], wB[U], xB[U][MT];
    auto load_blk = [&](bf16x8 (&w)[U], bf16x8 (&x)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (HALF) w[u] = w_live ? weight_load<PASSES>((const bf16x8*)(wp + (int64_t)(b * U + u) * 512)) : x_zero;
            else w[u] = weight_load<PASSES>((const bf16x8*)(wp + (int64_t)(b * U + u) * 512));
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const bf16x8* src = (const bf16x8*)(xp[i] + (int64_t)(b * U + u) * 512);
                if (p.x_mask) x[u][i] = x_live[i] ? *src : x_zero; else x[u][i] = *src;      // (uniform branch)
            }
        }
    };
    auto comp_blk = [&](bf16x8 (&w)[U], bf16x8 (&x)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < MT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u], x[u][i], acc[i], 0, 0, 0);
    };
    // MoE down: the live blocks of the wave's range, in order, through the same two register sets
    uint32_t live = 0;
    int bA = -1;
    if constexpr (MOE && EPI == EPI_RESID) {
        live = blk_live & (nblk >= 32 ? 0xffffffffu : ((1u << nblk) - 1u));
        live = __builtin_amdgcn_readfirstlane(live);
        if (live) { bA = __builtin_ctz(live); live &= live - 1; load_blk(wA, xA, bA); }
    } else {
        if (nblk > 0) load_blk(wA, xA, 0);
    }
    stamps.mark(1);                     // first block of loads issued
    // the residual operand of this wave's first epilogue element (deferred-norm producer form): fetched under the weight
    // stream, not as a dependent round trip after the K reduction (this launch is the only writer of these elements)
    f32x4 x_pre = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_RESID && p.rs_out && wave < NTW * MT) {
        const int t = wave / MT, i = wave - t * MT;
        const int m = 16 * (i + i_off) + fr, n = (nt0 + t) * 16 + 4 * fg;
        if (m < p.M && n < p.N && (!HALF || (fg >> 1) == half)) x_pre = *(const f32x4*)((const float*)p.C + (int64_t)m * p.ldc + n);
    }
    if constexpr (MOE && EPI == EPI_RESID) {
        while (bA >= 0) {
            int bB = -1;
            if (live) { bB = __builtin_ctz(live); live &= live - 1; load_blk(wB, xB, bB); }
            comp_blk(wA, xA);
            if (bB < 0) break;
            bA = -1;
            if (live) { bA = __builtin_ctz(live); live &= live - 1; load_blk(wA, xA, bA); }
            comp_blk(wB, xB);
        }
    } else {
    int b = 0;
    for (; b + 2 <= nblk; b += 2) {
        load_blk(wB, xB, b + 1);
        comp_blk(wA, xA);
        if (b == 0) stamps.mark_used(2, acc[0]);       // first block's data arrived and was used
        if (b + 2 < nblk) load_blk(wA, xA, b + 2);
        comp_blk(wB, xB);
    }
    if (b < nblk) comp_blk(wA, xA);
    }
    for (int s = nblk * U; s < (tail_live ? my_steps : 0); s++) {      // ragged tail: fewer than U k-steps
        const bf16x8 w = w_live ? weight_load<PASSES>((const bf16x8*)(wp + (int64_t)s * 512)) : x_zero;
#pragma unroll
        for (int i = 0; i < MT; i++)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x_live[i] ? *(const bf16x8*)(xp[i] + (int64_t)s * 512) : x_zero, acc[i], 0, 0, 0);
    }

    // ---- reduce the K slices in wave order ----
#pragma unroll
    for (int i = 0; i < MT; i++) red[((tile * ksplit + kw) * MT + i) * 64 + lane] = acc[i];
    stamps.mark(3);                     // wave 0's stream done
    __syncthreads();
    stamps.mark(4);                     // every wave's stream done
    auto ksum = [&](int t, int i) {
        f32x4 s = red[((t * ksplit) * MT + i) * 64 + lane];
        for (int w = 1; w < ksplit; w++) s += red[((t * ksplit + w) * MT + i) * 64 + lane];
        return s;
    };
    if (EPI == EPI_SWIGLU) {
        for (int i = wave; i < MT; i += NTW * ksplit) {
            const int f = (nt0 >> 1) * 16 + 4 * fg;    // NTW == 2: tiles [gate 16 | up 16] of features 8*nt0..
            epilogue_swiglu4<OutT, MOE>(p, 16 * i + fr, f, ksum(0, i), ksum(NTW - 1, i),
                                        RSTD_PRE && i == i_pre ? rstd_pre : deferred_rstd(p, 16 * i + fr));
        }
    } else if (EPI == EPI_RESID && nslices > 1) {
        // split-K across workgroups: publish the partial tile (slice 0 carries the bias); the following norm
        // kernel adds the slices into x in slice order
        for (int e = wave; e < NTW * MT; e += NTW * ksplit) {
            const int t = e / MT, i = e - t * MT;
            const int m = 16 * i + fr, n = (nt0 + t) * 16 + 4 * fg;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = ksum(t, i);
            if (p.bias && slice == 0) v += *(const f32x4*)(p.bias + n);    // N % 4 == 0 for residual widths
            *(f32x4*)(p.sk_part + ((int64_t)slice * p.M + m) * p.N + n) = v;
        }
    } else if (EPI == EPI_RESID && p.rs_out) {
        // x += alpha * (acc + bias) in place, and the deferred-RMSNorm operand + x^2 partials for the next projection
        for (int e = wave; e < NTW * MT; e += NTW * ksplit) {
            const int t = e / MT, i = e - t * MT;
            const int m = 16 * (i + i_off) + fr, n = (nt0 + t) * 16 + 4 * fg;      // N % 16 == 0 (hidden width)
            f32x4 v = ksum(t, i);
            float ss = 0.f;
            if (m < p.M && n < p.N && (!HALF || (fg >> 1) == half)) {
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                float* x = (float*)p.C + (int64_t)m * p.ldc + n;
                const f32x4 o = (e == wave ? x_pre : *(f32x4*)x) + p.alpha * v;
                *(f32x4*)x = o;
                act_store4<bf16_t>(p.nrm_xn, m, n, p.N, o * *(const f32x4*)(p.nrm_w + n));
                ss = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
            }
            ss += __shfl_xor(ss, 16, 64);
            if constexpr (HALF) {     // the two lane rows (fg = 2 half, 2 half + 1) of this workgroup's 8 columns
                if (fg == 2 * half && (nt0 + t) * 16 < p.N) p.rs_out[(int64_t)m * (p.N >> 3) + 2 * (nt0 + t) + half] = ss;
                continue;
            }
            ss += __shfl_xor(ss, 32, 64);
            if (fg == 0 && (nt0 + t) * 16 < p.N) p.rs_out[(int64_t)m * (p.N >> 4) + (nt0 + t)] = ss;
        }
    } else {
        for (int e = wave; e < NTW * MT; e += NTW * ksplit) {
            const int t = e / MT, i = e - t * MT;
            epilogue4<EPI, OutT>(p, 16 * i + fr, (nt0 + t) * 16 + 4 * fg, ksum(t, i),
                                 EPI != EPI_STORE ? 1.0f : (RSTD_PRE && i == i_pre ? rstd_pre : deferred_rstd(p, 16 * i + fr)));
        }
    }
}

// ------------------------------------------------------------------------------------------
// MoE decode routing in ONE launch (moe.go:57-103): router logits of a 16-row activation tile (E <= 64 experts, all of K),
// softmax, top-k, renormalised weights as the dense gate matrix GemmArgs::moe_gate of the expert projections.  Before:
// the router as a skinny GEMM launch (4.6 us) + moe_gate_kernel (4.5 us).  grid = 16-row tiles of the batch, 1024
// threads: wave w takes expert tile w % ET over K slice w / ET (fragments straight from L2 / HBM, as the skinny kernel),
// the K slices are summed through LDS in wave order, then wave r routes row r of the tile with a lane per expert.
// p: A (fragment-major operand, possibly the deferred-norm raw form with rs_in), W (router, fragment-major), M, N = E, K.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void moe_router_gate_kernel(GemmArgs p, int top_k, float* __restrict__ gate, float* __restrict__ logits_out, int ld_logits) {
    __shared__ f32x4 red[16][64];
    __shared__ float lg[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int E = p.N, ET = (E + 15) >> 4, KW = 16 / ET;         // expert tiles (1, 2 or 4), K slices
    const int tile = wave % ET, kw = wave / ET;
    const int row0 = blockIdx.x * 16;
    const int nks = p.K >> 5, q = nks / KW, rr = nks - q * KW;
    const int my_steps = kw < KW ? q + (kw < rr ? 1 : 0) : 0;      // (ET = 3 leaves a wave idle)
    const int ks0 = kw * q + (kw < rr ? kw : rr);
    const bf16_t* wp = (const bf16_t*)p.W + (((int64_t)tile * nks + ks0) * 64 + lane) * 8;
    const bf16_t* xp = (const bf16_t*)p.A + (((int64_t)blockIdx.x * nks + ks0) * 64 + lane) * 8;
    const bool x_live = row0 + fr < p.M;
    const bf16x8 zero = {};
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= my_steps; s += 4) {
        bf16x8 w[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            w[u] = *(const bf16x8*)(wp + (int64_t)(s + u) * 512);
            x[u] = x_live ? *(const bf16x8*)(xp + (int64_t)(s + u) * 512) : zero;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u], x[u], acc, 0, 0, 0);
    }
    for (; s < my_steps; s++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(wp + (int64_t)s * 512), x_live ? *(const bf16x8*)(xp + (int64_t)s * 512) : zero, acc, 0, 0, 0);
    // deferred RMSNorm: wave r owns row r of the tile — its x^2 partials, a lane per 16-byte chunk
    float ssq = 0.f;
    if (p.rs_in && row0 + wave < p.M) {
        const f32x4* src = (const f32x4*)(p.rs_in + (int64_t)(row0 + wave) * p.rs_tiles);
        f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = lane; c < (p.rs_tiles >> 2); c += 64) a4 += src[c];
        ssq = (a4[0] + a4[1]) + (a4[2] + a4[3]);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ssq += __shfl_xor(ssq, o, 64);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave < ET) {        // acc of lane (fr, fg): row fr, experts 16 tile + 4 fg .. + 3
        f32x4 v = red[wave][lane];
        for (int k = 1; k < KW; k++) v += red[k * ET + wave][lane];
#pragma unroll
        for (int r = 0; r < 4; r++) lg[fr][16 * wave + 4 * fg + r] = v[r];
    }
    __syncthreads();
    const int m = row0 + wave;
    if (m >= p.M) return;
    const float rstd = p.rs_in ? 1.0f / sqrtf(ssq * p.rs_inv_h + p.rs_eps) : 1.0f;
    const float v = (lane < E) ? lg[wave][lane] * rstd : -INFINITY;
    if (logits_out && lane < E) logits_out[(int64_t)m * ld_logits + lane] = v;
    float mx = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = (lane < E) ? expf(v - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float prob = (lane < E) ? e / sum : -1.f;
    int rank = 0;                                   // larger first, ties to the lower index (moe.go:75-92)
    const int pbits = __builtin_bit_cast(int, prob);
#pragma unroll 8
    for (int j = 0; j < E; j++) {                   // (lanes >= E hold -1: they never outrank an expert)
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, j));
        rank += (pj > prob || (pj == prob && j < lane)) ? 1 : 0;
    }
    float wsum = 0.f;
    for (int k = 0; k < top_k; k++) {               // moe.go:89-92: fp32 sum in rank order
        const int src = __ffsll((unsigned long long)__ballot(rank == k)) - 1;
        wsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, src));
    }
    if (lane < E) gate[(int64_t)m * E + lane] = rank < top_k ? prob / wsum : 0.f;      // moe.go:103
}

// ------------------------------------------------------------------------------------------
// fp32 kernel (parity mode): 64x64 tile, BK=16, 256 threads, 4x4 outputs per thread, k ascending.
// EPI_SWIGLU is not instantiated: the f32 path keeps W1 un-interleaved and applies the
// activation with swiglu_kernel.
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    __shared__ float As[16][64 + 4];
    __shared__ float Ws[16][64 + 4];
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + 63) / 64;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;
    const int tx = tid & 15, ty = tid >> 4;   // thread -> n = n0 + 4*tx.., m = m0 + 4*ty..
    int row_base = 0;
    if (p.seg) {
        row_base = p.seg[0];
        p.M = p.seg[1] - row_base;
        p.c_row0 = row_base;
    }
    if (m0 >= p.M) return;
    // (the f32 parity path keeps one launch per expert through `seg`; tile_map is bf16-only)

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;

    // each thread stages one float4 of A and one of W per k-step: row = tid/4, k = (tid%4)*4
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    int am = m0 + lrow;
    if (am > p.M - 1) am = p.M - 1;
    am += row_base;
    if (p.a_rows) am = p.a_rows[am];
    const float* ap = (const float*)p.A + (int64_t)am * p.lda + lk;
    const float* wp = (const float*)p.W + (int64_t)(n0 + lrow) * p.K + lk;   // W has N_pad (x128) rows

    for (int k0 = 0; k0 < p.K; k0 += 16) {
        const f32x4 av = *(const f32x4*)(ap + k0);
        const f32x4 wv = *(const f32x4*)(wp + k0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            As[lk + r][lrow] = av[r];
            Ws[lk + r][lrow] = wv[r];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = Ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        epilogue4<EPI, OutT>(p, m0 + ty * 4 + i, n0 + tx * 4, v);
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// bf16 skinny kernel, wide-N form (decode projections with >= 256 groups of 64 weight rows: FFN up, LM head).
// A wave owns NTB 16-row weight tiles over its K slice, so every activation fragment it fetches from L2 feeds
// NTB MFMAs instead of one: L2->CU traffic per weight byte drops from 1 + MT to 1 + MT/NTB (the activation
// fetches were costing ~15 % of the streaming rate, r01 probe).  8 waves per workgroup (256-VGPR budget): two
// register sets of U k-steps x (NTB weight + MT activation) fragments — for K = 2048 a wave's whole slice is in
// flight at once.  Same deterministic in-LDS K reduction and fused epilogues as the narrow form.
// ------------------------------------------------------------------------------------------
template <int MT, int NTB, int U, int EPI, typename OutT, bool PASSES = false, bool MOE = false>
__global__ __launch_bounds__(512) void gemm_skinny_wide_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = (f32x4*)smem;                       // [ksplit][NTB][MT][64]
    const int swg = blockIdx.x;
    NvlStamps stamps(p.stamps, swg);
    const int tid = threadIdx.x, lane = tid & 63, kw = tid >> 6;
    const int ksplit = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bx = blockIdx.x;
    if constexpr (PASSES) skinny_pass_remap(p, bx);     // (its own instantiation: rewriting the arguments costs the
                                                        // single-group kernels a few scratch dwords otherwise)
    const int nt0 = bx * NTB;
    if constexpr (MOE) { if (!moe_expert_live(p, (nt0 * 16) / (2 * p.moe_I))) return; }      // MoE decode: untouched expert (uniform)
    const int nks = p.K >> 5, q = nks / ksplit, rr = nks - q * ksplit;
    const int my_steps = q + (kw < rr ? 1 : 0);
    const int ks0 = kw * q + (kw < rr ? kw : rr);
    const int64_t tile_stride = (int64_t)nks * 512;  // elements between consecutive 16-row weight tiles
    // deferred RMSNorm: the scale of the rows this wave's epilogue elements belong to, fetched under the weight stream
    // (ahead of the first weight block: see the narrow kernel)
    const int i_pre = kw % MT;
    const float rstd_pre = (EPI == EPI_SWIGLU || EPI == EPI_STORE) ? deferred_rstd(p, 16 * i_pre + fr) : 1.0f;
    const bf16_t* wp = (const bf16_t*)p.W + (((int64_t)nt0 * nks + ks0) * 64 + lane) * 8;
    const bf16_t* xp = (const bf16_t*)p.A + ((int64_t)ks0 * 64 + lane) * 8;
    f32x4 acc[NTB][MT];
#pragma unroll
    for (int t = 0; t < NTB; t++)
#pragma unroll
        for (int i = 0; i < MT; i++) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (no row mask here, unlike the narrow kernel: this form runs at 17..64 rows, where at most one tile is ragged, and its
    // 256-register budget is full — the predicated loads cost 6 more spilled registers and 2 us per FFN-up launch at B = 32)
    const int nblk = my_steps / U;
    bf16x8 wA[U][NTB], xA[U][MT], wB[U][NTB], xB[U][MT];
    auto load_blk = [&](bf16x8 (&w)[U][NTB], bf16x8 (&x)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int t = 0; t < NTB; t++)
                w[u][t] = weight_load<PASSES>((const bf16x8*)(wp + t * tile_stride + (int64_t)(b * U + u) * 512));
#pragma unroll
            for (int i = 0; i < MT; i++) x[u][i] = *(const bf16x8*)(xp + i * tile_stride + (int64_t)(b * U + u) * 512);
        }
    };
    auto comp_blk = [&](bf16x8 (&w)[U][NTB], bf16x8 (&x)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < NTB; t++)
#pragma unroll
                for (int i = 0; i < MT; i++)
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][t], x[u][i], acc[t][i], 0, 0, 0);
    };
    if (nblk > 0) load_blk(wA, xA, 0);
    stamps.mark(1);
    int b = 0;
    for (; b + 2 <= nblk; b += 2) {
        load_blk(wB, xB, b + 1);
        comp_blk(wA, xA);
        if (b == 0) stamps.mark_used(2, acc[0][0]);
        if (b + 2 < nblk) load_blk(wA, xA, b + 2);
        comp_blk(wB, xB);
    }
    if (b < nblk) comp_blk(wA, xA);
    for (int s = nblk * U; s < my_steps; s++) {      // ragged tail
        bf16x8 xs[MT];
#pragma unroll
        for (int i = 0; i < MT; i++) xs[i] = *(const bf16x8*)(xp + i * tile_stride + (int64_t)s * 512);
#pragma unroll
        for (int t = 0; t < NTB; t++) {
            const bf16x8 w = weight_load<PASSES>((const bf16x8*)(wp + t * tile_stride + (int64_t)s * 512));
#pragma unroll
            for (int i = 0; i < MT; i++) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xs[i], acc[t][i], 0, 0, 0);
        }
    }

#pragma unroll
    for (int t = 0; t < NTB; t++)
#pragma unroll
        for (int i = 0; i < MT; i++) red[((kw * NTB + t) * MT + i) * 64 + lane] = acc[t][i];
    stamps.mark(3);
    __syncthreads();
    stamps.mark(4);
    auto ksum = [&](int t, int i) {
        f32x4 s = red[(t * MT + i) * 64 + lane];
        for (int w = 1; w < ksplit; w++) s += red[((w * NTB + t) * MT + i) * 64 + lane];
        return s;
    };
    if (EPI == EPI_SWIGLU) {
        // tiles come as [gate16 | up16] pairs of the same 16 features
        for (int e = kw; e < (NTB / 2) * MT; e += ksplit) {
            const int j = e / MT, i = e - j * MT;
            const int f = ((nt0 >> 1) + j) * 16 + 4 * fg;
            epilogue_swiglu4<OutT, MOE>(p, 16 * i + fr, f, ksum(2 * j, i), ksum(2 * j + 1, i),
                                        i == i_pre ? rstd_pre : deferred_rstd(p, 16 * i + fr));
        }
    } else {
        for (int e = kw; e < NTB * MT; e += ksplit) {
            const int t = e / MT, i = e - t * MT;
            epilogue4<EPI, OutT>(p, 16 * i + fr, (nt0 + t) * 16 + 4 * fg, ksum(t, i),
                                 EPI != EPI_STORE ? 1.0f : (i == i_pre ? rstd_pre : deferred_rstd(p, 16 * i + fr)));
        }
    }
}

static int g_chunk_max_m = 512;    // largest M that uses 64-row passes of the decode form at all (nvl_set_tuning key 11; 64 = off)
static int g_chunk_all_m = 64;     // up to here every projection does; above, only those whose 128x128 tile grid would
                                   // have fewer than g_chunk_min_tiles workgroups (keys 14, 12).  Measured: the mix wins
                                   // from 96 rows on (profiles/r01g_large_decode_batches.txt)
static int g_pass_interleave = 1;   // 64-row groups of one projection in ONE launch (key 13; 0 = one launch per group)
static int g_chunk_min_tiles = 160;
static int g_msplit_ks = 8;         // K split of the deferred-norm residual projections (nvl_set_tuning key 6)
static int g_narrow_waves = 2048;   // waves per narrow-form launch (nvl_set_tuning key 5).  Round-2 sweep on the final kernels
                                    // (profiles/r02_narrow_waves_sweep.txt): 2048 vs 1024: B=1 +5.4 %, B=8 +5.0 %, B=16 +2.4 %,
                                    // B=32 flat; 4096 = 2048.  (Round 1, before the activation-row mask: 1024 beat 4096 by 3-8 %.)
static int g_wide_ksplit = 0;   // experiment (nvl_set_tuning key 4): K split of the wide form when groups <= 512
static int g_x_mask = 1;            // nvl_set_tuning key 24: decode kernels skip the activation fetch of padded rows (M % 16 != 0)
static int g_force_ntw = 0, g_force_ksplit = 0;   // tuning overrides (nvl_bench_gemm only)
static int g_force_tile = 0;                      // 0 automatic, 1: 128x128x2st, 2: 256x128x3st, 3: 256x256x2st, 5: ping-pong 256x256

template <int EPI, typename OutT>
static inline bool launch_gemm_skinny_bf16(hipStream_t st, const GemmArgs& a);
constexpr int DEFER_MAX_M = 128;    // deferred RMSNorm (decode): rows of the x^2 partial buffer.  Above 64 rows the producer
                                    // runs as 64-row groups like every narrow projection (one workgroup per 16-row tile
                                    // was measured at M = 128: -13 %); above 128 rows FFN-up belongs on the tile kernels
static inline int skinny_rows(const GemmArgs& a) { return a.m_passes > 1 && a.M > 64 ? 64 : a.M; }   // rows a workgroup covers
// the groups of 64 rows as separate launches (weight-block count not a multiple of 8, or key 13 = 0)
template <int EPI, typename OutT>
static inline bool launch_skinny_passes_serial(hipStream_t st, const GemmArgs& a) {
    for (int r0 = 0; r0 < a.M; r0 += 64) {
        GemmArgs c = a;
        c.m_passes = 0;
        c.A = (const bf16_t*)a.A + (int64_t)r0 * a.K;
        c.M = a.M - r0 < 64 ? a.M - r0 : 64;
        c.c_row0 = a.c_row0 + r0;
        if (a.rs_in) c.rs_in = a.rs_in + (int64_t)r0 * a.rs_tiles;
        if (a.rs_out) {
            c.C = (float*)a.C + (int64_t)r0 * a.ldc; c.nrm_xn = a.nrm_xn + (int64_t)r0 * a.N;
            c.rs_out = a.rs_out + (int64_t)r0 * (a.N >> 4);
        }
        if (!launch_gemm_skinny_bf16<EPI, OutT>(st, c)) return false;
    }
    return true;
}

// skinny dispatch: M <= 64, no gather/segments.  Returns false when the shape is not eligible.
template <int NTW, int EPI, typename OutT>
static inline bool launch_gemm_skinny_ntw(hipStream_t st, const GemmArgs& a) {
    const int MT = skinny_rows(a) <= 16 ? 1 : (skinny_rows(a) <= 32 ? 2 : 4);
    if (EPI == EPI_RESID && a.m_split && a.rs_out && NTW == 1 && a.m_passes <= 1 && a.grid_y != 1) {   // (m_split: grid_y unused, keep it defined)
        GemmArgs c = a; c.grid_y = 1; return launch_gemm_skinny_ntw<NTW, EPI, OutT>(st, c);
    }
    const int U = MT <= 2 ? 4 : 2;                       // register budget: (1+MT)*U*2 fragments
    const int nblocks = cdiv(cdiv(a.N, 16), NTW);        // weight rows are padded to 128: all tiles exist
    const int KS = (EPI == EPI_RESID && a.sk_part && a.sk_slices > 1) ? a.sk_slices : 1;
    if (EPI == EPI_RESID && a.m_split && a.rs_out && NTW == 1 && a.m_passes <= 1) {
        // deferred-norm residual projection: one workgroup per (weight tile, 16-row activation tile), all of K each
        int ks = g_msplit_ks;
        while (ks > 1 && (a.K >> 5) / ks < 4) ks >>= 1;
        if (a.K % 32 != 0) return false;
        if constexpr (EPI == EPI_RESID) {
            if (a.moe_gate) {     // MoE down as the deferred-norm producer: all experts in one workgroup's K range, 16 waves
                ks = 16;
                const int nks = a.K >> 5, spe = a.moe_I >> 5;
                if (nks % (ks * 4) != 0 || spe % 4 != 0 || nks / ks > 4 * spe || nks / ks / 4 > 32) return false;
                if (a.rs_half) {
                    if (a.M > 16) return false;
                    hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, NTW, 4, EPI, OutT, false, true, true>), dim3(2 * nblocks, 1), dim3(NTW * ks * 64),
                                       (size_t)NTW * ks * 64 * 16, st, a);
                } else {
                    hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, NTW, 4, EPI, OutT, false, true>), dim3(nblocks, cdiv(a.M, 16)), dim3(NTW * ks * 64),
                                       (size_t)NTW * ks * 64 * 16, st, a);
                }
                return true;
            }
            if (a.rs_half) {      // <= 16 rows: 8 weight rows per workgroup, twice the workgroups (HALF above)
                if (a.M > 16) return false;
                hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, NTW, 4, EPI, OutT, false, false, true>), dim3(2 * nblocks, 1), dim3(NTW * ks * 64),
                                   (size_t)NTW * ks * 64 * 16, st, a);
                return true;
            }
        }
        hipLaunchKernelGGL((gemm_skinny_bf16_kernel<1, NTW, 4, EPI, OutT>), dim3(nblocks, cdiv(a.M, 16)), dim3(NTW * ks * 64),
                           (size_t)NTW * ks * 64 * 16, st, a);
        return true;
    }
    // waves per launch: enough to cover HBM latency on every CU (~2-4 thousand), not more — extra K slices only
    // shorten each wave's stream below the depth of its two-block pipeline (r01 sweeps: LM head, W1)
    int ksplit = 16;
    if (g_force_ksplit) ksplit = g_force_ksplit;
    else {
        // (the MoE gate/up projection counts every expert's workgroups, but those of experts no token picked exit at once:
        //  twice the budget — Granite-1B B=8 decode 6.50 K -> 6.60 K tok/s, profiles/r03_moe_decode_experiments.txt)
        const int64_t budget = (int64_t)g_narrow_waves * ((a.moe_gate && EPI == EPI_SWIGLU) ? 2 : 1);
        while (ksplit > 1 && (int64_t)nblocks * KS * NTW * ksplit > budget) ksplit >>= 1;
    }
    while (ksplit > 1 && ((a.K >> 5) / (ksplit * KS) < U || ksplit * NTW > 16)) ksplit >>= 1;   // >= one block of U k-steps per wave
    if (a.moe_gate && EPI == EPI_RESID) {      // MoE down: every wave's K range is exactly one expert (the skip test needs it)
        ksplit = a.moe_E / KS;
        if (ksplit < 1 || ksplit * KS != a.moe_E || ksplit > 16 || (a.moe_I >> 5) < 1) return false;
    }
    if (a.K % 32 != 0) return false;
    const size_t lds = (size_t)NTW * ksplit * MT * 64 * 16;
    if (a.m_passes > 1 && (nblocks % 8 != 0 || KS != 1)) return launch_skinny_passes_serial<EPI, OutT>(st, a);
    dim3 grid(nblocks * (a.m_passes > 1 ? a.m_passes : 1), KS), block(NTW * ksplit * 64);
    if (a.grid_y != KS) { GemmArgs c = a; c.grid_y = KS; return launch_gemm_skinny_ntw<NTW, EPI, OutT>(st, c); }
#define NVL_SK(MTv, Uv) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MTv, NTW, Uv, EPI, OutT>), grid, block, lds, st, a)
#define NVL_SKM(MTv, Uv) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MTv, NTW, Uv, EPI, OutT, false, true>), grid, block, lds, st, a)
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_RESID) {
        if (a.moe_gate) {                                 // MoE decode instantiation (expert skip + gate weight)
            if (a.m_passes > 1) return false;
            if (MT == 1) NVL_SKM(1, 4); else if (MT == 2) NVL_SKM(2, 4); else NVL_SKM(4, 2);
            return true;
        }
    }
    if (a.m_passes > 1) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<4, NTW, 2, EPI, OutT, true>), grid, block, lds, st, a);
    else if (MT == 1) NVL_SK(1, 4); else if (MT == 2) NVL_SK(2, 4); else NVL_SK(4, 2);
#undef NVL_SK
#undef NVL_SKM
    return true;
}
// Weight tiles per wave of the wide form (nvl_set_tuning key 36: 0 automatic, 2 / 4 forced).  Four tiles per wave quarter the
// activation fetches per weight byte, but an FFN-up launch is then only 256 workgroups of 2 waves — 64 KiB of weights in
// flight per CU.  Two tiles per wave (512 workgroups) for batches of 17..32 rows: Llama-3.2-1B B=32 decode 33.7 K -> 34.5 K
// tok/s (+2.4 %), B=20 +3.1 %, Falcon-7B B=32 +4.7 %, Llama-3-8B / Granite B=32 +0.8 %; the LM head (thousands of groups
// either way) is indifferent, and 64-row launches (MT = 4) lose 2 % (profiles/r03_wide_tiles_per_wave.txt).
static int g_wide_ntb = 0;
template <int EPI, typename OutT, int NTB = 4>
static inline bool launch_gemm_skinny_wide(hipStream_t st, const GemmArgs& a) {
    if constexpr (NTB == 4) {
        if (g_wide_ntb == 2 || (g_wide_ntb == 0 && skinny_rows(a) <= 32 && a.N < 65536)) return launch_gemm_skinny_wide<EPI, OutT, 2>(st, a);
    }
    if (a.K % 32 != 0 || a.sk_part) return false;
    const int groups = cdiv(cdiv(a.N, 16), NTB);     // weight rows are padded to 256: every tile of a group exists
    const int MT = skinny_rows(a) <= 16 ? 1 : (skinny_rows(a) <= 32 ? 2 : 4);
    const int U = MT <= 2 ? 4 : 2;
    // ~512 waves per launch (256-VGPR waves, each with up to 32 KiB of weights in flight) stream best from HBM: longer
    // per-wave K slices beat more waves (in the model, weights cold: FFN-up ksplit 2 > 4 > 8 > 1; LM head, 8B FFN-up 2)
    int ksplit = 8;
    if (g_force_ksplit) ksplit = g_force_ksplit <= 8 ? g_force_ksplit : 8;
    else if (g_wide_ksplit && groups <= 512) ksplit = g_wide_ksplit;
    else { while (ksplit > 2 && groups * ksplit > 512) ksplit >>= 1; if (groups >= 1024) ksplit = 1; }   // LM head: one wave per group (cold sweep: 96 vs 104 us)
    while (ksplit > 1 && (a.K >> 5) / ksplit < U) ksplit >>= 1;
    const size_t lds = (size_t)ksplit * NTB * MT * 64 * 16;
    if (a.m_passes > 1 && groups % 8 != 0) return launch_skinny_passes_serial<EPI, OutT>(st, a);
    const int wgs = groups * (a.m_passes > 1 ? a.m_passes : 1);
#define NVL_SKW(MTv, Uv, Pv)                                                                                             \
    do {                                                                                                               \
        NVL_LDS_ATTR((gemm_skinny_wide_bf16_kernel<MTv, NTB, Uv, EPI, OutT, Pv>), 8 * NTB * MTv * 64 * 16);            \
        hipLaunchKernelGGL((gemm_skinny_wide_bf16_kernel<MTv, NTB, Uv, EPI, OutT, Pv>), dim3(wgs), dim3(ksplit * 64),      \
                           lds, st, a);                                                                                \
    } while (0)
#define NVL_SKWM(MTv, Uv)                                                                                                \
    do {                                                                                                               \
        NVL_LDS_ATTR((gemm_skinny_wide_bf16_kernel<MTv, NTB, Uv, EPI, OutT, false, true>), 8 * NTB * MTv * 64 * 16);         \
        hipLaunchKernelGGL((gemm_skinny_wide_bf16_kernel<MTv, NTB, Uv, EPI, OutT, false, true>), dim3(wgs), dim3(ksplit * 64), \
                           lds, st, a);                                                                                \
    } while (0)
    if constexpr (EPI == EPI_SWIGLU) {
        if (a.moe_gate) {
            if (a.m_passes > 1) return false;
            if (MT == 1) NVL_SKWM(1, 4); else if (MT == 2) NVL_SKWM(2, 4); else NVL_SKWM(4, 2);
            return true;
        }
    }
    if (a.m_passes > 1) NVL_SKW(4, 2, true); else if (MT == 1) NVL_SKW(1, 4, false); else if (MT == 2) NVL_SKW(2, 4, false); else NVL_SKW(4, 2, false);
#undef NVL_SKW
#undef NVL_SKWM
    return true;
}
template <int EPI, typename OutT>
static inline bool launch_gemm_skinny_bf16(hipStream_t st, const GemmArgs& a0) {
    if (EPI == EPI_QKV) return false;               // prefill-only epilogue (a decode workgroup owns 16 columns, not a head)
    if (skinny_rows(a0) > 64 || a0.a_rows || a0.seg || a0.tile_map) return false;
    GemmArgs a = a0;
    a.x_mask = (g_x_mask && a0.M % 16 != 0) ? 1 : 0;     // padded rows of the last 16-row activation tile are not fetched
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_STORE || EPI == EPI_GELU) {
        // wide-N form once its 64-row groups fill the chip (g_force_ntw: 8 forces it, 1/2/4 force the narrow form)
        const bool wide_ok = skinny_rows(a) > 16 && cdiv(a.N, 64) >= 256 && !a.sk_part;   // M <= 16: the narrow form streams as fast
        if (g_force_ntw == 8 || (g_force_ntw == 0 && wide_ok)) { if (launch_gemm_skinny_wide<EPI, OutT>(st, a)) return true; }
    }
    if (EPI == EPI_SWIGLU) return launch_gemm_skinny_ntw<2, EPI, OutT>(st, a);
    if (g_force_ntw == 2) return launch_gemm_skinny_ntw<2, EPI, OutT>(st, a);
    if (g_force_ntw == 4) return launch_gemm_skinny_ntw<4, EPI, OutT>(st, a);
    return launch_gemm_skinny_ntw<1, EPI, OutT>(st, a);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, typename OutT>
static inline void launch_gemm_tile(hipStream_t st, const GemmArgs& a) {
    constexpr int lds = gemm_lds_bytes<BM, BN, STAGES>();
    NVL_LDS_ATTR((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, OutT>), lds);
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAGES, EPI, OutT>),
                       dim3(tiles, (EPI == EPI_RESID && a.sk_part) ? a.sk_slices : 1), dim3(WAVES_M * WAVES_N * 64), lds, st, a);
}

// how well `tiles` workgroups (one per CU at a time for the big tiles) fill 256 CUs
static inline double cu_fill(int tiles, int per_cu) {
    const int slots = 256 * per_cu;
    return (double)tiles / (double)(cdiv(tiles, slots) * slots);
}

template <int EPI, typename OutT>
static inline void launch_gemm_grouped(hipStream_t st, const GemmArgs& a, int max_mtiles) {
    if (a.grp_bm == 512) {      // prefill MoE, the default: 256-row m-tiles on the ping-pong kernel
        NVL_LDS_ATTR((gemm_bf16_pp_kernel<EPI, OutT, 4, 0>), 4 * 32 * 1024);
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OutT, 4, 0>), dim3(cdiv(a.N, 256) * max_mtiles), dim3(512), 4 * 32 * 1024, st, a);
        return;
    }
    if (a.grp_bm == 256) {      // prefill MoE with long expert segments: 256-row m-tiles, three stages
        constexpr int lds3 = gemm_lds_bytes<256, 128, 3>();
        NVL_LDS_ATTR((gemm_bf16_kernel<256, 128, 4, 2, 3, EPI, OutT>), lds3);
        hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 3, EPI, OutT>), dim3(cdiv(a.N, 128) * max_mtiles), dim3(512),
                           lds3, st, a);
        return;
    }
    if (a.grp_deep) {           // decode MoE: a handful of rows per expert, one workgroup per CU streaming 128 weight rows —
        constexpr int lds4 = gemm_lds_bytes<128, 128, 4>();   // latency-bound with two stages (16 K tiles, one round trip each)
        NVL_LDS_ATTR((gemm_bf16_kernel<128, 128, 2, 2, 4, EPI, OutT>), lds4);
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 4, EPI, OutT>), dim3(cdiv(a.N, 128) * max_mtiles), dim3(256),
                           lds4, st, a);
        return;
    }
    constexpr int lds = gemm_lds_bytes<128, 128, 2>();
    NVL_LDS_ATTR((gemm_bf16_kernel<128, 128, 2, 2, 2, EPI, OutT>), lds);
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2, EPI, OutT>), dim3(cdiv(a.N, 128) * max_mtiles), dim3(256),
                       lds, st, a);
}

template <int EPI, typename OutT>
static inline void launch_gemm_bf16(hipStream_t st, const GemmArgs& a) {
    if (a.tile_map) { launch_gemm_grouped<EPI, OutT>(st, a, a.M); return; }   // a.M carries the m-tile bound
    if (launch_gemm_skinny_bf16<EPI, OutT>(st, a)) return;
    if constexpr (EPI != EPI_QKV) {
        // 64 < M <= g_chunk_max_m (large decode batches): the tile kernels would launch only N/128 * M/128 workgroups, far
        // too few to stream the weights at HBM rate.  Run the decode form over ceil(M/64) groups of 64 activation rows
        // in one launch instead (GemmArgs::m_passes): the groups of a weight block run side by side on one XCD, so the
        // weights cross HBM once.  Only the narrow projections (QKV, O, FFN-down) do; FFN-up and the LM head have enough
        // tiles (g_chunk_all_m: below it every projection would).
        const bool few_tiles = cdiv(a.M, 128) * cdiv(a.N, 128) < g_chunk_min_tiles;
        if (a.M > 64 && a.M <= g_chunk_max_m && (a.M <= g_chunk_all_m || few_tiles || a.rs_in || a.rs_out) && !a.a_rows && !a.seg && !a.sk_part && a.K % 32 == 0 &&
            g_force_tile == 0) {
            GemmArgs c = a;
            c.m_passes = cdiv(a.M, 64);
            const bool ok = g_pass_interleave ? launch_gemm_skinny_bf16<EPI, OutT>(st, c)
                                              : launch_skinny_passes_serial<EPI, OutT>(st, c);
            if (ok) return;
        }
    }
    if (a.rs_in || a.rs_out) throw std::runtime_error("gemm: a deferred-RMSNorm projection did not fit the decode kernels");
    if constexpr (EPI == EPI_QKV) {
        if (a.qkv.hd == 128) {     // a wave must own a whole 128-column head: 256x256 tile as 4x2 waves of 64x128
            const int t3 = cdiv(a.M, 256) * cdiv(a.N, 256);
            if (g_force_tile == 0 && t3 >= 192 && cu_fill(t3, 1) >= 0.74 && a.K >= 160 && !a.seg && !a.a_rows) {
                NVL_LDS_ATTR((gemm_bf16_pp_kernel<EPI, OutT, 4, 0, 2>), 4 * 32 * 1024);     // the ping-pong form, wave tile 64x128
                hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OutT, 4, 0, 2>), dim3(t3), dim3(512), 4 * 32 * 1024, st, a);
                return;
            }
            launch_gemm_tile<256, 256, 4, 2, 2, EPI, OutT>(st, a);
            return;
        }
    }
    int tile = g_force_tile;
    if (tile == 0) {
        tile = 1;
        if (!a.seg && a.K >= 128) {
            // r01 sweeps (profiles/r01e_gemm_sweep_mid.txt): the ping-pong 256x256 form wins as soon as its tiles cover
            // 3/4 of the CUs; below that the 128x128 form (2 workgroups per CU) does; the lock-step 256-wide forms and a
            // ping-pong 256x128 variant (wave tile 64x64: 8 fragment reads per 16 MFMAs) never do
            const int t3 = cdiv(a.M, 256) * cdiv(a.N, 256);
            if (t3 >= 192 && cu_fill(t3, 1) >= 0.74) tile = (!a.seg && !a.a_rows && a.K >= 160) ? 5 : 3;   // 5: ping-pong form
        }
    }
#define NVL_PP(TILE, ST, PR)                                                                                          \
    if (tile == TILE && !a.seg && !a.a_rows) {                                                                         \
        NVL_LDS_ATTR((gemm_bf16_pp_kernel<EPI, OutT, ST, PR>), ST * 32 * 1024);                                        \
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, OutT, ST, PR>), dim3(cdiv(a.M, 256) * cdiv(a.N, 256)), dim3(512), \
                           ST * 32 * 1024, st, a);                                                                     \
        return;                                                                                                        \
    }
    NVL_PP(5, 4, 0)      // the default large-M form: 4-slot ring (5 slots: -3 %, 3 slots: -3 %, r01 sweep)
#undef NVL_PP
    if (tile == 3) launch_gemm_tile<256, 256, 2, 4, 2, EPI, OutT>(st, a);
    else if (tile == 2) launch_gemm_tile<256, 128, 4, 2, 3, EPI, OutT>(st, a);
    else launch_gemm_tile<128, 128, 2, 2, 2, EPI, OutT>(st, a);
}

template <int EPI, typename OutT>
static inline void launch_gemm_f32(hipStream_t st, const GemmArgs& a) {
    const int tiles = cdiv(a.M, 64) * cdiv(a.N, 64);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, OutT>), dim3(tiles), dim3(256), 0, st, a);
}

}  // namespace nvl
