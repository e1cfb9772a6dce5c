// attn.h — causal attention over the per-sequence KV slabs.  Replaces, for all three
// attention types, the reference's split/repeat/score/softmax/apply/merge loops:
//   GQA  attention.go:217-274 (+ :300-470)      group = nH/nKV
//   MHA  attention.go:34-64, :127-191           group = 1
//   MQA  mqa.go:47-99, :163-267                 group = nH (all heads share one K/V tile)
// No repeatKVHeads copy, no Concatenate re-copy, no [nH,S,T] score tensor.
//
// bf16 kernel (product path): flash-style, MFMA 16x16x32 bf16, fp32 online softmax.
// One workgroup = 4 waves = 64 "query rows" of one (sequence, kv head); a query row is a
// (position, head-in-group) pair, position-major, so every head of a GQA/MQA group reuses the
// same K/V tile from LDS.  Per wave 16 rows.  The products are issued transposed:
//     S^T[key, q] = K[key, :] · Q[q, :]^T          (A = K tile rows,   B = Q rows, from registers)
//     O^T[d,  q] += V^T[d, key] · P^T[key, q]      (A = V^T tile rows, B = P^T straight from the
//                                                   S^T accumulators — no LDS, no shuffles)
// so the softmax row (fixed q) lives on one lane column: the max/sum need two xor-shuffles, and the
// O rescale is lane-local.  The V cache is kept TRANSPOSED in HBM ([hd][T]) so the V^T tile is a
// coalesced load and its fragments are 8-byte LDS reads.
//
// f32 kernel (parity mode): one workgroup per (query position, head); scores in LDS; plain fp32.
#pragma once
#include "common.h"

namespace nvl {

struct AttnArgs {
    const void* q;        // [tokens][q_stride]; head h at column h*HD (bf16 or f32)
    int q_stride;
    void* out;            // [tokens][out_stride] (bf16: fragment-major, fp32: row-major); head h at column h*HD
    int out_stride;
    const void* kcache;   // layer base; (block, kvh) at block*slot_stride + kvh*Tmax*HD; [Tmax][HD]
    const void* vcache;   // bf16: V^T [HD][Tmax]; f32: [Tmax][HD]
    int64_t slot_stride;  // elements between consecutive KV blocks
    int Tmax;             // tokens per KV block (a multiple of 64): the whole slot in slab mode, 256 in paged mode
    const int32_t* seq_tok_start;
    const int32_t* seq_len;
    const int32_t* seq_pos;
    const int32_t* blk_table;   // [sequence in the batch][tbl_stride] block ids (slab mode: one entry, the slot)
    int tbl_stride;
    int bs_shift;               // log2(Tmax) when it is a power of two, else -1 (then the kernels divide)
    int nH, nKV, group;
    float scale;
    // fused decode (attn_decode_bf16_kernel<.., FUSED>): q/k/v of the NEW token come straight from the fp32 output of
    // the QKV projection ([tokens][qkv_stride] = [Q heads | K heads | V heads]); RoPE (rope.go:153-205) is applied here
    // and the new K/V row is appended to the slabs by this kernel (replaces rope_kv_kernel + the q round trip)
    const float* qkv;
    int qkv_stride;
    const float* cos_t;   // [max_seq][HD] or NULL
    const float* sin_t;
};

// index of the KV block that holds key `key`
__device__ __forceinline__ int kv_block_index(const AttnArgs& p, int key) { return p.bs_shift >= 0 ? (key >> p.bs_shift) : key / p.Tmax; }

template <int HD, int TQ>
__global__ __launch_bounds__(256) void attn_bf16_kernel(AttnArgs p) {
    // TQ = 16-row query sub-tiles per wave: every K / V^T fragment read from LDS feeds TQ MFMAs (the loop is LDS-bound at
    // TQ = 1: 1 KiB of fragment reads per MFMA).  A workgroup covers 64 * TQ query rows.
    constexpr int KROW = HD + 8;       // bf16 elements per K tile row (padded: 144 B for HD=64)
    constexpr int VROW = 64 + 8;       // bf16 elements per V^T tile row
    constexpr int KS = HD / 32;        // k-steps of the QK^T product
    constexpr int DT = HD / 16;        // 16-row d tiles of O^T
    constexpr int QROWS = 64 * TQ;     // query rows per workgroup
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KROW];
    __shared__ __attribute__((aligned(16))) bf16_t Vts[HD * VROW];

    const int seq = blockIdx.z, kvh = blockIdx.y, qt = blockIdx.x;
    const int S = p.seq_len[seq];
    const int R = S * p.group;                 // query rows of this (seq, kv head)
    if (qt * QROWS >= R) return;
    const int tok0 = p.seq_tok_start[seq], pos0 = p.seq_pos[seq];
    const int32_t* tbl = p.blk_table + (int64_t)seq * p.tbl_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane & 15, fg = lane >> 4;

    // ---- this lane's query rows (one per sub-tile) ----
    bool row_ok[TQ];
    int s_idx[TQ], head[TQ], limit[TQ];
    bf16x8 qf[TQ][KS];
#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        int row = qt * QROWS + (wave * TQ + qi) * 16 + fq;
        row_ok[qi] = row < R;
        if (!row_ok[qi]) row = qt * QROWS;     // clamp to a valid row; result discarded
        s_idx[qi] = row / p.group;
        head[qi] = kvh * p.group + (row - s_idx[qi] * p.group);
        limit[qi] = pos0 + s_idx[qi];          // last key this row may attend to (causal)
        const bf16_t* qp = (const bf16_t*)p.q + (int64_t)(tok0 + s_idx[qi]) * p.q_stride + head[qi] * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[qi][ks] = *(const bf16x8*)(qp + ks * 32 + fg * 8);
    }

    // last key any row of this workgroup needs
    int last_row = qt * QROWS + QROWS - 1;
    if (last_row > R - 1) last_row = R - 1;
    const int kmax = pos0 + last_row / p.group;
    const int n_kt = kmax / 64 + 1;

    f32x4 o[TQ][DT];
    float m_run[TQ], l_run[TQ];
#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        m_run[qi] = -INFINITY; l_run[qi] = 0.f;
#pragma unroll
        for (int d = 0; d < DT; d++) o[qi][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // softmax in the exp2 domain: scores are scaled by scale * log2(e) once, so every exponential is one v_exp_f32
    const float sl2 = p.scale * 1.4426950408889634f;
    // first key tile that any row of this WAVE may not see completely (causal): tiles below it need no masking
    const int wave_first_limit = pos0 + (qt * QROWS + wave * TQ * 16) / p.group;

    // K/V tiles are fetched one tile ahead into registers (64 B per thread), so the global latency of tile kt+1 runs
    // under the MFMAs and softmax of tile kt; the LDS image is refreshed between two barriers
    constexpr int KCH = HD / 8;                      // 16-B chunks per K row
    constexpr int KPT = 64 * KCH / 256, VPT = HD * 8 / 256;     // chunks per thread: K tile, V^T tile
    bf16x8 kreg[KPT], vreg[VPT];
    auto fetch = [&](int kt) {
        const int bi = kv_block_index(p, kt * 64), krow0 = kt * 64 - bi * p.Tmax;      // (64 | Tmax: no straddling)
        const int64_t blk_off = (int64_t)tbl[bi] * p.slot_stride + (int64_t)kvh * p.Tmax * HD;
        const bf16_t* kbase = (const bf16_t*)p.kcache + blk_off + (int64_t)krow0 * HD;     // K rows of the tile
        const bf16_t* vbase = (const bf16_t*)p.vcache + blk_off + krow0;                    // V^T columns of the tile
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            const int c = tid + i * 256, r = c / KCH, cc = c % KCH;
            kreg[i] = *(const bf16x8*)(kbase + (int64_t)r * HD + cc * 8);
        }
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            const int c = tid + i * 256, r = c >> 3, cc = c & 7;
            vreg[i] = *(const bf16x8*)(vbase + (int64_t)r * p.Tmax + cc * 8);
        }
    };
    fetch(0);
    for (int kt = 0; kt < n_kt; kt++) {
        __syncthreads();                             // every wave is done reading the previous tile's LDS image
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            const int c = tid + i * 256, r = c / KCH, cc = c % KCH;
            *(bf16x8*)(Ks + r * KROW + cc * 8) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            const int c = tid + i * 256, r = c >> 3, cc = c & 7;
            *(bf16x8*)(Vts + r * VROW + cc * 8) = vreg[i];
        }
        if (kt + 1 < n_kt) fetch(kt + 1);            // in flight during this tile's compute
        __syncthreads();

        // ---- S^T = K · Q^T : 4 sub-tiles of 16 keys, each K fragment used for all TQ query sub-tiles ----
        f32x4 s[TQ][4];
#pragma unroll
        for (int qi = 0; qi < TQ; qi++)
#pragma unroll
            for (int t = 0; t < 4; t++) s[qi][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (t * 16 + fq) * KROW + ks * 32 + fg * 8);
#pragma unroll
                for (int qi = 0; qi < TQ; qi++)
                    s[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], s[qi][t], 0, 0, 0);
            }
        // lane holds S^T[key = kt*64 + 16t + 4fg + r][q = fq] per sub-tile
        const bool whole = kt * 64 + 63 <= wave_first_limit;      // every row of the wave sees the whole tile
#pragma unroll
        for (int qi = 0; qi < TQ; qi++) {
            float tmax = -INFINITY;
            if (whole) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    s[qi][t] *= sl2;
                    tmax = fmaxf(tmax, fmaxf(fmaxf(s[qi][t][0], s[qi][t][1]), fmaxf(s[qi][t][2], s[qi][t][3])));
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int key = kt * 64 + t * 16 + fg * 4 + r;
                        float v = s[qi][t][r] * sl2;
                        v = (key <= limit[qi]) ? v : -INFINITY;      // reference: -1e10 then exp() == 0 exactly
                        s[qi][t][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run[qi], tmax);
            // a sub-tile whose rows see nothing yet (its causal limit lies before this tile AND before every earlier
            // one) cannot occur: tile 0 always holds key 0 <= limit, so m_new is finite from the first tile on
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __builtin_amdgcn_exp2f(s[qi][t][r] - m_new);
                    s[qi][t][r] = pv;
                    psum += pv;
                }
            if (__any(m_new != m_run[qi])) {                 // the running maximum moved for some row: rescale
                const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
                l_run[qi] *= alpha;
#pragma unroll
                for (int d = 0; d < DT; d++) o[qi][d] *= alpha;
            }
            l_run[qi] += psum;
            m_run[qi] = m_new;
        }

        // ---- O^T += V^T · P^T : k index permuted identically on both operands; each V^T fragment used TQ times ----
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 pf[TQ];
#pragma unroll
            for (int qi = 0; qi < TQ; qi++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pf[qi][r] = (bf16_t)s[qi][2 * u][r];          // keys 32u + 4fg + r
                    pf[qi][4 + r] = (bf16_t)s[qi][2 * u + 1][r];  // keys 32u + 16 + 4fg + r
                }
#pragma unroll
            for (int d = 0; d < DT; d++) {
                const bf16_t* vr = Vts + (d * 16 + fq) * VROW + u * 32 + fg * 4;
                const bf16x4 lo = *(const bf16x4*)(vr);
                const bf16x4 hi = *(const bf16x4*)(vr + 16);
                bf16x8 vf;
#pragma unroll
                for (int r = 0; r < 4; r++) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
#pragma unroll
                for (int qi = 0; qi < TQ; qi++)
                    o[qi][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qi], o[qi][d], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (!row_ok[qi]) continue;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < DT; d++)    // O[q][d = 16d + 4fg + r], written in the next GEMM's operand layout
            act_store4<bf16_t>((bf16_t*)p.out, tok0 + s_idx[qi], head[qi] * HD + d * 16 + fg * 4, p.out_stride, o[qi][d] * inv);
    }
}

// ------------------------------------------------------------------------------------------
// bf16 decode kernel (one new token per sequence, group <= 16 query heads per kv head).
// HBM/latency-bound KV read: one workgroup per (sequence, kv head); its NW waves take the 64-key
// tiles round-robin (split-T inside the workgroup), each with its own online-softmax state, K and
// V^T fragments loaded straight HBM -> VGPR (nothing is shared between waves, so no LDS staging
// and no barrier in the loop), and the NW partial results are merged once through LDS
// (flash-decoding combine, fixed wave order).  Same transposed MFMA dataflow as the prefill kernel.
// ------------------------------------------------------------------------------------------
// rotate one (lo, hi) pair of 8-wide chunks: lo = d0..d0+7 (< HD/2), hi = the same offsets + HD/2
__device__ __forceinline__ void rope_pair8(const float* row, const float* cs, const float* sn, int d0, int half,
                                           bf16x8& lo, bf16x8& hi) {
    const f32x4 a0 = *(const f32x4*)(row + d0), a1 = *(const f32x4*)(row + d0 + 4);
    const f32x4 b0 = *(const f32x4*)(row + d0 + half), b1 = *(const f32x4*)(row + d0 + half + 4);
    f32x4 yl0 = a0, yl1 = a1, yh0 = b0, yh1 = b1;
    if (cs) {
        const f32x4 c0 = *(const f32x4*)(cs + d0), c1 = *(const f32x4*)(cs + d0 + 4);
        const f32x4 s0 = *(const f32x4*)(sn + d0), s1 = *(const f32x4*)(sn + d0 + 4);
        yl0 = a0 * c0 + (-b0) * s0; yl1 = a1 * c1 + (-b1) * s1;     // x*cos + rotate_half(x)*sin (table halves duplicate)
        yh0 = b0 * c0 + a0 * s0;    yh1 = b1 * c1 + a1 * s1;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        lo[e] = (bf16_t)yl0[e]; lo[4 + e] = (bf16_t)yl1[e];
        hi[e] = (bf16_t)yh0[e]; hi[4 + e] = (bf16_t)yh1[e];
    }
}

template <int HD, int NW, bool FUSED>
__global__ __launch_bounds__(NW * 64) void attn_decode_bf16_kernel(AttnArgs p) {
    constexpr int KS = HD / 32, DT = HD / 16, HALF = HD / 2;
    __shared__ float red_m[NW][16];
    __shared__ float red_l[NW][16];
    __shared__ f32x4 red_o[NW][DT][64];
    const int seq = blockIdx.y, kvh = blockIdx.x;
    const int32_t* tbl = p.blk_table + (int64_t)seq * p.tbl_stride;
    const int tok = p.seq_tok_start[seq], pos0 = p.seq_pos[seq];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fq = lane & 15, fg = lane >> 4;
    const bool row_ok = fq < p.group;
    const int head = kvh * p.group + (row_ok ? fq : 0);
    bf16x8 qf[KS];
    const int n_kt = pos0 / 64 + 1;
    // FUSED: the new token's K row (RoPE applied) as this lane's operand chunks, and its V values for this lane's d
    bf16x8 knew[KS];
    bf16_t vnew[DT];
    if (FUSED) {
        const float* row = p.qkv + (int64_t)tok * p.qkv_stride;
        const float* cs = p.cos_t ? p.cos_t + (int64_t)pos0 * HD : nullptr;
        const float* sn = p.sin_t ? p.sin_t + (int64_t)pos0 * HD : nullptr;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            rope_pair8(row + head * HD, cs, sn, ks * 32 + fg * 8, HALF, qf[ks], qf[ks + KS / 2]);
            rope_pair8(row + (p.nH + kvh) * HD, cs, sn, ks * 32 + fg * 8, HALF, knew[ks], knew[ks + KS / 2]);
        }
        const float* vrow = row + (p.nH + p.nKV + kvh) * HD;
#pragma unroll
        for (int d = 0; d < DT; d++) vnew[d] = (bf16_t)vrow[d * 16 + fq];
    } else {
        const bf16_t* qp = (const bf16_t*)p.q + (int64_t)tok * p.q_stride + head * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[ks] = *(const bf16x8*)(qp + ks * 32 + fg * 8);
    }

    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // Each wave owns key tiles wave, wave+NW, ...; the loads of TWO of its tiles are issued before either is used
    // (a decode step is latency-bound: ~150 KB per workgroup), so sequences up to 2*NW*64 keys take one round trip.
    constexpr int NT2 = HD <= 64 ? 2 : 1;      // hd 128: one tile's operands already fill the register budget
    // the block ids of this wave's first tiles do not depend on the sequence length: fetch them beside it, not after it
    int first_blk[NT2];
#pragma unroll
    for (int h = 0; h < NT2; h++) first_blk[h] = tbl[min(kv_block_index(p, (wave + h * NW) * 64), p.tbl_stride - 1)];
    bf16x8 kf[NT2][4][KS];
    bf16x4 vlo[NT2][2][DT], vhi[NT2][2][DT];
    for (int kt0 = wave; kt0 < n_kt; kt0 += NT2 * NW) {
#pragma unroll
        for (int h = 0; h < NT2; h++) {
            const int kt = kt0 + h * NW;
            if (kt >= n_kt) continue;
            const int bi = kv_block_index(p, kt * 64), krow0 = kt * 64 - bi * p.Tmax;     // the KV block holding this tile
            const int blk = kt0 == wave ? first_blk[h] : tbl[bi];
            const int64_t blk_off = (int64_t)blk * p.slot_stride + (int64_t)kvh * p.Tmax * HD;
            const bf16_t* kbase = (const bf16_t*)p.kcache + blk_off + (int64_t)krow0 * HD;
            const bf16_t* vbase = (const bf16_t*)p.vcache + blk_off + krow0;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
                    kf[h][t][ks] = *(const bf16x8*)(kbase + (int64_t)(t * 16 + fq) * HD + ks * 32 + fg * 8);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int d = 0; d < DT; d++) {
                    const bf16_t* vr = vbase + (int64_t)(d * 16 + fq) * p.Tmax + u * 32 + fg * 4;
                    vlo[h][u][d] = *(const bf16x4*)(vr);
                    vhi[h][u][d] = *(const bf16x4*)(vr + 16);
                }
        }
#pragma unroll
        for (int h = 0; h < NT2; h++) {
            const int kt = kt0 + h * NW;
            if (kt >= n_kt) continue;
            if (FUSED && kt == n_kt - 1) {
                // the tile that holds key pos0: its slab row is not written yet (or not visible): patch the operands
                const int kl = pos0 & 63;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t == (kl >> 4) && fq == (kl & 15)) {
#pragma unroll
                        for (int ks = 0; ks < KS; ks++) kf[h][t][ks] = knew[ks];
                    }
                const int r32 = kl & 31, hi_half = r32 >> 4, fg0 = (r32 & 15) >> 2, j0 = r32 & 3;
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int d = 0; d < DT; d++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (u == (kl >> 5) && fg == fg0 && j == j0) {
                                if (hi_half) vhi[h][u][d][j] = vnew[d]; else vlo[h][u][d][j] = vnew[d];
                            }
            }
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[h][t][ks], qf[ks], s[t], 0, 0, 0);
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int key = kt * 64 + t * 16 + fg * 4 + r;
                    float v = s[t][r] * p.scale;
                    v = (key <= pos0) ? v : -INFINITY;
                    s[t][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);     // finite: every tile kt < n_kt holds key kt*64 <= pos0
            const float alpha = __expf(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __expf(s[t][r] - m_new);
                    s[t][r] = pv;
                    psum += pv;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < DT; d++) o[d] *= alpha;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; r++) { pf[r] = (bf16_t)s[2 * u][r]; pf[4 + r] = (bf16_t)s[2 * u + 1][r]; }
#pragma unroll
                for (int d = 0; d < DT; d++) {
                    bf16x8 vf;
#pragma unroll
                    for (int r = 0; r < 4; r++) { vf[r] = vlo[h][u][d][r]; vf[4 + r] = vhi[h][u][d][r]; }
                    o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
        }
    }
    if (FUSED) {
        // append the new key/value to the slabs for the following steps — after every load of this step (which used
        // the register copies), so the loads above are not ordered behind these stores
        int nblk, nrow;
        kv_locate(tbl, 0, pos0, p.Tmax, nblk, nrow);
        if (wave == 0 && fq == 0) {      // lanes fg = 0..3 of column 0 hold all of K_new between them
            bf16_t* kd = (bf16_t*)p.kcache + (int64_t)nblk * p.slot_stride + ((int64_t)kvh * p.Tmax + nrow) * HD;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) *(bf16x8*)(kd + ks * 32 + fg * 8) = knew[ks];
        }
        if (wave == 1 % NW && fg == 0) {  // lanes fq = 0..15 hold V_new[16d + fq]
            bf16_t* vd = (bf16_t*)p.vcache + (int64_t)nblk * p.slot_stride + (int64_t)kvh * p.Tmax * HD + nrow;
#pragma unroll
            for (int d = 0; d < DT; d++) vd[(int64_t)(d * 16 + fq) * p.Tmax] = vnew[d];
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (fg == 0) { red_m[wave][fq] = m_run; red_l[wave][fq] = l_run; }
#pragma unroll
    for (int d = 0; d < DT; d++) red_o[wave][d][lane] = o[d];
    __syncthreads();
    if (wave != 0 || !row_ok) return;
    float mstar = red_m[0][fq];
#pragma unroll
    for (int w = 1; w < NW; w++) mstar = fmaxf(mstar, red_m[w][fq]);
    float L = 0.f;
    f32x4 O[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) O[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const float sc = __expf(red_m[w][fq] - mstar);   // idle waves: exp(-inf) = 0
        L += red_l[w][fq] * sc;
#pragma unroll
        for (int d = 0; d < DT; d++) O[d] += red_o[w][d][lane] * sc;
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int d = 0; d < DT; d++)
        act_store4<bf16_t>((bf16_t*)p.out, tok, head * HD + d * 16 + fg * 4, p.out_stride, O[d] * inv);
}

// ------------------------------------------------------------------------------------------
// fp32 parity kernel: block = (query position, head), 256 threads, T <= 8192
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs p, int HD) {
    extern __shared__ float sc[];                 // [T_max_needed] scores then probs
    __shared__ float red[8];
    const int seq = blockIdx.z, head = blockIdx.y, s_idx = blockIdx.x;
    if (s_idx >= p.seq_len[seq]) return;
    const int tid = threadIdx.x;
    const int kvh = head / p.group;
    const int tok = p.seq_tok_start[seq] + s_idx;
    const int T = p.seq_pos[seq] + s_idx + 1;     // keys 0..limit
    const int32_t* tbl = p.blk_table + (int64_t)seq * p.tbl_stride;
    const float* q = (const float*)p.q + (int64_t)tok * p.q_stride + head * HD;
    auto kv_row = [&](const void* cache, int j) {       // row of key j: [block][kvh][Tmax][HD]
        const int bi = j / p.Tmax;
        return (const float*)cache + (int64_t)tbl[bi] * p.slot_stride + ((int64_t)kvh * p.Tmax + (j - bi * p.Tmax)) * HD;
    };

    float lmax = -INFINITY;
    for (int j = tid; j < T; j += 256) {
        const float* kr = kv_row(p.kcache, j);
        float sum = 0.f;
        for (int d = 0; d < HD; d++) sum = fmaf(q[d], kr[d], sum);
        sum *= p.scale;
        sc[j] = sum;
        lmax = fmaxf(lmax, sum);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int j = tid; j < T; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lsum;
    __syncthreads();
    const float tot = red[4] + red[5] + red[6] + red[7];
    for (int j = tid; j < T; j += 256) sc[j] = sc[j] / tot;      // attention.go:463-466
    __syncthreads();
    float* op = (float*)p.out + (int64_t)tok * p.out_stride + head * HD;
    for (int d = tid; d < HD; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < T; j++) acc = fmaf(sc[j], kv_row(p.vcache, j)[d], acc);
        op[d] = acc;
    }
}

}  // namespace nvl
