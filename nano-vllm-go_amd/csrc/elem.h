// elem.h — wavefront-reduction / element kernels of the forward path: embedding gather,
// RMSNorm/LayerNorm, RoPE + KV append, activations, softmax, MoE routing, argmax, and the
// load-time layout converters.  All HBM-bound; fp32 math; one 64-lane wave per row where a
// reduction is needed.  ActT is the activation storage type handed to the next GEMM
// (bf16 in the product path, float in the fp32 parity mode).
#pragma once
#include "common.h"

namespace nvl {

// ---------------------------------------------------------------------------------------
// embedWithOffset (generic_model.go:567-592) + EmbeddingMultiplier (:298-302)
// ---------------------------------------------------------------------------------------
// FM = true: the tables are in the fragment-major weight layout (fm_index below) — the tied LM head and
// the embedding table are ONE buffer (generic_loader.go:255-259 ties them; the reference transposes a copy).
template <typename WT, bool FM>
__global__ void embed_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ tok_pos,
                             const WT* __restrict__ emb, const WT* __restrict__ pos_emb,
                             int max_seq, float mult, float* __restrict__ x, int H) {
    const int t = blockIdx.x;
    const int tok = tokens[t];
    const int pos = tok_pos[t];
    const bool use_pos = pos_emb && pos < max_seq;
    float* xr = x + (int64_t)t * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float v = (float)emb[FM ? fm_index(tok, j, H) : (int64_t)tok * H + j];
        if (use_pos) v += (float)pos_emb[FM ? fm_index(pos, j, H) : (int64_t)pos * H + j];
        if (mult != 0.f) v *= mult;
        xr[j] = v;
    }
}

// ---------------------------------------------------------------------------------------
// LayerNorm / RMSNorm (tensor.go:193-250).  rows_idx (optional) gathers input rows (final norm on
// the last row of each sequence only).  Output: the next GEMM's operand (bf16 fragment-major / f32).
//   norm_kernel      : one wave per row, 4 rows per 256-thread block (prefill: thousands of rows)
//   norm_row_kernel  : one 256-thread block per row, the row held in registers between the
//                      reduction and the write (decode: a handful of rows -> latency-bound)
// ---------------------------------------------------------------------------------------
template <typename ActT>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x,
                                                   const int32_t* __restrict__ rows_idx,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps,
                                                   ActT* __restrict__ y, int rows, int H) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int src = rows_idx ? rows_idx[r] : r;
    const float* xr = x + (int64_t)src * H;
    const int H4 = H >> 2;  // H % 4 == 0 for every supported model
    if (b == nullptr) {
        float ss = 0.f;
        for (int j = lane; j < H4; j += 64) {
            const f32x4 v = *(const f32x4*)(xr + j * 4);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        ss = wave_sum(ss);
        const float rms = sqrtf(ss / (float)H + eps);
        for (int j = lane; j < H4; j += 64) {
            const f32x4 v = *(const f32x4*)(xr + j * 4);
            const f32x4 ww = *(const f32x4*)(w + j * 4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = (v[k] / rms) * ww[k];
            act_store4<ActT>(y, r, j * 4, H, o);
        }
    } else {
        float s = 0.f;
        for (int j = lane; j < H4; j += 64) {
            const f32x4 v = *(const f32x4*)(xr + j * 4);
            s += v[0] + v[1] + v[2] + v[3];
        }
        const float mean = wave_sum(s) / (float)H;
        float vs = 0.f;
        for (int j = lane; j < H4; j += 64) {
            const f32x4 v = *(const f32x4*)(xr + j * 4);
#pragma unroll
            for (int k = 0; k < 4; k++) { const float d = v[k] - mean; vs += d * d; }
        }
        const float var = wave_sum(vs) / (float)H;
        const float sd = sqrtf(var + eps);
        for (int j = lane; j < H4; j += 64) {
            const f32x4 v = *(const f32x4*)(xr + j * 4);
            const f32x4 ww = *(const f32x4*)(w + j * 4);
            const f32x4 bb = *(const f32x4*)(b + j * 4);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = ((v[k] - mean) / sd) * ww[k] + bb[k];
            act_store4<ActT>(y, r, j * 4, H, o);
        }
    }
}

constexpr int NORM_ROW_MAXCH = 5;   // float4 chunks per thread: H <= 256*4*5 = 5120
__device__ __forceinline__ float block256_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}
// Pending residual: the decode GEMMs that precede a norm (O projection, W2) may leave their output as
// split-K partial slices instead of adding it into x; this kernel then completes the residual add
// (generic_model.go:320-326,383-389) while it reads the row:  x[row] += alpha * sum_s part[s][row]  (slice order).
struct PendingResid {
    const float* part;   // [slices][rows_total][H] or NULL
    int slices;
    int rows_total;
    float alpha;
    // MoE combine folded in (moe.go:105-120): when slot_of != NULL the "slices" of row r are the top_k expert outputs
    // part[slot_of[r*slices + k]][H], weighted by gate_w[r*slices + k] — replaces moe_combine_kernel for decode
    const int32_t* slot_of;
    const float* gate_w;
};
// Prefill-sized batches, bf16 output: one 1024-thread workgroup per 16 output rows (one fragment-major row tile).  A wave
// normalises one row held in registers (same arithmetic as norm_kernel) into LDS; the workgroup then writes the tile's
// 1-KiB operand blocks with full-line stores (norm_kernel's lanes each write 8 bytes of a block: 16-byte fragments at a
// 256-byte stride).  H <= 64 * 4 * MAXCH (instances for 8 and 16 chunks per lane: H <= 2048 / 4096).
template <int MAXCH, bool PENDING = false>
__global__ __launch_bounds__(1024) void norm_tile16_kernel(float* __restrict__ x, const int32_t* __restrict__ rows_idx,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float eps, bf16_t* __restrict__ y, int rows, int H, PendingResid pr) {
    extern __shared__ __attribute__((aligned(16))) char smem_n[];
    bf16_t* tile = (bf16_t*)smem_n;                       // [16][H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 16 + wave;
    const int H4 = H >> 2;
    if (r < rows) {
        const int src = rows_idx ? rows_idx[r] : r;
        float* xr = x + (int64_t)src * H;
        f32x4 v[MAXCH];
#pragma unroll
        for (int c = 0; c < MAXCH; c++) {
            const int j = lane + c * 64;
            v[c] = j < H4 ? *(const f32x4*)(xr + j * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (PENDING) {     // complete the residual add a split-K projection left as slices (plain slices only: no MoE gather here)
            // slice loop outside, chunks unrolled inside, eight chunks at a time: the row and the running sums stay in registers
#pragma unroll
            for (int c0 = 0; c0 < MAXCH; c0 += 8) {
                f32x4 sl[8];
#pragma unroll
                for (int c = 0; c < 8; c++) sl[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < pr.slices; k++) {
                    const float* ps = pr.part + ((int64_t)k * pr.rows_total + src) * H;
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int j = lane + (c0 + c) * 64;
                        if (j < H4) sl[c] += *(const f32x4*)(ps + j * 4);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int j = lane + (c0 + c) * 64;
                    if (j < H4) {
                        v[c0 + c] += pr.alpha * sl[c];      // (slices in order, then alpha: as norm_row_kernel)
                        *(f32x4*)(xr + j * 4) = v[c0 + c];
                    }
                }
            }
        }
        float mean = 0.f, inv;
        if (b == nullptr) {
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < MAXCH; c++) ss += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
            inv = sqrtf(wave_sum(ss) / (float)H + eps);
        } else {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < MAXCH; c++) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
            mean = wave_sum(s) / (float)H;
            float vs = 0.f;
#pragma unroll
            for (int c = 0; c < MAXCH; c++) {
                const int j = lane + c * 64;
                if (j < H4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { const float d = v[c][k] - mean; vs += d * d; }
                }
            }
            inv = sqrtf(wave_sum(vs) / (float)H + eps);
        }
#pragma unroll
        for (int c = 0; c < MAXCH; c++) {
            const int j = lane + c * 64;
            if (j < H4) {
                const f32x4 ww = *(const f32x4*)(w + j * 4);
                bf16x4 o;
                if (b == nullptr) {
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = (bf16_t)((v[c][k] / inv) * ww[k]);
                } else {
                    const f32x4 bb = *(const f32x4*)(b + j * 4);
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = (bf16_t)(((v[c][k] - mean) / inv) * ww[k] + bb[k]);
                }
                *(bf16x4*)(tile + wave * H + j * 4) = o;
            }
        }
    }
    __syncthreads();
    // operand blocks of this row tile: block kb holds k = 32kb .. 32kb+31 of the 16 rows, lane l = row (l & 15), 8 k's (l >> 4)
    const int nkb = H >> 5;
    const int lr = lane & 15, lc = lane >> 4;
    if (blockIdx.x * 16 + lr < rows)
        for (int kb = wave; kb < nkb; kb += 16) {
            const bf16x8 f = *(const bf16x8*)(tile + lr * H + kb * 32 + lc * 8);
            *(bf16x8*)(y + (((int64_t)blockIdx.x * nkb + kb) * 64 + lane) * 8) = f;
        }
}

// the arithmetic of one row once it sits in registers (shared by norm_row_kernel and decode_seam_kernel)
template <typename ActT>
__device__ __forceinline__ void norm_row_finish(f32x4 (&v)[NORM_ROW_MAXCH], f32x4 (&ww)[NORM_ROW_MAXCH],
                                                f32x4 (&bb)[NORM_ROW_MAXCH], const float* b, float eps, ActT* y, int r,
                                                int H, float* red) {
    const int H4 = H >> 2;
    if (b == nullptr) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NORM_ROW_MAXCH; c++) ss += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
        const float rms = sqrtf(block256_sum(ss, red) / (float)H + eps);
#pragma unroll
        for (int c = 0; c < NORM_ROW_MAXCH; c++) {
            const int j = threadIdx.x + c * 256;
            if (j < H4) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = (v[c][k] / rms) * ww[c][k];
                act_store4<ActT>(y, r, j * 4, H, o);
            }
        }
    } else {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NORM_ROW_MAXCH; c++) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];   // zero-filled beyond H
        const float mean = block256_sum(s, red) / (float)H;
        float vs = 0.f;
#pragma unroll
        for (int c = 0; c < NORM_ROW_MAXCH; c++) {
            const int j = threadIdx.x + c * 256;
            if (j < H4) {
#pragma unroll
                for (int k = 0; k < 4; k++) { const float d = v[c][k] - mean; vs += d * d; }
            }
        }
        const float sd = sqrtf(block256_sum(vs, red) / (float)H + eps);
#pragma unroll
        for (int c = 0; c < NORM_ROW_MAXCH; c++) {
            const int j = threadIdx.x + c * 256;
            if (j < H4) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; k++) o[k] = ((v[c][k] - mean) / sd) * ww[c][k] + bb[c][k];
                act_store4<ActT>(y, r, j * 4, H, o);
            }
        }
    }
}

template <typename ActT>
__global__ __launch_bounds__(256) void norm_row_kernel(float* __restrict__ x,
                                                       const int32_t* __restrict__ rows_idx,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ b, float eps,
                                                       ActT* __restrict__ y, int H, PendingResid pr) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int src = rows_idx ? rows_idx[r] : r;
    float* xr = x + (int64_t)src * H;
    const int H4 = H >> 2;
    f32x4 v[NORM_ROW_MAXCH], ww[NORM_ROW_MAXCH], bb[NORM_ROW_MAXCH];
#pragma unroll
    for (int c = 0; c < NORM_ROW_MAXCH; c++) {
        const int j = threadIdx.x + c * 256;
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < H4) {
            v[c] = *(const f32x4*)(xr + j * 4);
            if (pr.part && pr.slot_of) {
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < pr.slices; k++)          // same order and arithmetic as moe_combine_kernel
                    s += pr.gate_w[(int64_t)src * pr.slices + k] *
                         *(const f32x4*)(pr.part + (int64_t)pr.slot_of[(int64_t)src * pr.slices + k] * H + j * 4);
                v[c] = v[c] + (pr.alpha != 0.f ? pr.alpha * s : s);
                *(f32x4*)(xr + j * 4) = v[c];
            } else if (pr.part) {
                f32x4 s = *(const f32x4*)(pr.part + (int64_t)src * H + j * 4);
                for (int k = 1; k < pr.slices; k++)
                    s += *(const f32x4*)(pr.part + ((int64_t)k * pr.rows_total + src) * H + j * 4);
                v[c] += pr.alpha * s;
                *(f32x4*)(xr + j * 4) = v[c];
            }
            ww[c] = *(const f32x4*)(w + j * 4);
            if (b) bb[c] = *(const f32x4*)(b + j * 4);
        }
    }
    norm_row_finish<ActT>(v, ww, bb, b, eps, y, r, H, red);
}

// Debug tap (nvl_set_debug mode 2): the residual stream after a layer as the NEXT kernel will see it — x plus whatever a
// split-K / MoE projection left pending for the following norm (same arithmetic and order as norm_row_kernel) — copied
// out WITHOUT touching x or the pending state, so the product path runs exactly as it does untapped.
__global__ __launch_bounds__(256) void tap_hidden_kernel(const float* __restrict__ x, float* __restrict__ dst, int H, PendingResid pr) {
    const int r = blockIdx.x;
    const float* xr = x + (int64_t)r * H;
    for (int j = threadIdx.x; j < (H >> 2); j += 256) {
        f32x4 v = *(const f32x4*)(xr + j * 4);
        if (pr.part && pr.slot_of) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < pr.slices; k++)
                s += pr.gate_w[(int64_t)r * pr.slices + k] * *(const f32x4*)(pr.part + (int64_t)pr.slot_of[(int64_t)r * pr.slices + k] * H + j * 4);
            v = v + (pr.alpha != 0.f ? pr.alpha * s : s);
        } else if (pr.part) {
            f32x4 s = *(const f32x4*)(pr.part + (int64_t)r * H + j * 4);
            for (int k = 1; k < pr.slices; k++) s += *(const f32x4*)(pr.part + ((int64_t)k * pr.rows_total + r) * H + j * 4);
            v += pr.alpha * s;
        }
        *(f32x4*)(dst + (int64_t)r * H + j * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------
// RoPE (rope.go:153-205) on Q and K as they leave the fused QKV GEMM, plus the KV append that
// replaces Concatenate (tensor.go:283-321): K -> slab[pos][hd], V -> slab[pos][hd] (VT = true would write V^T: unused).
// qkv is fp32 [tokens][(nH+2nKV)*HD] = [Q heads | K heads | V heads].
// grid = (tokens, nH + 2*nKV), block = HD/2 threads... one thread per rotation pair.
// ---------------------------------------------------------------------------------------
template <typename ActT, bool VT>
__global__ void rope_kv_kernel(const float* __restrict__ qkv, int qkv_stride,
                               const int32_t* __restrict__ tok_pos, const int32_t* __restrict__ tok_tbl,
                               const int32_t* __restrict__ blk_table,
                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                               ActT* __restrict__ q_out, int q_stride,
                               ActT* __restrict__ kcache, ActT* __restrict__ vcache,
                               int64_t slot_stride, int Tmax, int nH, int nKV, int HD) {
    const int t = blockIdx.x, h = blockIdx.y;
    const int pos = tok_pos[t];
    int slot, row;                                   // KV block of this position and the row inside it
    kv_locate(blk_table, tok_tbl[t], pos, Tmax, slot, row);
    const int half = HD >> 1;
    const float* src = qkv + (int64_t)t * qkv_stride + (int64_t)h * HD;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float x1 = src[i], x2 = src[i + half];
        float y1 = x1, y2 = x2;
        if (h < nH + nKV && cos_t) {
            const float c1 = cos_t[(int64_t)pos * HD + i], s1 = sin_t[(int64_t)pos * HD + i];
            const float c2 = cos_t[(int64_t)pos * HD + half + i], s2 = sin_t[(int64_t)pos * HD + half + i];
            y1 = x1 * c1 + (-x2) * s1;   // rope.go:196-202: x*cos + rotate_half(x)*sin
            y2 = x2 * c2 + x1 * s2;
        }
        if (h < nH) {
            ActT* d = q_out + (int64_t)t * q_stride + (int64_t)h * HD;
            ActIO<ActT>::st(d + i, y1);
            ActIO<ActT>::st(d + i + half, y2);
        } else if (h < nH + nKV) {
            const int kvh = h - nH;
            ActT* d = kcache + (int64_t)slot * slot_stride + ((int64_t)kvh * Tmax + row) * HD;
            ActIO<ActT>::st(d + i, y1);
            ActIO<ActT>::st(d + i + half, y2);
        } else {
            const int kvh = h - nH - nKV;
            if (VT) {
                ActT* d = vcache + (int64_t)slot * slot_stride + (int64_t)kvh * Tmax * HD;
                ActIO<ActT>::st(d + (int64_t)i * Tmax + row, y1);
                ActIO<ActT>::st(d + (int64_t)(i + half) * Tmax + row, y2);
            } else {
                ActT* d = vcache + (int64_t)slot * slot_stride + ((int64_t)kvh * Tmax + row) * HD;
                ActIO<ActT>::st(d + i, y1);
                ActIO<ActT>::st(d + i + half, y2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// un-fused activations for the fp32 parity mode and the op-level entry points
// ---------------------------------------------------------------------------------------
// h [rows][2F] = [gate | up] -> y [rows][F] = silu(gate)*up   (transformer.go:50-66)
template <typename ActT>
__global__ void swiglu_kernel(const float* __restrict__ h, ActT* __restrict__ y, int64_t rows, int F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * F) return;
    const int64_t r = i / F; const int f = (int)(i - r * F);
    const float g = h[r * 2 * F + f], u = h[r * 2 * F + F + f];
    ActIO<ActT>::st(y + i, (g / (1.0f + expf(-g))) * u);
}
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = gelu_tanh_f(x[i]);        // (the same function as the GEMM epilogue: what the op test checks is what the model runs)
}
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = v / (1.0f + expf(-v));
}

// Softmax over the last dim (tensor.go:128-160): one wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (int64_t)r * cols;
    float* yr = y + (int64_t)r * cols;
    float mx = -INFINITY;
    for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, xr[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) { const float e = expf(xr[j] - mx); yr[j] = e; s += e; }
    s = wave_sum(s);
    for (int j = lane; j < cols; j += 64) yr[j] = yr[j] / s;
}

// ---------------------------------------------------------------------------------------
// greedy argmax (cmd/ask/main.go:389-402: first strict maximum) + LogitsScaling divide
// (generic_model.go:473-477).  Two passes so a handful of rows still fill the chip:
//   pass 1: grid (chunks, rows): each block scans ARGMAX_CHUNK logits (scaled in place when
//           logits_scaling != 0) and writes its (max, first index);
//   pass 2: one wave per row reduces the chunk winners (ties -> lower index).
// ---------------------------------------------------------------------------------------
constexpr int ARGMAX_CHUNK = 4096;

__device__ __forceinline__ void argmax_merge(float& best, int& bi, float ov, int oi) {
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
}

__global__ __launch_bounds__(256) void argmax_partial_kernel(float* __restrict__ logits, int ld, int V,
                                                             float logits_scaling, float* __restrict__ pval,
                                                             int32_t* __restrict__ pidx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    float* row = logits + (int64_t)blockIdx.y * ld;
    const int c0 = blockIdx.x * ARGMAX_CHUNK;
    const int c1 = min(V, c0 + ARGMAX_CHUNK);
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll 4
    for (int j = c0 + threadIdx.x; j < c1; j += 256) {
        float v = row[j];
        if (logits_scaling != 0.f) { v = v / logits_scaling; row[j] = v; }
        if (v > best) { best = v; bi = j; }      // strict >, ascending j => first max per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) argmax_merge(best, bi, sv[k], si[k]);
        pval[blockIdx.y * gridDim.x + blockIdx.x] = best;
        pidx[blockIdx.y * gridDim.x + blockIdx.x] = bi;
    }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pval,
                                                          const int32_t* __restrict__ pidx, int chunks,
                                                          int32_t* __restrict__ out) {
    const int r = blockIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < chunks; c += 64) argmax_merge(best, bi, pval[r * chunks + c], pidx[r * chunks + c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (threadIdx.x == 0) out[r] = (bi == 0x7fffffff) ? 0 : bi;   // all -inf / NaN: reference returns index 0
}

// logits /= LogitsScaling (generic_model.go:472-477) for the rows a sampler reads (the argmax path folds it into
// argmax_partial_kernel)
__global__ void scale_rows_kernel(float* __restrict__ x, int ld, int n, float div) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { float* p = x + (int64_t)blockIdx.y * ld + j; *p = *p / div; }
}

// ---------------------------------------------------------------------------------------
// Decode seam (nvl_decode_greedy, bf16 path): the tail of step s and the head of step s+1 as ONE launch, one
// 256-thread block per sequence: argmax over the chunk partials (argmax_final_kernel), the token fed back + every
// position advanced by one (cmd/ask/main.go:315-320), the embedding gather of that token (embed_kernel,
// generic_model.go:567-592) and layer 0's norm of the row (norm_row_kernel) — same arithmetic, same summation order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_seam_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                          int chunks, const int32_t* __restrict__ sampled,
                                                          int32_t* __restrict__ hist, int64_t hist_stride,
                                                          int32_t* __restrict__ hist_len, int32_t* __restrict__ argmax_out,
                                                          int32_t* __restrict__ ring_out, const int32_t* __restrict__ ring_pos0,
                                                          int ring_rows, int32_t* __restrict__ tokens,
                                                          int32_t* __restrict__ tok_pos, int32_t* __restrict__ seq_pos,
                                                          const bf16_t* __restrict__ emb, const bf16_t* __restrict__ pos_emb,
                                                          int max_seq, float mult, float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float eps, bf16_t* __restrict__ y, int H) {
    __shared__ float red[4];
    __shared__ int tok_s, pos_s;
    const int r = blockIdx.x;
    if (threadIdx.x < 64) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (!sampled) {
            for (int c = threadIdx.x; c < chunks; c += 64) argmax_merge(best, bi, pval[r * chunks + c], pidx[r * chunks + c]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
        }
        if (threadIdx.x == 0) {
            int t = (bi == 0x7fffffff) ? 0 : bi;
            if (sampled) {                 // the token comes from sample_row_kernel; it joins the sequence's history
                t = sampled[r];
                const int n = hist_len[r];
                hist[(int64_t)r * hist_stride + n] = t;
                hist_len[r] = n + 1;
            }
            const int pos = tok_pos[r] + 1;
            // ring row = steps since the loop started (from the positions, so every step's launch has the same arguments)
            argmax_out[r] = t; ring_out[(int64_t)(pos - 1 - ring_pos0[r]) * ring_rows + r] = t; tokens[r] = t;
            tok_pos[r] = pos; seq_pos[r] = pos;          // (one token per sequence: row r IS sequence r)
            tok_s = t; pos_s = pos;
        }
    }
    __syncthreads();
    const int tok = tok_s, pos = pos_s;
    const bool use_pos = pos_emb && pos < max_seq;
    float* xr = x + (int64_t)r * H;
    const int H4 = H >> 2;
    f32x4 v[NORM_ROW_MAXCH], ww[NORM_ROW_MAXCH], bb[NORM_ROW_MAXCH];
#pragma unroll
    for (int c = 0; c < NORM_ROW_MAXCH; c++) {
        const int j = threadIdx.x + c * 256;
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < H4) {
            const bf16x4 e = *(const bf16x4*)(emb + fm_index(tok, j * 4, H));
            bf16x4 pe = {};
            if (use_pos) pe = *(const bf16x4*)(pos_emb + fm_index(pos, j * 4, H));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float f = (float)e[k];
                if (use_pos) f += (float)pe[k];
                if (mult != 0.f) f *= mult;
                v[c][k] = f;
            }
            *(f32x4*)(xr + j * 4) = v[c];
            ww[c] = *(const f32x4*)(w + j * 4);
            if (b) bb[c] = *(const f32x4*)(b + j * 4);
        }
    }
    norm_row_finish<bf16_t>(v, ww, bb, b, eps, y, r, H, red);
}

// ---------------------------------------------------------------------------------------
// MoE routing (moe.go:63-92): softmax over E, top-k by descending prob (ties: lower expert
// index first — the reference's sort.Slice is unstable, this is the documented tie rule),
// weights renormalised over the chosen k.  One wave per token, E <= 64.
// ---------------------------------------------------------------------------------------
// routing of one token by one wave (moe.go:63-103): softmax over E <= 64 experts, top-k by descending probability
// (ties: lower expert index), weights renormalised over the chosen k
__device__ __forceinline__ void moe_route_row(const float* __restrict__ logits_row, int E, int top_k, int lane,
                                              int32_t* __restrict__ ids_row, float* __restrict__ w_row) {
    const float v = (lane < E) ? logits_row[lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = (lane < E) ? expf(v - mx) : 0.f;
    const float s = wave_sum(e);
    float prob = (lane < E) ? e / s : -1.f;
    // rank of this lane's probability: larger first, ties to the lower index — the order the reference's repeated
    // argmax visits them (moe.go:75-92).  64 register broadcasts instead of top_k dependent butterfly reductions.
    int rank = 0;
    const int pbits = __builtin_bit_cast(int, prob);
#pragma unroll 8
    for (int j = 0; j < E; j++) {       // (lanes >= E hold -1: they never outrank an expert)
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, j));
        rank += (pj > prob || (pj == prob && j < lane)) ? 1 : 0;
    }
    float wsum = 0.f;
    for (int k = 0; k < top_k; k++) {     // moe.go:89-92: fp32 sum in rank order
        const int src = __ffsll((unsigned long long)__ballot(rank == k)) - 1;
        wsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, src));
    }
    if (rank < top_k && lane < E) {
        ids_row[rank] = lane;
        w_row[rank] = prob / wsum;        // moe.go:103
    }
}
__global__ __launch_bounds__(256) void moe_route_kernel(const float* __restrict__ router_logits, int ld,
                                                        int rows, int E, int top_k,
                                                        int32_t* __restrict__ expert_ids,   // [rows][k]
                                                        float* __restrict__ expert_w) {     // [rows][k]
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    moe_route_row(router_logits + (int64_t)r * ld, E, top_k, lane, expert_ids + (int64_t)r * top_k, expert_w + (int64_t)r * top_k);
}

// Decode ("dense-masked" MoE form, gemm.h GemmArgs::moe_gate): the routing result as a dense [rows][E] matrix of
// renormalised weights, 0 for the experts a token did not pick.  One wave per token, many workgroups.
__global__ __launch_bounds__(256) void moe_gate_kernel(const float* __restrict__ router_logits, int ld, int rows, int E,
                                                       int top_k, float* __restrict__ gate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* logits_row = router_logits + (int64_t)r * ld;
    const float v = (lane < E) ? logits_row[lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = (lane < E) ? expf(v - mx) : 0.f;
    const float s = wave_sum(e);
    const float prob = (lane < E) ? e / s : -1.f;
    int rank = 0;                                   // as moe_route_row: larger first, ties to the lower index
    const int pbits = __builtin_bit_cast(int, prob);
#pragma unroll 8
    for (int j = 0; j < E; j++) {       // (lanes >= E hold -1: they never outrank an expert)
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, j));
        rank += (pj > prob || (pj == prob && j < lane)) ? 1 : 0;
    }
    float wsum = 0.f;
    for (int k = 0; k < top_k; k++) {               // moe.go:89-92: fp32 sum in rank order
        const int src = __ffsll((unsigned long long)__ballot(rank == k)) - 1;
        wsum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(pbits, src));
    }
    if (lane < E) gate[(int64_t)r * E + lane] = rank < top_k ? prob / wsum : 0.f;      // moe.go:103
}
// W_down of all experts concatenated along K, fragment-major [H_pad][E * I]: block (nt, e * I/32 + kb) = block (nt, kb) of
// expert e's [H][I] matrix.  grid (H / 16, E); one-off at finalize.
__global__ __launch_bounds__(256) void moe_cat_down_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int H, int I, int E) {
    const int nt = blockIdx.x, e = blockIdx.y, kbs = I >> 5;
    const bf16x8* s8 = (const bf16x8*)(src + ((int64_t)e * H * I + (int64_t)nt * kbs * 512));
    bf16x8* d8 = (bf16x8*)(dst + ((int64_t)nt * E * kbs + (int64_t)e * kbs) * 512);
    for (int i = threadIdx.x; i < kbs * 64; i += 256) d8[i] = s8[i];
}

// Small batches (decode: rows * top_k <= MOE_PLAN_MAX_PAIRS): routing, the per-expert histogram, the segment scan with
// the grouped GEMM's tile map, and the scatter of (token, expert) pairs into expert order — moe_route / hist / scan /
// scatter above and the memset of the counters — as ONE single-workgroup launch.
constexpr int MOE_PLAN_MAX_PAIRS = 2048, MOE_PLAN_MAX_E = 64;
__global__ __launch_bounds__(1024) void moe_plan_kernel(const float* __restrict__ router_logits, int ld, int rows, int E,
                                                       int top_k, int BM, int32_t* __restrict__ expert_ids,
                                                       float* __restrict__ expert_w, int32_t* __restrict__ seg_start,
                                                       int32_t* __restrict__ tile_map, int32_t* __restrict__ n_mtiles,
                                                       int32_t* __restrict__ perm_token, int32_t* __restrict__ slot_of) {
    __shared__ int32_t ids[MOE_PLAN_MAX_PAIRS];
    __shared__ float gw[MOE_PLAN_MAX_PAIRS];
    __shared__ int counts[MOE_PLAN_MAX_E], cursor[MOE_PLAN_MAX_E];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pairs = rows * top_k, nw = blockDim.x >> 6;
    if (threadIdx.x < MOE_PLAN_MAX_E) counts[threadIdx.x] = 0;
    for (int r = wave; r < rows; r += nw)         // one wave per token; results stay in LDS for the phases below
        moe_route_row(router_logits + (int64_t)r * ld, E, top_k, lane, ids + r * top_k, gw + r * top_k);
    __syncthreads();
    for (int i = threadIdx.x; i < pairs; i += blockDim.x) {
        const int e = ids[i];
        expert_ids[i] = e;
        expert_w[i] = gw[i];
        atomicAdd(&counts[e], 1);
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive scans over the experts (E <= 64: one lane each): first row and first m-tile of every segment
        const int c = lane < E ? counts[lane] : 0;
        const int nt_e = (c + BM - 1) / BM;
        int row = c, tile = nt_e;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int yr = __shfl_up(row, o, 64), yt = __shfl_up(tile, o, 64);
            if (lane >= o) { row += yr; tile += yt; }
        }
        const int row0 = row - c, tile0 = tile - nt_e;
        if (lane < E) {
            seg_start[lane] = row0;
            cursor[lane] = row0;
            for (int t = 0; t < nt_e; t++) {
                const int r = t * BM;
                tile_map[4 * (tile0 + t)] = lane; tile_map[4 * (tile0 + t) + 1] = row0 + r;
                tile_map[4 * (tile0 + t) + 2] = (c - r < BM) ? (c - r) : BM; tile_map[4 * (tile0 + t) + 3] = 0;
            }
        }
        if (lane == 63) { seg_start[E] = row; *n_mtiles = tile; }
    }
    __syncthreads();
    // pairs of one expert keep their (token, rank) order: positions are handed out by a per-expert scan, not by atomics,
    // so the row order inside a segment — and with it every output bit — is the same on every run
    for (int e = wave; e < E; e += nw) {
        int base = cursor[e];
        for (int i0 = 0; i0 < pairs; i0 += 64) {
            const int i = i0 + lane;
            const bool mine = i < pairs && ids[i] == e;
            const unsigned long long b = __ballot(mine);
            if (mine) {
                const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
                perm_token[pos] = i / top_k;
                slot_of[i] = pos;
            }
            base += __popcll(b);
        }
    }
}

// out[t] (+)= resid_mult * sum_rank w[t][rank] * eo[slot_of(t,rank)]   in rank order
// (moe.go:117-119 accumulates in expert-rank order; generic_model.go:383-389 adds the residual)
__global__ void moe_combine_kernel(const float* __restrict__ eo, const int32_t* __restrict__ slot_of,
                                   const float* __restrict__ w, int top_k, float resid_mult,
                                   float* __restrict__ x, int H, int accumulate_into_x) {
    const int t = blockIdx.x;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < top_k; k++) {
            const int sl = slot_of[(int64_t)t * top_k + k];
            acc += w[(int64_t)t * top_k + k] * eo[(int64_t)sl * H + j];
        }
        float* xp = x + (int64_t)t * H + j;
        if (accumulate_into_x) *xp = *xp + (resid_mult != 0.f ? resid_mult * acc : acc);
        else *xp = acc;
    }
}

// Counting sort of the (token, rank) pairs by expert, three small launches:
//   moe_hist_kernel    per-expert pair counts (atomics on E counters; zeroed by the host per call)
//   moe_scan_kernel    one wave: segment starts, the per-expert placement cursors, and the m-tile table of
//                      the grouped GEMM launch: tile_map[i] = {expert, row0, nrows, 0}, *n_mtiles
//   moe_scatter_kernel each pair takes the next row of its expert's segment (atomic cursor).  The row a pair
//                      lands on can differ from run to run, the VALUES cannot: every pair is computed
//                      independently and moe_combine_kernel reads them back through slot_of in rank order.
// prefill-sized batches: per-workgroup counts in LDS first (E <= 64), so the global counters see one atomic per
// (workgroup, expert) instead of one per pair (32768 pairs on 32 addresses: 57 -> ~10 us)
constexpr int MOE_PAIRS_PER_WG = 1024;
__global__ __launch_bounds__(256) void moe_hist_kernel(const int32_t* __restrict__ expert_ids, int n_pairs, int32_t* __restrict__ counts) {
    __shared__ int lc[64];
    if (threadIdx.x < 64) lc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MOE_PAIRS_PER_WG / 256; j++) {
        const int i = blockIdx.x * MOE_PAIRS_PER_WG + j * 256 + threadIdx.x;
        if (i < n_pairs) atomicAdd(&lc[expert_ids[i]], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64 && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
}
// prefill: rows of the fragment-major activation matrix copied into expert order (dst row i = src row perm[i]), one
// workgroup per 16 destination rows; a wave moves whole 1-KiB operand blocks (coalesced stores, 16-byte gathers)
__global__ __launch_bounds__(256) void moe_gather_fm_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ perm,
                                                            bf16_t* __restrict__ dst, int n_rows, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 16 + (lane & 15), fg = lane >> 4;
    const int nks = K >> 5;
    if (r >= n_rows) return;                       // rows past the end of the last tile are never stored by the GEMM
    const int tok = perm[r];
    const bf16_t* s = src + ((int64_t)(tok >> 4) * nks * 64 + (tok & 15) + 16 * fg) * 8;
    bf16_t* d = dst + ((int64_t)blockIdx.x * nks * 64 + lane) * 8;
    for (int ks = wave; ks < nks; ks += 4)
        *(bf16x8*)(d + (int64_t)ks * 512) = *(const bf16x8*)(s + (int64_t)ks * 512);
}
__global__ __launch_bounds__(64) void moe_scan_kernel(const int32_t* __restrict__ counts, int E, int BM,
                                                      int32_t* __restrict__ seg_start, int32_t* __restrict__ cursor,
                                                      int32_t* __restrict__ tile_map, int32_t* __restrict__ n_mtiles) {
    // exclusive scans over the experts (E <= 64: one lane each): first row and first m-tile of every segment
    const int lane = threadIdx.x;
    const int c = lane < E ? counts[lane] : 0;
    const int nt_e = (c + BM - 1) / BM;
    int row = c, tile = nt_e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int yr = __shfl_up(row, o, 64), yt = __shfl_up(tile, o, 64);
        if (lane >= o) { row += yr; tile += yt; }
    }
    const int row0 = row - c, tile0 = tile - nt_e;
    if (lane < E) {
        seg_start[lane] = row0;
        cursor[lane] = row0;
        for (int t = 0; t < nt_e; t++) {
            const int r = t * BM;
            tile_map[4 * (tile0 + t)] = lane; tile_map[4 * (tile0 + t) + 1] = row0 + r;
            tile_map[4 * (tile0 + t) + 2] = (c - r < BM) ? (c - r) : BM; tile_map[4 * (tile0 + t) + 3] = 0;
        }
    }
    if (lane == 63) { seg_start[E] = row; *n_mtiles = tile; }
}
__global__ __launch_bounds__(256) void moe_scatter_kernel(const int32_t* __restrict__ expert_ids, int n_pairs, int top_k,
                                   int32_t* __restrict__ cursor, int32_t* __restrict__ perm_token,
                                   int32_t* __restrict__ slot_of) {
    // rank inside the workgroup from LDS atomics, one global reservation per (workgroup, expert).  The row order inside
    // an expert's segment varies from run to run; no output depends on it (rows are independent, the combine sums a
    // token's slots in rank order through slot_of).
    __shared__ int lc[64], base[64];
    if (threadIdx.x < 64) lc[threadIdx.x] = 0;
    __syncthreads();
    int e[MOE_PAIRS_PER_WG / 256], r[MOE_PAIRS_PER_WG / 256];
#pragma unroll
    for (int j = 0; j < MOE_PAIRS_PER_WG / 256; j++) {
        const int i = blockIdx.x * MOE_PAIRS_PER_WG + j * 256 + threadIdx.x;
        e[j] = 0; r[j] = 0;
        if (i < n_pairs) { e[j] = expert_ids[i]; r[j] = atomicAdd(&lc[e[j]], 1); }
    }
    __syncthreads();
    if (threadIdx.x < 64 && lc[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], lc[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MOE_PAIRS_PER_WG / 256; j++) {
        const int i = blockIdx.x * MOE_PAIRS_PER_WG + j * 256 + threadIdx.x;
        if (i < n_pairs) {
            const int pos = base[e[j]] + r[j];
            perm_token[pos] = i / top_k;
            slot_of[i] = pos;
        }
    }
}

// ---------------------------------------------------------------------------------------
// load-time converters (generic_loader.go:642-663 dtype expansion, tensor.go:112-125 Transpose)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_as_f32(const void* p, int dtype, int64_t i) {
    if (dtype == 0) return ((const float*)p)[i];
    if (dtype == 1) return (float)((const bf16_t*)p)[i];
    return (float)((const _Float16*)p)[i];
}
// dst rows [dn0, dn0+N) x cols [0, K) of a [.][K] matrix  <-  the sub-block rows [sn0, sn0+N) x cols [sk0, sk0+K)
// of the logical [SN][SK] source, stored IN_OUT ([SK][SN]) or OUT_IN ([SN][SK]).  The sub-block form is how a
// tensor-parallel rank takes its shard while the tensor is converted; FM selects the fragment-major
// destination layout (bf16 MFMA operands), else row-major.
struct Slice2D { int64_t SN, SK, sn0, sk0, dn0; };
template <typename DT, bool FM>
__global__ void convert_2d_kernel(const void* __restrict__ src, int dtype, int transpose_in,
                                  DT* __restrict__ dst, int64_t N, int64_t K, Slice2D sl) {
    __shared__ float tile[32][33];
    const int64_t n0 = (int64_t)blockIdx.y * 32, k0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    if (transpose_in) {
        for (int r = ty; r < 32; r += 8) {
            const int64_t k = k0 + r, n = n0 + tx;
            tile[r][tx] = (k < K && n < N) ? ld_as_f32(src, dtype, (sl.sk0 + k) * sl.SN + sl.sn0 + n) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int64_t n = n0 + r, k = k0 + tx;
            if (n < N && k < K) dst[FM ? fm_index(sl.dn0 + n, k, K) : (sl.dn0 + n) * K + k] = (DT)tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int64_t n = n0 + r, k = k0 + tx;
            if (n < N && k < K)
                dst[FM ? fm_index(sl.dn0 + n, k, K) : (sl.dn0 + n) * K + k] =
                    (DT)ld_as_f32(src, dtype, (sl.sn0 + n) * sl.SK + sl.sk0 + k);
        }
    }
}
__global__ void convert_1d_kernel(const void* __restrict__ src, int dtype, float* __restrict__ dst, int64_t n,
                                  int64_t src_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ld_as_f32(src, dtype, src_off + i);
}

// tensor parallel: x[m][:] += alpha * (part[m][:] (+ bias))  after the all-reduce of a row-parallel projection
__global__ void axpy_rows_kernel(float* __restrict__ x, const float* __restrict__ part, const float* __restrict__ bias,
                                 float alpha, int64_t rows, int H) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * H) return;
    f32x4 v = *(const f32x4*)(part + i);
    if (bias) v += *(const f32x4*)(bias + (i % H));
    f32x4 o = *(f32x4*)(x + i);
    o += alpha * v;
    *(f32x4*)(x + i) = o;
}
// local tensor-parallel emulation (several shard models of one process on one GPU): dst = sum_r src[r]
struct PtrList8 { const float* p[8]; };
__global__ void sum_bufs_kernel(float* __restrict__ dst, PtrList8 srcs, int n_src, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = *(const f32x4*)(srcs.p[0] + i);
    for (int r = 1; r < n_src; r++) s += *(const f32x4*)(srcs.p[r] + i);
    *(f32x4*)(dst + i) = s;
}
// dst row r (of K elements) = src row idx[r], or zeros when idx[r] < 0
template <typename DT>
__global__ void gather_rows_kernel(const DT* __restrict__ src, const int32_t* __restrict__ idx,
                                   DT* __restrict__ dst, int64_t K) {
    const int64_t r = blockIdx.x;
    const int s = idx[r];
    for (int64_t k = threadIdx.x; k < K; k += blockDim.x)
        dst[r * K + k] = (s >= 0) ? src[(int64_t)s * K + k] : (DT)0.f;
}

}  // namespace nvl
