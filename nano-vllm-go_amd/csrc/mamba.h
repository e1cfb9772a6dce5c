// mamba.h — Mamba2 (selective state-space) blocks of the Granite-4 hybrid models: the element / scan kernels between
// the two projections of Mamba2Layer.Forward (purego/tensor/mamba2.go:74-181).  The projections themselves
// (in_proj, out_proj) are the GEMMs of gemm.h.
//
//   in_proj GEMM -> proj [tokens][gate EH | xBC conv_dim | dt nh]   (fp32)
//   mamba_conv_kernel      causal conv1d + bias + SiLU on xBC, softplus(dt + bias)        mamba2.go:107-131, 183-254, 360-376
//   mamba_scan_kernel      per (sequence, head): state = abar*state + dt*B*u; y = C·state + D*u   mamba2.go:256-351
//   mamba_gate_norm_kernel y *= SiLU(gate); RMS norm over EH (eps 1e-5); * norm weight -> out_proj operand   :135-170
//   out_proj GEMM (+ residual)
//
// State: the reference keeps SSMState ON THE LAYER (mamba2.go:29-30), i.e. one state for whatever sequence ran last —
// wrong for more than one live sequence.  Here every KV slot (sequence) owns its state,
// [slot][mamba layer][head][head_dim][state] fp32 in HBM, zeroed when the sequence (re)starts at position 0.
// The reference carries NO convolution state across calls (ConvCache is never used): a decode call convolves its one
// token with zeros.  That is mirrored — the conv window never reaches before the call's first token of a sequence.
#pragma once
#include "common.h"

namespace nvl {

struct MambaArgs {
    const float* proj;       // [tokens][P]  P = EH + conv_dim + nh
    int P, EH, conv_dim, nh, hd, ss, ng, K;
    const float* conv_w;     // [conv_dim][K]
    const float* conv_b;     // [conv_dim] or NULL
    const float* a_log;      // [nh] or NULL
    const float* Dskip;      // [nh] or NULL
    const float* dt_bias;    // [nh] or NULL
    const float* norm_w;     // [EH] or NULL
    float* xbc;              // [tokens][conv_dim]  SiLU(conv(xBC))
    float* delta;            // [tokens][nh]
    float* y;                // [tokens][EH]
    float* state;            // this layer's states: slot s at state + s * state_slot_stride
    int64_t state_slot_stride;
    const int32_t* tok_pos;        // per token: absolute position
    const int32_t* tok_seq;        // per token: index of its sequence in the batch
    const int32_t* seq_tok_start;  // per sequence
    const int32_t* seq_len;
    const int32_t* seq_pos;        // first position of the sequence in this call
    const int32_t* seq_slot;       // KV slot of the sequence (blk_table in slab mode)
    // A history longer than max_batch_tokens is prefilled in chunks by the runner (nvllm.hip runner_impl); the reference
    // runs it as ONE Forward whose convolution window spans the whole history.  For those chunks only, the raw xBC rows of
    // the last K-1 tokens of the previous chunk are kept per (slot, Mamba2 layer) and stand in for the zero padding:
    // chain 0 = off (every other call: the reference's zero padding), 1 = first chunk (save the tail), 2 = later chunk
    // (use the saved tail, then save the new one).
    float* tail;                   // this layer's tails: slot s at tail + s * tail_slot_stride, [K-1][conv_dim]
    int64_t tail_slot_stride;
    int chain;
    // chunk-parallel SSD scan (mamba_ssd_kernel MODE 1 / 2): per (sequence, chunk, head) state scratch and chunk decay
    float* ssd_z;                  // [sequences][max_chunks][nh][hd][ss]
    float* ssd_decay;              // [sequences][max_chunks][nh]
};

__device__ __forceinline__ float silu_ref(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float softplus_ref(float x) { return x > 20.f ? x : log1pf(__expf(x)); }    // mamba2.go:370-376

// grid (tokens), 256 threads over channels (+ the nh dt values)
__global__ __launch_bounds__(256) void mamba_conv_kernel(MambaArgs p) {
    const int t = blockIdx.x;
    const int j = p.tok_pos[t] - p.seq_pos[p.tok_seq[t]];      // tokens of this sequence before t IN THIS CALL
    const float* row = p.proj + (int64_t)t * p.P + p.EH;        // xBC of token t
    for (int c = threadIdx.x; c < p.conv_dim; c += blockDim.x) {
        float sum = 0.f;
        for (int k = 0; k < p.K; k++) {                         // out[t] = sum_k x[t - (K-1) + k] * w[c][k], zero before the call
            const int back = p.K - 1 - k;
            if (back <= j) sum = fmaf(row[c - (int64_t)back * p.P], p.conv_w[(int64_t)c * p.K + k], sum);
            else if (p.chain == 2)     // continued chunk: the window reaches into the previous chunk's last K-1 tokens
                sum = fmaf(p.tail[(int64_t)p.seq_slot[p.tok_seq[t]] * p.tail_slot_stride + (int64_t)(p.K - 1 - (back - j)) * p.conv_dim + c],
                           p.conv_w[(int64_t)c * p.K + k], sum);
        }
        if (p.conv_b) sum += p.conv_b[c];
        p.xbc[(int64_t)t * p.conv_dim + c] = silu_ref(sum);
    }
    for (int h = threadIdx.x; h < p.nh; h += blockDim.x) {
        float v = p.proj[(int64_t)t * p.P + p.EH + p.conv_dim + h];
        if (p.dt_bias) v += p.dt_bias[h];
        p.delta[(int64_t)t * p.nh + h] = softplus_ref(v);
    }
}

// grid (sequences), 256 threads over channels: after the convolution of a chunk of a chained prefill (chain >= 1), keep the
// raw xBC rows of the sequence's last K-1 tokens for the next chunk (a chunk shorter than K-1 keeps the newest old rows).
__global__ __launch_bounds__(256) void mamba_tail_kernel(MambaArgs p) {
    const int seq = blockIdx.x;
    const int t0 = p.seq_tok_start[seq], n = p.seq_len[seq];
    float* tl = p.tail + (int64_t)p.seq_slot[seq] * p.tail_slot_stride;
    const bool have_old = p.chain == 2 && p.seq_pos[seq] > 0;
    for (int c = threadIdx.x; c < p.conv_dim; c += blockDim.x) {
        float old[8], nw[8];
        for (int i = 0; i < p.K - 1; i++) old[i] = have_old ? tl[(int64_t)i * p.conv_dim + c] : 0.f;
        for (int i = 0; i < p.K - 1; i++) {
            const int rel = n - (p.K - 1) + i;              // token of this call that lands in tail row i
            nw[i] = rel >= 0 ? p.proj[(int64_t)(t0 + rel) * p.P + p.EH + c] : old[i + n];
        }
        for (int i = 0; i < p.K - 1; i++) tl[(int64_t)i * p.conv_dim + c] = nw[i];
    }
}

// grid (nh, sequences), 256 threads = head_dim x SG state groups; PER = ss / SG states per thread in registers.
// Sequential over the sequence's tokens of this call (the recurrence); everything else is parallel.
template <int PER>
__global__ __launch_bounds__(256) void mamba_scan_kernel(MambaArgs p) {
    const int h = blockIdx.x, seq = blockIdx.y;
    const int SG = 256 / p.hd;                       // host guarantees hd | 256, SG a power of two <= 64, SG * PER == ss
    const int d = threadIdx.x / SG, sg = threadIdx.x % SG;
    const int t0 = p.seq_tok_start[seq], n = p.seq_len[seq];
    float* st = p.state + (int64_t)p.seq_slot[seq] * p.state_slot_stride + ((int64_t)h * p.hd + d) * p.ss + sg * PER;
    float s[PER];
    if (p.seq_pos[seq] == 0) {                       // a sequence that (re)starts: zero state (ResetState, generic_model.go:285-292)
#pragma unroll
        for (int i = 0; i < PER; i++) s[i] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < PER; i++) s[i] = st[i];
    }
    const float A = p.a_log ? -__expf(p.a_log[h]) : 0.f;          // mamba2.go:286
    const float Dh = p.Dskip ? p.Dskip[h] : 0.f;
    int g = h * p.ng / p.nh;                                      // :301-304
    if (g >= p.ng) g = p.ng - 1;
    for (int i = 0; i < n; i++) {
        const int t = t0 + i;
        const float dt = p.delta[(int64_t)t * p.nh + h];
        const float abar = p.a_log ? __expf(A * dt) : 1.0f;       // :287
        const float* xr = p.xbc + (int64_t)t * p.conv_dim;
        const float u = xr[h * p.hd + d];
        const float* Bt = xr + p.EH + g * p.ss + sg * PER;
        const float* Ct = Bt + p.ng * p.ss;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const float v = fmaf(abar, s[q], (dt * Bt[q]) * u);   // :325
            s[q] = v;
            acc = fmaf(Ct[q], v, acc);                            // :333-336
        }
        for (int msk = 1; msk < SG; msk <<= 1) acc += __shfl_xor(acc, msk, 64);
        if (sg == 0) p.y[(int64_t)t * p.EH + h * p.hd + d] = acc + Dh * u;      // :339-345
    }
#pragma unroll
    for (int i = 0; i < PER; i++) st[i] = s[i];
}

// ------------------------------------------------------------------------------------------
// Chunked ("SSD") form of the same recurrence for prefill-sized calls in the bf16 mode: one workgroup of four waves per
// (sequence, head) walks its tokens in chunks of Q = 64 and does each chunk with MFMA products instead of 64 dependent
// steps (mamba2.go:256-351 is the sequential statement; this is its algebra).  With a_j = A dt_j, s_t = sum_{j<=t} a_j
// (inclusive, inside the chunk) and S_in the state entering the chunk:
//     state_t = exp(s_t) S_in + sum_{j<=t} exp(s_t - s_j) dt_j  u_j (x) B_j
//     y_t     = exp(s_t) (C_t . S_in) + sum_{j<=t} [exp(s_t - s_j) dt_j (C_t . B_j)] u_j + D u_t
//     S_out   = exp(s_Q) S_in + sum_j [exp(s_Q - s_j) dt_j u_j] (x) B_j
// i.e. four products per chunk, all v_mfma_f32_16x16x32_bf16 with fp32 accumulation:
//     G = C B^T (Q x Q, k = state);  Y_intra = (G o L) U (k = token);  Y_inter = C S_in^T (k = state);  S += (U o w)^T B (k = token)
// Every decay factor is exp of a NON-POSITIVE fp32 number (s decreases), the state lives in fp32 accumulators across the
// chunks and is rounded to bf16 only as the Y_inter operand; u, B, C and the masked decay matrix G o L are rounded to bf16 as
// MFMA operands — the same kind of rounding point the projections around this kernel already have; the
// fp32 parity mode keeps mamba_scan_kernel.  Operands sit in LDS K-contiguous ([row][k], rows padded by 16 bytes: the
// fragment reads are conflict-free), B and U additionally transposed.  A ragged last chunk runs with dt = 0 for the missing
// tokens: they neither decay nor feed the state, and their rows are not stored.
// ------------------------------------------------------------------------------------------
template <int HD, int SS>
constexpr int mamba_ssd_lds_bytes() {
    return 2 * ((64 * (SS + 8)) * 2 + SS * (64 + 8) + 2 * HD * (64 + 8) + 64 * (64 + 8) + HD * (SS + 8)) + 64 * 4 * 3;
}
// MODE 0: one workgroup per (sequence, head) walks the chunks in order (the state stays in its accumulators).
// With few (sequence, head) pairs and long prompts that leaves most of the chip idle, so the chunks can run in parallel:
// MODE 1: grid.z = chunks; each workgroup computes only ITS chunk's contribution to the state from a zero state,
//         Z_c = (U o w)^T B, and the chunk's total decay exp(s_Q) -> scratch;
// mamba_ssd_prefix_kernel: per (sequence, head) the short recurrence S_{c+1} = decay_c S_c + Z_c over the chunks (elementwise),
//         leaving S_c (the state ENTERING chunk c) in the scratch slot of chunk c and the final state in the slot state;
// MODE 2: grid.z = chunks; each workgroup loads the state entering its chunk and computes the chunk's outputs.
template <int HD, int SS, int MODE = 0>
__global__ __launch_bounds__(256) void mamba_ssd_kernel(MambaArgs p) {
    constexpr int Q = 64, DT = HD / 16, NT = SS / 16, NP = 4 / DT, NTW = NT / NP;     // d tiles, state tiles, state parts, state tiles per wave
    constexpr int LC = SS + 8, LQ = Q + 8;                                             // padded row lengths (elements)
    static_assert(HD == 32 || HD == 64, "head_dim 32 / 64");
    static_assert(NT % NP == 0, "state tiles must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    bf16_t* Cs = (bf16_t*)smem_m;                 // [Q][LC]    C_t[n]
    bf16_t* Bs = Cs + Q * LC;                     // [Q][LC]    B_j[n]
    bf16_t* Bt = Bs + Q * LC;                     // [SS][LQ]   B^T
    bf16_t* Ut = Bt + SS * LQ;                    // [HD][LQ]   U^T
    bf16_t* Uw = Ut + HD * LQ;                    // [HD][LQ]   (w_j u_j)^T
    bf16_t* Ms = Uw + HD * LQ;                    // [Q][LQ]    G o L
    bf16_t* Sb = Ms + Q * LQ;                     // [HD][LC]   bf16 copy of the state entering the chunk
    float* s_cum = (float*)(Sb + HD * LC);        // [Q] inclusive prefix sums of a_j
    float* s_dt = s_cum + Q;                      // [Q] dt_j
    float* s_w = s_dt + Q;                        // [Q] exp(s_Q - s_j) dt_j
    const int h = blockIdx.x, seq = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int t0 = p.seq_tok_start[seq], n = p.seq_len[seq];
    if (MODE != 0 && (int)blockIdx.z * Q >= n) return;          // (uniform: before any barrier)
    const float A = p.a_log ? -__expf(p.a_log[h]) : 0.f;
    const float Dh = p.Dskip ? p.Dskip[h] : 0.f;
    int g = h * p.ng / p.nh;
    if (g >= p.ng) g = p.ng - 1;
    // ---- the state: wave w owns d tile w % DT and state tiles [(w / DT) * NTW, +NTW) in fp32 accumulators ----
    const int sd = wave % DT, sn0 = (wave / DT) * NTW;
    float* st = p.state + (int64_t)p.seq_slot[seq] * p.state_slot_stride + (int64_t)h * HD * SS;
    // chunk-parallel modes: this (sequence, head, chunk)'s scratch slot
    float* zs = MODE != 0 ? p.ssd_z + (((int64_t)seq * gridDim.z + blockIdx.z) * p.nh + h) * (HD * SS) : nullptr;
    f32x4 sacc[NTW];
    const bool fresh = p.seq_pos[seq] == 0;
#pragma unroll
    for (int j = 0; j < NTW; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int64_t o = (int64_t)(16 * sd + 4 * fg + r) * SS + 16 * (sn0 + j) + fr;
            sacc[j][r] = MODE == 1 ? 0.f : (MODE == 2 ? zs[o] : (fresh ? 0.f : st[o]));
        }

    for (int c0 = MODE == 0 ? 0 : (int)blockIdx.z * Q; c0 < (MODE == 0 ? n : (int)blockIdx.z * Q + 1); c0 += Q) {
        const int nv = min(Q, n - c0);                          // valid tokens of this chunk
        __syncthreads();                                        // the previous chunk's LDS readers are done
        // ---- decay bookkeeping: wave 0 scans a_j = A dt_j ----
        if (wave == 0) {
            const float dt = lane < nv ? p.delta[(int64_t)(t0 + c0 + lane) * p.nh + h] : 0.f;
            float s = A * dt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const float v = __shfl_up(s, o, 64); if (lane >= o) s += v; }
            s_cum[lane] = s; s_dt[lane] = dt;
            const float sq = __shfl(s, 63, 64);
            s_w[lane] = __expf(sq - s) * dt;
        }
        // ---- the state entering the chunk, as the bf16 operand of Y_inter ----
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < NTW; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) Sb[(16 * sd + 4 * fg + r) * LC + 16 * (sn0 + j) + fr] = (bf16_t)sacc[j][r];
        }
        __syncthreads();
        // ---- stage u, B, C of the chunk (fp32 rows of xbc -> bf16, K-contiguous; B and u also transposed) ----
        for (int i = tid; i < Q * (SS / 4); i += 256) {
            const int t = i / (SS / 4), q4 = (i - t * (SS / 4)) * 4;
            f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f}, c = b;
            if (t < nv) {
                const float* xr = p.xbc + (int64_t)(t0 + c0 + t) * p.conv_dim + p.EH + g * SS + q4;
                b = *(const f32x4*)xr;
                if (MODE != 1) c = *(const f32x4*)(xr + p.ng * SS);
            }
            bf16x4 bb, cc;
#pragma unroll
            for (int r = 0; r < 4; r++) { bb[r] = (bf16_t)b[r]; cc[r] = (bf16_t)c[r]; if (MODE != 2) Bt[(q4 + r) * LQ + t] = (bf16_t)b[r]; }
            if (MODE != 1) { *(bf16x4*)(Bs + t * LC + q4) = bb; *(bf16x4*)(Cs + t * LC + q4) = cc; }
        }
        for (int i = tid; i < Q * (HD / 4); i += 256) {
            const int t = i / (HD / 4), d4 = (i - t * (HD / 4)) * 4;
            f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < nv) u = *(const f32x4*)(p.xbc + (int64_t)(t0 + c0 + t) * p.conv_dim + h * HD + d4);
            const float w = s_w[t];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (MODE != 1) Ut[(d4 + r) * LQ + t] = (bf16_t)u[r];
                if (MODE != 2) Uw[(d4 + r) * LQ + t] = (bf16_t)(u[r] * w);
            }
        }
        __syncthreads();
        if (MODE != 1) {
            // ---- G = C B^T for this wave's 16 tokens (row tile `wave`), causal column tiles only; M = G o L -> LDS ----
            {
                const int tt = wave;
                float srow[4];
#pragma unroll
                for (int r = 0; r < 4; r++) srow[r] = s_cum[16 * tt + 4 * fg + r];
                for (int jt = 0; jt < 4; jt++) {
                    f32x4 gacc = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (jt <= tt) {
#pragma unroll
                        for (int ks = 0; ks < SS / 32; ks++) {
                            const bf16x8 af = *(const bf16x8*)(Cs + (16 * tt + fr) * LC + ks * 32 + fg * 8);      // A: rows t, k = state
                            const bf16x8 bf = *(const bf16x8*)(Bs + (16 * jt + fr) * LC + ks * 32 + fg * 8);      // B: cols j, k = state
                            gacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, gacc, 0, 0, 0);
                        }
                    }
                    const int j = 16 * jt + fr;                       // this lane's column (token j), rows t = 16 tt + 4 fg + r
                    const float sj = s_cum[j], dj = s_dt[j];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int t = 16 * tt + 4 * fg + r;
                        const float v = (j <= t) ? gacc[r] * (__expf(srow[r] - sj) * dj) : 0.f;
                        Ms[t * LQ + j] = (bf16_t)v;
                    }
                }
            }
            __syncthreads();
            // ---- y rows of this wave: Y_intra = M U (k = token) and Y_inter = C S_in^T (k = state) ----
            {
                const int tt = wave;
                f32x4 yi[DT], ye[DT];
#pragma unroll
                for (int d = 0; d < DT; d++) { yi[d] = f32x4{0.f, 0.f, 0.f, 0.f}; ye[d] = yi[d]; }
#pragma unroll
                for (int ks = 0; ks < Q / 32; ks++) {
                    if (ks * 32 > 16 * tt + 15) continue;             // tokens beyond this tile's last row: M is zero there
                    const bf16x8 af = *(const bf16x8*)(Ms + (16 * tt + fr) * LQ + ks * 32 + fg * 8);
#pragma unroll
                    for (int d = 0; d < DT; d++) {
                        const bf16x8 bf = *(const bf16x8*)(Ut + (16 * d + fr) * LQ + ks * 32 + fg * 8);
                        yi[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, yi[d], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < SS / 32; ks++) {
                    const bf16x8 af = *(const bf16x8*)(Cs + (16 * tt + fr) * LC + ks * 32 + fg * 8);
#pragma unroll
                    for (int d = 0; d < DT; d++) {
                        const bf16x8 bf = *(const bf16x8*)(Sb + (16 * d + fr) * LC + ks * 32 + fg * 8);
                        ye[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, ye[d], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int t = 16 * tt + 4 * fg + r;
                    if (t >= nv) continue;
                    const float es = __expf(s_cum[t]);
                    const float* ur = p.xbc + (int64_t)(t0 + c0 + t) * p.conv_dim + h * HD;
                    float* yr = p.y + (int64_t)(t0 + c0 + t) * p.EH + h * HD;
#pragma unroll
                    for (int d = 0; d < DT; d++) yr[16 * d + fr] = yi[d][r] + es * ye[d][r] + Dh * ur[16 * d + fr];
                }
            }
        }
        if (MODE != 2) {
            // ---- S_out = exp(s_Q) S_in + (U o w)^T B  (k = token) ----
            const float eq = __expf(s_cum[Q - 1]);
#pragma unroll
            for (int j = 0; j < NTW; j++) sacc[j] *= eq;
#pragma unroll
            for (int ks = 0; ks < Q / 32; ks++) {
                const bf16x8 af = *(const bf16x8*)(Uw + (16 * sd + fr) * LQ + ks * 32 + fg * 8);              // A: rows d, k = token
#pragma unroll
                for (int j = 0; j < NTW; j++) {
                    const bf16x8 bf = *(const bf16x8*)(Bt + (16 * (sn0 + j) + fr) * LQ + ks * 32 + fg * 8);   // B: cols n, k = token
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, sacc[j], 0, 0, 0);
                }
            }
            if (MODE == 1 && tid == 0) p.ssd_decay[((int64_t)seq * gridDim.z + blockIdx.z) * p.nh + h] = eq;
        }
    }
    if (MODE == 2) return;
    float* dst = MODE == 1 ? zs : st;
#pragma unroll
    for (int j = 0; j < NTW; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(int64_t)(16 * sd + 4 * fg + r) * SS + 16 * (sn0 + j) + fr] = sacc[j][r];
}

// the chunk recurrence of the chunk-parallel form: grid (nh, sequences, state_elems / 1024), one f32x4 of the head's hd x ss
// state per thread; the chunks are walked four at a time so that four slots' loads are in flight per round trip
__global__ __launch_bounds__(256) void mamba_ssd_prefix_kernel(MambaArgs p, int max_chunks, int state_elems) {
    const int h = blockIdx.x, seq = blockIdx.y;
    const int e = (blockIdx.z * 256 + threadIdx.x) * 4;
    if (e >= state_elems) return;
    const int nc = (p.seq_len[seq] + 63) / 64;
    float* st = p.state + (int64_t)p.seq_slot[seq] * p.state_slot_stride + (int64_t)h * state_elems + e;
    f32x4 run = p.seq_pos[seq] == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)st;
    const int64_t slot0 = (int64_t)seq * max_chunks * p.nh + h;          // slot of chunk c: slot0 + c * nh
    for (int c0 = 0; c0 < nc; c0 += 4) {
        f32x4 z[4];
        float dc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = min(c0 + i, nc - 1);
            z[i] = *(const f32x4*)(p.ssd_z + (slot0 + (int64_t)c * p.nh) * state_elems + e);
            dc[i] = p.ssd_decay[slot0 + (int64_t)c * p.nh];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (c0 + i >= nc) break;
            *(f32x4*)(p.ssd_z + (slot0 + (int64_t)(c0 + i) * p.nh) * state_elems + e) = run;      // the state entering chunk c
            run = dc[i] * run + z[i];
        }
    }
    *(f32x4*)st = run;
}

// grid (tokens), 256 threads: y *= SiLU(gate); rms over EH; * norm weight; written as the out_proj GEMM's operand
template <typename ActT>
__global__ __launch_bounds__(256) void mamba_gate_norm_kernel(MambaArgs p, ActT* __restrict__ out) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    const float* gate = p.proj + (int64_t)t * p.P;
    float* yr = p.y + (int64_t)t * p.EH;
    float ss = 0.f;
    for (int c = threadIdx.x; c < p.EH; c += 256) {
        const float v = yr[c] * silu_ref(gate[c]);
        yr[c] = v;
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rms = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.EH + 1e-5f);      // eps fixed in mamba2.go:147
    for (int c = threadIdx.x * 4; c < p.EH; c += 1024) {       // EH % 4 == 0 (host)
        f32x4 v = *(const f32x4*)(yr + c) * rms;
        if (p.norm_w) v *= *(const f32x4*)(p.norm_w + c);
        act_store4<ActT>(out, t, c, p.EH, v);
    }
}

}  // namespace nvl
