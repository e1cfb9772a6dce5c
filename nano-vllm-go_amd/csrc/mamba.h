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
