// sample.h — tensor.SampleWithHistory (purego/tensor/sampling.go:33-102) on the device, so that only token ids
// (not B x V logits) cross PCIe after a forward pass.  One 1024-thread workgroup per logits row; the row stays in
// L2 (V x 4 B <= 513 KB) over the ~20 passes below.  Every reduction runs in a fixed order and the histograms use
// integer atomics, so a call is reproducible bit for bit.  The only random input, rand.Float32() (sampling.go:205),
// is an argument: the Go host keeps drawing it from math/rand in sequence order, its RNG stream is unchanged.
//
// What cannot be bit-identical to the reference: its sums run sequentially in index (or sorted) order over V
// elements, ours in a fixed tree order; sort.Slice is unstable, so which of several EQUAL probabilities survive a
// top-k / top-p cut is unspecified there (we keep the lowest indices).  tests/test_sampling_gpu.py states the
// resulting tolerance (5e-4 of probability mass: the error bound of the reference's own sequential fp32 sums).
#pragma once

#include "common.h"

namespace nvl {

struct SampleArgs {
    const float* logits; int64_t ld;     // [rows][ld]: final logits (LogitsScaling already applied)
    float* work;                          // [rows][V] scratch: the probability vector
    int32_t* cnt;                         // [rows][V] zero-filled scratch for the repetition counts; left zero-filled
    const int32_t* hist;                  // token histories, concatenated
    const int32_t* hist_off;              // [rows + 1] offsets into hist
    const float* uniforms;                // [rows] the rand.Float32() draw of each row
    int32_t* out;                         // [rows] sampled token id
    float* probs_out; int64_t ldp;        // optional [rows][ldp]: the final distribution (parity tap)
    int V, top_k;
    float temperature, top_p, rep_penalty;
};

constexpr int SAMPLE_THREADS = 1024, SAMPLE_WAVES = SAMPLE_THREADS / 64;
constexpr double SAMPLE_FX = 1099511627776.0;   // 2^40: probabilities as fixed point for order-independent mass sums

struct SampleShared {
    float red[SAMPLE_WAVES];
    unsigned long long hist[256];
    unsigned long long above;          // count / mass strictly above the current radix prefix
    unsigned long long ties;           // count / mass of elements equal to the final threshold
    uint32_t digit;
    int seg_cnt[SAMPLE_WAVES];
    float seg_sum[SAMPLE_WAVES];
    int idx;
};

__device__ __forceinline__ float sample_block_sum(float v, SampleShared& sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SAMPLE_WAVES; w++) t += sh.red[w];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float sample_block_max(float v, SampleShared& sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh.red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = sh.red[0];
#pragma unroll
    for (int w = 1; w < SAMPLE_WAVES; w++) t = fmaxf(t, sh.red[w]);
    __syncthreads();
    return t;
}

// Radix select over the bit patterns of the (non-negative) probabilities, most significant byte first.
// COUNT mode: the threshold T with  #(p > T) < target <= #(p >= T);   MASS mode: the same with probability mass
// (2^-40 fixed point).  Leaves sh.above = amount strictly above T and sh.ties = amount equal to T.  If the target is
// never reached T = 0 (everything is kept), which is what the reference's "cutoff = len" does.
template <bool MASS>
__device__ uint32_t sample_radix_select(const float* w, int V, unsigned long long target, SampleShared& sh) {
    uint32_t pref = 0, mask = 0;
    if (threadIdx.x == 0) sh.above = 0;
    for (int d = 24; d >= 0; d -= 8) {
        for (int b = threadIdx.x; b < 256; b += SAMPLE_THREADS) sh.hist[b] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < V; j += SAMPLE_THREADS) {
            const float p = w[j];
            const uint32_t bits = __float_as_uint(p);
            if ((bits & mask) == pref)
                atomicAdd(&sh.hist[(bits >> d) & 255u], MASS ? (unsigned long long)((double)p * SAMPLE_FX) : 1ull);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long acc = sh.above;
            int bin = 255;
            for (; bin > 0; bin--) {
                if (acc + sh.hist[bin] >= target) break;
                acc += sh.hist[bin];
            }
            sh.above = acc; sh.ties = sh.hist[bin]; sh.digit = (uint32_t)bin;
        }
        __syncthreads();
        pref |= sh.digit << d;
        mask |= 255u << d;
    }
    return pref;
}

// Keep p > T and the first `need` elements with p == T in index order; zero the rest (topKFiltering / topPFiltering's
// "result[indexed[i].idx] = indexed[i].prob", sampling.go:150-153,189-192).
__device__ void sample_apply_threshold(float* w, int V, uint32_t T, int need, SampleShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = ((V + SAMPLE_WAVES - 1) / SAMPLE_WAVES + 63) & ~63;
    const int j0 = wave * seg, j1 = min(V, j0 + seg);
    int c = 0;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        const bool tie = j < j1 && __float_as_uint(w[j]) == T;
        c += __popcll(__ballot(tie));
    }
    if (lane == 0) sh.seg_cnt[wave] = c;
    __syncthreads();
    int ord = 0;
    for (int k = 0; k < wave; k++) ord += sh.seg_cnt[k];
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        const uint32_t bits = j < j1 ? __float_as_uint(w[j]) : 0u;
        const bool tie = j < j1 && bits == T;
        const unsigned long long b = __ballot(tie);
        const int my = ord + __popcll(b & ((1ull << lane) - 1ull));
        if (j < j1 && !(bits > T || (tie && my < need))) w[j] = 0.f;
        ord += __popcll(b);
    }
    __syncthreads();
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_row_kernel(SampleArgs a) {
    __shared__ SampleShared sh;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    const float* lg = a.logits + (int64_t)row * a.ld;
    float* w = a.work + (int64_t)row * V;
    int32_t* cnt = a.cnt + (int64_t)row * V;

    // ---- copy + repetition penalty (sampling.go:43-68): count x3 for the last 10 history tokens ----
    for (int j = tid; j < V; j += SAMPLE_THREADS) w[j] = lg[j];
    const int h0 = a.hist_off[row], hn = a.hist_off[row + 1] - h0;
    if (a.rep_penalty != 1.0f && hn > 0) {
        for (int i = tid; i < hn; i += SAMPLE_THREADS) {
            const int t = a.hist[h0 + i];
            if (t >= 0 && t < V) atomicAdd(&cnt[t], i >= hn - 10 ? 3 : 1);
        }
        __syncthreads();
        for (int i = tid; i < hn; i += SAMPLE_THREADS) {
            const int t = a.hist[h0 + i];
            if (t < 0 || t >= V) continue;
            const int c = atomicExch(&cnt[t], 0);          // exactly one thread per distinct token sees the count
            if (c > 0) {
                const float pen = a.rep_penalty * (float)c;
                const float l = w[t];
                w[t] = l > 0.f ? l / pen : l * pen;
            }
        }
    }
    __syncthreads();

    // ---- temperature (:71-75) and softmax (:105-127: exp in float64, narrowed) ----
    const bool scale = a.temperature > 0.f && a.temperature != 1.0f;
    float mx = -INFINITY;
    for (int j = tid; j < V; j += SAMPLE_THREADS) {
        float l = w[j];
        if (scale) { l = l / a.temperature; w[j] = l; }
        mx = fmaxf(mx, l);
    }
    mx = sample_block_max(mx, sh);
    float part = 0.f;
    for (int j = tid; j < V; j += SAMPLE_THREADS) {
        const float e = (float)exp((double)(w[j] - mx));
        w[j] = e;
        part += e;
    }
    const float denom = sample_block_sum(part, sh);
    for (int j = tid; j < V; j += SAMPLE_THREADS) w[j] = w[j] / denom;
    __syncthreads();

    // ---- top-k (:78-80, :130-156) ----
    if (a.top_k > 0 && a.top_k < V) {
        const uint32_t T = sample_radix_select<false>(w, V, (unsigned long long)a.top_k, sh);
        const int need = a.top_k - (int)sh.above;
        sample_apply_threshold(w, V, T, need, sh);
    }
    // ---- top-p (:83-85, :159-195): the shortest descending prefix whose mass reaches p ----
    if (a.top_p < 1.0f) {
        // p <= 0: the reference's first cumulative sum already reaches p -> only the largest survives (target: any mass)
        const unsigned long long target = max(1ull, (unsigned long long)((double)fmaxf(a.top_p, 0.f) * SAMPLE_FX));
        const uint32_t T = sample_radix_select<true>(w, V, target, sh);
        int need = 0x7fffffff;                               // T == 0: the mass never reached p, keep everything
        if (T != 0u) {
            const double t = (double)__uint_as_float(T), above = (double)sh.above / SAMPLE_FX;
            const double nties = (double)sh.ties / ((double)(unsigned long long)(t * SAMPLE_FX));
            double jn = ceil(((double)a.top_p - above) / t);
            if (jn < 1.0) jn = 1.0;
            if (jn > nties + 0.5) jn = nties + 0.5;
            need = (int)jn;
        }
        sample_apply_threshold(w, V, T, need, sh);
    }

    // ---- renormalise (:88-96) ----
    part = 0.f;
    for (int j = tid; j < V; j += SAMPLE_THREADS) part += w[j];
    const float total = sample_block_sum(part, sh);
    if (total > 0.f)
        for (int j = tid; j < V; j += SAMPLE_THREADS) w[j] = w[j] / total;
    __syncthreads();
    if (a.probs_out)
        for (int j = tid; j < V; j += SAMPLE_THREADS) a.probs_out[(int64_t)row * a.ldp + j] = w[j];

    // ---- multinomial (:198-217): first index whose running sum (index order) reaches r = u * sum ----
    const int seg = ((V + SAMPLE_WAVES - 1) / SAMPLE_WAVES + 63) & ~63;
    const int j0 = wave * seg, j1 = min(V, j0 + seg);
    auto wave_scan = [&](float v) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float y = __shfl_up(v, o, 64);
            if (lane >= o) v += y;
        }
        return v;
    };
    float carry = 0.f;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        const float x = wave_scan(j < j1 ? w[j] : 0.f);
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) sh.seg_sum[wave] = carry;
    if (tid == 0) sh.idx = 0x7fffffff;
    __syncthreads();
    float off = 0.f, grand = 0.f;
    for (int k = 0; k < SAMPLE_WAVES; k++) {
        if (k == wave) off = grand;
        grand += sh.seg_sum[k];
    }
    const float r = a.uniforms[row] * grand;
    carry = off;
    for (int base = j0; base < j1; base += 64) {
        const int j = base + lane;
        const float x = wave_scan(j < j1 ? w[j] : 0.f);
        const unsigned long long hit = __ballot(j < j1 && carry + x >= r);
        if (hit) {
            if (lane == 0) atomicMin(&sh.idx, base + (int)__ffsll((long long)hit) - 1);
            break;
        }
        carry += __shfl(x, 63, 64);
    }
    __syncthreads();
    if (tid == 0) a.out[row] = sh.idx == 0x7fffffff ? V - 1 : sh.idx;
}

}  // namespace nvl
