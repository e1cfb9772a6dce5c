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
    float* work;                          // [rows][round_up(V, 4)] scratch: the probability vector (16-byte aligned rows)
    int32_t* cnt;                         // [rows][V] zero-filled scratch for the repetition counts; left zero-filled
    const int32_t* hist;                  // token histories, concatenated
    const int32_t* hist_off;              // [rows + 1] offsets into hist, or NULL:
    int64_t hist_stride;                  //   then row r's history is hist[r * hist_stride .. + hist_len[r]) (device-kept
    const int32_t* hist_len;              //   histories of the fused sampled decode loop)
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

// elementwise pass over a row in 16-byte pieces (the row base is 16-byte aligned: work rows have a stride of
// round_up(V, 4) floats) plus a scalar tail: `f4` maps a float4, `f1` a float
template <typename F4, typename F1>
__device__ __forceinline__ void sample_for_each(float* w, int V, F4 f4, F1 f1) {
    const int V4 = V >> 2;
    f32x4* w4 = (f32x4*)w;
#pragma unroll 4
    for (int j = threadIdx.x; j < V4; j += SAMPLE_THREADS) w4[j] = f4(w4[j]);
    for (int j = (V4 << 2) + threadIdx.x; j < V; j += SAMPLE_THREADS) w[j] = f1(w[j]);
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_row_kernel(SampleArgs a) {
    __shared__ SampleShared sh;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    const float* lg = a.logits + (int64_t)row * a.ld;
    float* w = a.work + (int64_t)row * ((V + 3) & ~3);
    int32_t* cnt = a.cnt + (int64_t)row * V;

    // ---- copy + repetition penalty (sampling.go:43-68): count x3 for the last 10 history tokens ----
    if ((a.ld & 3) == 0) {
        const f32x4* lg4 = (const f32x4*)lg;
        f32x4* w4 = (f32x4*)w;
#pragma unroll 4
        for (int j = tid; j < (V >> 2); j += SAMPLE_THREADS) w4[j] = lg4[j];
        for (int j = (V & ~3) + tid; j < V; j += SAMPLE_THREADS) w[j] = lg[j];
    } else {
#pragma unroll 8
        for (int j = tid; j < V; j += SAMPLE_THREADS) w[j] = lg[j];
    }
    const int64_t h0 = a.hist_off ? (int64_t)a.hist_off[row] : (int64_t)row * a.hist_stride;
    const int hn = a.hist_off ? a.hist_off[row + 1] - a.hist_off[row] : a.hist_len[row];
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

    // ---- temperature (:71-75) and softmax (:105-127) ----
    const bool scale = a.temperature > 0.f && a.temperature != 1.0f;
    const float temp = a.temperature;
    float mx = -INFINITY;
    sample_for_each(w, V,
        [&](f32x4 v) { if (scale) v = f32x4{v[0] / temp, v[1] / temp, v[2] / temp, v[3] / temp};
                       mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); return v; },
        [&](float v) { if (scale) v = v / temp; mx = fmaxf(mx, v); return v; });
    mx = sample_block_max(mx, sh);
    float part = 0.f;
    // (the reference narrows a float64 exp: <= 1 ulp from expf, far inside the tolerance)
    sample_for_each(w, V,
        [&](f32x4 v) { v = f32x4{expf(v[0] - mx), expf(v[1] - mx), expf(v[2] - mx), expf(v[3] - mx)};
                       part += (v[0] + v[1]) + (v[2] + v[3]); return v; },
        [&](float v) { v = expf(v - mx); part += v; return v; });
    const float denom = sample_block_sum(part, sh);
    sample_for_each(w, V, [&](f32x4 v) { return f32x4{v[0] / denom, v[1] / denom, v[2] / denom, v[3] / denom}; },
                    [&](float v) { return v / denom; });
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
    sample_for_each(w, V, [&](f32x4 v) { part += (v[0] + v[1]) + (v[2] + v[3]); return v; }, [&](float v) { part += v; return v; });
    const float total = sample_block_sum(part, sh);
    if (total > 0.f)
        sample_for_each(w, V, [&](f32x4 v) { return f32x4{v[0] / total, v[1] / total, v[2] / total, v[3] / total}; },
                        [&](float v) { return v / total; });
    __syncthreads();
    if (a.probs_out) {
#pragma unroll 8
        for (int j = tid; j < V; j += SAMPLE_THREADS) a.probs_out[(int64_t)row * a.ldp + j] = w[j];
    }

    // ---- multinomial (:198-217): first index whose running sum (index order) reaches r = u * sum ----
    // every wave owns a contiguous segment; a lane takes 4 consecutive entries per step (one 16-byte load)
    const int seg = ((V + SAMPLE_WAVES - 1) / SAMPLE_WAVES + 255) & ~255;
    const int j0 = wave * seg, j1 = min(V, j0 + seg);
    auto load4 = [&](int j) {                   // entries j .. j+3 (zero beyond the segment)
        if (j + 3 < j1) return *(const f32x4*)(w + j);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4; k++) if (j + k < j1) v[k] = w[j + k];
        return v;
    };
    auto lane_scan = [&](f32x4 v, float& lane_total) {    // inclusive prefix inside the lane, then across the wave
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        float t = v[3];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float y = __shfl_up(t, o, 64);
            if (lane >= o) t += y;
        }
        lane_total = t;                                      // inclusive over lanes 0..lane
        const float excl = t - v[3];
        return f32x4{excl + v[0], excl + v[1], excl + v[2], excl + v[3]};
    };
    float carry = 0.f;
    for (int base = j0; base < j1; base += 256) {
        float t;
        (void)lane_scan(load4(base + lane * 4), t);
        carry += __shfl(t, 63, 64);
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
    for (int base = j0; base < j1; base += 256) {
        const int j = base + lane * 4;
        float t;
        const f32x4 c = lane_scan(load4(j), t);
        int k = 4;                                           // first of my 4 entries whose running sum reaches r
        for (int q = 3; q >= 0; q--) if (j + q < j1 && carry + c[q] >= r) k = q;
        const unsigned long long hit = __ballot(k < 4);
        if (hit) {
            const int first = (int)__ffsll((long long)hit) - 1;
            if (lane == first) atomicMin(&sh.idx, j + k);
            break;
        }
        carry += __shfl(t, 63, 64);
    }
    __syncthreads();
    if (tid == 0) a.out[row] = sh.idx == 0x7fffffff ? V - 1 : sh.idx;
}

}  // namespace nvl
