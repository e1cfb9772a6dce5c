// tp_p2p.h — tensor-parallel all-reduce (sum) over the xGMI mesh by direct peer stores, with the residual add — and, in
// decode, the RMSNorm that follows — fused.  ONE launch per all-reduce, replayable from a hipGraph.
//
// The reference has no tensor parallelism (nanovllm/config.go:61 is inert).  The sharded forward (nvllm.hip) leaves, after
// every row-parallel projection (O, FFN-down), an fp32 partial [tokens][H] per rank that must be summed over the T ranks
// and added to the residual stream.  xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU): a ring is
// per-link bound, direct stores use every link at once.
//
//   one-shot (decode-sized payloads, latency-bound): every rank stores its partial into EVERY peer's inbox slot
//       (T-1 links in parallel), signals, waits for the T arrivals, sums the T inbox slots in rank order and adds
//       alpha * sum into x.  1 exchange.  NORM: the same pass also emits the deferred-RMSNorm operand of the next projection,
//       xn_raw = bf16(x * w_norm) (fragment-major) + the per-tile sums of x^2 (gemm.h GemmArgs::rs_in) — what the
//       single-GPU path's residual projections produce in their epilogue — so a tensor-parallel decode step needs no norm
//       launches either: per layer QKV, attention, O, all-reduce, FFN-up, FFN-down, all-reduce = 7 launches (5 unsharded).
//   two-shot (prefill-sized payloads, bandwidth-bound): reduce-scatter + all-gather on the mesh.  Rank o owns chunk o:
//       every rank stores chunk o of its partial into o's inbox; o sums the T copies and stores the reduced chunk into
//       every peer's result buffer; every rank adds alpha * result into x.  Each link carries 1/T of the payload per
//       phase.  2 exchanges, still one launch (two waits).
//
// Payload type PT: bf16 in the bf16 product mode (fp32 accumulation, one rounding of the partial before it crosses the
// link and — two-shot — one of the reduced value; halves the link bytes), fp32 in the fp32 parity mode.  Every rank adds
// the SAME rounded values in the SAME rank order, so all ranks hold bit-identical residual streams.  Peer stores are
// 16 bytes per lane (8 bf16 / 4 fp32): the comm buffers are uncached, and a scalar 2-byte store to such memory costs
// ~12x the per-byte time of a 16-byte one (MI355X guide, store table).
//
// Synchronisation.  All state lives on the device, so a captured launch replays correctly:
//   * call counters n1 (one-shot calls so far) / n2 (two-shot) in the rank's own buffer: a call reads its counter at the
//     start (parity = n & 1 selects the inbox half, (n >> 1) + 1 is how many calls of that parity there have been) and
//     the last workgroup to finish increments it;
//   * arrival counters ctr[parity][kind] in the RECEIVER's buffer, one arrival per sending rank per call: the workgroups of
//     a sender count themselves on a local word after a release fence, the last one signals every peer with ONE
//     system-scope atomic — so the expected count is ((n >> 1) + 1) * T whatever the grids were;
//   * receivers poll with bounded spins (s_sleep; on timeout an error word is set, every later wait falls through and the
//     host call fails instead of hanging the GPU — nvl_tp_p2p_rearm clears the group after such a failure).
// Inboxes are double-buffered by call parity: a rank can start call n+1 while a peer still reads call n, and cannot start
// n+2 before that peer has sent n+1.  A launch spins inside the kernel, so every workgroup of it must be resident: grids
// are capped at 512 workgroups of 256 threads (the chip holds 2048).
// The comm buffers are uncached device allocations exported with hipIpcGetMemHandle (one process per GPU).
#pragma once
#include "common.h"

namespace nvl {

struct P2PArgs {
    int T, rank;
    int oneshot;                          // 1: one-shot, 0: reduce-scatter + all-gather
    int64_t count, chunk;                 // elements of the payload; elements per owner (two-shot)
    const float* part;                    // this rank's fp32 partial [count]
    float* x;                             // residual stream [count]
    float alpha;
    char* peer[8];                        // comm buffer base of every rank (peer[rank] = own)
    int64_t off_in1, off_in2, off_res;    // byte offsets inside a comm buffer
    int64_t in1_stride, in2_stride, res_stride;            // elements per (parity) / per source slot
    int spin_limit;
    // NORM (one-shot only): deferred RMSNorm of the completed rows
    int H;                                // row length (count = rows * H)
    const float* nrm_w;                   // [H]
    bf16_t* nrm_xn;                       // xn_raw, fragment-major [rows_pad16][H]
    float* rs_out;                        // [rows][H / rs_cols] sums of x^2
    int rs_cols;                          // 16, or 8 (the half-tile layout of <= 16-row batches: gemm.h rs_half)
};
// header of a comm buffer (1 KiB): words on their own 64-byte lines
constexpr int P2P_OFF_CTR = 0;            // ctr[parity][kind] at (parity * 3 + kind) * 64: written by the peers
constexpr int P2P_OFF_CALLS = 512;        // own: n1 at +0, n2 at +64
constexpr int P2P_OFF_DONE = 640;         // own: workgroup counters of the launch phases, 3 x 64 bytes
constexpr int P2P_OFF_ERR = 960;

__device__ __forceinline__ unsigned long long* p2p_ctr(const P2PArgs& p, int r, int parity, int kind) {
    return (unsigned long long*)(p.peer[r] + P2P_OFF_CTR + (int64_t)(parity * 3 + kind) * 64);
}
template <typename PT> struct P2PVec;
template <> struct P2PVec<float> { static constexpr int N = 4; typedef f32x4 V; };
template <> struct P2PVec<bf16_t> { static constexpr int N = 8; typedef bf16x8 V; };

// this workgroup's peer stores are out; the LAST workgroup of the launch to get here signals every rank (one arrival per
// sender per call) and re-arms the local counter for the next launch / replay
__device__ __forceinline__ void p2p_signal_all(const P2PArgs& p, int parity, int kind) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope
    __syncthreads();
    __shared__ int last;
    unsigned* done = (unsigned*)(p.peer[p.rank] + P2P_OFF_DONE + kind * 64);
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == gridDim.x - 1);
        if (last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last && threadIdx.x < (unsigned)p.T)
        __hip_atomic_fetch_add(p2p_ctr(p, (int)threadIdx.x, parity, kind), 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_wait(const P2PArgs& p, int parity, int kind, unsigned long long target) {
    if (threadIdx.x == 0) {
        const unsigned long long* c = p2p_ctr(p, p.rank, parity, kind);
        volatile int* err = (volatile int*)(p.peer[p.rank] + P2P_OFF_ERR);
        int it = 0;
        // (once a wait has timed out the group is dead: later waits fall through at once, the host reports the error)
        while (*err == 0 && __hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
            __builtin_amdgcn_s_sleep(32);
            if (++it > p.spin_limit) { *err = 1; break; }      // give up, do not hang
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// the call is over on this rank: the last workgroup bumps the call counter (after every workgroup has read it)
__device__ __forceinline__ void p2p_finish(const P2PArgs& p, int which) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* done = (unsigned*)(p.peer[p.rank] + P2P_OFF_DONE + 2 * 64 + 32);       // (its own word: the phase counters are re-armed above)
        const unsigned t = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long* n = (unsigned long long*)(p.peer[p.rank] + P2P_OFF_CALLS + which * 64);
            __hip_atomic_fetch_add(n, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename PT>
__device__ __forceinline__ typename P2PVec<PT>::V p2p_pack(const float* src) {
    typename P2PVec<PT>::V o;
    const f32x4 a = *(const f32x4*)src;
    if constexpr (P2PVec<PT>::N == 4) {
        o = a;
    } else {
        const f32x4 b = *(const f32x4*)(src + 4);
#pragma unroll
        for (int k = 0; k < 4; k++) { o[k] = (bf16_t)a[k]; o[4 + k] = (bf16_t)b[k]; }
    }
    return o;
}

// ONE launch = one all-reduce.  grid <= 512 workgroups of 256 threads (all resident: the kernel spins), count % 8 == 0.
template <typename PT, bool NORM>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs p) {
    constexpr int VN = P2PVec<PT>::N;
    typedef typename P2PVec<PT>::V Vec;
    const int which = p.oneshot ? 0 : 1;
    const unsigned long long n = __hip_atomic_load((const unsigned long long*)(p.peer[p.rank] + P2P_OFF_CALLS + which * 64),
                                                   __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    const int parity = (int)(n & 1);
    const unsigned long long target = ((n >> 1) + 1) * (unsigned long long)p.T;
    const int64_t stride = (int64_t)gridDim.x * 256 * VN;
    if (p.oneshot) {
        // send: my partial -> inbox1[parity][my rank] of every rank (my own included: everyone sums the same rounded values)
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VN; i < p.count; i += stride) {
            const Vec o = p2p_pack<PT>(p.part + i);
            for (int r = 0; r < p.T; r++)
                *(Vec*)((PT*)(p.peer[r] + p.off_in1) + ((int64_t)parity * p.T + p.rank) * p.in1_stride + i) = o;
        }
        p2p_signal_all(p, parity, 0);
        p2p_wait(p, parity, 0, target);
        // apply: x += alpha * sum_r inbox1[parity][r] in rank order (+ the deferred norm of the completed rows)
        const PT* in = (const PT*)(p.peer[p.rank] + p.off_in1) + (int64_t)parity * p.T * p.in1_stride;
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < p.count; i += (int64_t)gridDim.x * 256 * 8) {
            // 8 columns per thread whatever the payload type: two adjacent threads cover a 16-column norm tile
            float s[8];
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = 0.f;
            for (int r = 0; r < p.T; r++) {
                const PT* src = in + (int64_t)r * p.in1_stride + i;
#pragma unroll
                for (int h = 0; h < 8 / VN; h++) {
                    const Vec v = *(const Vec*)(src + h * VN);
#pragma unroll
                    for (int k = 0; k < VN; k++) s[h * VN + k] += (float)v[k];
                }
            }
            f32x4 x0 = *(f32x4*)(p.x + i), x1 = *(f32x4*)(p.x + i + 4);
#pragma unroll
            for (int k = 0; k < 4; k++) { x0[k] += p.alpha * s[k]; x1[k] += p.alpha * s[4 + k]; }
            *(f32x4*)(p.x + i) = x0; *(f32x4*)(p.x + i + 4) = x1;
            if constexpr (NORM) {
                const int64_t m = i / p.H;
                const int nn = (int)(i - m * p.H);
                act_store4<bf16_t>(p.nrm_xn, m, nn, p.H, x0 * *(const f32x4*)(p.nrm_w + nn));
                act_store4<bf16_t>(p.nrm_xn, m, nn + 4, p.H, x1 * *(const f32x4*)(p.nrm_w + nn + 4));
                float ss = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) ss += x0[k] * x0[k] + x1[k] * x1[k];
                if (p.rs_cols == 8) {
                    p.rs_out[m * (p.H >> 3) + (nn >> 3)] = ss;
                } else {                                       // the pair of threads of a 16-column tile (H % 16 == 0: same row)
                    ss += __shfl_xor(ss, 1, 64);
                    if (((nn >> 3) & 1) == 0) p.rs_out[m * (p.H >> 4) + (nn >> 4)] = ss;
                }
            }
        }
        p2p_finish(p, 0);
        return;
    }
    // ---- two-shot ----
    // reduce-scatter send: element i goes to its owner o = i / chunk, into inbox2[parity][my rank][i - o*chunk] on rank o
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VN; i < p.count; i += stride) {
        const int o = (int)(i / p.chunk);                     // chunk % 8 == 0: the elements of a vector share an owner
        *(Vec*)((PT*)(p.peer[o] + p.off_in2) + ((int64_t)parity * p.T + p.rank) * p.in2_stride + (i - (int64_t)o * p.chunk)) = p2p_pack<PT>(p.part + i);
    }
    p2p_signal_all(p, parity, 1);
    p2p_wait(p, parity, 1, target);
    {   // owner: sum the T copies of my chunk in rank order, store the reduced chunk into every rank's result buffer
        const int64_t c0 = (int64_t)p.rank * p.chunk;
        const int64_t nmine = p.count - c0 < p.chunk ? (p.count - c0 > 0 ? p.count - c0 : 0) : p.chunk;
        const PT* in = (const PT*)(p.peer[p.rank] + p.off_in2) + (int64_t)parity * p.T * p.in2_stride;
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VN; i < nmine; i += stride) {
            float s[VN];
#pragma unroll
            for (int k = 0; k < VN; k++) s[k] = 0.f;
            for (int r = 0; r < p.T; r++) {
                const Vec v = *(const Vec*)(in + (int64_t)r * p.in2_stride + i);
#pragma unroll
                for (int k = 0; k < VN; k++) s[k] += (float)v[k];
            }
            Vec o;
#pragma unroll
            for (int k = 0; k < VN; k++) o[k] = (PT)s[k];
            for (int r = 0; r < p.T; r++)
                *(Vec*)((PT*)(p.peer[r] + p.off_res) + (int64_t)parity * p.res_stride + c0 + i) = o;
        }
    }
    p2p_signal_all(p, parity, 2);
    p2p_wait(p, parity, 2, target);
    {   // everyone: x += alpha * result
        const PT* res = (const PT*)(p.peer[p.rank] + p.off_res) + (int64_t)parity * p.res_stride;
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VN; i < p.count; i += stride) {
            const Vec v = *(const Vec*)(res + i);
#pragma unroll
            for (int h = 0; h < VN / 4; h++) {
                f32x4 xv = *(f32x4*)(p.x + i + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; k++) xv[k] += p.alpha * (float)v[4 * h + k];
                *(f32x4*)(p.x + i + 4 * h) = xv;
            }
        }
    }
    p2p_finish(p, 1);
}

}  // namespace nvl
