// tp_p2p.h — tensor-parallel all-reduce (sum) over the xGMI mesh by direct peer stores, with the residual add fused.
//
// The reference has no tensor parallelism (nanovllm/config.go:61 is inert).  The sharded forward (nvllm.hip) leaves, after
// every row-parallel projection (O, FFN-down), an fp32 partial [tokens][H] per rank that must be summed over the T ranks
// and added to the residual stream.  xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU): a ring is
// per-link bound, direct stores use every link at once.
//
//   one-shot (decode-sized payloads, latency-bound): every rank stores its partial into EVERY peer's inbox slot
//       (T-1 links in parallel), signals, then sums the T inbox slots in rank order and adds alpha * sum into x.
//       1 exchange, 2 launches.
//   two-shot (prefill-sized payloads, bandwidth-bound): reduce-scatter + all-gather on the mesh.  Rank o owns chunk o:
//       every rank stores chunk o of its partial into o's inbox; o sums the T copies and stores the reduced chunk into
//       every peer's result buffer; every rank adds alpha * result into x.  Each link carries 1/T of the payload per
//       phase.  2 exchanges, 3 launches.
//
// Payload type PT: bf16 in the bf16 product mode (fp32 accumulation, one rounding of the partial before it crosses the
// link and — two-shot — one of the reduced value; halves the link bytes), fp32 in the fp32 parity mode.  Every rank adds
// the SAME rounded values in the SAME rank order, so all ranks hold bit-identical residual streams.
//
// Synchronisation: monotonic 64-bit arrival counters in the receiver's buffer, bumped by one system-scope release atomic
// per sending workgroup after a system fence; receivers poll with bounded spins (s_sleep; on timeout an error word is set
// and the forward call fails instead of hanging the GPU).  Inboxes are double-buffered by call parity: a rank can start
// call n+1 while a peer still reads call n, and cannot start n+2 before that peer has sent n+1.
// The comm buffers are uncached device allocations exported with hipIpcGetMemHandle (one process per GPU).
#pragma once
#include "common.h"

namespace nvl {

struct P2PArgs {
    int T, rank, parity;
    int64_t count, chunk;                 // elements of the payload; elements per owner (two-shot)
    const float* part;                    // this rank's fp32 partial [count]
    float* x;                             // residual stream [count]
    float alpha;
    char* peer[8];                        // comm buffer base of every rank (peer[rank] = own)
    int64_t off_ctr, off_err, off_in1, off_in2, off_res;   // byte offsets inside a comm buffer
    int64_t in1_stride, in2_stride, res_stride;            // elements per (parity) / per source slot
    unsigned long long target;            // arrival count to wait for
    int spin_limit;
};
// comm buffer: counters ctr[parity][kind] (kind 0 one-shot, 1 reduce-scatter, 2 all-gather), each on its own 64-byte line
__device__ __forceinline__ unsigned long long* p2p_ctr(const P2PArgs& p, int r, int kind) {
    return (unsigned long long*)(p.peer[r] + p.off_ctr + (int64_t)(p.parity * 3 + kind) * 64);
}
template <typename PT> __device__ __forceinline__ PT p2p_cvt(float v);
template <> __device__ __forceinline__ float p2p_cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t p2p_cvt<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ void p2p_signal_all(const P2PArgs& p, int kind) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: this workgroup's peer stores are out
    __syncthreads();
    if (threadIdx.x < (unsigned)p.T)
        __hip_atomic_fetch_add(p2p_ctr(p, (int)threadIdx.x, kind), 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_wait(const P2PArgs& p, int kind) {
    if (threadIdx.x == 0) {
        const unsigned long long* c = p2p_ctr(p, p.rank, kind);
        volatile int* err = (volatile int*)(p.peer[p.rank] + p.off_err);
        int it = 0;
        // (once a wait has timed out the group is dead: later waits fall through at once, the host reports the error)
        while (*err == 0 && __hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < p.target) {
            __builtin_amdgcn_s_sleep(32);
            if (++it > p.spin_limit) { *err = 1; break; }      // give up, do not hang
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// ---- one-shot -----------------------------------------------------------------------------------------------------
// send: my partial -> inbox1[parity][my rank] of every rank (my own included: everyone sums the same rounded values)
template <typename PT>
__global__ __launch_bounds__(256) void p2p_oneshot_send_kernel(P2PArgs p) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < p.count; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *(const f32x4*)(p.part + i);          // count % 4 == 0 (H % 64 == 0)
        PT o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = p2p_cvt<PT>(v[k]);
        for (int r = 0; r < p.T; r++) {
            PT* dst = (PT*)(p.peer[r] + p.off_in1) + ((int64_t)p.parity * p.T + p.rank) * p.in1_stride + i;
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = o[k];
        }
    }
    p2p_signal_all(p, 0);
}
// receive: wait for T x (sender workgroups) arrivals, x += alpha * sum_r inbox1[parity][r] in rank order
template <typename PT>
__global__ __launch_bounds__(256) void p2p_oneshot_apply_kernel(P2PArgs p) {
    p2p_wait(p, 0);
    const PT* in = (const PT*)(p.peer[p.rank] + p.off_in1) + (int64_t)p.parity * p.T * p.in1_stride;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < p.count; i += (int64_t)gridDim.x * 1024) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < p.T; r++) {
            const PT* src = in + (int64_t)r * p.in1_stride + i;
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] += (float)src[k];
        }
        f32x4 xv = *(f32x4*)(p.x + i);
        xv += p.alpha * s;
        *(f32x4*)(p.x + i) = xv;
    }
}

// ---- two-shot -----------------------------------------------------------------------------------------------------
// reduce-scatter send: element i goes to its owner o = i / chunk, into inbox2[parity][my rank][i - o*chunk] on rank o
template <typename PT>
__global__ __launch_bounds__(256) void p2p_rs_send_kernel(P2PArgs p) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < p.count; i += (int64_t)gridDim.x * 1024) {
        const int o = (int)(i / p.chunk);                     // chunk % 4 == 0: the 4 elements share an owner
        const f32x4 v = *(const f32x4*)(p.part + i);
        PT* dst = (PT*)(p.peer[o] + p.off_in2) + ((int64_t)p.parity * p.T + p.rank) * p.in2_stride + (i - (int64_t)o * p.chunk);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = p2p_cvt<PT>(v[k]);
    }
    p2p_signal_all(p, 1);
}
// owner: wait, sum the T copies of my chunk in rank order, store the reduced chunk into every rank's result buffer
template <typename PT>
__global__ __launch_bounds__(256) void p2p_rs_reduce_bcast_kernel(P2PArgs p) {
    p2p_wait(p, 1);
    const int64_t c0 = (int64_t)p.rank * p.chunk;
    const int64_t n = p.count - c0 < p.chunk ? (p.count - c0 > 0 ? p.count - c0 : 0) : p.chunk;
    const PT* in = (const PT*)(p.peer[p.rank] + p.off_in2) + (int64_t)p.parity * p.T * p.in2_stride;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < p.T; r++) {
            const PT* src = in + (int64_t)r * p.in2_stride + i;
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] += (float)src[k];
        }
        PT o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = p2p_cvt<PT>(s[k]);
        for (int r = 0; r < p.T; r++) {
            PT* dst = (PT*)(p.peer[r] + p.off_res) + (int64_t)p.parity * p.res_stride + c0 + i;
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = o[k];
        }
    }
    p2p_signal_all(p, 2);
}
// everyone: wait for the T owners, x += alpha * result
template <typename PT>
__global__ __launch_bounds__(256) void p2p_ag_apply_kernel(P2PArgs p) {
    p2p_wait(p, 2);
    const PT* res = (const PT*)(p.peer[p.rank] + p.off_res) + (int64_t)p.parity * p.res_stride;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < p.count; i += (int64_t)gridDim.x * 1024) {
        f32x4 xv = *(f32x4*)(p.x + i);
#pragma unroll
        for (int k = 0; k < 4; k++) xv[k] += p.alpha * (float)res[i + k];
        *(f32x4*)(p.x + i) = xv;
    }
}

}  // namespace nvl
