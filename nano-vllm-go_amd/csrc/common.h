// common.h — shared device/host helpers for libnvllm_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

namespace nvl {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define NVL_WAVE 64
// weight matrices are allocated with their row count padded to the largest GEMM N-tile (zero rows), so every
// tile instance may read a whole tile; outputs beyond N are never stored
constexpr int64_t W_ROW_PAD = 256;

// ---- bf16 conversions -------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE, NaN stays NaN (v_cvt_pk_bf16_f32)

template <typename T> struct ActIO;
template <> struct ActIO<float> {
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ActIO<bf16_t> {
    __device__ static __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// ---- operand layouts ------------------------------------------------------------------------
// bf16 MFMA operands (weights AND the activations handed from kernel to kernel) live in HBM in
// FRAGMENT-MAJOR order: the [R][K] matrix is cut into 16-row x 32-k operand blocks of
// v_mfma_f32_16x16x32_bf16, each stored as its 64 lanes' 16-byte fragments back to back (1 KiB),
// blocks ordered [r/16][k/32].  A wave-instruction then reads/writes one contiguous KiB; an LDS
// image filled by global_load_lds is lane-linear, so fragment ds_read_b128s are conflict-free.
// fp32 (parity mode) operands stay row-major.
__host__ __device__ __forceinline__ int64_t fm_index(int64_t r, int64_t k, int64_t K) {
    return ((((r >> 4) * (K >> 5) + (k >> 5)) * 64) + (r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
}
template <typename T> __device__ __forceinline__ int64_t act_index(int64_t r, int64_t k, int64_t K);
template <> __device__ __forceinline__ int64_t act_index<float>(int64_t r, int64_t k, int64_t K) { return r * K + k; }
template <> __device__ __forceinline__ int64_t act_index<bf16_t>(int64_t r, int64_t k, int64_t K) { return fm_index(r, k, K); }

// store 4 consecutive-k activations of row r (k % 4 == 0): one 8-byte (bf16) / 16-byte (fp32) store
template <typename T> __device__ __forceinline__ void act_store4(T* base, int64_t r, int64_t k, int64_t K, f32x4 v);
template <> __device__ __forceinline__ void act_store4<float>(float* base, int64_t r, int64_t k, int64_t K, f32x4 v) {
    *(f32x4*)(base + r * K + k) = v;
}
template <> __device__ __forceinline__ void act_store4<bf16_t>(bf16_t* base, int64_t r, int64_t k, int64_t K, f32x4 v) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (bf16_t)v[i];
    *(bf16x4*)(base + fm_index(r, k, K)) = o;
}

// ---- KV placement ---------------------------------------------------------------------------
// A sequence's keys live in blocks of `bs` tokens named by its block table blk_table[tbl ...] (paged mode: the host's
// 256-token blocks, block_manager.go; slab mode: ONE block spanning the whole slot).  Position pos -> (block, row).
__device__ __forceinline__ void kv_locate(const int32_t* __restrict__ blk_table, int tbl, int pos, int bs, int& blk, int& row) {
    const int bi = pos / bs;
    blk = blk_table[tbl + bi];
    row = pos - bi * bs;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // tensor.go:181-190: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), the tanh in float64 there.
    // tanh(u) = 1 - 2 / (exp(2u) + 1) on the hardware exp2 / rcp (both ~1 ulp): ABSOLUTE error ~1e-7, and only the absolute
    // error of tanh enters 1 + tanh; exp -> inf gives exactly 1, exp -> 0 exactly -1.  (libm's tanhf is ~40 instructions per
    // element; in the prefill epilogue of a K = 768 projection — GPT-2's FFN-up: 64 K elements per 256 x 256 tile — it cost
    // a third of the launch: 117 us against 81 us for the same-FLOP FFN-down, profiles/r02_gpt2_phase_breakdown.txt.)
    const float x3 = x * x * x;
    const float inner = 0.7978845608028654f * (x + 0.044715f * x3);
    const float e = __builtin_amdgcn_exp2f(inner * 2.8853900817779268f);      // exp(2 inner)
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return 0.5f * x * (1.0f + th);
}

// ---- wave reductions (64 lanes) ----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- in-kernel time stamps (diagnostic runs only: nvl_set_debug mode 4) ------------------------------------------
// A launch whose argument block carries a stamp buffer writes, per workgroup, the 100 MHz constant clock
// (s_memrealtime: one counter for the whole chip) at up to 8 points of its life: scripts/decode_timeline.py turns a decode
// step's stamps into dispatch spread / prologue / first data / stream / reduce / epilogue / gap to the next kernel.
// Production launches pass a null pointer: one scalar branch per stamp site, nothing else.
// The times are kept in registers and written when the object dies (the kernel's exits): a store to global memory at the
// top of a kernel would make every later uniform load a vector load (the compiler may no longer assume the loaded
// memory unclobbered, so it cannot use the scalar cache) — the instrumentation would change what it measures.
#ifndef NVL_STAMPS
// the product library: no stamp site at all.  (Round 3 first shipped the sites compiled in and disabled by a NULL buffer:
// each one still is a branch with two scheduling barriers inside, which splits the kernel's scheduling regions — every
// decode projection ran 3-7 % slower than in round 2, 31.0 K against 32.1 K decode tok/s at B = 32 on the same box,
// profiles/r03_ab_vs_r02.txt.)  `make diag` builds libnvllm_hip_diag.so with -DNVL_STAMPS for scripts/decode_timeline.py.
struct NvlStamps {
    __device__ __forceinline__ NvlStamps(unsigned long long*, int) {}
    __device__ __forceinline__ void mark(int) {}
    template <typename T> __device__ __forceinline__ void mark_used(int, const T&) {}
};
#else
struct NvlStamps {
    unsigned long long* buf; int wg; unsigned long long t[6];
    __device__ __forceinline__ NvlStamps(unsigned long long* b, int w) : buf(b), wg(w) {
#pragma unroll
        for (int i = 0; i < 6; i++) t[i] = 0;
        mark(0);
    }
    // (inline asm WITHOUT a memory clobber, fenced for the scheduler only: the builtin counts as a write to unknown memory,
    //  after which the compiler turns every uniform load of the kernel into a vector load — in production builds too)
    __device__ __forceinline__ void mark(int slot) {
        if (buf) {
            unsigned long long v;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v));
            __builtin_amdgcn_sched_barrier(0);
            t[slot] = v;
        }
    }
    // stamp once `v` has been computed (pins the stamp behind the instructions that produce it)
    template <typename T> __device__ __forceinline__ void mark_used(int slot, const T& v) { asm volatile("" :: "v"(v)); mark(slot); }
    __device__ __forceinline__ ~NvlStamps() {
        if (!buf) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are retired
        mark(5);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 6; i++) buf[(size_t)wg * 8 + i] = t[i];
        }
    }
};
#endif
// compile-time switch for kernels with uniform global loads (decode attention): even a disabled stamp site is an opaque
// side effect to the compiler, after which uniform loads may no longer use the scalar path
template <bool ON> struct NvlStampsT;
template <> struct NvlStampsT<true> : NvlStamps { __device__ __forceinline__ NvlStampsT(unsigned long long* b, int w) : NvlStamps(b, w) {} };
template <> struct NvlStampsT<false> {
    __device__ __forceinline__ NvlStampsT(unsigned long long*, int) {}
    __device__ __forceinline__ void mark(int) {}
    template <typename T> __device__ __forceinline__ void mark_used(int, const T&) {}
};
constexpr int STAMP_MAX_WG = 2048, STAMP_MAX_LAUNCH = 128;

// ---- host error plumbing -------------------------------------------------------
struct HipError {
    hipError_t code;
    const char* what;
    const char* file;
    int line;
};

#define NVL_HIP(expr)                                                                 \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) throw ::nvl::HipError{_e, #expr, __FILE__, __LINE__};   \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, DEVICE): a process may open model handles on
// several devices (nvl_runtime_opts.device) and from several threads; a repeated set is harmless, a missing one is a
// launch failure on the second device.
struct LdsAttrOnce { std::atomic<uint32_t> done{0}; };
static inline void ensure_lds_attr(LdsAttrOnce& g, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    if (g.done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    g.done.fetch_or(bit, std::memory_order_release);
}
#define NVL_LDS_ATTR(fn, bytes)                                        \
    do {                                                               \
        static ::nvl::LdsAttrOnce _once;                               \
        ::nvl::ensure_lds_attr(_once, (const void*)(fn), (int)(bytes)); \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace nvl
