// gemm.h — projection GEMMs of the forward path (replaces tensor.MatMul,
// purego/tensor/tensor.go:62-88, at every call site listed in SURVEY.md §8 a3).
//
//   C[M,N] = A[M,K] · W[N,K]^T   (+ fused epilogue)
//
// Device weight layout: W is stored [N_pad][K], K contiguous (the checkpoint's
// own [out,in] order), so both MFMA operands are read K-contiguous from LDS with
// ds_read_b128 and no transpose is ever needed.  The reference keeps [in,out]
// (generic_loader.go:398-403); nvl_upload_tensor converts once at load.
//
// bf16 path (NVL_PRECISION_BF16): 128x128x64 tiles, 4 waves (2x2), each wave a
// 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators; operands staged
// HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip), double-buffered,
// XOR-swizzled through the per-lane SOURCE address so the ds_read_b128 fragment
// reads are bank-conflict-free (cdna_hip_programming.md §5.4 rule 21, T2).
// The MFMA is issued "swapped" (weights as the A operand) so each lane ends up
// with 4 consecutive N elements of one output row: 8/16-byte epilogue accesses.
//
// f32 path (NVL_PRECISION_F32): plain LDS-tiled fp32 FMA kernel, k ascending —
// the tight-tolerance parity mode, not a performance path.
#pragma once
#include "common.h"

namespace nvl {

enum GemmEpi {
    EPI_STORE = 0,   // C = acc (+bias)
    EPI_RESID = 1,   // X += alpha * (acc (+bias))      X fp32, in place   (generic_model.go:320-326)
    EPI_SWIGLU = 2,  // C[m, f] = silu(gate) * up        W rows interleaved (transformer.go:50-66)
    EPI_GELU = 3,    // C = gelu_tanh(acc + bias)                           (transformer.go:67-78)
};

struct GemmArgs {
    const void* A;          // [M][lda] bf16 (bf16 path) or fp32 (f32 path)
    int lda;
    const int32_t* a_rows;  // optional gather: logical row r reads A[a_rows[r]] (MoE), else NULL
    const void* W;          // [N_pad][K]
    void* C;                // output, leading dimension ldc (elements)
    int ldc;
    const float* bias;      // [N] or NULL
    float alpha;            // EPI_RESID multiplier
    int M, N, K;            // logical sizes; N_pad = round_up(N, 128) rows exist in W
    const int32_t* seg;     // optional device {start,end}: rows [start,end) of A (via a_rows if set)
                            // and of C; M is then only the grid bound (MoE expert segments)
};

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // tensor.go:181-190
    float x3 = x * x * x;
    float inner = 0.7978845608028654f * (x + 0.044715f * x3);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// ------------------------------------------------------------------------------------------
// epilogue shared by both kernels: 4 consecutive n for one row m
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT>
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M) return;
    if (EPI == EPI_SWIGLU) return;  // handled by the caller (needs two accumulators)
    if (n >= p.N) return;
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
    if (EPI == EPI_RESID) {
        float* x = (float*)p.C + (int64_t)m * p.ldc + n;
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
    OutT* c = (OutT*)p.C + (int64_t)m * p.ldc + n;
    if (full) {
        if (sizeof(OutT) == 4) {
            *(f32x4*)c = v;
        } else {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (bf16_t)v[r];
            *(bf16x4*)c = o;
        }
    } else {
        for (int r = 0; r < 4; r++)
            if (n + r < p.N) ActIO<OutT>::st(c + r, v[r]);
    }
}

template <typename OutT>
__device__ __forceinline__ void epilogue_swiglu4(const GemmArgs& p, int m, int f, f32x4 g, f32x4 u) {
    // p.N counts fused rows (2F); output has F = N/2 columns, ldc = F
    if (m >= p.M || f >= (p.N >> 1)) return;
    OutT* c = (OutT*)p.C + (int64_t)m * p.ldc + f;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = silu_f(g[r]) * u[r];
    if (sizeof(OutT) == 4) {
        *(f32x4*)c = v;
    } else {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = (bf16_t)v[r];
        *(bf16x4*)c = o;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------
constexpr int G_BM = 128, G_BN = 128, G_BK = 64;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;          // 16 KiB per operand tile
constexpr int G_LDS_BYTES = 2 * 2 * G_TILE_BYTES;      // 2 buffers x (A,W) = 64 KiB

// XCD-aware block remap (bijective for any grid size): blocks b and b+8 share an XCD, so give
// each XCD a contiguous run of tiles -> neighbouring tiles (same A row panel) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

template <int EPI, typename OutT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + G_BN - 1) / G_BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int wm = wave >> 1, wn = wave & 1;
    int row_base = 0;
    if (p.seg) {
        row_base = p.seg[0];
        p.M = p.seg[1] - row_base;
        p.C = (char*)p.C + (int64_t)row_base * p.ldc * (EPI == EPI_RESID ? 4 : (int)sizeof(OutT));
    }
    if (m0 >= p.M) return;

    // ---- staging addresses: one global_load_lds_dwordx4 moves 8 rows x 128 B (1 KiB) ----
    const int r_in = lane >> 3, pos = lane & 7;
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (wave * 4 + i) * 8 + r_in;          // tile row 0..127
        const int chunk = pos ^ ((row >> 1) & 7);           // swizzle on the SOURCE address
        int am = m0 + row;
        if (am > p.M - 1) am = p.M - 1;                     // clamp: rows >= M are never stored
        am += row_base;
        if (p.a_rows) am = p.a_rows[am];
        a_src[i] = (const bf16_t*)p.A + (int64_t)am * p.lda + chunk * 8;
        w_src[i] = (const bf16_t*)p.W + (int64_t)(n0 + row) * p.K + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * 2 * G_TILE_BYTES + wave * 4096;
        char* wbase = abase + G_TILE_BYTES;
        const int koff = kt * G_BK;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(a_src[i] + koff),
                (__attribute__((address_space(3))) void*)(abase + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(w_src[i] + koff),
                (__attribute__((address_space(3))) void*)(wbase + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    const int fr = lane & 15, fg = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int a_row_off = (wm * 64 + fr) * 128;   // activation rows (MFMA B operand -> output col m)
    const int w_row_off = (wn * 64 + fr) * 128;   // weight rows     (MFMA A operand -> output row n)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / G_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nt; t++) {
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char* abuf = smem + (t & 1) * 2 * G_TILE_BYTES;
        const char* wbuf = abuf + G_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int coff = (((ks * 4 + fg) ^ swz) << 4);
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(abuf + a_row_off + i * 2048 + coff);
#pragma unroll
            for (int j = 0; j < 4; j++) wf[j] = *(const bf16x8*)(wbuf + w_row_off + j * 2048 + coff);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds rows n = 4*fg + r of column m = fr of each 16x16 tile ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int ntile = (n0 + wn * 64 + j * 16) >> 4;   // even: gate block, +1: up block
                const int f = (ntile >> 1) * 16 + 4 * fg;
                epilogue_swiglu4<OutT>(p, m, f, acc[i][j], acc[i][j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = n0 + wn * 64 + j * 16 + 4 * fg;
                epilogue4<EPI, OutT>(p, m, n, acc[i][j]);
            }
        }
    }
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
        p.C = (char*)p.C + (int64_t)row_base * p.ldc * 4;
    }
    if (m0 >= p.M) return;

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
// bf16 skinny kernel (decode: M <= 64 rows).  The projection is then a weight-streaming problem
// (HBM-bound: every weight byte is read once, activations are a few hundred KB in L2), so the
// shape of the kernel is set by memory-level parallelism, not by MFMA:
//   * one workgroup owns 16*BNT weight rows (output columns) over ALL of K;
//   * its KSPLIT waves each stream a disjoint K slice of those rows straight HBM -> VGPR with
//     non-temporal 16-byte loads (no LDS round trip: nothing is shared between waves), several
//     k-steps in flight per wave, thousands of waves per launch;
//   * activations (<= 64 x K bf16) are read as MFMA fragments from L2;
//   * the K slices are reduced through LDS in fixed wave order (deterministic, no atomics) and the
//     fused epilogue (bias / residual / SwiGLU / GELU) runs once per output element.
// ------------------------------------------------------------------------------------------
template <int MT, int BNT, int EPI, typename OutT>
__global__ __launch_bounds__(1024) void gemm_skinny_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = (f32x4*)smem;                       // [ksplit][MT*BNT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksplit = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * (16 * BNT);
    const int kslice = p.K / ksplit;
    const int kbeg = wave * kslice;

    const bf16_t* wp[BNT];
    const bf16_t* xp[MT];
#pragma unroll
    for (int j = 0; j < BNT; j++) wp[j] = (const bf16_t*)p.W + (int64_t)(n0 + 16 * j + fr) * p.K + kbeg + 8 * fg;
#pragma unroll
    for (int i = 0; i < MT; i++) {
        int m = 16 * i + fr;
        if (m > p.M - 1) m = p.M - 1;
        xp[i] = (const bf16_t*)p.A + (int64_t)m * p.lda + kbeg + 8 * fg;
    }
    f32x4 acc[MT][BNT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < BNT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // k loop: blocks of U k-steps (U*32 of K), two register sets -> the next block's loads are in
    // flight while the current block's MFMAs issue (the compiler emits counted vmcnt for these).
    constexpr int U = 4;
    const int nblk = kslice / (32 * U);
    bf16x8 wA[U][BNT], xA[U][MT], wB[U][BNT], xB[U][MT];
    auto load_blk = [&](bf16x8 (&w)[U][BNT], bf16x8 (&x)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int j = 0; j < BNT; j++)
                w[u][j] = __builtin_nontemporal_load((const bf16x8*)(wp[j] + (b * U + u) * 32));
#pragma unroll
            for (int i = 0; i < MT; i++) x[u][i] = *(const bf16x8*)(xp[i] + (b * U + u) * 32);
        }
    };
    auto comp_blk = [&](bf16x8 (&w)[U][BNT], bf16x8 (&x)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int j = 0; j < BNT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][j], x[u][i], acc[i][j], 0, 0, 0);
    };
    load_blk(wA, xA, 0);
    int b = 0;
    for (; b + 2 <= nblk; b += 2) {
        load_blk(wB, xB, b + 1);
        comp_blk(wA, xA);
        if (b + 2 < nblk) load_blk(wA, xA, b + 2);
        comp_blk(wB, xB);
    }
    if (b < nblk) comp_blk(wA, xA);
    // ---- reduce the K slices in wave order ----
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < BNT; j++) red[(wave * (MT * BNT) + i * BNT + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    for (int i = wave; i < MT; i += ksplit) {
        f32x4 sum[BNT];
#pragma unroll
        for (int j = 0; j < BNT; j++) {
            sum[j] = red[(i * BNT + j) * 64 + lane];
            for (int w = 1; w < ksplit; w++) sum[j] += red[(w * (MT * BNT) + i * BNT + j) * 64 + lane];
        }
        const int m = 16 * i + fr;
        if (EPI == EPI_SWIGLU) {
            const int f = (n0 >> 5) * 16 + 4 * fg;      // BNT == 2: rows [gate 16 | up 16]
            epilogue_swiglu4<OutT>(p, m, f, sum[0], sum[BNT - 1]);
        } else {
#pragma unroll
            for (int j = 0; j < BNT; j++) epilogue4<EPI, OutT>(p, m, n0 + 16 * j + 4 * fg, sum[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT>
static inline bool launch_gemm_skinny_bf16(hipStream_t st, const GemmArgs& a);

template <int EPI, typename OutT>
static inline void launch_gemm_bf16(hipStream_t st, const GemmArgs& a) {
    if (launch_gemm_skinny_bf16<EPI, OutT>(st, a)) return;
    const int tiles = cdiv(a.M, G_BM) * cdiv(a.N, G_BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI, OutT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OutT>), dim3(tiles), dim3(256), G_LDS_BYTES, st, a);
}
// skinny dispatch: M <= 64, no gather/segments.  Returns false when the shape is not eligible.
template <int EPI, typename OutT>
static inline bool launch_gemm_skinny_bf16(hipStream_t st, const GemmArgs& a) {
    if (a.M > 64 || a.a_rows || a.seg) return false;
    constexpr int BNT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int MT = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int nblocks = cdiv(a.N, 16 * BNT);
    int ksplit = 16;
    while (ksplit > 1 && (a.K % (ksplit * 128) != 0 || ksplit * MT * BNT > 64 || (int64_t)nblocks * ksplit > 16384))
        ksplit >>= 1;
    if (a.K % (ksplit * 128) != 0) return false;      // K slices are whole blocks of 4 k-steps
    const size_t lds = (size_t)ksplit * MT * BNT * 64 * 16;
    dim3 grid(nblocks), block(ksplit * 64);
#define NVL_SK(MTv) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MTv, BNT, EPI, OutT>), grid, block, lds, st, a)
    if (MT == 1) NVL_SK(1); else if (MT == 2) NVL_SK(2); else NVL_SK(4);
#undef NVL_SK
    return true;
}

template <int EPI, typename OutT>
static inline void launch_gemm_f32(hipStream_t st, const GemmArgs& a) {
    const int tiles = cdiv(a.M, 64) * cdiv(a.N, 64);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, OutT>), dim3(tiles), dim3(256), 0, st, a);
}

}  // namespace nvl
