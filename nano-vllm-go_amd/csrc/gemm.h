// gemm.h — projection GEMMs of the forward path (replaces tensor.MatMul,
// purego/tensor/tensor.go:62-88, at every call site listed in SURVEY.md §8 a3).
//
//   C[M,N] = A[M,K] · W[N,K]^T   (+ fused epilogue)
//
// bf16 path (NVL_PRECISION_BF16).  Both operands live in HBM in the fragment-major layout of
// common.h (fm_index): 16-row x 32-k MFMA operand blocks, 1 KiB contiguous each.  The reference
// keeps weights [in,out] fp32 (generic_loader.go:398-403); nvl_upload_tensor converts once at load,
// and every kernel that produces a GEMM input (norm, attention, SwiGLU/GELU epilogue) writes it
// directly in this layout.
//   * gemm_bf16_kernel   (prefill, M > 64): 128x128x64 tiles, 4 waves (2x2), each wave a 64x64
//     sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators; operand blocks go HBM->LDS with
//     global_load_lds_dwordx4 (one contiguous KiB per wave-instruction, no VGPR round trip),
//     double-buffered; the LDS image is lane-linear so the fragment ds_read_b128s are
//     bank-conflict-free with no swizzle.  MFMA-bound.
//   * gemm_skinny_bf16_kernel (decode, M <= 64): weight-streaming, HBM-bound (see below).
// The MFMA is issued "swapped" (weights as the A operand) so each lane ends up with 4 consecutive N
// elements of one output row: 8/16-byte epilogue accesses, and the bf16 epilogues can write the
// next GEMM's fragment-major operand directly.
//
// f32 path (NVL_PRECISION_F32): plain LDS-tiled fp32 FMA kernel on row-major operands, k ascending —
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
    const void* A;          // bf16: fragment-major [M_pad16][K]; f32: row-major [M][lda]
    int lda;
    const int32_t* a_rows;  // optional gather: logical row r reads A row a_rows[r] (MoE), else NULL
    const void* W;          // bf16: fragment-major [N_pad128][K]; f32: row-major [N_pad128][K]
    void* C;                // fp32 outputs row-major [.][ldc]; bf16 outputs fragment-major [.][ldc]
    int ldc;
    const float* bias;      // [N] or NULL
    float alpha;            // EPI_RESID multiplier
    int M, N, K;            // logical sizes
    const int32_t* seg;     // optional device {start,end}: rows [start,end) of A (via a_rows if set)
                            // and of C; M is then only the grid bound (MoE expert segments)
    int c_row0;             // row offset added to every output row (set from seg inside the kernel)
};

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + __expf(-g)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // tensor.go:181-190
    float x3 = x * x * x;
    float inner = 0.7978845608028654f * (x + 0.044715f * x3);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// ------------------------------------------------------------------------------------------
// epilogue shared by all kernels: 4 consecutive n for one row m
// ------------------------------------------------------------------------------------------
template <int EPI, typename OutT>
__device__ __forceinline__ void epilogue4(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M || n >= p.N) return;
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
    const int64_t row = (int64_t)p.c_row0 + m;
    if (EPI == EPI_RESID) {
        float* x = (float*)p.C + row * p.ldc + n;
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
    if (sizeof(OutT) == 4) {
        float* c = (float*)p.C + row * p.ldc + n;
        if (full) *(f32x4*)c = v;
        else
            for (int r = 0; r < 4; r++)
                if (n + r < p.N) c[r] = v[r];
    } else {
        // bf16 activation for the next GEMM: widths are multiples of 64, so the group is always full
        act_store4<bf16_t>((bf16_t*)p.C, row, n, p.ldc, v);
    }
}

template <typename OutT>
__device__ __forceinline__ void epilogue_swiglu4(const GemmArgs& p, int m, int f, f32x4 g, f32x4 u) {
    // p.N counts fused rows (2F); the output has F = N/2 columns, ldc = F
    if (m >= p.M || f >= (p.N >> 1)) return;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = silu_f(g[r]) * u[r];
    act_store4<OutT>((OutT*)p.C, (int64_t)p.c_row0 + m, f, p.ldc, v);
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA kernel (prefill)
// ------------------------------------------------------------------------------------------
constexpr int G_BM = 128, G_BN = 128, G_BK = 64;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;          // 16 KiB per operand tile = 16 blocks of 1 KiB
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
        p.c_row0 = row_base;
    }
    if (m0 >= p.M) return;

    // ---- staging: each operand tile is 16 blocks (8 row-tiles x 2 k-steps) of 1 KiB; wave w moves
    // blocks 4w..4w+3 of A and of W per K tile, one global_load_lds_dwordx4 each ----
    const int fr = lane & 15, fg = lane >> 4;
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int blk = wave * 4 + i, rt = blk >> 1, ksb = blk & 1;
        int am = m0 + rt * 16 + fr;
        if (am > p.M - 1) am = p.M - 1;                     // clamp: rows >= M are never stored
        am += row_base;
        if (p.a_rows) am = p.a_rows[am];
        a_src[i] = (const bf16_t*)p.A + ((((int64_t)(am >> 4) * (p.K >> 5) + ksb) * 64) + (am & 15) + 16 * fg) * 8;
        w_src[i] = (const bf16_t*)p.W + (((int64_t)((n0 >> 4) + rt) * (p.K >> 5) + ksb) * 64 + lane) * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* abase = smem + buf * 2 * G_TILE_BYTES + wave * 4096;
        char* wbase = abase + G_TILE_BYTES;
        const int64_t koff = (int64_t)kt * 1024;            // 2 k-steps x 512 elements per K tile
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

    // ---- fragment reads: LDS image [row-tile 0..7][k-step 0..1][lane][16 B] ----
    const int a_blk_off = wm * 4 * 2048 + lane * 16;   // activation rows (MFMA B operand -> output col m)
    const int w_blk_off = wn * 4 * 2048 + lane * 16;   // weight rows     (MFMA A operand -> output row n)

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
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) af[i] = *(const bf16x8*)(abuf + a_blk_off + i * 2048 + ks * 1024);
#pragma unroll
            for (int j = 0; j < 4; j++) wf[j] = *(const bf16x8*)(wbuf + w_blk_off + j * 2048 + ks * 1024);
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
// bf16 skinny kernel (decode: M <= 64 rows).  The projection is then a weight-streaming problem
// (HBM-bound: every weight byte is read once, activations are a few hundred KB in L2), so the
// shape of the kernel is set by memory-level parallelism, not by MFMA:
//   * one workgroup owns NTW 16-row weight tiles (16*NTW output columns) over ALL of K;
//   * it runs NTW*KSPLIT waves; wave w streams tile w/KSPLIT over K slice w%KSPLIT straight
//     HBM -> VGPR with non-temporal 16-byte loads — one contiguous KiB per instruction thanks to the
//     fragment-major layout; no LDS round trip, nothing is shared between waves; two register sets
//     keep the next block of k-steps in flight behind the current block's MFMAs;
//   * activations (<= 64 x K bf16, fragment-major too) are read as MFMA fragments from L2;
//   * the K slices are reduced through LDS in fixed wave order (deterministic, no atomics) and the
//     fused epilogue (bias / residual / SwiGLU / GELU) runs once per output element.  For SwiGLU
//     NTW = 2: tile 0 is the gate block, tile 1 the up block of the same 16 features.
// ------------------------------------------------------------------------------------------
template <int MT, int NTW, int U, int EPI, typename OutT>
__global__ __launch_bounds__(1024) void gemm_skinny_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* red = (f32x4*)smem;                       // [NTW][ksplit][MT][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksplit = (blockDim.x >> 6) / NTW;
    const int tile = wave / ksplit, kw = wave - tile * ksplit;
    const int fr = lane & 15, fg = lane >> 4;
    const int nt0 = blockIdx.x * NTW;                // first 16-row weight tile of this workgroup
    const int kslice = p.K / ksplit;
    const int ks0 = (kw * kslice) >> 5;              // first k-step of this wave

    const bf16_t* wp = (const bf16_t*)p.W + (((int64_t)(nt0 + tile) * (p.K >> 5) + ks0) * 64 + lane) * 8;
    const bf16_t* xp[MT];
#pragma unroll
    for (int i = 0; i < MT; i++)   // rows >= M of the last 16-row tile exist (padded allocation) and are never stored
        xp[i] = (const bf16_t*)p.A + (((int64_t)i * (p.K >> 5) + ks0) * 64 + lane) * 8;
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // k loop: blocks of U k-steps (U*32 of K), two register sets -> the next block's loads are in
    // flight while the current block's MFMAs issue (the compiler emits counted vmcnt for these).
    const int nblk = kslice / (32 * U);
    bf16x8 wA[U], xA[U][MT], wB[U], xB[U][MT];
    auto load_blk = [&](bf16x8 (&w)[U], bf16x8 (&x)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            w[u] = __builtin_nontemporal_load((const bf16x8*)(wp + (int64_t)(b * U + u) * 512));
#pragma unroll
            for (int i = 0; i < MT; i++) x[u][i] = *(const bf16x8*)(xp[i] + (int64_t)(b * U + u) * 512);
        }
    };
    auto comp_blk = [&](bf16x8 (&w)[U], bf16x8 (&x)[U][MT]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < MT; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u], x[u][i], acc[i], 0, 0, 0);
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
    for (int i = 0; i < MT; i++) red[((tile * ksplit + kw) * MT + i) * 64 + lane] = acc[i];
    __syncthreads();
    auto ksum = [&](int t, int i) {
        f32x4 s = red[((t * ksplit) * MT + i) * 64 + lane];
        for (int w = 1; w < ksplit; w++) s += red[((t * ksplit + w) * MT + i) * 64 + lane];
        return s;
    };
    if (EPI == EPI_SWIGLU) {
        for (int i = wave; i < MT; i += NTW * ksplit) {
            const int f = (nt0 >> 1) * 16 + 4 * fg;    // NTW == 2: tiles [gate 16 | up 16] of features 8*nt0..
            epilogue_swiglu4<OutT>(p, 16 * i + fr, f, ksum(0, i), ksum(NTW - 1, i));
        }
    } else {
        for (int e = wave; e < NTW * MT; e += NTW * ksplit) {
            const int t = e / MT, i = e - t * MT;
            epilogue4<EPI, OutT>(p, 16 * i + fr, (nt0 + t) * 16 + 4 * fg, ksum(t, i));
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
        p.c_row0 = row_base;
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
// host launchers
// ------------------------------------------------------------------------------------------
static int g_force_ntw = 0, g_force_ksplit = 0;   // tuning overrides (nvl_bench_gemm only)

// skinny dispatch: M <= 64, no gather/segments.  Returns false when the shape is not eligible.
template <int NTW, int EPI, typename OutT>
static inline bool launch_gemm_skinny_ntw(hipStream_t st, const GemmArgs& a) {
    const int MT = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int U = MT <= 2 ? 4 : 2;                       // register budget: (1+MT)*U*2 fragments
    const int nblocks = cdiv(cdiv(a.N, 16), NTW);        // weight rows are padded to 128: all tiles exist
    int ksplit = g_force_ksplit ? g_force_ksplit : 16;
    while (ksplit > 1 && (a.K % (ksplit * 32 * U) != 0 || ksplit * NTW > 16)) ksplit >>= 1;
    if (a.K % (ksplit * 32 * U) != 0) return false;      // K slices are whole blocks of U k-steps
    const size_t lds = (size_t)NTW * ksplit * MT * 64 * 16;
    dim3 grid(nblocks), block(NTW * ksplit * 64);
#define NVL_SK(MTv, Uv) hipLaunchKernelGGL((gemm_skinny_bf16_kernel<MTv, NTW, Uv, EPI, OutT>), grid, block, lds, st, a)
    if (MT == 1) NVL_SK(1, 4); else if (MT == 2) NVL_SK(2, 4); else NVL_SK(4, 2);
#undef NVL_SK
    return true;
}
template <int EPI, typename OutT>
static inline bool launch_gemm_skinny_bf16(hipStream_t st, const GemmArgs& a) {
    if (a.M > 64 || a.a_rows || a.seg) return false;
    if (EPI == EPI_SWIGLU) return launch_gemm_skinny_ntw<2, EPI, OutT>(st, a);
    if (g_force_ntw == 2) return launch_gemm_skinny_ntw<2, EPI, OutT>(st, a);
    if (g_force_ntw == 4) return launch_gemm_skinny_ntw<4, EPI, OutT>(st, a);
    return launch_gemm_skinny_ntw<1, EPI, OutT>(st, a);
}

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

template <int EPI, typename OutT>
static inline void launch_gemm_f32(hipStream_t st, const GemmArgs& a) {
    const int tiles = cdiv(a.M, 64) * cdiv(a.N, 64);
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, OutT>), dim3(tiles), dim3(256), 0, st, a);
}

}  // namespace nvl
