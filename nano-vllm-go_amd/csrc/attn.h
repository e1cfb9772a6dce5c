// attn.h — causal attention over the per-sequence KV slabs.  Replaces, for all three
// attention types, the reference's split/repeat/score/softmax/apply/merge loops:
//   GQA  attention.go:217-274 (+ :300-470)      group = nH/nKV
//   MHA  attention.go:34-64, :127-191           group = 1
//   MQA  mqa.go:47-99, :163-267                 group = nH (all heads share one K/V tile)
// No repeatKVHeads copy, no Concatenate re-copy, no [nH,S,T] score tensor.
//
// bf16 kernels (product path): flash-style, MFMA 16x16x32 bf16, fp32 online softmax.  A "query row" is a
// (position, head-in-group) pair, position-major, so every head of a GQA/MQA group reuses the same K/V tile.
// The products are issued transposed:
//     S^T[key, q] = K[key, :] · Q[q, :]^T          (A = K tile rows,   B = Q rows, from registers)
//     O^T[d,  q] += V^T[d, key] · P^T[key, q]      (A = V^T tile rows, B = P^T straight from the
//                                                   S^T accumulators — no LDS, no shuffles)
// so the softmax row (fixed q) lives on one lane column: the row maximum needs two cross-lane steps, the O rescale is
// lane-local.  K and V are both kept ROW-MAJOR in HBM ([T][hd], the reference's own cache layout, kv_cache.go:5-6):
// appends are contiguous rows; the V^T operand comes out of an LDS image of the row-major tile by transposed reads
// (ds_read_b64_tr_b16) — in the prefill kernel the image is filled by LDS-DMA (8 waves, 256 query rows per workgroup),
// in the decode kernel by the wave that loaded the rows (one new token per sequence, split-T over the waves).
//
// f32 kernel (parity mode): one workgroup per (query position, head); scores in LDS; plain fp32.
#pragma once
#include "common.h"

namespace nvl {

struct AttnArgs {
    const void* q;        // [tokens][q_stride]; head h at column h*HD (bf16 or f32)
    int q_stride;
    void* out;            // [tokens][out_stride] (bf16: fragment-major, fp32: row-major); head h at column h*HD
    int out_stride;
    const void* kcache;   // layer base; (block, kvh) at block*slot_stride + kvh*Tmax*HD; [Tmax][HD]
    const void* vcache;   // same layout as kcache: [Tmax][HD]
    int64_t slot_stride;  // elements between consecutive KV blocks
    int Tmax;             // tokens per KV block (a multiple of 64): the whole slot in slab mode, 256 in paged mode
    const int32_t* seq_tok_start;
    const int32_t* seq_len;
    const int32_t* seq_pos;
    const int32_t* blk_table;   // [sequence in the batch][tbl_stride] block ids (slab mode: one entry, the slot)
    int tbl_stride;
    int bs_shift;               // log2(Tmax) when it is a power of two, else -1 (then the kernels divide)
    int nH, nKV, group;
    float scale;
    // fused decode (attn_decode_bf16_kernel<.., FUSED>): q/k/v of the NEW token come straight from the fp32 output of
    // the QKV projection ([tokens][qkv_stride] = [Q heads | K heads | V heads]); RoPE (rope.go:153-205) is applied here
    // and the new K/V row is appended to the slabs by this kernel (replaces rope_kv_kernel + the q round trip)
    const float* qkv;
    int qkv_stride;
    const float* cos_t;   // [max_seq][HD] or NULL
    const float* sin_t;
    // decode kernel with gridDim.z > 1 (keys split over workgroups): workgroup z leaves its partial result here —
    // [(seq * gridDim.x + blockIdx.x) * gridDim.z + z] records of {O^T [HD/16][64 lanes] f32x4 (unnormalised), m[16], l[16]} — and
    // takes a ticket from part_cnt[seq * gridDim.x + blockIdx.x]; the workgroup that draws the last one combines them in z order.
    float* part;
    int32_t* part_cnt;    // zero between launches
    unsigned long long* stamps;   // diagnostic runs: per-workgroup time stamps (common.h nvl_stamp), else NULL
    int kv_nt;                    // decode: non-temporal K/V loads (set by the host: attention() in nvllm.hip)
};
template <int HD> constexpr int attn_part_floats() { return (HD / 16) * 64 * 4 + 32; }

// index of the KV block that holds key `key`
__device__ __forceinline__ int kv_block_index(const AttnArgs& p, int key) { return p.bs_shift >= 0 ? (key >> p.bs_shift) : key / p.Tmax; }

// ------------------------------------------------------------------------------------------
// LDS image of one 64-key K or V tile ([64 keys][HD] bf16, rows of CPR = HD/8 16-byte chunks).  The tile arrives by
// LDS-DMA (global_load_lds_dwordx4: lane l of a 1-KiB piece lands at byte 16*l of the piece), so the image is a
// permutation of 16-byte chunks chosen through the SOURCE address: chunk c of row r sits at chunk position
//     r * CPR + (((c >> 1) ^ swz(r)) << 1 | (c & 1)),      swz(r) = (r >> 1) & 3 (HD 64),  r & 7 (HD 128)
// i.e. the 32-byte column segments of a row are XOR-ed with a row-dependent pattern.  With it both read patterns are
// bank-conflict-free: the K fragment ds_read_b128s (16 rows x the same chunk per 16-lane group) and the V
// ds_read_b64_tr_b16 transposed reads (8 consecutive rows x one 32-byte segment per half wave).
// ------------------------------------------------------------------------------------------
template <int HD> __device__ __forceinline__ int kv_swz(int row) { return HD == 64 ? ((row >> 1) & 3) : (row & 7); }
template <int HD> __device__ __forceinline__ int kv_img_chunk(int row, int c) {     // chunk position inside the image
    return row * (HD / 8) + ((((c >> 1) ^ kv_swz<HD>(row)) << 1) | (c & 1));
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 lds_read_tr4(const char* addr) {
    // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major
    // (lane i gets column i of the 4 rows).  EXEC must be all ones.
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
    return __builtin_bit_cast(bf16x4, v);
}

// One LDS-DMA piece: 64 lanes x 16 bytes from sbase + voff (per lane) to LDS lds_addr + lane * 16.  Issued by inline
// assembly ON PURPOSE: the compiler's wait-count pass knows the builtin form writes LDS and puts s_waitcnt vmcnt(0) in
// front of the next transposed LDS read it sees — i.e. it waits for the tile that was issued a moment ago, two tiles ahead
// of the one being read, and the ring degenerates to load-then-compute.  Through the assembly the pieces are invisible
// to that pass; the kernel's own counted s_waitcnt vmcnt + barrier order them (vector loads return in order).
// (s_nop 4: SGPR written by VALU -> read by VMEM needs 5 wait states, M0 write -> LDS-DMA needs 1; the hazard recogniser
// does not look inside the assembly.  M0 is a reserved register and cannot be named as a clobber: the kernels that call
// this use no other M0 consumer — no builtin LDS-DMA, no movrel, no GWS.)
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

// single-instruction maxima: fmaxf on MFMA outputs makes the compiler canonicalise each input first (v_max x, x)
// maximum of the 16 scores a lane holds, straight off the MFMA accumulators: two interleaved v_max3 chains in ONE asm
// block.  The compiler inserts no hazard wait states for inline assembly (an MFMA result needs 8 before a VALU reads
// it): the s_nop is that wait.
__device__ __forceinline__ float vmax16(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) {
    float r0, r1;
    asm("s_nop 7\n\t"
        "v_max3_f32 %0, %2, %3, %4\n\t"
        "v_max3_f32 %1, %5, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %1, %1, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %14, %15\n\t"
        "v_max3_f32 %0, %0, %16, %17\n\t"
        "v_max_f32 %0, %0, %1"
        : "=&v"(r0), "=&v"(r1)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
          "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
    return r0;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// maximum over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48) without going through LDS
__device__ __forceinline__ float rows_max(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float m = vmax2(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return vmax2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ------------------------------------------------------------------------------------------
// bf16 prefill kernel.  One workgroup = 8 waves = 256 query rows of one (sequence, kv head); each wave owns 32 rows as
// two 16-row sub-tiles, so every K / V fragment read from LDS feeds two MFMAs.  K and V tiles (64 keys) stream
// HBM -> LDS by LDS-DMA through a 3-slot ring, two tiles ahead of the compute, with ONE barrier per tile:
//     wait (my pieces of tile kt landed)  |  barrier  |  issue tile kt+2 into the slot tile kt-1 used  |  compute tile kt
// (a slot is re-filled only after every wave passed the barrier that follows its last read; LDS-DMA is ordered by the
// issuing wave's vmcnt + that barrier).  hd 64: four waves per SIMD (two workgroups per CU), so one wave's softmax VALU
// runs beside another's MFMAs; hd 128: two.
// ------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(512, HD == 64 ? 4 : 2) void attn_prefill_bf16_kernel(AttnArgs p) {
    constexpr int KS = HD / 32, DT = HD / 16, CPR = HD / 8, TQ = 2;
    constexpr int TILE_BYTES = 64 * HD * 2, PIECES = TILE_BYTES / 1024, PW = 2 * PIECES / 8, NBUF = 3;
    constexpr int QROWS = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // NBUF x [K image | V image]
    __shared__ int32_t sblk[128];          // this sequence's block table (<= 128 blocks: host-checked)

    // grid (kv head, sequence, query tile): consecutive block ids differ in the kv head, so with 8 | nKV every query tile of
    // one (sequence, kv head) lands on the same XCD (ids that differ by 8 share one) and re-reads its K/V tiles from that
    // XCD's L2 instead of from HBM / the memory-side cache.
    // Query tiles are the SLOWEST grid index and taken last first: under the causal mask tile qt walks qt+1 key-tile
    // groups, so the whole batch is dispatched longest-workgroup-first and the short ones fill the tail.  (Sequence-major
    // order left the last sequence's longest workgroups starting late: 61 % average occupancy at 8 x 2048.)
    const int seq = blockIdx.y, kvh = blockIdx.x, qt = gridDim.z - 1 - blockIdx.z;
    const int S = p.seq_len[seq];
    const int R = S * p.group;                 // query rows of this (seq, kv head)
    if (qt * QROWS >= R) return;
    const int tok0 = p.seq_tok_start[seq], pos0 = p.seq_pos[seq];
    const int32_t* tbl = p.blk_table + (int64_t)seq * p.tbl_stride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fq = lane & 15, fg = lane >> 4;
    // The block ids go through LDS: inside the tile loop the ONLY vector-memory operations may be the LDS-DMA pieces —
    // a register-returning load there (or a load still pending at loop entry) makes the compiler wait with vmcnt(0), which
    // also waits for the DMA issued a moment ago and serialises the whole ring (measured: 4200 cycles per tile instead of ~1300)
    if (tid < p.tbl_stride && tid < 128) sblk[tid] = tbl[tid];

    // ---- this lane's query rows (one per sub-tile); position and head are re-derived where needed (registers) ----
    bf16x8 qf[TQ][KS];
    auto row_of = [&](int qi, bool& ok) {
        int row = qt * QROWS + (wave * TQ + qi) * 16 + fq;
        ok = row < R;
        return ok ? row : qt * QROWS;          // clamp to a valid row; result discarded
    };
#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        bool ok;
        const int row = row_of(qi, ok), si = row / p.group, hd_ = kvh * p.group + (row - si * p.group);
        const bf16_t* qp = (const bf16_t*)p.q + (int64_t)(tok0 + si) * p.q_stride + hd_ * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[qi][ks] = *(const bf16x8*)(qp + ks * 32 + fg * 8);
    }
    // the Q fragments are complete before the first DMA is issued (an empty asm that "rewrites" them: the compiler waits
    // here, once, and no register is pending on a vector load when the loop starts)
#pragma unroll
    for (int qi = 0; qi < TQ; qi++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) asm volatile("" : "+v"(qf[qi][ks]));
    __syncthreads();                      // sblk is visible
    int last_row = qt * QROWS + QROWS - 1;
    if (last_row > R - 1) last_row = R - 1;
    const int n_kt = (pos0 + last_row / p.group) / 64 + 1;
    // key tiles this WAVE needs: up to its last valid row's causal limit (none when the wave has no valid row)
    const int w_row0 = qt * QROWS + wave * 32;
    int w_last = w_row0 + 31;
    if (w_last > R - 1) w_last = R - 1;
    const int wave_n_kt = w_row0 < R ? (pos0 + w_last / p.group) / 64 + 1 : 0;
    const int wave_first_limit = pos0 + w_row0 / p.group;     // tiles wholly below it need no masking

    // ---- LDS-DMA: wave w moves pieces w*PW .. w*PW+PW-1 of each tile's [K pieces | V pieces] list ----
    int src_off[PW];                 // element offset of this lane's 16 bytes inside the (K or V) tile
#pragma unroll
    for (int i = 0; i < PW; i++) {
        const int jj = (wave * PW + i) % PIECES;
        const int pos16 = jj * 64 + lane, row = pos16 / CPR, pc = pos16 % CPR;
        const int c = (((pc >> 1) ^ kv_swz<HD>(row)) << 1) | (pc & 1);
        src_off[i] = (row * HD + c * 8) * 2;            // bytes
    }
    const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int kt, int buf) {
        const int bi = kv_block_index(p, kt * 64), krow0 = kt * 64 - bi * p.Tmax;      // (64 | Tmax: no straddling)
        const int blk = __builtin_amdgcn_readfirstlane(sblk[bi]);
        const int64_t blk_off = (int64_t)blk * p.slot_stride + ((int64_t)kvh * p.Tmax + krow0) * HD;
        const uint32_t dst = smem_lds + buf * (2 * TILE_BYTES) + wave * (PW * 1024);  // (K pieces first, then V pieces)
#pragma unroll
        for (int i = 0; i < PW; i++) {
            const bool is_v = (wave * PW + i) >= PIECES;
            const bf16_t* base = (const bf16_t*)(is_v ? p.vcache : p.kcache) + blk_off;
            lds_dma16(base, (uint32_t)src_off[i], dst + i * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a K / V image) ----
    int k_off[KS];                    // row 16t + fq: + t * 16 rows
#pragma unroll
    for (int ks = 0; ks < KS; ks++) k_off[ks] = 16 * kv_img_chunk<HD>(fq, ks * 4 + fg);
    const int q4 = fq >> 2, p4 = fq & 3;
    int v_off[DT];                    // row 4 fg + q4 (+ 32 u, + 16): 32-byte segment d, half p4 & 1
#pragma unroll
    for (int d = 0; d < DT; d++) v_off[d] = 16 * kv_img_chunk<HD>(4 * fg + q4, 2 * d + (p4 >> 1)) + 8 * (p4 & 1);

    f32x4 o[TQ][DT];
    f32x4 lacc[TQ];                   // softmax denominators out of the MFMA pipe: ones · P^T, every row the complete sum
    float m_run[TQ];
#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        m_run[qi] = -INFINITY;
        lacc[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < DT; d++) o[qi][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float sl2 = p.scale * 1.4426950408889634f;     // exp2 domain

    issue(0, 0);
    if (n_kt > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < n_kt; kt++) {
        // my pieces of tile kt have landed (tile kt+1's may stay in flight)
        if (kt + 1 < n_kt) {
            if (PW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < n_kt) { int nb = buf + 2; if (nb >= NBUF) nb -= NBUF; issue(kt + 2, nb); }
        const char* kimg = smem + buf * (2 * TILE_BYTES);
        const char* vimg = kimg + TILE_BYTES;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        if (kt >= wave_n_kt) continue;                   // (wave-uniform) every row of this wave is masked out here

        // ---- S^T = K · Q^T : 4 sub-tiles of 16 keys, each K fragment used for both query sub-tiles ----
        f32x4 s[TQ][4];
#pragma unroll
        for (int qi = 0; qi < TQ; qi++)
#pragma unroll
            for (int t = 0; t < 4; t++) s[qi][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                const bf16x8 kf = *(const bf16x8*)(kimg + k_off[ks] + t * (16 * CPR * 16));
#pragma unroll
                for (int qi = 0; qi < TQ; qi++)
                    s[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], s[qi][t], 0, 0, 0);
            }
        // lane holds S^T[key = kt*64 + 16t + 4fg + r][q = fq] per sub-tile
        const bool whole = kt * 64 + 63 <= wave_first_limit;      // every row of the wave sees the whole tile
        if (!whole) {                                             // a tile on the causal diagonal
#pragma unroll
            for (int qi = 0; qi < TQ; qi++) {
                bool ok;
                const int lim = pos0 + row_of(qi, ok) / p.group - kt * 64 - fg * 4;   // last key (tile-relative) this row may see
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        s[qi][t][r] = (t * 16 + r <= lim) ? s[qi][t][r] : -INFINITY;  // reference: -1e10 then exp() == 0 exactly
            }
        }
        // scores stay RAW; the softmax scale (sl2 > 0: checked by the host) is folded into the exponent's fma:
        // p = exp2(s * sl2 - m), m = running max of s * sl2 (finite from the first tile on: key 0 <= limit)
        float m_new[TQ];
        bool moved = false;
#pragma unroll
        for (int qi = 0; qi < TQ; qi++) {
            const float tmax = rows_max(vmax16(s[qi][0], s[qi][1], s[qi][2], s[qi][3]));
            m_new[qi] = vmax2(m_run[qi], tmax * sl2);
            moved |= m_new[qi] != m_run[qi];
        }
        if (__any(moved)) {                                       // the running maximum moved for some row: rescale
#pragma unroll
            for (int qi = 0; qi < TQ; qi++) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new[qi]);
                lacc[qi] *= alpha;
#pragma unroll
                for (int d = 0; d < DT; d++) o[qi][d] *= alpha;
                m_run[qi] = m_new[qi];
            }
        }
#pragma unroll
        for (int qi = 0; qi < TQ; qi++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) s[qi][t][r] = __builtin_amdgcn_exp2f(fmaf(s[qi][t][r], sl2, -m_new[qi]));

        // ---- O^T += V^T · P^T : the V^T operand comes out of the row-major V image by transposed LDS reads; the k index
        // (keys 32u + 4fg + r, then 32u + 16 + 4fg + r) is permuted identically on both operands ----
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 pf[TQ];
#pragma unroll
            for (int qi = 0; qi < TQ; qi++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pf[qi][r] = (bf16_t)s[qi][2 * u][r];          // keys 32u + 4fg + r
                    pf[qi][4 + r] = (bf16_t)s[qi][2 * u + 1][r];  // keys 32u + 16 + 4fg + r
                }
#pragma unroll
            for (int d = 0; d < DT; d++) {
                const bf16x4 lo = lds_read_tr4(vimg + v_off[d] + u * (32 * CPR * 16));
                const bf16x4 hi = lds_read_tr4(vimg + v_off[d] + u * (32 * CPR * 16) + 16 * CPR * 16);
                bf16x8 vf;
#pragma unroll
                for (int r = 0; r < 4; r++) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
#pragma unroll
                for (int qi = 0; qi < TQ; qi++)
                    o[qi][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qi], o[qi][d], 0, 0, 0);
            }
            // the row sums of P (as rounded for the product above) ride the MFMA pipe too: 32 f32 adds less per tile on
            // the vector ALU, which is the busier unit here
            bf16x8 ones;
#pragma unroll
            for (int r = 0; r < 8; r++) ones[r] = (bf16_t)1.0f;
#pragma unroll
            for (int qi = 0; qi < TQ; qi++)
                lacc[qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qi], lacc[qi], 0, 0, 0);
        }
    }

#pragma unroll
    for (int qi = 0; qi < TQ; qi++) {
        const float l = lacc[qi][0];
        bool ok;
        const int row = row_of(qi, ok), si = row / p.group, hd_ = kvh * p.group + (row - si * p.group);
        if (!ok) continue;
        const float inv = 1.0f / l;
#pragma unroll
        for (int d = 0; d < DT; d++)    // O[q][d = 16d + 4fg + r], written in the next GEMM's operand layout
            act_store4<bf16_t>((bf16_t*)p.out, tok0 + si, hd_ * HD + d * 16 + fg * 4, p.out_stride, o[qi][d] * inv);
    }
}
template <int HD> constexpr int attn_prefill_lds_bytes() { return 3 * 2 * 64 * HD * 2; }

// ------------------------------------------------------------------------------------------
// bf16 decode kernel (one new token per sequence; 16 query heads of a kv head's group per workgroup).
// HBM/latency-bound KV read: one workgroup per (sequence, kv head); its NW waves take the 64-key tiles round-robin
// (split-T inside the workgroup), each with its own online-softmax state; nothing is shared between waves, so there
// is no LDS staging and no barrier in the loop, and the NW partial results are merged once through LDS
// (flash-decoding combine, fixed wave order).
//   S^T = K · Q^T on MFMA (K fragments straight HBM -> VGPR: 16 rows x 64 B per instruction);
//   V is read as WHOLE ROWS: lane (ksub = lane / CPR, dch = lane % CPR) loads chunk dch of key row i*KPI + ksub, so
//   every load instruction is one contiguous KiB of the row-major V slab and the append of the new token's V is one
//   contiguous row; the rows then pass through a wave-private LDS image (the prefill kernel's image) and come back
//   as the V^T MFMA operand by transposed reads, so P·V is 8 MFMAs per tile with P^T straight from the accumulators.
// ------------------------------------------------------------------------------------------
// rotate one (lo, hi) pair of 8-wide chunks: lo = d0..d0+7 (< HD/2), hi = the same offsets + HD/2
__device__ __forceinline__ void rope_pair8(const float* row, const float* cs, const float* sn, int d0, int half,
                                           bf16x8& lo, bf16x8& hi) {
    const f32x4 a0 = *(const f32x4*)(row + d0), a1 = *(const f32x4*)(row + d0 + 4);
    const f32x4 b0 = *(const f32x4*)(row + d0 + half), b1 = *(const f32x4*)(row + d0 + half + 4);
    f32x4 yl0 = a0, yl1 = a1, yh0 = b0, yh1 = b1;
    if (cs) {
        const f32x4 c0 = *(const f32x4*)(cs + d0), c1 = *(const f32x4*)(cs + d0 + 4);
        const f32x4 s0 = *(const f32x4*)(sn + d0), s1 = *(const f32x4*)(sn + d0 + 4);
        yl0 = a0 * c0 + (-b0) * s0; yl1 = a1 * c1 + (-b1) * s1;     // x*cos + rotate_half(x)*sin (table halves duplicate)
        yh0 = b0 * c0 + a0 * s0;    yh1 = b1 * c1 + a1 * s1;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        lo[e] = (bf16_t)yl0[e]; lo[4 + e] = (bf16_t)yl1[e];
        hi[e] = (bf16_t)yh0[e]; hi[4 + e] = (bf16_t)yh1[e];
    }
}

// the same in two steps — issue the loads, do the arithmetic later (the decode kernel puts its K/V stream in between)
struct RopeRaw { f32x4 a0, a1, b0, b1, c0, c1, s0, s1; };
__device__ __forceinline__ void rope_load8(const float* row, const float* cs, const float* sn, int d0, int half, RopeRaw& r) {
    r.a0 = *(const f32x4*)(row + d0); r.a1 = *(const f32x4*)(row + d0 + 4);
    r.b0 = *(const f32x4*)(row + d0 + half); r.b1 = *(const f32x4*)(row + d0 + half + 4);
    if (cs) {
        r.c0 = *(const f32x4*)(cs + d0); r.c1 = *(const f32x4*)(cs + d0 + 4);
        r.s0 = *(const f32x4*)(sn + d0); r.s1 = *(const f32x4*)(sn + d0 + 4);
    }
}
__device__ __forceinline__ void rope_math8(const RopeRaw& r, bool rot, bf16x8& lo, bf16x8& hi) {
    f32x4 yl0 = r.a0, yl1 = r.a1, yh0 = r.b0, yh1 = r.b1;
    if (rot) {
        yl0 = r.a0 * r.c0 + (-r.b0) * r.s0; yl1 = r.a1 * r.c1 + (-r.b1) * r.s1;
        yh0 = r.b0 * r.c0 + r.a0 * r.s0;    yh1 = r.b1 * r.c1 + r.a1 * r.s1;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        lo[e] = (bf16_t)yl0[e]; lo[4 + e] = (bf16_t)yl1[e];
        hi[e] = (bf16_t)yh0[e]; hi[4 + e] = (bf16_t)yh1[e];
    }
}

// combine of a split decode launch: the nsplit partial results of one (sequence, kv head, query tile), in workgroup order
// whichever workgroup runs it.  One wave, the lanes of real heads; lane (fq, fg) as in the kernel's own combine.
template <int HD>
__device__ __forceinline__ void attn_decode_merge(const AttnArgs& p, int pair, int tok, int head, int nsplit, int lane) {
    constexpr int DT = HD / 16;
    const int fq = lane & 15, fg = lane >> 4;
    const float* rec0 = p.part + (int64_t)pair * nsplit * attn_part_floats<HD>();
    float mstar = -INFINITY;
    for (int z = 0; z < nsplit; z++) mstar = fmaxf(mstar, rec0[(int64_t)z * attn_part_floats<HD>() + DT * 256 + fq]);
    float L = 0.f;
    f32x4 O[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) O[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nsplit; z++) {
        const float* rec = rec0 + (int64_t)z * attn_part_floats<HD>();
        const float mz = rec[DT * 256 + fq];
        if (mz == -INFINITY) continue;                 // that workgroup had no tile
        const float sc = __builtin_amdgcn_exp2f(mz - mstar);
        L += rec[DT * 256 + 16 + fq] * sc;
#pragma unroll
        for (int d = 0; d < DT; d++) O[d] += ((const f32x4*)rec)[d * 64 + lane] * sc;
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int d = 0; d < DT; d++)
        act_store4<bf16_t>((bf16_t*)p.out, tok, head * HD + d * 16 + fg * 4, p.out_stride, O[d] * inv);
}

// The six leading arguments repeat fields of AttnArgs: the library is built with -amdgpu-kernarg-preload-count=6, so they
// sit in SGPRs when a wave starts and the two scalar loads every wave needs before it can issue its first K/V load (the
// sequence's position, the block id of its first tile) leave at once — with everything inside the struct the compiler
// fetched the argument block in two dependent pieces first (in-kernel stamps: 2.4 us from entry to the first vector load
// at B = 1, against 0.4 us in the projection kernels).
template <int HD, int NW, bool FUSED, bool STAMP = false>
__global__ __launch_bounds__(NW * 64) void attn_decode_bf16_kernel(const int32_t* __restrict__ a_seq_pos, const int32_t* __restrict__ a_blk_table,
                                                                   int a_tbl_stride, int a_bs_shift, int a_Tmax, int a_group, AttnArgs p) {
    constexpr int KS = HD / 32, DT = HD / 16, HALF = HD / 2, CPR = HD / 8, KPI = 64 / CPR, NVL = 64 / KPI;
    constexpr int NT2 = HD <= 64 ? 2 : 1;                    // key tiles whose loads are in flight together, per wave
    constexpr int TILE_BYTES = 64 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [NW] wave-private V images (later: the partial O^T)
    __shared__ float red_m[NW][16];
    __shared__ float red_l[NW][16];
    // grid.x = kv head x query tile: a workgroup takes 16 of the group's query heads (GQA / MHA: one tile; Falcon's MQA:
    // 71 heads = 5 tiles that stream the same K/V — it is tiny next to the weights — from L2)
    const int swg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    NvlStampsT<STAMP> stamps(p.stamps, swg);          // (diagnostic instantiation only: see common.h)
    const int QT = (a_group + 15) >> 4;
    const int seq = blockIdx.y, kvh = blockIdx.x / QT, qt = blockIdx.x - kvh * QT;
    const int32_t* tbl = a_blk_table + (int64_t)seq * a_tbl_stride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ONE scalar round trip for everything the first K/V loads need: the token row, the position, and the block ids of this
    // wave's first tiles (which do not depend on the position).  Left to itself the compiler sinks the block-id loads behind
    // the position's wait and the tile-count branch (two more dependent scalar round trips before the first vector load:
    // +0.7 us per launch at B = 1).
    constexpr int NT2_ = HD <= 64 ? 2 : 1;
    const int vw_ = blockIdx.z * NW + wave, nvw_ = NW * gridDim.z;
    const int tok = seq, pos0 = a_seq_pos[seq];              // (decode launches carry one token per sequence, in sequence order)
    int first_blk[NT2_];
#pragma unroll
    for (int h = 0; h < NT2_; h++) {
        const int key = (vw_ + h * nvw_) * 64;
        const int bi = a_bs_shift >= 0 ? (key >> a_bs_shift) : key / a_Tmax;
        first_blk[h] = tbl[min(bi, a_tbl_stride - 1)];
    }
    const int fq = lane & 15, fg = lane >> 4;                // MFMA role: query head qt*16 + fq of the group, k group fg
    const int dch = lane % CPR, ksub = lane / CPR;           // row role: 16-byte chunk dch of key row ksub (+ i*KPI)
    const bool row_ok = qt * 16 + fq < p.group;
    const int head = kvh * p.group + (row_ok ? qt * 16 + fq : 0);
    char* vimg = smem + wave * TILE_BYTES;
    bf16x8 qf[KS];
    const int n_kt = pos0 / 64 + 1;
    // key tiles are dealt round-robin over "virtual waves": the NW waves of each of the gridDim.z workgroups of this
    // (sequence, kv head).  gridDim.z > 1 (long contexts in small batches: one workgroup would pull the whole K/V of its
    // head through one CU, ~50 GB/s) leaves partial results for attn_decode_merge_kernel.
    const int vw = blockIdx.z * NW + wave, nvw = NW * gridDim.z;
    // (FUSED: the new token's q / k / v are fetched and rotated AFTER the first round's K/V loads are issued: below)
    bf16x8 knew[KS];
    bf16x8 vnew8;
    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;     // exp2 domain
    const float sl2 = p.scale * 1.4426950408889634f;
    // LDS offsets: where this lane's V chunks go (row i*KPI + ksub, chunk dch) and where its transposed reads come from
    // (row i*KPI + ksub: the swizzle pattern only depends on i & 1, so two bases + compile-time row offsets)
    const int w_base[2] = {16 * kv_img_chunk<HD>(ksub, dch), 16 * kv_img_chunk<HD>(KPI + ksub, dch) - 16 * KPI * CPR};
    const int q4 = fq >> 2, p4 = fq & 3;
    int v_off[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) v_off[d] = 16 * kv_img_chunk<HD>(4 * fg + q4, 2 * d + (p4 >> 1)) + 8 * (p4 & 1);

    bf16x8 kf[NT2][4][KS];
    bf16x8 vch[NT2][NVL];
    // the K / V loads of one round (NT2 tiles of this wave): addresses depend on scalars only
    // NT: non-temporal loads (AttnArgs::kv_nt, chosen by the host) — a (sequence, kv head)'s K/V are read ONCE per step when
    // one workgroup serves the whole group (GQA / MHA), and lines that stay out of L2 leave it to the projections'
    // activations: decode +5 % at B = 32 (32.0 K -> 33.7 K tok/s), GPT-2 B = 128 +3.2 %; it loses 2 % at 32-64 workgroups
    // (profiles/r03_kv_nontemporal.txt).  Falcon's MQA (5 workgroups share one K/V stream through L2) keeps the default policy.
    auto issue_round_t = [&](int kt0, auto nt_tag) {
        constexpr bool NT = decltype(nt_tag)::value;
#pragma unroll
        for (int h = 0; h < NT2; h++) {
            const int kt = kt0 + h * nvw;
            if (kt >= n_kt) continue;
            const int bi = kv_block_index(p, kt * 64), krow0 = kt * 64 - bi * p.Tmax;     // the KV block holding this tile
            const int blk = kt0 == vw ? first_blk[h] : tbl[bi];
            const int64_t blk_off = (int64_t)blk * p.slot_stride + ((int64_t)kvh * p.Tmax + krow0) * HD;
            const bf16_t* kbase = (const bf16_t*)p.kcache + blk_off;
            const bf16_t* vbase = (const bf16_t*)p.vcache + blk_off;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
                {
                    const bf16x8* src = (const bf16x8*)(kbase + (int64_t)(t * 16 + fq) * HD + ks * 32 + fg * 8);
                    if constexpr (NT) kf[h][t][ks] = __builtin_nontemporal_load(src); else kf[h][t][ks] = *src;
                }
#pragma unroll
            for (int i = 0; i < NVL; i++) {                                                                 // 1 KiB, contiguous
                const bf16x8* src = (const bf16x8*)(vbase + (int64_t)lane * 8 + i * 512);
                if constexpr (NT) vch[h][i] = __builtin_nontemporal_load(src); else vch[h][i] = *src;
            }
        }
    };
    auto issue_round = [&](int kt0) {
        if (p.kv_nt) issue_round_t(kt0, std::true_type{}); else issue_round_t(kt0, std::false_type{});
    };
    // Load order (vector loads return in issue order): the new token's fp32 q/k/v row + cos/sin FIRST (a dozen small loads),
    // the round's K/V stream right behind them with no wait in between, THEN the RoPE arithmetic — it waits for the row only
    // (counted vmcnt) and runs, like the first QK^T MFMAs after it, while the K/V tiles are still arriving.  Round 2 waited
    // for the row with vmcnt(0) before issuing the first K load; the first round-3 order (K/V first, row behind) made the
    // RoPE and every MFMA wait for the whole K/V transfer (in-kernel stamps, scripts/decode_timeline.py: 2.3 us of compute
    // after 6 us of transfer, nothing overlapped).  Only the 8-wave hd-64 instantiation (GQA models at decode batches that
    // fill the chip with one workgroup per CU) has the 26 registers to spare.
    constexpr bool ROW_FIRST = FUSED && HD == 64;
    RopeRaw rq[KS / 2], rk[KS / 2];
    f32x4 v0, v1;
    const float* cs = nullptr;
    if (ROW_FIRST) {
        const float* row = p.qkv + (int64_t)tok * p.qkv_stride;
        cs = p.cos_t ? p.cos_t + (int64_t)pos0 * HD : nullptr;
        const float* sn = p.sin_t ? p.sin_t + (int64_t)pos0 * HD : nullptr;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            rope_load8(row + head * HD, cs, sn, ks * 32 + fg * 8, HALF, rq[ks]);
            rope_load8(row + (p.nH + kvh) * HD, nullptr, nullptr, ks * 32 + fg * 8, HALF, rk[ks]);     // (cos / sin: the query's copy)
        }
        const float* vrow = row + (p.nH + p.nKV + kvh) * HD + dch * 8;
        v0 = *(const f32x4*)vrow; v1 = *(const f32x4*)(vrow + 4);
    }
    // every other instantiation: the new token's row and its rotation BEFORE the K/V stream, as in round 2 — behind it the
    // row's temporaries are live together with the round's 32 K/V fragments (hd 64, 2 / 4 waves — MHA models: 237 -> 276
    // registers, one wave per SIMD instead of two, GPT-2 B = 128 attention 46 -> 51 us; hd 128, 8 waves: 22 spilled)
    if (!ROW_FIRST && FUSED) {
        const float* row = p.qkv + (int64_t)tok * p.qkv_stride;
        const float* cs2 = p.cos_t ? p.cos_t + (int64_t)pos0 * HD : nullptr;
        const float* sn = p.sin_t ? p.sin_t + (int64_t)pos0 * HD : nullptr;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            rope_pair8(row + head * HD, cs2, sn, ks * 32 + fg * 8, HALF, qf[ks], qf[ks + KS / 2]);
            rope_pair8(row + (p.nH + kvh) * HD, cs2, sn, ks * 32 + fg * 8, HALF, knew[ks], knew[ks + KS / 2]);
        }
        const float* vrow = row + (p.nH + p.nKV + kvh) * HD + dch * 8;
        const f32x4 w0 = *(const f32x4*)vrow, w1 = *(const f32x4*)(vrow + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { vnew8[e] = (bf16_t)w0[e]; vnew8[4 + e] = (bf16_t)w1[e]; }
    } else if (!FUSED) {
        const bf16_t* qp = (const bf16_t*)p.q + (int64_t)tok * p.q_stride + head * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) qf[ks] = *(const bf16x8*)(qp + ks * 32 + fg * 8);
    }
    if (vw < n_kt) issue_round(vw);
    stamps.mark(1);                     // first round's K/V loads issued
    if (ROW_FIRST) {
#pragma unroll
        for (int ks = 0; ks < KS / 2; ks++) {
            rk[ks].c0 = rq[ks].c0; rk[ks].c1 = rq[ks].c1; rk[ks].s0 = rq[ks].s0; rk[ks].s1 = rq[ks].s1;
            rope_math8(rq[ks], cs != nullptr, qf[ks], qf[ks + KS / 2]);
            rope_math8(rk[ks], cs != nullptr, knew[ks], knew[ks + KS / 2]);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { vnew8[e] = (bf16_t)v0[e]; vnew8[4 + e] = (bf16_t)v1[e]; }
    }
    stamps.mark(2);                     // q/k/v of the new token rotated
    // one round: scores, online softmax, O^T += V^T P^T of the tiles whose K/V fragments issue_round left in kf / vch.
    // (The first round is peeled — its loads went out above, ahead of the RoPE arithmetic — instead of making the issue
    // conditional inside ONE loop: there every fragment register became a loop-carried value with a select per iteration,
    // 213 -> 252 registers on the 2- and 4-wave hd-64 kernels (MHA models: one wave per SIMD instead of two) and 25
    // spilled on the 8-wave hd-128 kernel.)
    auto do_round = [&](int kt0) __attribute__((always_inline)) {
        // ---- scores of the round's tiles ----
        f32x4 s[NT2][4];
        float tmax = -INFINITY;
#pragma unroll
        for (int h = 0; h < NT2; h++) {
            const int kt = kt0 + h * nvw;
            if (kt >= n_kt) {
#pragma unroll
                for (int t = 0; t < 4; t++) s[h][t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                continue;
            }
            if (FUSED && kt == n_kt - 1) {
                // the tile that holds key pos0: its slab row is not written yet (or not visible): patch the operands
                const int kl = pos0 & 63;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t == (kl >> 4) && fq == (kl & 15)) {
#pragma unroll
                        for (int ks = 0; ks < KS; ks++) kf[h][t][ks] = knew[ks];
                    }
#pragma unroll
                for (int i = 0; i < NVL; i++)
                    if (i == kl / KPI && ksub == kl % KPI) vch[h][i] = vnew8;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[h][t][ks], qf[ks], acc, 0, 0, 0);
                const int lim = pos0 - kt * 64 - fg * 4;      // last visible key, relative to this lane's first
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[r] * sl2;
                    v = (t * 16 + r <= lim) ? v : -INFINITY;
                    acc[r] = v;
                    tmax = fmaxf(tmax, v);
                }
                s[h][t] = acc;
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);     // finite: the round's first tile kt0 < n_kt holds key kt0*64 <= pos0
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int h = 0; h < NT2; h++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __builtin_amdgcn_exp2f(s[h][t][r] - m_new);
                    s[h][t][r] = pv;
                    psum += pv;
                }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; d++) o[d] *= alpha;
        // ---- O^T += V^T · P^T: the tile's V rows go through this wave's LDS image and come back as the transposed
        // MFMA operand (ds_read_b64_tr_b16); a wave's LDS accesses execute in order, so no barrier ----
#pragma unroll
        for (int h = 0; h < NT2; h++) {
            if (kt0 + h * nvw >= n_kt) continue;
#pragma unroll
            for (int i = 0; i < NVL; i++) *(bf16x8*)(vimg + w_base[i & 1] + i * (16 * KPI * CPR)) = vch[h][i];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                bf16x8 pf;
#pragma unroll
                for (int r = 0; r < 4; r++) { pf[r] = (bf16_t)s[h][2 * u][r]; pf[4 + r] = (bf16_t)s[h][2 * u + 1][r]; }
#pragma unroll
                for (int d = 0; d < DT; d++) {
                    const bf16x4 lo = lds_read_tr4(vimg + v_off[d] + u * (32 * CPR * 16));
                    const bf16x4 hi = lds_read_tr4(vimg + v_off[d] + u * (32 * CPR * 16) + 16 * CPR * 16);
                    bf16x8 vf;
#pragma unroll
                    for (int r = 0; r < 4; r++) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
                    o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[d], 0, 0, 0);
                }
            }
        }
    };
    if (vw < n_kt) do_round(vw);
    for (int kt0 = vw + NT2 * nvw; kt0 < n_kt; kt0 += NT2 * nvw) {
        issue_round(kt0);
        do_round(kt0);
    }
    if (FUSED && blockIdx.z == 0 && qt == 0) {
        // append the new key/value to the slabs for the following steps — after every load of this step (which used
        // the register copies), so the loads above are not ordered behind these stores
        int nblk, nrow;
        kv_locate(tbl, 0, pos0, p.Tmax, nblk, nrow);
        const int64_t noff = (int64_t)nblk * p.slot_stride + ((int64_t)kvh * p.Tmax + nrow) * HD;
        if (wave == 0 && fq == 0) {      // lanes fg = 0..3 of column 0 hold all of K_new between them
#pragma unroll
            for (int ks = 0; ks < KS; ks++) *(bf16x8*)((bf16_t*)p.kcache + noff + ks * 32 + fg * 8) = knew[ks];
        }
        if (wave == 1 % NW && ksub == 0)  // lanes dch = 0..CPR-1 hold the V row: one contiguous row
            *(bf16x8*)((bf16_t*)p.vcache + noff + dch * 8) = vnew8;
    }
    // ---- flash-decoding combine through LDS (the partial O^T overwrites this wave's own V image), fixed wave order ----
    stamps.mark_used(3, o[0]);          // wave 0's tiles done
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (fg == 0) { red_m[wave][fq] = m_run; red_l[wave][fq] = l_run; }
    f32x4* my_o = (f32x4*)vimg;                              // [DT][64 lanes]
#pragma unroll
    for (int d = 0; d < DT; d++) my_o[d * 64 + lane] = o[d];
    __syncthreads();
    stamps.mark(4);                     // every wave's tiles done
    if (gridDim.z == 1 && NW >= DT) {
        // whole (sequence, kv head) in this workgroup: wave d combines output block d (same sums, same order as below;
        // one wave doing all DT blocks: B = 32 33.55 K -> 33.80 K decode tok/s, B = 8 +0.9 %, same-box A/B)
        if (wave >= DT || !row_ok) return;
        float ms = red_m[0][fq];
#pragma unroll
        for (int w = 1; w < NW; w++) ms = fmaxf(ms, red_m[w][fq]);
        float Ls = 0.f;
        f32x4 Od = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const float sc = __builtin_amdgcn_exp2f(red_m[w][fq] - ms);
            Ls += red_l[w][fq] * sc;
            Od += ((const f32x4*)(smem + w * TILE_BYTES))[wave * 64 + lane] * sc;
        }
        act_store4<bf16_t>((bf16_t*)p.out, tok, head * HD + wave * 16 + fg * 4, p.out_stride, Od * (1.0f / Ls));
        return;
    }
    if (wave != 0 || !row_ok) return;
    float mstar = red_m[0][fq];
#pragma unroll
    for (int w = 1; w < NW; w++) mstar = fmaxf(mstar, red_m[w][fq]);
    float L = 0.f;
    f32x4 O[DT];
#pragma unroll
    for (int d = 0; d < DT; d++) O[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mstar > -INFINITY) {       // (a workgroup of a split launch may have had no tile at all)
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const float sc = __builtin_amdgcn_exp2f(red_m[w][fq] - mstar);   // idle waves: exp2(-inf) = 0
            L += red_l[w][fq] * sc;
            const f32x4* wo = (const f32x4*)(smem + w * TILE_BYTES);
#pragma unroll
            for (int d = 0; d < DT; d++) O[d] += wo[d * 64 + lane] * sc;
        }
    }
    if (gridDim.z > 1) {
        const int pair = seq * gridDim.x + blockIdx.x;               // (sequence, kv head, query tile)
        float* rec = p.part + ((int64_t)pair * gridDim.z + blockIdx.z) * attn_part_floats<HD>();
#pragma unroll
        for (int d = 0; d < DT; d++) ((f32x4*)rec)[d * 64 + lane] = O[d];
        if (fg == 0) { rec[DT * 256 + fq] = mstar; rec[DT * 256 + 16 + fq] = L; }
        // Last arriver combines (no second launch: a merge kernel cost 5-7 us of the 8-11 the split saves).  Release my
        // record device-wide (the workgroups of a pair may sit on different XCDs, whose L2s are not coherent), draw a
        // ticket, and if it is the last one acquire the others' records.  The order of the sum is z order regardless.
        __threadfence();
        int ticket = 0;
        int32_t* cnt = p.part_cnt + pair;
        if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != (int)gridDim.z - 1) return;
        __threadfence();
        attn_decode_merge<HD>(p, pair, tok, head, gridDim.z, lane);
        if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // for the next launch
        return;
    }
    const float inv = 1.0f / L;
#pragma unroll
    for (int d = 0; d < DT; d++)
        act_store4<bf16_t>((bf16_t*)p.out, tok, head * HD + d * 16 + fg * 4, p.out_stride, O[d] * inv);
}

// ------------------------------------------------------------------------------------------
// fp32 parity kernel: block = (query position, head), 256 threads, T <= 8192
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs p, int HD) {
    extern __shared__ float sc[];                 // [T_max_needed] scores then probs
    __shared__ float red[8];
    const int seq = blockIdx.z, head = blockIdx.y, s_idx = blockIdx.x;
    if (s_idx >= p.seq_len[seq]) return;
    const int tid = threadIdx.x;
    const int kvh = head / p.group;
    const int tok = p.seq_tok_start[seq] + s_idx;
    const int T = p.seq_pos[seq] + s_idx + 1;     // keys 0..limit
    const int32_t* tbl = p.blk_table + (int64_t)seq * p.tbl_stride;
    const float* q = (const float*)p.q + (int64_t)tok * p.q_stride + head * HD;
    auto kv_row = [&](const void* cache, int j) {       // row of key j: [block][kvh][Tmax][HD]
        const int bi = j / p.Tmax;
        return (const float*)cache + (int64_t)tbl[bi] * p.slot_stride + ((int64_t)kvh * p.Tmax + (j - bi * p.Tmax)) * HD;
    };

    float lmax = -INFINITY;
    for (int j = tid; j < T; j += 256) {
        const float* kr = kv_row(p.kcache, j);
        float sum = 0.f;
        for (int d = 0; d < HD; d++) sum = fmaf(q[d], kr[d], sum);
        sum *= p.scale;
        sc[j] = sum;
        lmax = fmaxf(lmax, sum);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int j = tid; j < T; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = lsum;
    __syncthreads();
    const float tot = red[4] + red[5] + red[6] + red[7];
    for (int j = tid; j < T; j += 256) sc[j] = sc[j] / tot;      // attention.go:463-466
    __syncthreads();
    float* op = (float*)p.out + (int64_t)tok * p.out_stride + head * HD;
    for (int d = tid; d < HD; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < T; j++) acc = fmaf(sc[j], kv_row(p.vcache, j)[d], acc);
        op[d] = acc;
    }
}

}  // namespace nvl
