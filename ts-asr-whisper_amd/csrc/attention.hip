// Flash-style attention for head_dim 64 on gfx950: forward and backward, dense / causal / rectangular.
//
// Replaces the SDPA / flash-attn call the reference reaches through HF WhisperAttention
// (HF:modeling_whisper.py:337-351; encoder self-attention encoder.py:216-221, SE-DiCoW enrollment cross-attention
// layers.py:152-157, decoder self/cross attention HF:469-494).  q arrives pre-scaled (HF:309), scaling = 1.
//
// Forward structure (one workgroup = 4 waves = 128 query rows of one (batch, head); wave = 32 query rows):
//   S^T[key][q] = K . Q^T      "swapped" MFMA (a = K rows from LDS, b = Q rows held in registers) so that a lane
//                              owns ONE query column: row max / sum / rescale are lane-local (+1 half-wave exchange)
//   P^T -> bf16 in registers   the accumulator layout of S^T *is* a valid B-operand layout for the next MFMA as
//                              long as the A operand uses the same k-slot permutation
//                              key(x, half, e) = 16x + 8(e>>2) + 4 half + (e&3)   -- no cross-lane traffic at all
//   O^T[d][q] += V^T . P^T     A operand = V^T fragments fetched with ds_read_b64_tr_b16 (transposing LDS read)
//                              from the row-major [key][d] V tile
// K/V tiles (64 keys) stream through a 2-stage LDS ring filled by global_load_lds DMA (16 B/lane), swizzled on
// the source address (K: ds_read_b128 conflict-free; V: tr-read conflict-free).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gsrc, (lds_void_t*)lds_dst, 16, 0, 0);
}

#define HD 64
#define KV_TILE 64
#define TILE_BYTES (KV_TILE * HD * 2)        // 8 KiB
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// K-style image ([rows][64] bf16, 128-B rows): chunk c of row r at c ^ ((r>>1)&7)   (ds_read_b128 fragments)
__device__ __forceinline__ int kswz(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
// V-style image: chunk c of row r at c ^ (((r>>1)&1)<<2)                              (ds_read_b64_tr_b16 fragments)
__device__ __forceinline__ int vswz(int row, int c) { return row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4); }

// ---- K/V/Q/dO tile staging through a buffer descriptor.  A tile source holds the descriptor of one (batch, head) slice
// (num_records ends with the last valid row, so rows past the end read as ZERO -- they are masked anyway) and this wave's
// two per-lane byte offsets (row-in-tile * row stride + swizzled 16-B chunk), computed once.  A tile then costs two DMA
// instructions with a SCALAR row offset and no VALU address arithmetic -- plain VALU instructions share the SIMD's issue
// port with the MFMAs (tools/probe_overlap.hip), so per-tile pointer math was ~20 % of the loop's VALU work.
enum { SWZ_K = 0, SWZ_V = 1, SWZ_U = 2 };
struct tile_src_t { __amdgpu_buffer_rsrc_t rs; unsigned vo[2]; int row_bytes; };
__device__ __forceinline__ int rev3(int x);
template <int SWZ, int NI = 2>      // NI = DMA instructions per wave and tile: 2 when four waves share a tile, 1 when eight do
__device__ __forceinline__ tile_src_t make_tile_src(const unsigned short* base, int64_t rs, int nrows, int wave, int lane) {
    tile_src_t t;
    t.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(base), 0,
                                             (unsigned)(((int64_t)(nrows - 1) * rs + HD) * 2), 0x00020000);
    t.row_bytes = (int)(rs * 2);
    const int rr = lane >> 3, p = lane & 7;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = (wave * NI + i) * 8 + rr;
        const int c = SWZ == SWZ_V ? (p ^ (((row >> 1) & 1) << 2)) : SWZ == SWZ_K ? (p ^ ((row >> 1) & 7)) : (p ^ rev3((row >> 1) & 7));
        t.vo[i] = (unsigned)(row * (int)(rs * 2) + c * 16);
    }
    return t;
}
template <int NI = 2>
__device__ __forceinline__ void stage_tile(const tile_src_t& t, int row0, char* lds, int wave) {
    const int so = row0 * t.row_bytes;
#pragma unroll
    for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rs, (lds_void_t*)(lds + (wave * NI + i) * 1024), 16, t.vo[i], so, 0, 0);
}

// LDS-DMA the compiler must not see.  hipcc tracks every `buffer_load ... lds` it emits and puts s_waitcnt vmcnt(0) in front of the next
// LDS load whose address it cannot tell apart from the DMA's destination -- in practice every typed LDS load: the prefetch of the
// next tile, the flag poll and the incoming dQ sum would be waited for at the first fragment read of every block (attn_bwd_dkv_kernel
// paid exactly that, ~800 cycles per tile, until round 6).  The backward kernels issue their DMA from inline assembly (m0 = wave-uniform LDS address;
// nothing else in the kernel uses m0) and orders it by hand: s_waitcnt vmcnt + s_barrier.
template <int SC>      // SC: 0 = default cache policy, 1 = sc0 sc1 (the hand-off traffic: the reader's L1 is bypassed)
__device__ __forceinline__ void dma16x(unsigned lds_addr, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    if constexpr (SC) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int SC>
__device__ __forceinline__ void dma4x(unsigned lds_addr, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    if constexpr (SC) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen sc0 sc1 lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int NI = 2>
__device__ __forceinline__ void stage_tile_x(const tile_src_t& t, int row0, char* lds, int wave) {
    const unsigned so = (unsigned)(row0 * t.row_bytes), la = (unsigned)(uintptr_t)lds + (unsigned)(wave * NI) * 1024u;
#pragma unroll
    for (int i = 0; i < NI; ++i) dma16x<0>(la + i * 1024u, t.rs, t.vo[i], so);
}

// 8 transposing reads (one 32-key block x 64 d) + wait, as ONE asm statement (see gemm.hip for the rationale).
// a0/a1 = lane base addresses for d-block 0/1; OFF selects the 32-key block inside the tile.
template <int OFF>
__device__ __forceinline__ void tr_read_block(bf16x8_t (&f)[2][2], unsigned a0, unsigned a1) {
    bf16x4_t r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 1024), "i"(OFF + 2048), "i"(OFF + 3072)
        : "memory");
    // f[x][dblk]
    f[0][0] = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    f[0][1] = __builtin_shufflevector(r2, r3, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][0] = __builtin_shufflevector(r4, r5, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][1] = __builtin_shufflevector(r6, r7, 0, 1, 2, 3, 4, 5, 6, 7);
}

// lane base address (bytes, LDS) of the transposing read for d-block `dblk` of a V-style tile at `s`
__device__ __forceinline__ unsigned tr_base(const char* s, int lane, int dblk) {
    const int G = lane >> 4, u = lane & 15, hh = G >> 1;
    const int row = 4 * hh + (u >> 2);
    const int c = (dblk * 4 + 2 * (G & 1) + ((u & 3) >> 1)) ^ (((u >> 3) & 1) << 2);
    return (unsigned)(uintptr_t)(s + row * 128 + (c << 4) + ((u & 1) << 3));
}

// Sum over the 32 lanes of each half-wave with DPP (no LDS): quad swaps, half-row / row mirrors, then row_bcast15 into
// the odd rows; lane 31 ends up with the total of lanes 0-31 and lane 63 with that of lanes 32-63.
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    int x;
#define DPP_ADD(ctrl, rmask) \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false); v += __int_as_float(x);
    DPP_ADD(0xB1, 0xf)    // quad_perm [1,0,3,2]
    DPP_ADD(0x4E, 0xf)    // quad_perm [2,3,0,1]
    DPP_ADD(0x141, 0xf)   // row_half_mirror
    DPP_ADD(0x140, 0xf)   // row_mirror        -> every lane holds its 16-lane row sum
    DPP_ADD(0x142, 0xa)   // row_bcast15 into rows 1,3
#undef DPP_ADD
    return v;
}
// Fused bias gradient: column sums of a wave's [32 rows][64 d] output tile (values as stored, rows past the end excluded)
// -> one partial row of the workspace: ws[row_id][h*64 + d].  vals[d][r]: the MFMA accumulator layout of the epilogues.
__device__ __forceinline__ void tile_colsum_partial(const f32x16_t (&acc)[2], float scale, bool row_ok, float* ws_row, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // sum the bf16-rounded values that are actually stored (what a separate column-sum pass would read)
            const float v = row_ok ? bf2f(f2bf(acc[d][r] * scale)) : 0.f;
            const float t = half_wave_sum_dpp(v);
            if ((lane & 31) == 31) ws_row[d * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)] = t;
        }
}

// ---- split issue / wait forms of the transposing reads: the 8 reads of a 32-row block are issued EARLY (before the
// MFMAs / softmax that do not depend on them) and waited for right before their first consumer; the wait names the
// destination registers ("+v") so that no consumer is scheduled above it (cdna_hip_programming.md section 5.7 form (ii)).
struct tr8_t { bf16x4_t r0, r1, r2, r3, r4, r5, r6, r7; };
template <int OFF>   // V-style image (forward): a0/a1 = d-block bases
__device__ __forceinline__ void tr_issue_v(tr8_t& t, unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:%13"
        : "=&v"(t.r0), "=&v"(t.r1), "=&v"(t.r2), "=&v"(t.r3), "=&v"(t.r4), "=&v"(t.r5), "=&v"(t.r6), "=&v"(t.r7)
        : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 1024), "i"(OFF + 2048), "i"(OFF + 3072)
        : "memory");
}
template <int OFF>   // U image (backward): a{dblk}{sec}
__device__ __forceinline__ void tr_issue_u(tr8_t& t, unsigned a00, unsigned a01, unsigned a10, unsigned a11) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%13"
        : "=&v"(t.r0), "=&v"(t.r1), "=&v"(t.r2), "=&v"(t.r3), "=&v"(t.r4), "=&v"(t.r5), "=&v"(t.r6), "=&v"(t.r7)
        : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "i"(OFF), "i"(OFF + 2048)
        : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait(tr8_t& t) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(t.r0), "+v"(t.r1), "+v"(t.r2), "+v"(t.r3), "+v"(t.r4), "+v"(t.r5), "+v"(t.r6), "+v"(t.r7)
                 : "i"(N) : "memory");
}
__device__ __forceinline__ void tr_pack(bf16x8_t (&f)[2][2], const tr8_t& t) {       // f[x][dblk]
    f[0][0] = __builtin_shufflevector(t.r0, t.r1, 0, 1, 2, 3, 4, 5, 6, 7);
    f[0][1] = __builtin_shufflevector(t.r2, t.r3, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][0] = __builtin_shufflevector(t.r4, t.r5, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][1] = __builtin_shufflevector(t.r6, t.r7, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& s, int r0) {
    bf16x8_t o;
    unsigned* ou = reinterpret_cast<unsigned*>(&o);
#pragma unroll
    for (int e = 0; e < 4; ++e) ou[e] = pack_bf16x2(s[r0 + 2 * e], s[r0 + 2 * e + 1]);
    return o;
}

// XCD-aware workgroup -> (row block, head, batch) map.  Workgroups are dealt round-robin to the 8 XCDs (dispatch id L
// lands on XCD L % 8), each with a private 4 MiB L2; the natural (x = row block fastest) order therefore scatters the
// row blocks that share one head's K/V (or Q/dO) over all 8 L2s and every one of them fetches the panels again
// (measured: ~3.5x the algorithmic bytes on the fabric side of L2).  Here all row blocks of one (batch, head) get dispatch
// ids in one residue class mod 8: a (batch, head) pair is served by a single XCD and its panels are fetched once.
__device__ __forceinline__ void attn_block_coords(int nblk, int H, int B, int& blk, int& h, int& b) {
    const int L = blockIdx.x, nbh = H * B;
    const int xcd = L & 7, idx = L >> 3;
    const int full = (nbh >> 3) * nblk;               // dispatch slots per XCD covered by whole groups of 8 pairs
    int bh;
    if (idx < full) { bh = (idx / nblk) * 8 + xcd; blk = idx % nblk; }
    else {                                            // the last (nbh % 8) pairs: plain row-block-fastest order
        const int rem = (idx - full) * 8 + xcd;
        bh = (nbh & ~7) + rem / nblk; blk = rem % nblk;
    }
    h = bh % H; b = bh / H;
}

#ifndef ATTN_DKV_JIT
#define ATTN_DKV_JIT 0
#endif
#ifndef ATTN_BWD_DKV_NW8
#define ATTN_BWD_DKV_NW8 0
#endif
#ifdef DICOW_EXPERIMENTS
#include "experiments/attention_fwd_variants.inc"      // the measured-and-rejected forward kernels (see the file)
#else
#define ATTN_FWD_NW8 0
#define ATTN_FWD_PIPE 0
#endif


// ---- four workgroups per CU (round 3).  Every restructuring of the forward loop that costs the third wave per SIMD loses ~12 %
// (profiles/r03_attn_fwd_variants.txt): the loop lives on the interleave of independent waves.  This form goes the other way:
// K fragments are read just in time instead of one tile ahead (32 registers less), the ring has two slots (32 KB), so that four
// workgroups = four waves per SIMD fit (<= 128 VGPRs, 128 KB of LDS).
#ifndef ATTN_Q_NT
#define ATTN_Q_NT 0        // nontemporal Q loads / O stores of the four-workgroup forward kernel (each line is touched once)
#endif
#ifndef ATTN_O_NT
#define ATTN_O_NT 0
#endif
typedef __attribute__((ext_vector_type(2))) unsigned u32x2n_t;
#ifndef ATTN_OCC4_TAIL
#define ATTN_OCC4_TAIL 1   // skip the all-padding second key block of the last tile and the all-padding waves of the last query block
#endif
#ifndef ATTN_DQ_WGS
#define ATTN_DQ_WGS 2      // resident workgroups per CU the dq kernel's registers are cut for (3: 168 VGPRs)
#endif
#ifndef ATTN_BWD_TAIL
#define ATTN_BWD_TAIL 1
#endif
#ifndef ATTN_DKV_CTS
#define ATTN_DKV_CTS 0     // dkv kernel: the ring slot as a COMPILE-TIME constant (tile loop unrolled by two): every LDS address = lane register + immediate
#endif
#ifndef ATTN_DKV_1BAR
#define ATTN_DKV_1BAR 0    // with ATTN_DKV_CTS: ONE barrier per query tile (the next tile is requested after it, not before)
#endif
template <int N> struct attn_ic { static constexpr int value = N; };
#ifndef ATTN_OCC4_PRIO
#define ATTN_OCC4_PRIO 0   // experiments: 1 = priority 1 in the S^T segment, 2 = in the softmax + PV segment, 3 / 4 = static by workgroup parity
#endif
#ifndef ATTN_OCC4_SPEC
#define ATTN_OCC4_SPEC 1
#endif
#ifndef ATTN_OCC4_PKSUM
#define ATTN_OCC4_PKSUM 0
#endif
#ifndef ATTN_FWD_OCC4
#define ATTN_FWD_OCC4 1
#endif
// SPEC (with LOG2): a speculative first pass over all tiles WITHOUT any row maximum -- p = 2^s against exponent zero, which is
// exact while every row's sum stays inside the fp32 / bf16 exponent range; the maximum (25 of the loop's ~140 VALU instructions,
// and the loop is VALU-bound) was only a guard.  The guard moves to the end: a row whose sum left [2^-100, 2^100] (or is not
// finite) makes its workgroup vote for a second, ordinary pass (the loop below, with the moving reference exponent) that
// recomputes the block from scratch.  Both passes run one barrier and one staging step per tile in every wave; the vote is
// workgroup-uniform.
template <bool LOG2, bool SPEC = false>
__global__ void __launch_bounds__(256, 4) attn_fwd_occ4_kernel(const dicow_attn_fwd_args a) {
    static_assert(!SPEC || LOG2, "the speculative pass needs base-2 scores");
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];      // two (K, V) slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    int qblk, h, b;
    attn_block_coords((a.Lq + 127) / 128, a.H, a.B, qblk, h, b);
    const int q0 = qblk * 128;
    const unsigned short* Q = reinterpret_cast<const unsigned short*>(a.q) + (int64_t)b * a.q_bs + h * HD;
    const unsigned short* K = reinterpret_cast<const unsigned short*>(a.k) + (int64_t)b * a.k_bs + h * HD;
    const unsigned short* V = reinterpret_cast<const unsigned short*>(a.v) + (int64_t)b * a.v_bs + h * HD;
    int qrow = q0 + wave * 32 + (lane & 31);
    const int qrow_c = qrow < a.Lq ? qrow : a.Lq - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        qf[kk] = ATTN_Q_NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(Q + (int64_t)qrow_c * a.q_rs + kk * 16 + hh * 8))
                           : *reinterpret_cast<const bf16x8_t*>(Q + (int64_t)qrow_c * a.q_rs + kk * 16 + hh * 8);
    f32x16_t o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_ref = LOG2 ? 0.f : -INFINITY, l_run = 0.f;
    bool zero_ref = true;                                     // LOG2: m_ref == 0 in every lane of the wave (wave-uniform)
    int kv_end = a.Lk;
    if (a.causal) { const int lim = q0 + 128 < a.Lk ? q0 + 128 : a.Lk; kv_end = lim; }
    const int nt = (kv_end + KV_TILE - 1) / KV_TILE;
    const tile_src_t srcK = make_tile_src<SWZ_K>(K, a.k_rs, a.Lk, wave, lane), srcV = make_tile_src<SWZ_V>(V, a.v_rs, a.Lk, wave, lane);
    stage_tile(srcK, 0, smem, wave);
    stage_tile(srcV, 0, smem + TILE_BYTES, wave);
    if (nt > 1) {
        stage_tile(srcK, KV_TILE, smem + 2 * TILE_BYTES, wave);
        stage_tile(srcV, KV_TILE, smem + 3 * TILE_BYTES, wave);
    }
    bool general = true;
    if constexpr (SPEC) {
        if (ATTN_OCC4_PRIO == 3 && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(1);
        if (ATTN_OCC4_PRIO == 4 && (blockIdx.x & 2)) __builtin_amdgcn_s_setprio(1);
        // Padding: Lk = 1500 is 23.44 tiles of 64 keys and 11.7 blocks of 128 queries.  When the LAST tile's second key block lies
        // entirely past Lk it is not computed (NKB = 1: 1/48 of the matrix and softmax work of every workgroup), and a wave whose 32
        // query rows all lie past Lq (the fourth wave of the last query block) only keeps the barriers and its share of the staging.
        const bool tail_half = ATTN_OCC4_TAIL && !a.causal && nt * KV_TILE - a.Lk >= 32;
        const bool wave_live = !ATTN_OCC4_TAIL || q0 + wave * 32 < a.Lq;
        auto spec_tile = [&](int t, auto nkb_tag) {
            constexpr int NKB = decltype(nkb_tag)::value;
            char* sK = smem + (t & 1) * 2 * TILE_BYTES;
            char* sV = sK + TILE_BYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t >= 1 && t + 1 < nt) {
                char* nK = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
                stage_tile(srcK, (t + 1) * KV_TILE, nK, wave);
                stage_tile(srcV, (t + 1) * KV_TILE, nK + TILE_BYTES, wave);
            }
            if (!wave_live) return;
            if (ATTN_OCC4_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            if (ATTN_OCC4_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            f32x16_t s[2];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                bf16x8_t kf[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const bf16x8_t*>(sK + kswz(kb * 32 + (lane & 31), kk * 2 + hh));
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s[kb], 0, 0, 0);
            }
            const unsigned va0 = tr_base(sV, lane, 0), va1 = tr_base(sV, lane, 1);
            tr8_t tv0, tv1;
            tr_issue_v<0>(tv0, va0, va1);
            const int k0 = t * KV_TILE;
            const bool need_mask = (k0 + NKB * 32 > a.Lk) || (a.causal && (k0 + KV_TILE - 1 > q0 + wave * 32));
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key >= a.Lk || (a.causal && key > qrow)) s[kb][r] = -INFINITY;
                    }
            }
            if (ATTN_OCC4_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (ATTN_OCC4_PRIO == 2) __builtin_amdgcn_s_setprio(1);
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][r]);
                    s[kb][r] = p;
                    psum += p;
                }
            l_run += psum;
            bf16x8_t vf[2][2];
            tr_wait<0>(tv0);
            if constexpr (NKB == 2) tr_issue_v<4096>(tv1, va0, va1);
            tr_pack(vf, tv0);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bf16x8_t pf = pack8(s[0], 8 * x);
#pragma unroll
                for (int d = 0; d < 2; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[x][d], pf, o[d], 0, 0, 0);
            }
            if constexpr (NKB == 2) {
                tr_wait<0>(tv1);
                tr_pack(vf, tv1);
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    const bf16x8_t pf = pack8(s[1], 8 * x);
#pragma unroll
                    for (int d = 0; d < 2; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[x][d], pf, o[d], 0, 0, 0);
                }
            }
        };
        const int nfull = tail_half ? nt - 1 : nt;
        for (int t = 0; t < nfull; ++t) spec_tile(t, attn_ic<2>{});
        if (tail_half) spec_tile(nt - 1, attn_ic<1>{});
        // ---- the guard, and the workgroup's vote
        const float lt = l_run + __shfl_xor(l_run, 32, 64);
        const bool row_bad = qrow < a.Lq && !(lt > 0x1p-100f && lt < 0x1p100f);       // (false for NaN, too)
        const bool wave_bad = __builtin_amdgcn_ballot_w64(row_bad) != 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave has left the last tile: the LDS is free
        asm volatile("" ::: "memory");
        if (lane == 0) reinterpret_cast<volatile int*>(smem)[wave] = wave_bad ? 1 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const volatile int* vt = reinterpret_cast<const volatile int*>(smem);
        general = (vt[0] | vt[1] | vt[2] | vt[3]) != 0;
        general = __builtin_amdgcn_readfirstlane(general ? 1 : 0) != 0;
        if (general) {                                        // start over: accumulators, sum, the first two tiles
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // the votes have been read
            asm volatile("" ::: "memory");
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
            l_run = 0.f;
            stage_tile(srcK, 0, smem, wave);
            stage_tile(srcV, 0, smem + TILE_BYTES, wave);
            if (nt > 1) {
                stage_tile(srcK, KV_TILE, smem + 2 * TILE_BYTES, wave);
                stage_tile(srcV, KV_TILE, smem + 3 * TILE_BYTES, wave);
            }
        }
    }
    if (general)
    for (int t = 0; t < nt; ++t) {
        char* sK = smem + (t & 1) * 2 * TILE_BYTES;
        char* sV = sK + TILE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile t landed (this wave's share)
        __builtin_amdgcn_s_barrier();                         // ... everybody's, and every wave has left tile t-1
        asm volatile("" ::: "memory");
        if (t >= 1 && t + 1 < nt) {
            char* nK = smem + ((t + 1) & 1) * 2 * TILE_BYTES; // slot of tile t-1
            stage_tile(srcK, (t + 1) * KV_TILE, nK, wave);
            stage_tile(srcV, (t + 1) * KV_TILE, nK + TILE_BYTES, wave);
        }
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8_t kf[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<const bf16x8_t*>(sK + kswz(kb * 32 + (lane & 31), kk * 2 + hh));
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s[kb], 0, 0, 0);
        }
        const unsigned va0 = tr_base(sV, lane, 0), va1 = tr_base(sV, lane, 1);
        tr8_t tv0, tv1;
        tr_issue_v<0>(tv0, va0, va1);                         // (the second block's fragments are requested behind the first's wait: 16 registers)
        const int k0 = t * KV_TILE;
        const bool need_mask = (k0 + KV_TILE > a.Lk) || (a.causal && (k0 + KV_TILE - 1 > q0 + wave * 32));
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= a.Lk || (a.causal && key > qrow)) s[kb][r] = -INFINITY;
                }
        }
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float psum = 0.f;
        if constexpr (LOG2) {
            // the reference exponent starts at zero and p = 2^s needs nothing in front of the exponential; it moves (lazily, as in
            // the other mode) only when a row maximum leaves [-60, 60] -- upwards at any tile, downwards at a row's first tile --
            // and from then on this wave pays a subtraction pass per tile
            const bool need = mx > m_ref + 60.0f || (t == 0 && mx < -60.0f && mx > -INFINITY);
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                const float m_new = need ? (t == 0 ? mx : fmaxf(m_ref, mx)) : m_ref;
                const float alpha = __builtin_amdgcn_exp2f(m_ref - m_new);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                m_ref = m_new;
                zero_ref = false;
            }
            if (!zero_ref) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] -= m_ref;
            }
#if ATTN_OCC4_PKSUM
            f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(s[kb][r]), p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                    s[kb][r] = p0; s[kb][r + 1] = p1;
                    ps2 += f32x2_t{p0, p1};                       // v_pk_add_f32: 16 instructions for the 32 row-sum adds
                }
            psum = ps2.x + ps2.y;
#else
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][r]);
                    s[kb][r] = p;
                    psum += p;
                }
#endif
        } else {
        if (__builtin_amdgcn_ballot_w64(mx > m_ref + 8.0f * LN2) != 0) {
            const float m_new = fmaxf(m_ref, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_ref - m_use) * LOG2E);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            m_ref = m_new;
        }
        const float mL = (m_ref == -INFINITY) ? 0.f : m_ref * LOG2E;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, -mL));
                s[kb][r] = p;
                psum += p;
            }
        }
        l_run += psum;
        bf16x8_t vf[2][2];
        tr_wait<0>(tv0);
        tr_issue_v<4096>(tv1, va0, va1);
        tr_pack(vf, tv0);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8_t pf = pack8(s[0], 8 * x);
#pragma unroll
            for (int d = 0; d < 2; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[x][d], pf, o[d], 0, 0, 0);
        }
        tr_wait<0>(tv1);
        tr_pack(vf, tv1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const bf16x8_t pf = pack8(s[1], 8 * x);
#pragma unroll
            for (int d = 0; d < 2; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[x][d], pf, o[d], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < a.Lq) {
        unsigned short* O = reinterpret_cast<unsigned short*>(a.o) + (int64_t)b * a.o_bs + (int64_t)qrow * a.o_rs + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = d * 32 + 8 * q4 + 4 * hh;
                const u32x2n_t ov = {pack_bf16x2(o[d][4 * q4] * inv_l, o[d][4 * q4 + 1] * inv_l),
                                     pack_bf16x2(o[d][4 * q4 + 2] * inv_l, o[d][4 * q4 + 3] * inv_l)};
                if (ATTN_O_NT) __builtin_nontemporal_store(ov, reinterpret_cast<u32x2n_t*>(O + col));
                else *reinterpret_cast<u32x2n_t*>(O + col) = ov;
            }
        if (a.lse && hh == 0)
            a.lse[((int64_t)b * a.H + h) * a.Lq + qrow] = LOG2 ? (m_ref + __builtin_amdgcn_logf(l_tot)) * LN2      // (v_log_f32 is log2)
                                                               : m_ref + __builtin_amdgcn_logf(l_tot) * LN2;
    }
}

static int check_strides(int64_t rs, const char* n) {
    if (rs % 8 != 0) { dicow_set_error("attention: %s row stride must be a multiple of 8 elements", n); return 0; }
    return 1;
}

extern "C" int dicow_attn_fwd(const dicow_attn_fwd_args* a, void* stream) {
    DICOW_REQUIRE(a && a->q && a->k && a->v && a->o, "attn_fwd: null operand");
    DICOW_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0, "attn_fwd: empty problem");
    DICOW_REQUIRE(a->H <= 65535 && a->B <= 65535, "attn_fwd: B/H too large for the grid");
    if (!check_strides(a->q_rs, "q") || !check_strides(a->k_rs, "k") || !check_strides(a->v_rs, "v") ||
        !check_strides(a->o_rs, "o")) return DICOW_ERR_INVALID;
    DICOW_REQUIRE(a->q_bs % 8 == 0 && a->k_bs % 8 == 0 && a->v_bs % 8 == 0 && a->o_bs % 4 == 0, "attn_fwd: batch strides must keep 16-byte alignment");
    dim3 grid(dicow_cdiv(a->Lq, 128) * a->H * a->B);
#ifdef DICOW_EXPERIMENTS
    static const int ncu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
    const int64_t wg8 = (int64_t)dicow_cdiv(a->Lq, 256) * a->H * a->B;
    if (ATTN_FWD_NW8 && !a->q_log2 && wg8 >= 2 * ncu) {
        hipLaunchKernelGGL((attn_fwd_kernel<false, 8>), dim3((unsigned)wg8), dim3(512), 0, (hipStream_t)stream, *a);
        DICOW_CHECK_LAUNCH("attn_fwd (8 waves)");
        return DICOW_OK;
    }
    if (!ATTN_FWD_OCC4) {
        if (a->q_log2) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        else if (ATTN_FWD_PIPE && !a->causal) hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        DICOW_CHECK_LAUNCH("attn_fwd (experiment)");
        return DICOW_OK;
    }
#endif
    if (a->q_log2 && ATTN_OCC4_SPEC) hipLaunchKernelGGL((attn_fwd_occ4_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, *a);
    else if (a->q_log2) hipLaunchKernelGGL((attn_fwd_occ4_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_fwd_occ4_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, *a);
    DICOW_CHECK_LAUNCH("attn_fwd");
    return DICOW_OK;
}

// ================================================================================================ backward
// Three kernels (deterministic, no atomics):
//   (delta[b,h,q] = sum_d dO[q,d] * O[q,d] is computed in the dq kernel prologue and published with -lse for the dkv kernel)
//   attn_bwd_dq_kernel: per 128-query block, loop over key tiles   (S^T, dP^T, dQ^T += K^T . dS^T)
//   attn_bwd_dkv_kernel: per 128-key block, loop over query tiles  (S, dP, dV^T += dO^T . P, dK^T += Q^T . dS)
// All LDS tiles use the "universal" image U: 16-B chunk c of row r stored at c ^ rev3((r>>1)&7), which is
// conflict-free both for ds_read_b128 row fragments and for ds_read_b64_tr_b16 column fragments.
__device__ __forceinline__ int rev3(int x) { return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1); }
__device__ __forceinline__ int uswz(int row, int c) { return row * 128 + ((c ^ rev3((row >> 1) & 7)) << 4); }

// lane base address of a transposing read on a U image: rows 4*half + (u>>2) (+8*sec), 16 columns of d-block dblk
__device__ __forceinline__ unsigned tr_base_u(const char* s, int lane, int dblk, int sec) {
    const int G = lane >> 4, u = lane & 15, hh = G >> 1;
    const int row = 8 * sec + 4 * hh + (u >> 2);
    const int c = (dblk * 4 + 2 * (G & 1) + ((u & 3) >> 1)) ^ rev3((row >> 1) & 7);
    return (unsigned)(uintptr_t)(s + row * 128 + (c << 4) + ((u & 1) << 3));
}

// 8 transposing reads of one 32-row block (rows OFF/128 .. +31) x 64 columns; f[x][dblk], x = 16-row half
template <int OFF>
__device__ __forceinline__ void tr_read_block_u(bf16x8_t (&f)[2][2], unsigned a00, unsigned a01, unsigned a10, unsigned a11) {
    // a{dblk}{sec}
    bf16x4_t r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "i"(OFF), "i"(OFF + 2048)
        : "memory");
    f[0][0] = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    f[0][1] = __builtin_shufflevector(r2, r3, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][0] = __builtin_shufflevector(r4, r5, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][1] = __builtin_shufflevector(r6, r7, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------ dQ
template <bool LOG2>        // q carries log2 e (compile-time: the per-score multiply in front of the exponential disappears)
__global__ void __launch_bounds__(256, ATTN_DQ_WGS) attn_bwd_dq_kernel(const dicow_attn_bwd_args a) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];      // K0 V0 K1 V1 (U images)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    int qblk, h, b;
    attn_block_coords((a.Lq + 127) / 128, a.H, a.B, qblk, h, b);
    const int q0 = qblk * 128;
    const unsigned short* Q = reinterpret_cast<const unsigned short*>(a.q) + (int64_t)b * a.q_bs + h * HD;
    const unsigned short* K = reinterpret_cast<const unsigned short*>(a.k) + (int64_t)b * a.k_bs + h * HD;
    const unsigned short* V = reinterpret_cast<const unsigned short*>(a.v) + (int64_t)b * a.v_bs + h * HD;
    const unsigned short* dO = reinterpret_cast<const unsigned short*>(a.d_o) + (int64_t)b * a.do_bs + h * HD;

    const int qrow = q0 + wave * 32 + (lane & 31);
    const int qrow_c = qrow < a.Lq ? qrow : a.Lq - 1;
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *reinterpret_cast<const bf16x8_t*>(Q + (int64_t)qrow_c * a.q_rs + kk * 16 + hh * 8);
        dof[kk] = *reinterpret_cast<const bf16x8_t*>(dO + (int64_t)qrow_c * a.do_rs + kk * 16 + hh * 8);
    }
    const int64_t stat = ((int64_t)b * a.H + h) * a.Lq + qrow_c;
    // -lse and -delta seed the S and dP accumulators, so the MFMA chain itself delivers (s - lse) and (dP - delta): plain
    // VALU work and MFMAs share one issue port per SIMD (tools/probe_overlap.hip: they do not overlap, transcendentals do),
    // which makes every VALU instruction shaved off the softmax recompute a direct saving.
    // delta[q] = rowsum(dO * O) is computed right here from the row this lane pair already holds (it used to be a kernel of
    // its own); the negated (delta, lse) planes are published for attn_bwd_dkv_kernel, which is launched after this one.
    const unsigned short* Op = reinterpret_cast<const unsigned short*>(a.o) + (int64_t)b * a.o_bs + h * HD;
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(Op + (int64_t)qrow_c * a.o_rs + kk * 16 + hh * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            dsum = fmaf(bfbits2f((unsigned short)of[e]), bfbits2f((unsigned short)dof[kk][e]), dsum);
    }
    dsum += __shfl_xor(dsum, 32, 64);
    // q_log2: the scores are base-2 exponents already (q carries log2 e): seed with -lse in base-2 units, exponent scale 1
    const float nlse = LOG2 ? -a.lse[stat] * LOG2E : -a.lse[stat];
    const float ndlt = -dsum;
    if (hh == 0 && qrow < a.Lq) {
        a.delta[stat] = ndlt;
        a.delta[(int64_t)a.B * a.H * a.Lq + stat] = nlse;
    }

    f32x16_t dq[2], seed_s, seed_p;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { seed_s[r] = nlse; seed_p[r] = ndlt; }

    int kv_end = a.Lk;
    if (a.causal) kv_end = q0 + 128 < a.Lk ? q0 + 128 : a.Lk;
    const int nt = (kv_end + KV_TILE - 1) / KV_TILE;

    const tile_src_t srcK = make_tile_src<SWZ_U>(K, a.k_rs, a.Lk, wave, lane), srcV = make_tile_src<SWZ_U>(V, a.v_rs, a.Lk, wave, lane);
    // lane bases of the transposing reads, once (the stage only adds a scalar)
    const unsigned kb00 = tr_base_u(smem, lane, 0, 0), kb01 = tr_base_u(smem, lane, 0, 1);
    const unsigned kb10 = tr_base_u(smem, lane, 1, 0), kb11 = tr_base_u(smem, lane, 1, 1);
    stage_tile(srcK, 0, smem, wave);
    stage_tile(srcV, 0, smem + TILE_BYTES, wave);
    // (padding, as in attn_fwd_occ4_kernel: the last tile's all-padding second key block is not computed, an all-padding wave of the
    // last query block keeps only the barriers and its share of the staging)
    const bool tail_half = ATTN_BWD_TAIL && !a.causal && nt * KV_TILE - a.Lk >= 32;
    const bool wave_live = !ATTN_BWD_TAIL || q0 + wave * 32 < a.Lq;
    auto dq_tile = [&](int t, auto nkb_tag) {
        constexpr int NKB = decltype(nkb_tag)::value;
        char* sK = smem + (t & 1) * 2 * TILE_BYTES;
        char* sV = sK + TILE_BYTES;
        if (t + 1 < nt) {
            char* nK = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
            stage_tile(srcK, (t + 1) * KV_TILE, nK, wave);
            stage_tile(srcV, (t + 1) * KV_TILE, nK + TILE_BYTES, wave);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        if (wave_live) {
        const int k0 = t * KV_TILE;
        const bool need_mask = (k0 + NKB * 32 > a.Lk) || (a.causal && (k0 + KV_TILE - 1 > q0 + wave * 32));
        // K^T fragments for the dQ product: issued now, consumed after S / dP / dS
        const unsigned stg = (unsigned)((t & 1) * 2 * TILE_BYTES);
        const unsigned k00 = kb00 + stg, k01 = kb01 + stg, k10 = kb10 + stg, k11 = kb11 + stg;
        tr8_t tk0, tk1;
        tr_issue_u<0>(tk0, k00, k01, k10, k11);
        if constexpr (NKB == 2) tr_issue_u<4096>(tk1, k00, k01, k10, k11);
        f32x16_t ds[2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            // the first MFMA of each chain takes the loop-invariant seed registers as C and writes a fresh D: no per-tile
            // v_mov of 2 x 16 seed values (they were a quarter of this loop's VALU instructions)
            f32x16_t s = seed_s, dp = seed_p;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + uswz(kb * 32 + (lane & 31), kk * 2 + hh));
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + uswz(kb * 32 + (lane & 31), kk * 2 + hh));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[kk], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = LOG2 ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * LOG2E);
            if (need_mask) {                          // one branch per block: a test inside the score loop becomes 16
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= a.Lk || (a.causal && key > qrow)) s[r] = 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[kb][r] = s[r] * dp[r];
        }
        // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
        {
            bf16x8_t ktf[2][2];
            tr_wait<0>(tk0);      // (the compiler-scheduled K/V row reads above share the LDS queue: drain it)
            tr_pack(ktf, tk0);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bf16x8_t pf = pack8(ds[0], 8 * x);
#pragma unroll
                for (int d = 0; d < 2; ++d) dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[x][d], pf, dq[d], 0, 0, 0);
            }
            if constexpr (NKB == 2) {
            tr_wait<0>(tk1);
            tr_pack(ktf, tk1);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bf16x8_t pf = pack8(ds[1], 8 * x);
#pragma unroll
                for (int d = 0; d < 2; ++d) dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[x][d], pf, dq[d], 0, 0, 0);
            }
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        const int nfull = tail_half ? nt - 1 : nt;
        for (int t = 0; t < nfull; ++t) dq_tile(t, attn_ic<2>{});
        if (tail_half) dq_tile(nt - 1, attn_ic<1>{});
    }
    if (a.dq_colsum) {        // q_proj bias gradient, fused: partial row (b, q block, wave) of the first workspace plane
        const int nqb = (a.Lq + 127) / 128;
        float* wsr = reinterpret_cast<float*>(a.cs_ws) + ((int64_t)((b * nqb + qblk) * 4 + wave) * a.H + h) * HD;
        tile_colsum_partial(dq, a.dq_scale, qrow < a.Lq, wsr, lane);
    }
    if (qrow < a.Lq) {
        unsigned short* DQ = reinterpret_cast<unsigned short*>(a.dq) + (int64_t)b * a.dq_bs + (int64_t)qrow * a.dq_rs + h * HD;
        const float sc = a.dq_scale;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = d * 32 + 8 * q4 + 4 * hh;
                *reinterpret_cast<uint2*>(DQ + col) =
                    make_uint2(pack_bf16x2(dq[d][4 * q4] * sc, dq[d][4 * q4 + 1] * sc),
                               pack_bf16x2(dq[d][4 * q4 + 2] * sc, dq[d][4 * q4 + 3] * sc));
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
// lse / delta of a 64-query tile -> LDS (one 4-byte DMA per lane; waves 0/2 fetch lse, waves 1/3 delta, so every wave
// issues the same number of VMEM ops and one counted vmcnt serves all)
__device__ __forceinline__ void stage_stats64(const float* lse, const float* delta, int q0, int Lq, char* dst, int wave, int lane) {
    int q = q0 + lane; q = q < Lq ? q : Lq - 1;
    const float* src = (wave & 1) ? delta + q : lse + q;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(dst + (wave & 1) * 256), 4, 0, 0);
}

template <int NW, bool LOG2>      // NW waves = NW * 32 keys per workgroup, sharing each Q / dO tile (8: half the DMA instructions per wave); LOG2 as in attn_bwd_dq_kernel
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 1 : (ATTN_DKV_JIT ? 3 : 2)) attn_bwd_dkv_kernel(const dicow_attn_bwd_args a) {
    constexpr int KB = NW * 32, NI = 8 / NW;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES + 1024];   // Q0 dO0 Q1 dO1 (U images) + lse/delta x2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    int kblk, h, b;
    attn_block_coords((a.Lk + KB - 1) / KB, a.H, a.B, kblk, h, b);
    const int kblk0 = kblk * KB;
    const unsigned short* Q = reinterpret_cast<const unsigned short*>(a.q) + (int64_t)b * a.q_bs + h * HD;
    const unsigned short* K = reinterpret_cast<const unsigned short*>(a.k) + (int64_t)b * a.k_bs + h * HD;
    const unsigned short* V = reinterpret_cast<const unsigned short*>(a.v) + (int64_t)b * a.v_bs + h * HD;
    const unsigned short* dO = reinterpret_cast<const unsigned short*>(a.d_o) + (int64_t)b * a.do_bs + h * HD;
    const float* lse = a.delta + (int64_t)a.B * a.H * a.Lq + ((int64_t)b * a.H + h) * a.Lq;     // -lse plane of the workspace
    const float* delta = a.delta + ((int64_t)b * a.H + h) * a.Lq;                                // -delta plane

    // this wave's 32 keys as B operands (column = key, k-slots = d)
    const int key = kblk0 + wave * 32 + (lane & 31);
    const int key_c = key < a.Lk ? key : a.Lk - 1;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8_t*>(K + (int64_t)key_c * a.k_rs + kk * 16 + hh * 8);
        vf[kk] = *reinterpret_cast<const bf16x8_t*>(V + (int64_t)key_c * a.v_rs + kk * 16 + hh * 8);
    }
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    // q_log2: q = log2(e) q_true and the -lse plane was published in base-2 units by the dq kernel: exponent scale 1, and
    // dK = dS^T q_true = ln 2 * dS^T q
    constexpr float dk_mul = LOG2 ? LN2 : 1.0f;

    const int t0 = a.causal ? (kblk0 / KV_TILE) : 0;                 // query tiles entirely before the key block see none of it
    const int nt = (a.Lq + KV_TILE - 1) / KV_TILE;
    const tile_src_t srcQ = make_tile_src<SWZ_U, NI>(Q, a.q_rs, a.Lq, wave, lane), srcdO = make_tile_src<SWZ_U, NI>(dO, a.do_rs, a.Lq, wave, lane);
    // (round 6: the DMA is issued from inline assembly -- see dma16x: the compiler's own LDS-DMA made it wait for the NEXT tile's DMA in
    // front of the first seed read of every block)
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0) as a builtin: the compiler's scoreboard learns that the K / V fragment loads are done
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(delta), 0,
                                                                         (unsigned)(((int64_t)a.B * a.H * a.Lq + a.Lq) * 4), 0x00020000);
    const unsigned stat_plane = (wave & 1) ? 0u : (unsigned)((int64_t)a.B * a.H * a.Lq * 4);
    auto stage_stats_x = [&](int q0, char* dst) {
        int q = q0 + lane; q = q < a.Lq ? q : a.Lq - 1;
        dma4x<0>((unsigned)(uintptr_t)dst + (unsigned)(wave & 1) * 256u, rsS, stat_plane + (unsigned)q * 4u, 0u);
    };
    if (t0 < nt) {
        stage_tile_x<NI>(srcQ, t0 * KV_TILE, smem, wave);
        stage_tile_x<NI>(srcdO, t0 * KV_TILE, smem + TILE_BYTES, wave);
        stage_stats_x(t0 * KV_TILE, smem + 4 * TILE_BYTES);
    }
    // lane-derived row-fragment offsets (swizzle XORs) computed once: plain VALU instructions share the SIMD's issue port
    // with the MFMAs, so per-tile address arithmetic is pure loss
    int fo[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fo[qb][kk] = uswz(qb * 32 + (lane & 31), kk * 2 + hh);
    const unsigned qb00 = tr_base_u(smem, lane, 0, 0), qb01 = tr_base_u(smem, lane, 0, 1);
    const unsigned qb10 = tr_base_u(smem, lane, 1, 0), qb11 = tr_base_u(smem, lane, 1, 1);
    // (padding, as in the other attention kernels: the all-padding second query block of the last tile is not computed; a wave whose
    // 32 keys all lie past Lk keeps only the barriers and its share of the staging)
    const bool tail_half = ATTN_BWD_TAIL && nt * KV_TILE - a.Lq >= 32 && nt - 1 > t0;
    const bool wave_live = !ATTN_BWD_TAIL || kblk0 + wave * 32 < a.Lk;
#if ATTN_DKV_CTS
#include "experiments/attention_dkv_cts.inc"       // compile-time ring slot / one barrier per tile: measured equal (profiles/r04_attn_bwd_variants.txt)
#else
    auto dkv_tile = [&](int t, auto nqb_tag) {
        constexpr int NQB = decltype(nqb_tag)::value;
        char* sQ = smem + ((t - t0) & 1) * 2 * TILE_BYTES;
        char* sdO = sQ + TILE_BYTES;
        if (t + 1 < nt) {
            char* nQ = smem + ((t - t0 + 1) & 1) * 2 * TILE_BYTES;
            stage_tile_x<NI>(srcQ, (t + 1) * KV_TILE, nQ, wave);
            stage_tile_x<NI>(srcdO, (t + 1) * KV_TILE, nQ + TILE_BYTES, wave);
            stage_stats_x((t + 1) * KV_TILE, smem + 4 * TILE_BYTES + ((t - t0 + 1) & 1) * 512);
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * NI + 1) : "memory");     // this tile landed; the next one (2 NI tile pieces + 1 statistics piece) may fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        const int qt0 = t * KV_TILE;
        const float* sStat = reinterpret_cast<const float*>(smem + 4 * TILE_BYTES + ((t - t0) & 1) * 512);
        const bool need_mask = (qt0 + KV_TILE > a.Lq) || (kblk0 + KB > a.Lk) || (a.causal && (kblk0 + wave * 32 + 31 > qt0));
        const unsigned stg = (unsigned)(((t - t0) & 1) * 2 * TILE_BYTES);
        const unsigned q00 = qb00 + stg, q01 = qb01 + stg, q10 = qb10 + stg, q11 = qb11 + stg;
        const unsigned o00 = q00 + TILE_BYTES, o01 = q01 + TILE_BYTES, o10 = q10 + TILE_BYTES, o11 = q11 + TILE_BYTES;
#if ATTN_DKV_JIT
#include "experiments/attention_dkv_jit.inc"       // three waves per SIMD with just-in-time fragments: measured equal (profiles/r03_attn_bwd_variants.txt)
#else
#define DKV_QBLOCK(QB)                                                                                                  \
        {                                                                                                               \
            tr8_t tdo, tq;                                                                                              \
            tr_issue_u<(QB) * 4096>(tdo, o00, o01, o10, o11);                                                           \
            tr_issue_u<(QB) * 4096>(tq, q00, q01, q10, q11);                                                            \
            f32x16_t s, dp;                       /* accumulators seeded with -lse[q], -delta[q] (see attn_bwd_dq_kernel) */ \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                          \
                const float4 lv = *reinterpret_cast<const float4*>(sStat + (QB) * 32 + 8 * q4 + 4 * hh);                 \
                const float4 dv4 = *reinterpret_cast<const float4*>(sStat + 64 + (QB) * 32 + 8 * q4 + 4 * hh);          \
                s[4 * q4] = lv.x; s[4 * q4 + 1] = lv.y; s[4 * q4 + 2] = lv.z; s[4 * q4 + 3] = lv.w;                     \
                dp[4 * q4] = dv4.x; dp[4 * q4 + 1] = dv4.y; dp[4 * q4 + 2] = dv4.z; dp[4 * q4 + 3] = dv4.w;             \
            }                                                                                                           \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                          \
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(sQ + fo[QB][kk]);                                 \
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(sdO + fo[QB][kk]);                                \
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kk], s, 0, 0, 0);                                    \
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kk], dp, 0, 0, 0);                                  \
            }                                                                                                           \
            f32x16_t pv, dsv;                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) pv[r] = (LOG2 ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * LOG2E));                 \
            if (need_mask) {                      /* ONE branch per block: a test inside the score loop becomes 16 */   \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
                    const int qq = qt0 + (QB) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);                                   \
                    if (qq >= a.Lq || key >= a.Lk || (a.causal && key > qq)) pv[r] = 0.f;                               \
                }                                                                                                       \
            }                                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) dsv[r] = pv[r] * dp[r];                                      \
            bf16x8_t qtf[2][2], dotf[2][2];                                                                             \
            tr_wait<0>(tdo);                                                                                            \
            tr_wait<0>(tq);                                                                                             \
            tr_pack(dotf, tdo);                                                                                         \
            tr_pack(qtf, tq);                                                                                           \
            _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                             \
                const bf16x8_t pf = pack8(pv, 8 * x);                                                                   \
                const bf16x8_t df = pack8(dsv, 8 * x);                                                                  \
                _Pragma("unroll") for (int d = 0; d < 2; ++d) {                                                         \
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[x][d], pf, dv[d], 0, 0, 0);                    \
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[x][d], df, dk[d], 0, 0, 0);                     \
                }                                                                                                       \
            }                                                                                                           \
        }
#endif
        if (wave_live) {
            DKV_QBLOCK(0)
            if constexpr (NQB == 2) DKV_QBLOCK(1)
        }
#undef DKV_QBLOCK
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        const int nfull = tail_half ? nt - 1 : nt;
        for (int t = t0; t < nfull; ++t) dkv_tile(t, attn_ic<2>{});
        if (tail_half) dkv_tile(nt - 1, attn_ic<1>{});
    }
#endif
    if (a.dv_colsum) {        // v_proj bias gradient, fused: second workspace plane, partial row (b, key block, wave)
        // partial rows are indexed by 128-key block and wave-in-block whatever the workgroup size (the reduction's layout)
        const int nqb = (a.Lq + 127) / 128, nkb = (a.Lk + 127) / 128;
        const int kb128 = (kblk0 >> 7) + (wave >> 2);
        float* wsr = reinterpret_cast<float*>(a.cs_ws) + (int64_t)a.B * nqb * 4 * a.H * HD +
                     ((int64_t)((b * nkb + kb128) * 4 + (wave & 3)) * a.H + h) * HD;
        if (kb128 < nkb) tile_colsum_partial(dv, 1.0f, key < a.Lk, wsr, lane);
    }
    if (key < a.Lk) {
        unsigned short* DK = reinterpret_cast<unsigned short*>(a.dk) + (int64_t)b * a.dk_bs + (int64_t)key * a.dk_rs + h * HD;
        unsigned short* DV = reinterpret_cast<unsigned short*>(a.dv) + (int64_t)b * a.dv_bs + (int64_t)key * a.dv_rs + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = d * 32 + 8 * q4 + 4 * hh;
                *reinterpret_cast<uint2*>(DK + col) = make_uint2(pack_bf16x2(dk[d][4 * q4] * dk_mul, dk[d][4 * q4 + 1] * dk_mul),
                                                                 pack_bf16x2(dk[d][4 * q4 + 2] * dk_mul, dk[d][4 * q4 + 3] * dk_mul));
                *reinterpret_cast<uint2*>(DV + col) = make_uint2(pack_bf16x2(dv[d][4 * q4], dv[d][4 * q4 + 1]),
                                                                 pack_bf16x2(dv[d][4 * q4 + 2], dv[d][4 * q4 + 3]));
            }
    }
}

// ------------------------------------------------------------------------------------------------ fused backward (5 matrix passes)
// One kernel for dQ, dK and dV of a dense (non-causal) problem that fills the chip (the encoder's 1500 x 1500 self-attention and
// the SE-DiCoW enrollment cross-attention).  The two-kernel form above computes S and dP twice (7 matrix passes per score
// block); here a workgroup owns 128 keys (wave = 32 keys, as attn_bwd_dkv_kernel), walks the 64-query tiles ONCE and produces
//   per 32 x 32 block:   S = Q K^T, dP = dO V^T          (A = Q / dO row fragments from LDS, B = K / V held in registers)
//                        dV^T += dO^T P, dK^T += Q^T dS   (A = transposing reads, B = P / dS straight from the accumulators)
//   per 64-query tile:   dS^T (bf16) of the whole workgroup -> LDS image [128 keys][64 q]   (the lane owns a KEY column of dS,
//                        the dQ product contracts over keys: the transpose goes through LDS, 16 KB per tile, double-buffered)
//                        dQ^T[d][q] = K^T[d][key] . dS^T[key][q] over the workgroup's 128 keys: wave w computes the block
//                        (q block w >> 1, d block w & 1); its K^T fragments are loop-invariant (32 registers, read once)
// The dQ tile of a (batch, head) is the sum over its key blocks = over WORKGROUPS.  No atomics (bit-reproducible): the key
// blocks of a (batch, head) -- a "pair" -- run on ONE XCD (pair p -> XCD p % 8: the grid is padded to 8 x ceil(BH / 8) x nkb and
// mapped inside the kernel; XCC_ID is checked) and pass the running fp32 sum of a tile along a CHAIN through that XCD's L2:
// the workgroup of rank r for a tile reads what rank r - 1 left in the workspace, adds its share, stores it back and bumps the
// tile's counter (one u32 per (pair, tile, wave block)); rank 0 stores without reading, the last rank scales, rounds and writes
// the bf16 dQ rows (and the q-bias column sums).  Plain stores keep the lines in L2, the consumer's loads are `sc0 sc1`
// (tools/probe_handoff.hip: 1.1-1.4 us per hop, bit-exact).  ROTATION: workgroup j starts at tile j * R (R = nt / nkb) and walks
// the tiles cyclically; its rank for a tile = the number of workgroups whose walk reached that tile earlier (`rank` in the loop below) -- the workgroups of a pair sit on different
// tiles and nobody waits for a workgroup dispatched after it: the chain cannot deadlock while workgroups are dispatched in order.
// A wait that exceeds its budget (or a pair found on two XCDs) raises the status word of the workspace instead of hanging
// (dicow_attn_bwd_fused_status) and sets the abort word every spinner looks at.  The workspace tiles are in FRAGMENT order
// ([q4][lane][4 floats] per wave block): every hand-off load / store instruction moves 1 KB contiguous.
// The hand-off is PIPELINED around the next tile's A phase (profiles/r06_attn_fused.txt):
//   top of A(t+1)   : the counter of tile t+1 is polled by a 4-byte LDS-DMA (asynchronous; a register poll from inline assembly is
//                     copied by the compiler before it lands, a builtin load makes the compiler drain every DMA of the tile)
//   middle of A(t+1): vmcnt(0) -- the stores of tile t have been acknowledged -> publish tile t (counter store); look at the poll:
//                     if the predecessor is through, request the 4 KB sum by LDS-DMA into this wave's landing zone (sZ)
//   end of A(t+1)   : second-chance poll in front of barrier X; barrier X (dS^T image complete); dS fragment reads; barrier Y
//                     (the image may be overwritten by A(t+2)); DMA of the Q / dO tile t+3
//   B(t+1)          : dQ product; if the sum has not been requested yet: bounded spin on the counter (slow path, ~17 % of the tiles),
//                     then read the landing zone, add, 16 stores in fragment order.
// All tile / landing / poll DMA is issued from inline assembly (dma16x / dma4x): behind a BUILTIN LDS-DMA hipcc puts
// `s_waitcnt vmcnt(0)` in front of the next typed LDS read, which would serialise every prefetch (common.h dicow_dma16).
#define DS_BYTES (128 * 128)
#define FUSED_WS_HDR 4096                    // status words (int[0] = error bits, int[1] = abort), then per-pair XCD ids; flags follow
#define FUSED_ERR_TIMEOUT 1
#define FUSED_ERR_XCD 2
#ifndef ATTN_FUSED_MIN_WGS
#define ATTN_FUSED_MIN_WGS 1024              // (512: whisper-base B = 8 -- 768 workgroups -- measured in profiles/r06_base_kernel_table.txt)
#endif
#ifndef ATTN_FUSED_SPIN
#define ATTN_FUSED_SPIN (1 << 15)
#endif

// -delta[q] = -rowsum(dO * O) and -lse (base-2 units in q_log2 mode) for every (b, h, q): the seeds of the dP / S accumulators
// (computed in the dq kernel's prologue in the two-kernel form); 8 lanes per row.  Also clears the header and flags of the workspace.
template <bool LOG2>
__global__ void __launch_bounds__(256) attn_bwd_stats_kernel(const dicow_attn_bwd_args a, unsigned* zero_words, int n_zero) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < n_zero) zero_words[gid] = 0u;
    const int64_t row = gid >> 3, nrows = (int64_t)a.B * a.H * a.Lq;
    const int p = (int)(gid & 7);
    if (row >= nrows) return;                                  // (rows are 8-lane groups: a group leaves together)
    const int q = (int)(row % a.Lq), bh = (int)(row / a.Lq), h = bh % a.H, b = bh / a.H;
    const unsigned short* Op = reinterpret_cast<const unsigned short*>(a.o) + (int64_t)b * a.o_bs + (int64_t)q * a.o_rs + h * HD + p * 8;
    const unsigned short* dOp = reinterpret_cast<const unsigned short*>(a.d_o) + (int64_t)b * a.do_bs + (int64_t)q * a.do_rs + h * HD + p * 8;
    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(Op), df = *reinterpret_cast<const bf16x8_t*>(dOp);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(bfbits2f((unsigned short)of[e]), bfbits2f((unsigned short)df[e]), s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (p == 0) {
        a.delta[row] = -s;
        a.delta[nrows + row] = LOG2 ? -a.lse[row] * LOG2E : -a.lse[row];
    }
}

// 8 transposing reads of ONE 32-column block (bases a0 / a1 = sections 0 / 1) over two consecutive 32-row blocks at OFF:
// frag(rb, x) = (r[4 rb + 2 x], r[4 rb + 2 x + 1]),  rb = row block, x = 16-row half
template <int OFF>
__device__ __forceinline__ void tr_issue_c(tr8_t& t, unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
        "ds_read_b64_tr_b16 %1, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%11\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:%13"
        : "=&v"(t.r0), "=&v"(t.r1), "=&v"(t.r2), "=&v"(t.r3), "=&v"(t.r4), "=&v"(t.r5), "=&v"(t.r6), "=&v"(t.r7)
        : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 2048), "i"(OFF + 4096), "i"(OFF + 6144)
        : "memory");
}
__device__ __forceinline__ void tr_pack_c(bf16x8_t (&f)[2][2], const tr8_t& t) {     // f[row block][x]
    f[0][0] = __builtin_shufflevector(t.r0, t.r1, 0, 1, 2, 3, 4, 5, 6, 7);
    f[0][1] = __builtin_shufflevector(t.r2, t.r3, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][0] = __builtin_shufflevector(t.r4, t.r5, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1][1] = __builtin_shufflevector(t.r6, t.r7, 0, 1, 2, 3, 4, 5, 6, 7);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4f_t;

// poll one flag word until it reads `expect` (every lane loads the same word, L1 bypassed); false = gave up
__device__ __forceinline__ bool fused_wait_flag(__amdgpu_buffer_rsrc_t rsF, unsigned fo, unsigned expect, int* status) {
    for (int spin = 0;; ++spin) {
        unsigned f;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(fo), "s"(rsF) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)f) == expect) return true;
        if ((spin & 63) == 63) {
            const int ab = __hip_atomic_load(status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readfirstlane(ab) != 0 || spin >= ATTN_FUSED_SPIN) {
                if ((threadIdx.x & 63) == 0) { atomicOr(status, FUSED_ERR_TIMEOUT); __hip_atomic_store(status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

#ifndef ATTN_FUSED_PROFILE
#define ATTN_FUSED_PROFILE 0
#endif
#ifndef ATTN_FUSED_ABL
#define ATTN_FUSED_ABL 0      // ablation builds (results are garbage, only the time matters): 1 no hand-off, 2 no dQ product, 4 no dS^T writes, 8 no exponentials, 16 no DMA after the prologue, 32 no dV / dK MFMAs
#endif
#if ATTN_FUSED_PROFILE
// per-workgroup cycle accounting (wave 0): stamps 0..7 per tile, the deltas to the previous stamp summed over the tiles; record of 16
// int64 per workgroup behind the tiles of the workspace (tools/bench_attn.py ATTN_PROFILE_FUSED=1)
#define FPROF_DECL long long fp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int fp_slow = 0; long long fp_prev = __builtin_readcyclecounter(); const long long fp_t0 = fp_prev, fp_r0 = wall_clock64();
#define FPROF(i) { const long long n_ = __builtin_readcyclecounter(); fp_acc[i] += n_ - fp_prev; fp_prev = n_; }
#define FPROF_DEP(x) asm volatile("" : "+v"(x));
#define FPROF_SLOW ++fp_slow;
#define FPROF_END { if (tid == 0) { long long* rec = reinterpret_cast<long long*>(tiles + nflags * 4096) + (long long)blockIdx.x * 16; \
      for (int i = 0; i < 9; ++i) rec[i] = fp_acc[i]; rec[9] = __builtin_readcyclecounter() - fp_t0; rec[10] = wall_clock64() - fp_r0; rec[11] = kblk; rec[12] = fp_r0; rec[13] = fp_slow; } }
#else
#define FPROF_DECL
#define FPROF(i)
#define FPROF_DEP(x)
#define FPROF_SLOW
#define FPROF_END
#endif

// LDS reads the compiler must not see.  A typed LDS load that follows an LDS-DMA issue makes hipcc insert s_waitcnt vmcnt(0) in front of
// it (SIInsertWaitcnts cannot tell the DMA's destination from the load's address): the prefetch of the next tile, the flag poll and
// the incoming dQ sum would all be waited for at the first seed read of every block.  (attn_bwd_dkv_kernel pays exactly that: its seed
// reads wait for the next tile's DMA, ~800 cycles per tile.)
struct seeds_t { f32x4_t l0, l1, l2, l3, d0, d1, d2, d3; };
template <int OFF>      // -lse[q] / -delta[q] of one 32-query block: addr = statistics base + 16 * half
__device__ __forceinline__ void seeds_issue(seeds_t& t, unsigned addr) {
    asm volatile(
        "ds_read_b128 %0, %8 offset:%9\n\t"
        "ds_read_b128 %1, %8 offset:%10\n\t"
        "ds_read_b128 %2, %8 offset:%11\n\t"
        "ds_read_b128 %3, %8 offset:%12\n\t"
        "ds_read_b128 %4, %8 offset:%13\n\t"
        "ds_read_b128 %5, %8 offset:%14\n\t"
        "ds_read_b128 %6, %8 offset:%15\n\t"
        "ds_read_b128 %7, %8 offset:%16"
        : "=&v"(t.l0), "=&v"(t.l1), "=&v"(t.l2), "=&v"(t.l3), "=&v"(t.d0), "=&v"(t.d1), "=&v"(t.d2), "=&v"(t.d3)
        : "v"(addr), "i"(OFF), "i"(OFF + 32), "i"(OFF + 64), "i"(OFF + 96), "i"(OFF + 256), "i"(OFF + 288), "i"(OFF + 320), "i"(OFF + 352)
        : "memory");
}
template <int N>
__device__ __forceinline__ void seeds_wait(seeds_t& t) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(t.l0), "+v"(t.l1), "+v"(t.l2), "+v"(t.l3), "+v"(t.d0), "+v"(t.d1), "+v"(t.d2), "+v"(t.d3) : "i"(N) : "memory");
}
__device__ __forceinline__ f32x16_t cat16(f32x4_t a, f32x4_t b, f32x4_t c, f32x4_t d) {
    typedef __attribute__((ext_vector_type(8))) float f32x8_t;
    const f32x8_t lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}
__device__ __forceinline__ unsigned lds_read_u32_now(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
// 4 transposing reads = the fragments of ONE 16-row half of a 32-row block of a U image, both 32-column blocks: f[dblk] = (r[2 dblk], r[2 dblk + 1])
struct tr4_t { bf16x4_t r0, r1, r2, r3; };
template <int OFF>
__device__ __forceinline__ void tr_issue_h(tr4_t& t, unsigned a00, unsigned a01, unsigned a10, unsigned a11) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4 offset:%8\n\t"
        "ds_read_b64_tr_b16 %1, %5 offset:%8\n\t"
        "ds_read_b64_tr_b16 %2, %6 offset:%8\n\t"
        "ds_read_b64_tr_b16 %3, %7 offset:%8"
        : "=&v"(t.r0), "=&v"(t.r1), "=&v"(t.r2), "=&v"(t.r3) : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void tr_wait_h2(tr4_t& a, tr4_t& b) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.r0), "+v"(a.r1), "+v"(a.r2), "+v"(a.r3), "+v"(b.r0), "+v"(b.r1), "+v"(b.r2), "+v"(b.r3) : "i"(N) : "memory");
}
__device__ __forceinline__ void tr_pack_h(bf16x8_t (&f)[2], const tr4_t& t) {
    f[0] = __builtin_shufflevector(t.r0, t.r1, 0, 1, 2, 3, 4, 5, 6, 7);
    f[1] = __builtin_shufflevector(t.r2, t.r3, 0, 1, 2, 3, 4, 5, 6, 7);
}
struct land_t { f32x4_t p0, p1, p2, p3; };
__device__ __forceinline__ void land_read_now(land_t& t, unsigned addr) {      // 4 x [lane][4 floats] at 1 KB strides
    asm volatile(
        "ds_read_b128 %0, %4 offset:0\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t.p0), "=&v"(t.p1), "=&v"(t.p2), "=&v"(t.p3) : "v"(addr) : "memory");
}

template <bool LOG2>
__global__ void __launch_bounds__(256, 2) attn_bwd_fused_kernel(const dicow_attn_bwd_args a, char* fws, int nt, int nkb, int rstride) {
    constexpr int KB = 128;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE_BYTES + 2 * DS_BYTES + 2048];   // Q0 dO0 Q1 dO1 | dS^T x2 | lse/delta x2 | flag polls
    char* const sDS = smem + 4 * TILE_BYTES;               // the dS^T image [128 keys][64 q] (ONE buffer: two barriers per tile)
    char* const sZ = sDS + DS_BYTES;                         // landing zone of the incoming dQ sums: 4 KB per wave
    char* const sST = sZ + DS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    // every key block of a (batch, head) on ONE XCD, in dispatch order (dispatch id L runs on XCD L % 8): pair p lives on XCD p % 8;
    // the grid is padded to whole groups of 8 pairs and the workgroups of the pairs that do not exist leave at once
    int kblk, h, b;
    {
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
        const int bh = (idx / nkb) * 8 + xcd;
        if (bh >= a.H * a.B) return;
        kblk = idx % nkb; h = bh % a.H; b = bh / a.H;
    }
    const int kblk0 = kblk * KB;
    const unsigned short* Q = reinterpret_cast<const unsigned short*>(a.q) + (int64_t)b * a.q_bs + h * HD;
    const unsigned short* K = reinterpret_cast<const unsigned short*>(a.k) + (int64_t)b * a.k_bs + h * HD;
    const unsigned short* V = reinterpret_cast<const unsigned short*>(a.v) + (int64_t)b * a.v_bs + h * HD;
    const unsigned short* dO = reinterpret_cast<const unsigned short*>(a.d_o) + (int64_t)b * a.do_bs + h * HD;
    const float* lse = a.delta + (int64_t)a.B * a.H * a.Lq + ((int64_t)b * a.H + h) * a.Lq;     // -lse plane of the workspace
    const float* delta = a.delta + ((int64_t)b * a.H + h) * a.Lq;                                // -delta plane
    int* const status = reinterpret_cast<int*>(fws);
    const int pair = b * a.H + h;

    // this wave's 32 keys as B operands (column = key, k-slots = d)
    const int key = kblk0 + wave * 32 + (lane & 31);
    const int key_c = key < a.Lk ? key : a.Lk - 1;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8_t*>(K + (int64_t)key_c * a.k_rs + kk * 16 + hh * 8);
        vf[kk] = *reinterpret_cast<const bf16x8_t*>(V + (int64_t)key_c * a.v_rs + kk * 16 + hh * 8);
    }
    // K^T fragments of the dQ product: M = this wave's 32 d columns (d block wave & 1), k = the workgroup's 128 keys.  The key
    // block passes through the (still unused) ring slots once, as a U image; rows past Lk read as zero.
    const int dblk = wave & 1, qsub = wave >> 1;
    bf16x8_t ktf[4][2];
    {
        const tile_src_t srcKb = make_tile_src<SWZ_U>(K + (int64_t)kblk0 * a.k_rs, a.k_rs, a.Lk - kblk0, wave, lane);
        stage_tile_x(srcKb, 0, smem, wave);
        stage_tile_x(srcKb, KV_TILE, smem + TILE_BYTES, wave);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ka0 = tr_base_u(smem, lane, dblk, 0), ka1 = tr_base_u(smem, lane, dblk, 1);
        tr8_t t0, t1;
        tr_issue_c<0>(t0, ka0, ka1);
        tr_issue_c<8192>(t1, ka0, ka1);
        tr_wait<0>(t0);
        tr_wait<0>(t1);
        bf16x8_t f[2][2];
        tr_pack_c(f, t0); ktf[0][0] = f[0][0]; ktf[0][1] = f[0][1]; ktf[1][0] = f[1][0]; ktf[1][1] = f[1][1];
        tr_pack_c(f, t1); ktf[2][0] = f[0][0]; ktf[2][1] = f[0][1]; ktf[3][0] = f[1][0]; ktf[3][1] = f[1][1];
        // a wave whose keys all lie past Lk never writes its rows of the dS^T images: they stay zero
        const bool live0 = kblk0 + wave * 32 < a.Lk;
        if (!live0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<uint4*>(sDS + (wave * 32) * 128 + c * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // the ring slots are free again
        asm volatile("" ::: "memory");
    }
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }
    constexpr float dk_mul = LOG2 ? LN2 : 1.0f;

    // ---- the order of the tiles.  Key block j starts at tile s_j = floor(j nt / nkb) and wraps around (nt >= nkb; otherwise every block
    // starts at tile 0).  The visitors of a tile then arrive a whole stretch of tiles apart -- key block k = the last one with s_k <= tile
    // first (rank 0: stores without reading), then k - 1, ..., 0, nkb - 1, ..., k + 1 (rank nkb - 1: writes the bf16 rows) -- so a
    // hand-off has (nt / nkb) tile times to land before it is needed, and every workgroup is the first visitor of some tiles and the
    // last of others (balanced).  rank(tile) = (k - j) mod nkb is a fixed function of the shape: the sum order is the same in every run.
    const int R = rstride > 0 ? rstride : nt / nkb;          // tiles between the starts of consecutive key blocks (0: nt < nkb -- every block starts at tile 0)
    const bool rot = R > 0 && R * (nkb - 1) < nt;
#ifndef ATTN_FUSED_REVERSE
#define ATTN_FUSED_REVERSE 0      // (measured: the middle blocks 17 -> 14 % of their tiles on the slow path, block 0 -- which then follows the LAST block -- 69 %: no gain)
#endif
    // (the rotation runs on a VIRTUAL index vj = nkb - 1 - j: the visitor before key block j on a tile is then key block j - 1, dispatched
    // BEFORE it -- its lead is the rotation's R tiles plus the dispatch gap instead of minus it; only block 0 follows the last one)
    const int vj = (rot && ATTN_FUSED_REVERSE) ? nkb - 1 - kblk : kblk;
    const int s_j = rot ? vj * R : 0;
    auto tile_of = [&](int i) { const int x = s_j + i; return x >= nt ? x - nt : x; };

    const tile_src_t srcQ = make_tile_src<SWZ_U>(Q, a.q_rs, a.Lq, wave, lane), srcdO = make_tile_src<SWZ_U>(dO, a.do_rs, a.Lq, wave, lane);
    // -lse / -delta of a 64-query tile: one 4-byte DMA per lane (waves 0 / 2 fetch lse, waves 1 / 3 delta: the same number of VMEM
    // instructions in every wave); one descriptor over this (batch, head)'s rows of both planes of the workspace
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(delta), 0,
                                                                         (unsigned)(((int64_t)a.B * a.H * a.Lq + a.Lq) * 4), 0x00020000);
    const unsigned stat_plane = (wave & 1) ? 0u : (unsigned)((int64_t)a.B * a.H * a.Lq * 4);
    auto stage_stats_x = [&](int q0, char* dst) {
        int q = q0 + lane; q = q < a.Lq ? q : a.Lq - 1;
        dma4x<0>((unsigned)(uintptr_t)dst + (unsigned)(wave & 1) * 256u, rsS, stat_plane + (unsigned)q * 4u, 0u);
    };
    stage_tile_x(srcQ, tile_of(0) * KV_TILE, smem, wave);
    stage_tile_x(srcdO, tile_of(0) * KV_TILE, smem + TILE_BYTES, wave);
    stage_stats_x(tile_of(0) * KV_TILE, sST);
    if (nt > 1) {
        stage_tile_x(srcQ, tile_of(1) * KV_TILE, smem + 2 * TILE_BYTES, wave);
        stage_tile_x(srcdO, tile_of(1) * KV_TILE, smem + 3 * TILE_BYTES, wave);
        stage_stats_x(tile_of(1) * KV_TILE, sST + 512);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // the first tile landed, the second may fly
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int fo[4];                                             // row fragment offsets of q block 0 (q block 1: + 4096, the same swizzle)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = uswz(lane & 31, kk * 2 + hh);
    const unsigned qb00 = tr_base_u(smem, lane, 0, 0), qb01 = tr_base_u(smem, lane, 0, 1);
    const unsigned qb10 = tr_base_u(smem, lane, 1, 0), qb11 = tr_base_u(smem, lane, 1, 1);
    // dS^T image: this lane's key row; chunk c (4 queries = 8 bytes per half) at  dsw ^ (c << 4)
    const int dsrow = wave * 32 + (lane & 31);
    const int dsw = (dsrow * 128 + 8 * hh) ^ (rev3((dsrow >> 1) & 7) << 4);      // byte offset inside an image
    const unsigned dsr0 = tr_base_u(sDS, lane, qsub, 0), dsr1 = tr_base_u(sDS, lane, qsub, 1);
    const bool tail_half = ATTN_BWD_TAIL && nt * KV_TILE - a.Lq >= 32 && nt > 1;
    const bool wave_live = kblk0 + wave * 32 < a.Lk && !((ATTN_FUSED_ABL & 64) && wave == 3);      // (ablation 64: three working waves per workgroup)

    // hand-off addressing: flag word and 4 KB fragment tile of (pair, tile, this wave)
    unsigned* const flags = reinterpret_cast<unsigned*>(fws + FUSED_WS_HDR);
    const int64_t nflags = (int64_t)a.B * a.H * nt * 4;
    char* const tiles = fws + FUSED_WS_HDR + ((nflags * 4 + 4095) & ~(int64_t)4095);
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(flags, 0, (unsigned)(nflags * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(tiles, 0, (unsigned)(nflags * 4096), 0x00020000);
    bool chain_ok = true;
    if (nkb > 1 && tid == 0) {                               // every key block of a (batch, head) on ONE XCD?  (the hand-off rests on it)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int mine = (int)(xcc & 15u) + 1;
        const int seen = atomicCAS(status + 16 + pair, 0, mine);        // (the header holds <= 1000 pairs: see the launcher)
        if (seen != 0 && seen != mine) atomicOr(status, FUSED_ERR_XCD);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), as a builtin: the compiler's own scoreboard learns that the prologue's loads are done
    const unsigned voff = (unsigned)lane * 16u;
    char* const sFL = sST + 1024 + wave * 256;               // landing zone of this wave's flag polls (one dword per lane, all the same)
    const unsigned stat_addr = (unsigned)(uintptr_t)sST + 16u * (unsigned)hh, fl_addr = (unsigned)(uintptr_t)sFL;
    // ---- a hand-off (tile i of this wave's (q block, d block) of dQ), pipelined so that none of its steps waits on the one before:
    //   top of A(i)      the flag of tile i is polled (a 4-byte DMA into LDS: no register, no wait)
    //   middle of A(i)   the poll has landed; flag up -> the sum so far is requested: DMA into the wave's 4 KB of the landing zone
    //   end of B(i)      sum + this tile's product -> the workspace tile (or, for the last visitor, the bf16 rows of dQ)
    //   middle of A(i+1) the s_waitcnt vmcnt(0) there has seen the stores acknowledged: the flag of tile i goes up
    // A visitor needs the one before it to be about one and a half tile times ahead; the rotation puts nt / nkb tiles between them.
    int pub_idx = -1, pub_val = 0;                           // stores in flight: the flag to raise at the next vmcnt(0)
    // rank(t) = (k(t) - j) mod nkb, k(t) = min(nkb - 1, t / R) = the last key block that starts at or before tile t, carried incrementally
    int rk_k = vj, rk_rem = 0, cur_t = s_j;
    const unsigned z_addr = (unsigned)(uintptr_t)sZ + (unsigned)wave * 4096u;

    FPROF_DECL
    // one tile; SLOT = i & 1 at compile time: every LDS address of the body is a lane register + an immediate
    auto tile_body = [&](const int i, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const int t = cur_t;
        char* const sQ = smem + SLOT * 2 * TILE_BYTES;
        char* const sdO = sQ + TILE_BYTES;
        const int qt0 = t * KV_TILE;
        const bool need_mask = (qt0 + KV_TILE > a.Lq) || (kblk0 + KB > a.Lk);
        const bool two_blocks = !(tail_half && t == nt - 1);
        const int q32 = qt0 + qsub * 32;                      // first query of this wave's block of the tile's dQ
        const bool blk_live = q32 < a.Lq;
        const int rank = rot ? (rk_k >= vj ? rk_k - vj : rk_k + nkb - vj) : kblk;
        const bool first = rank == 0, last = rank == nkb - 1;
        const unsigned fidx = (unsigned)((pair * nt + t) * 4 + wave);
        const bool handoff = blk_live && !first && !(ATTN_FUSED_ABL & 1);
        // the poll: a 4-byte DMA into LDS nobody waits for before the middle of the A phase.  (NOT a load into a register: the compiler
        // believes an asm output is there at once and is free to copy it -- at the merge of two branches it did, before the load had
        // landed, and the late arrival then overwrote a register that had been given to a dS value.)
        if (handoff) dma4x<1>(fl_addr, rsF, 0u, fidx * 4u);      // (a DMA into LDS, not a load into a register: a compiler-issued load made hipcc put s_waitcnt vmcnt(0) -- the store acknowledgements -- at the top of every tile; an asm load's output gets copied before it lands)
        FPROF(0)
#define FUSED_QBLOCK(QB)                                                                                                \
        {                                                                                                               \
            constexpr int TOFF = SLOT * 2 * TILE_BYTES + (QB) * 4096;                                                   \
            tr4_t tdo, tq;                        /* dO^T / Q^T fragments of ONE 16-query half at a time (registers) */ \
            seeds_t sd;                                                                                                 \
            seeds_issue<SLOT * 512 + (QB) * 128>(sd, stat_addr);                                                        \
            tr_issue_h<TOFF + TILE_BYTES>(tdo, qb00, qb01, qb10, qb11);                                                 \
            tr_issue_h<TOFF>(tq, qb00, qb01, qb10, qb11);                                                               \
            seeds_wait<8>(sd);                                                                                          \
            f32x16_t s = cat16(sd.l0, sd.l1, sd.l2, sd.l3), dp = cat16(sd.d0, sd.d1, sd.d2, sd.d3);                    \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                          \
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(sQ + fo[kk] + (QB) * 4096);                       \
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(sdO + fo[kk] + (QB) * 4096);                      \
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kk], s, 0, 0, 0);                                    \
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[kk], dp, 0, 0, 0);                                  \
            }                                                                                                           \
            f32x16_t pv, dsv;                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) pv[r] = (ATTN_FUSED_ABL & 8) ? s[r] : (LOG2 ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * LOG2E)); \
            if (need_mask) {                                                                                            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
                    const int qq = qt0 + (QB) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);                                   \
                    if (qq >= a.Lq || key >= a.Lk) pv[r] = 0.f;                                                         \
                }                                                                                                       \
            }                                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) dsv[r] = (ATTN_FUSED_ABL & 8) ? dp[r] : pv[r] * dp[r];       \
            bf16x8_t qtf[2], dotf[2];                                                                                   \
            _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                             \
                tr_wait_h2<0>(tdo, tq);                                                                                 \
                tr_pack_h(dotf, tdo);                                                                                   \
                tr_pack_h(qtf, tq);                                                                                     \
                const bf16x8_t pf = pack8(pv, 8 * x);                                                                   \
                const bf16x8_t df = pack8(dsv, 8 * x);                                                                  \
                const uint4 du = __builtin_bit_cast(uint4, df);                                                         \
                if (!(ATTN_FUSED_ABL & 4)) {                                                                            \
                *reinterpret_cast<uint2*>(sDS + (dsw ^ ((4 * (QB) + 2 * x) << 4))) = make_uint2(du.x, du.y);            \
                *reinterpret_cast<uint2*>(sDS + (dsw ^ ((4 * (QB) + 2 * x + 1) << 4))) = make_uint2(du.z, du.w); }      \
                _Pragma("unroll") for (int d = 0; d < 2; ++d) {                                                         \
                    if (ATTN_FUSED_ABL & 32) { dv[d][0] += bfbits2f((unsigned short)(dotf[d][0] ^ pf[0])); dk[d][0] += bfbits2f((unsigned short)(qtf[d][0] ^ df[0])); } else { \
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[d], pf, dv[d], 0, 0, 0);                       \
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[d], df, dk[d], 0, 0, 0); }                      \
                }                                                                                                       \
                if (x == 0) {                     /* the second half's fragments: requested once the first half's MFMAs have read theirs */ \
                    asm volatile("" : "+v"(dv[0]), "+v"(dk[0]), "+v"(dv[1]), "+v"(dk[1]));                               \
                    tr_issue_h<TOFF + TILE_BYTES + 2048>(tdo, qb00, qb01, qb10, qb11);                                  \
                    tr_issue_h<TOFF + 2048>(tq, qb00, qb01, qb10, qb11);                                                \
                }                                                                                                       \
            }                                                                                                           \
        }
        if (wave_live) FUSED_QBLOCK(0)
        FPROF(1)
        // ---- middle of the A phase: everything requested so far has long landed (the next tile's DMA dates from the B phase before),
        // so this wait is all but free -- and it has seen the stores of the tile before acknowledged: its flag goes up.  The poll of
        // this tile's flag is in: flag up -> the sum so far is requested now and has the rest of the tile to land.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FPROF(2)
        if (pub_idx >= 0) {
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)pub_val, rsF, (unsigned)pub_idx * 4u, 0, 0);
            pub_idx = -1;
        }
        bool have = false, again = false;
        if (handoff) {
            have = chain_ok && (unsigned)__builtin_amdgcn_readfirstlane((int)lds_read_u32_now(fl_addr)) == (unsigned)rank;
            if (have) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) dma16x<1>(z_addr + q4 * 1024u, rsW, voff + q4 * 1024u, fidx * 4096u);
            } else if (chain_ok) {                            // not yet: a second poll, looked at in front of the barrier
                again = true;
                dma4x<1>(fl_addr, rsF, 0u, fidx * 4u);
            }
        }
        if (wave_live && two_blocks) FUSED_QBLOCK(1)
#undef FUSED_QBLOCK
        FPROF(3)
        // ---- barrier X: dS^T complete, this slot free for the tile after next, the next tile landed (its DMA is older than the
        // four loads of the sum, which may stay in flight: loads complete in order)
        if (have) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (again && (unsigned)__builtin_amdgcn_readfirstlane((int)lds_read_u32_now(fl_addr)) == (unsigned)rank) {     // second chance: the sum travels under the B phase
            have = true;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) dma16x<1>(z_addr + q4 * 1024u, rsW, voff + q4 * 1024u, fidx * 4096u);
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        FPROF(4)
        // ---- B phase: the next-but-one tile's DMA, the dQ product, the hand-off
        const bool more = i + 2 < nt && !(ATTN_FUSED_ABL & 16);
        const bool prod = blk_live && !(ATTN_FUSED_ABL & 2);
        // this wave's columns of dS^T: all 16 fragment reads up front, then barrier Y -- once every wave HAS them the image may be
        // rewritten, and the rest of the B phase (the MFMAs, the wait for the sum, the stores) runs un-synchronised, beside the
        // other waves' next A phase instead of in a burst with their stores and DMA
        tr8_t u0, u1;
        if (prod) {
            tr_issue_c<0>(u0, dsr0, dsr1);
            tr_issue_c<8192>(u1, dsr0, dsr1);
        }
        if (more) {
            int t2 = t + 2; t2 = t2 >= nt ? t2 - nt : t2; t2 *= KV_TILE;
            stage_tile_x(srcQ, t2, sQ, wave);
            stage_tile_x(srcdO, t2, sdO, wave);
            stage_stats_x(t2, sST + SLOT * 512);
        }
        if (prod) { tr_wait<0>(u0); tr_wait<0>(u1); }
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ---- barrier Y
        asm volatile("" ::: "memory");
        FPROF(5)
        if (prod) {
            f32x16_t dq;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = 0.f;
            bf16x8_t f[2][2];
            tr_pack_c(f, u0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0][0], f[0][0], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0][1], f[0][1], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1][0], f[1][0], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1][1], f[1][1], dq, 0, 0, 0);
            tr_pack_c(f, u1);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[2][0], f[0][0], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[2][1], f[0][1], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[3][0], f[1][0], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[3][1], f[1][1], dq, 0, 0, 0);
            FPROF_DEP(dq[0])
            FPROF(6)
            if (handoff) {
                if (!have) {                                  // the slow path: the visitor before had not published by the middle of the A phase
                    FPROF_SLOW
                    if (chain_ok) chain_ok = fused_wait_flag(rsF, fidx * 4u, (unsigned)rank, status);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) dma16x<1>(z_addr + q4 * 1024u, rsW, voff + q4 * 1024u, fidx * 4096u);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (more) {
                    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // only the 5 DMA instructions of the next-but-one tile are younger than the sum
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                FPROF(7)
                land_t pr;
                land_read_now(pr, z_addr + voff);
#pragma unroll
                for (int e = 0; e < 4; ++e) { dq[e] += pr.p0[e]; dq[4 + e] += pr.p1[e]; dq[8 + e] += pr.p2[e]; dq[12 + e] += pr.p3[e]; }
            }
            FPROF_DEP(dq[0])
            if (ATTN_FUSED_ABL & 1) {
            } else if (!last) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const u32x4f_t sv = {__float_as_uint(dq[4 * q4]), __float_as_uint(dq[4 * q4 + 1]), __float_as_uint(dq[4 * q4 + 2]), __float_as_uint(dq[4 * q4 + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(sv, rsW, voff + q4 * 1024u, fidx * 4096u, 0);
                }
                pub_idx = (int)fidx; pub_val = rank + 1;      // (raised once the stores have been acknowledged: the next vmcnt(0))
            } else {
                const int qrow = q32 + (lane & 31);
                if (a.dq_colsum) {                            // q_proj bias gradient: partial row (b, 32-query block), columns h*64 + dblk*32 ..
                    float* wsr = reinterpret_cast<float*>(a.cs_ws) + ((int64_t)(b * 2 * nt + (q32 >> 5)) * a.H + h) * HD + dblk * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = qrow < a.Lq ? bf2f(f2bf(dq[r] * a.dq_scale)) : 0.f;
                        const float tsum = half_wave_sum_dpp(v);
                        if ((lane & 31) == 31) wsr[8 * (r >> 2) + 4 * hh + (r & 3)] = tsum;
                    }
                }
                if (qrow < a.Lq) {
                    unsigned short* DQ = reinterpret_cast<unsigned short*>(a.dq) + (int64_t)b * a.dq_bs + (int64_t)qrow * a.dq_rs + h * HD + dblk * 32;
                    const float sc = a.dq_scale;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        *reinterpret_cast<uint2*>(DQ + 8 * q4 + 4 * hh) =
                            make_uint2(pack_bf16x2(dq[4 * q4] * sc, dq[4 * q4 + 1] * sc), pack_bf16x2(dq[4 * q4 + 2] * sc, dq[4 * q4 + 3] * sc));
                }
            }
        } else if (last && a.dq_colsum) {                     // an all-padding 32-query block of the last tile: its partial row is zero
            float* wsr = reinterpret_cast<float*>(a.cs_ws) + ((int64_t)(b * 2 * nt + (q32 >> 5)) * a.H + h) * HD + dblk * 32;
            if (lane < 32) wsr[lane] = 0.f;
        }
        FPROF(8)
        // the next tile and its rank
        ++cur_t; ++rk_rem;
        if (rk_rem == R) { rk_rem = 0; rk_k = rk_k + 1 < nkb ? rk_k + 1 : rk_k; }
        if (cur_t == nt) { cur_t = 0; rk_k = 0; rk_rem = 0; }
    };
    {
        int i = 0;
        for (; i + 1 < nt; i += 2) { tile_body(i, attn_ic<0>{}); tile_body(i + 1, attn_ic<1>{}); }
        if (i < nt) tile_body(i, attn_ic<0>{});
    }
    if (pub_idx >= 0) {                                      // the last tile's flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)pub_val, rsF, (unsigned)pub_idx * 4u, 0, 0);
    }
    FPROF_END
    if (a.dv_colsum) {        // v_proj bias gradient, fused: second workspace plane, partial row (b, key block, wave)
        const int nqb = (a.Lq + 127) / 128, nkb128 = (a.Lk + 127) / 128;
        float* wsr = reinterpret_cast<float*>(a.cs_ws) + (int64_t)a.B * nqb * 4 * a.H * HD +
                     ((int64_t)((b * nkb128 + kblk) * 4 + wave) * a.H + h) * HD;
        tile_colsum_partial(dv, 1.0f, key < a.Lk, wsr, lane);
    }
    if (key < a.Lk) {
        unsigned short* DK = reinterpret_cast<unsigned short*>(a.dk) + (int64_t)b * a.dk_bs + (int64_t)key * a.dk_rs + h * HD;
        unsigned short* DV = reinterpret_cast<unsigned short*>(a.dv) + (int64_t)b * a.dv_bs + (int64_t)key * a.dv_rs + h * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = d * 32 + 8 * q4 + 4 * hh;
                *reinterpret_cast<uint2*>(DK + col) = make_uint2(pack_bf16x2(dk[d][4 * q4] * dk_mul, dk[d][4 * q4 + 1] * dk_mul),
                                                                 pack_bf16x2(dk[d][4 * q4 + 2] * dk_mul, dk[d][4 * q4 + 3] * dk_mul));
                *reinterpret_cast<uint2*>(DV + col) = make_uint2(pack_bf16x2(dv[d][4 * q4], dv[d][4 * q4 + 1]),
                                                                 pack_bf16x2(dv[d][4 * q4 + 2], dv[d][4 * q4 + 3]));
            }
    }
}

extern "C" int64_t dicow_attn_bwd_colsum_ws_bytes(int B, int H, int Lq, int Lk) {
    return (int64_t)B * (dicow_cdiv(Lq, 128) + dicow_cdiv(Lk, 128)) * 4 * H * HD * 4;
}

extern "C" int64_t dicow_attn_bwd_fused_ws_bytes(int B, int H, int Lq, int Lk) {
    (void)Lk;
    const int64_t nflags = (int64_t)B * H * dicow_cdiv(Lq, KV_TILE) * 4;
    return FUSED_WS_HDR + ((nflags * 4 + 4095) & ~(int64_t)4095) + nflags * 4096
#if ATTN_FUSED_PROFILE
           + (int64_t)8 * dicow_cdiv((int64_t)B * H, 8) * dicow_cdiv(Lk, 128) * 128
#endif
        ;
}
// error bits a fused launch left in its workspace (0 = fine; read AFTER the stream has been synchronised): 1 = a hand-off wait
// ran out of its budget, 2 = the key blocks of one (batch, head) were not all on one XCD.  Either way dq is not to be trusted.
extern "C" int dicow_attn_bwd_fused_status(const void* fused_ws) {
    int st[2] = {0, 0};
    if (!fused_ws) return 0;
    if (hipMemcpy(st, fused_ws, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return st[0];
}
static bool attn_bwd_use_fused(const dicow_attn_bwd_args* a) {
    if (!a->fused_ws || a->causal) return false;
    const int64_t nkb = dicow_cdiv(a->Lk, 128), wgs = nkb * a->H * a->B;
    if ((int64_t)a->B * a->H > 1000) return false;                                       // (the header's per-pair words)
    if (a->fused_mode != 1 && (wgs < ATTN_FUSED_MIN_WGS || a->Lq < 256)) return false;  // below two rounds of the chip the two-kernel form stays
    const int64_t nflags = (int64_t)a->B * a->H * dicow_cdiv(a->Lq, KV_TILE) * 4;
    if (nflags * 4096 >= ((int64_t)1 << 32)) return false;
    return a->fused_ws_bytes >= dicow_attn_bwd_fused_ws_bytes(a->B, a->H, a->Lq, a->Lk);
}

extern "C" int dicow_attn_bwd(const dicow_attn_bwd_args* a, void* stream) {
    DICOW_REQUIRE(!(a && (a->dq_colsum || a->dv_colsum)) ||
                  (a->cs_ws && a->cs_ws_bytes >= dicow_attn_bwd_colsum_ws_bytes(a->B, a->H, a->Lq, a->Lk)),
                  "attn_bwd: fused column sums need cs_ws of dicow_attn_bwd_colsum_ws_bytes() bytes");
    DICOW_REQUIRE(a && a->q && a->k && a->v && a->o && a->d_o && a->lse && a->delta && a->dq && a->dk && a->dv,
                  "attn_bwd: null operand");
    DICOW_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0, "attn_bwd: empty problem");
    const int64_t rs[] = {a->q_rs, a->k_rs, a->v_rs, a->o_rs, a->do_rs, a->dq_rs, a->dk_rs, a->dv_rs,
                          a->q_bs, a->k_bs, a->v_bs, a->o_bs, a->do_bs, a->dq_bs, a->dk_bs, a->dv_bs};
    for (int i = 0; i < 16; ++i) DICOW_REQUIRE(rs[i] % 4 == 0, "attn_bwd: strides must keep 8-byte alignment");
    DICOW_REQUIRE(a->q_rs % 8 == 0 && a->k_rs % 8 == 0 && a->v_rs % 8 == 0 && a->do_rs % 8 == 0, "attn_bwd: q/k/v/dO row strides %% 8");
    hipStream_t st = (hipStream_t)stream;
    DICOW_REQUIRE(!a->fused_ws || (((uintptr_t)a->fused_ws & 4095) == 0), "attn_bwd: fused_ws must be 4096-byte aligned");
    if (attn_bwd_use_fused(a)) {
        const int nt = dicow_cdiv(a->Lq, KV_TILE), nkb = dicow_cdiv(a->Lk, 128);
        const int64_t nflags = (int64_t)a->B * a->H * nt * 4;
        const int n_zero = (int)((FUSED_WS_HDR + nflags * 4) / 4);
        const int64_t nthr = (int64_t)a->B * a->H * a->Lq * 8;
        const unsigned sblocks = (unsigned)dicow_cdiv(nthr > n_zero ? nthr : n_zero, 256);
        char* fws = reinterpret_cast<char*>(a->fused_ws);
        if (a->q_log2) hipLaunchKernelGGL(attn_bwd_stats_kernel<true>, dim3(sblocks), dim3(256), 0, st, *a, reinterpret_cast<unsigned*>(fws), n_zero);
        else hipLaunchKernelGGL(attn_bwd_stats_kernel<false>, dim3(sblocks), dim3(256), 0, st, *a, reinterpret_cast<unsigned*>(fws), n_zero);
        DICOW_CHECK_LAUNCH("attn_bwd_stats");
        const dim3 grid((unsigned)(8 * dicow_cdiv((int64_t)a->H * a->B, 8) * nkb));      // whole groups of 8 (batch, head) pairs: one pair per XCD
#ifdef DICOW_ABLATIONS
        static const int rstride = [] { const char* e = getenv("DICOW_ATTN_FUSED_RSTRIDE"); return e ? atoi(e) : 0; }();      // (diagnostic builds: the tile stride of the rotation)
#else
        const int rstride = 0;                                // nt / nkb
#endif
        if (a->q_log2) hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, grid, dim3(256), 0, st, *a, fws, nt, nkb, rstride);
        else hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, grid, dim3(256), 0, st, *a, fws, nt, nkb, rstride);
        DICOW_CHECK_LAUNCH("attn_bwd_fused");
        if (a->dq_colsum || a->dv_colsum) {
            const int64_t D = (int64_t)a->H * HD;
            const int rq = a->B * 2 * nt, rk = a->B * dicow_cdiv(a->Lk, 128) * 4;
            const float* ws = reinterpret_cast<const float*>(a->cs_ws);
            const int64_t dv_plane = (int64_t)a->B * dicow_cdiv(a->Lq, 128) * 4 * D;
            int rc = DICOW_OK;
            if (a->dq_colsum && a->dv_colsum && rq == rk) {
                float* outs[2] = {a->dq_colsum, a->dv_colsum};
                return dicow_launch_reduce_multi(ws, rq, D, dv_plane, outs, 2, D, st);
            }
            if (a->dq_colsum) rc = dicow_launch_reduce_parts(ws, rq, D, a->dq_colsum, D, st);
            if (rc == DICOW_OK && a->dv_colsum) rc = dicow_launch_reduce_parts(ws + dv_plane, rk, D, a->dv_colsum, D, st);
            return rc;
        }
        return DICOW_OK;
    }
    if (a->q_log2) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, dim3(dicow_cdiv(a->Lq, 128) * a->H * a->B), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, dim3(dicow_cdiv(a->Lq, 128) * a->H * a->B), dim3(256), 0, st, *a);
    DICOW_CHECK_LAUNCH("attn_bwd_dq");
    if (ATTN_BWD_DKV_NW8 && (int64_t)dicow_cdiv(a->Lk, 256) * a->H * a->B >= 1024) {
        if (a->q_log2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<8, true>), dim3(dicow_cdiv(a->Lk, 256) * a->H * a->B), dim3(512), 0, st, *a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<8, false>), dim3(dicow_cdiv(a->Lk, 256) * a->H * a->B), dim3(512), 0, st, *a);
    } else {
        if (a->q_log2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<4, true>), dim3(dicow_cdiv(a->Lk, 128) * a->H * a->B), dim3(256), 0, st, *a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<4, false>), dim3(dicow_cdiv(a->Lk, 128) * a->H * a->B), dim3(256), 0, st, *a);
    }
    DICOW_CHECK_LAUNCH("attn_bwd_dkv");
    if (a->dq_colsum || a->dv_colsum) {               // add the per-wave partial rows up (no atomics)
        const int64_t D = (int64_t)a->H * HD;
        const int rq = a->B * dicow_cdiv(a->Lq, 128) * 4, rk = a->B * dicow_cdiv(a->Lk, 128) * 4;
        const float* ws = reinterpret_cast<const float*>(a->cs_ws);
        int rc = DICOW_OK;
        if (a->dq_colsum && a->dv_colsum && rq == rk) {       // self-attention: both column sums in ONE reduce launch
            float* outs[2] = {a->dq_colsum, a->dv_colsum};
            return dicow_launch_reduce_multi(ws, rq, D, (int64_t)rq * D, outs, 2, D, st);
        }
        if (a->dq_colsum) rc = dicow_launch_reduce_parts(ws, rq, D, a->dq_colsum, D, st);
        if (rc == DICOW_OK && a->dv_colsum) rc = dicow_launch_reduce_parts(ws + (int64_t)rq * D, rk, D, a->dv_colsum, D, st);
        return rc;
    }
    return DICOW_OK;
}
