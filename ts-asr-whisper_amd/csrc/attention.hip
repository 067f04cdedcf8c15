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
    if (t0 < nt) {
        stage_tile<NI>(srcQ, t0 * KV_TILE, smem, wave);
        stage_tile<NI>(srcdO, t0 * KV_TILE, smem + TILE_BYTES, wave);
        stage_stats64(lse, delta, t0 * KV_TILE, a.Lq, smem + 4 * TILE_BYTES, wave, lane);
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
            stage_tile<NI>(srcQ, (t + 1) * KV_TILE, nQ, wave);
            stage_tile<NI>(srcdO, (t + 1) * KV_TILE, nQ + TILE_BYTES, wave);
            stage_stats64(lse, delta, (t + 1) * KV_TILE, a.Lq, smem + 4 * TILE_BYTES + ((t - t0 + 1) & 1) * 512, wave, lane);
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

extern "C" int64_t dicow_attn_bwd_colsum_ws_bytes(int B, int H, int Lq, int Lk) {
    return (int64_t)B * (dicow_cdiv(Lq, 128) + dicow_cdiv(Lk, 128)) * 4 * H * HD * 4;
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
