// NT GEMM with a DEFERRED bias + GELU epilogue (round 4): C = gelu(bf16(A B^T + bias)) [, aux = gelu'(...)], persistent LDS ring.
//
// gemm_ntr_kernel (gemm_ntr.inc) runs ONE wave per SIMD, so a tile's epilogue runs with the matrix pipe idle: 9.5 us (inference
// fc1) / 12.9 us (training fc1: + the saved derivative) per 27-us k-loop, almost all of it VALU issue (the exact-erf GELU is ~12
// instructions + 2 transcendentals per element).  Here the finished 192 x 320 tile is HELD as packed bf16 pre-activations -- what
// the AMP Linear output is anyway -- in 120 registers (the k-loop needs ~110 + 240 accumulators), and its GELU + stores are
// threaded between the MFMAs of the NEXT tile's k-loop: one 32 x 32 block (4 quads per lane) per k-step in steps 1..15, a quad per
// MFMA slice; the stores go out in the first half of the following step (they must be older than that step's A-stream requests:
// the loop's counted wait, vmcnt(8), may only leave those in flight).  The held registers are indexed by the step (uniform,
// s_set_gpr_idx), so the loop is not unrolled over blocks.  The last tile of a workgroup is flushed after the loop.
// Same ring, request order, counted waits and tile walk as gemm_ntr_kernel<., 3, 5>; scalar (not packed) fp32 arithmetic in the
// loop (packed VALU beside MFMAs is dearer than two scalar ones: MI355X_MICROARCH.md), hence this file is compiled with
// -fno-slp-vectorize (tools/build_ntd.sh -> tools/libv_ntd.so; the switch is DICOW_NT_DEFER=1|2|3 at run time).  Needs K >= 16
// k-steps of 64, N % 320 == 0, 32-bit byte offsets, one batch.
// RESULT (round 4, profiles/r04_ntd_deferred_epilogue.txt): bit-identical to gemm_ntr_kernel on every shape tried, and 1.5-2 x SLOWER:
// the GELU of a tile is ~3500 VALU / transcendental issues against 1200 MFMA gaps that already carry ~2.4 issues each -- beyond the
// ~5 per gap a lone wave can hide, so the deferred steps become issue-bound.  Not built into the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/dicow_hip.h"
#include "../common.h"

#define BK 64
typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned u32x32_t __attribute__((ext_vector_type(32)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// (same tile walk as gemm.hip: grouped + XCD-aware, each XCD a contiguous chunk of the grouped order)
__device__ __forceinline__ void ntd_tile_coords(int ntm, int ntn, int bid, int& tm, int& tn) {
    const int nwg = ntm * ntn;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int GM = 8;
    const int per_group = GM * ntn;
    const int group = id / per_group, rem = id - group * per_group;
    const int first_m = group * GM;
    const int gsize = (ntm - first_m) < GM ? (ntm - first_m) : GM;
    tm = first_m + rem % gsize;
    tn = rem / gsize;
}
// LDS image rows of 128 B (64 k), 16-B chunk c of row r at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ bf16x8_t ntd_frag(const char* s, int row, int c) {
    return *reinterpret_cast<const bf16x8_t*>(s + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}

#define NTD_SLOT 32768
#define NTD_LDS (5 * NTD_SLOT)
#define NTD_C16_NT 2            // cache policy of the bf16 output stores (nt: touched once), as gemm_ntr.inc
#define NTD_X_NT 0
#define NTD_C32_NT 0
#define NTD_RES_NT 2
#ifndef NTD_DROP_DEFERRED
#define NTD_DROP_DEFERRED 0     // diagnostic: every in-loop load / store of the deferred epilogue goes out of range (dropped): what do THEY cost?
#endif
#ifndef NTD_DIAG
#define NTD_DIAG 0                // diagnostic builds (results WRONG): 1 = the deferred steps run as plain steps, 2 = + no hold / immediate blocks, 3 = + no flush
#endif
#ifndef NTD_VALU_MASK
#define NTD_VALU_MASK 0x402     // sched_group_barrier classes of the deferred arithmetic: VALU | transcendental
#endif
#ifndef NTD_VALU_PER_GAP
#define NTD_VALU_PER_GAP 4     // ... and how many of them follow each MFMA of a deferred slice
#endif

template <int FLAGS>
__global__ void __launch_bounds__(256) gemm_ntd_kernel(const dicow_gemm_args a) {
    constexpr int NJ = 3, NI = 5, BMT = 192, BNT = 320, WMR = 96, WNC = 160;
    constexpr bool GELU_K = (FLAGS & DICOW_EPI_GELU) != 0;          // heavy deferred work (GELU per element); else "light": stores [+ residual add]
    constexpr bool DAUX = (FLAGS & DICOW_EPI_GELU_DAUX) != 0;
    constexpr bool BIAS_K = (FLAGS & DICOW_EPI_BIAS) != 0, SCALE_K = (FLAGS & DICOW_EPI_SCALE_N) != 0;
    constexpr bool RES_K = (FLAGS & DICOW_EPI_RESIDUAL) != 0;      // C (fp32) = bf16(acc + bias) + residual
    constexpr int ESZ = (FLAGS & DICOW_EPI_OUT_F32) ? 4 : 2;
    static_assert(!RES_K || ESZ == 4, "residual epilogue: fp32 output");
    static_assert(RES_K || ESZ == 2, "bf16 output unless residual");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (a.M + BMT - 1) / BMT, ntn = a.N / BNT;
    const int total = ntm * ntn;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = a.K / BK;                          // >= 16 (host-checked)

    int v = blockIdx.x;
    int tm, tn;
    ntd_tile_coords(ntm, ntn, v, tm, tn);
    int m0 = tm * BMT, n0 = tn * BNT;
    unsigned offA[8], offB[8];
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int ka = 0, kb = 0;
    const bool h0_is_b = wave == 3;                   // half 0 = 192 A rows + 64 B rows (column block 0 of both wave columns)
#define NTD_OFFS_ROWS(OFF, ROWEXPR, LIM, LD)                                                                  \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                          \
        const int r_ = (wave * 8 + e) * 8 + (lane >> 3), c_ = (lane & 7) ^ ((r_ >> 1) & 7);                  \
        int g_ = (ROWEXPR); g_ = g_ < (LIM) ? g_ : (LIM) - 1;                                                \
        OFF[e] = (unsigned)(((int64_t)g_ * (LD) + c_ * 8) * 2);                                              \
    }
#define NTD_OFFS_H0(M0_, N0_)                                                                                \
    if (h0_is_b) { NTD_OFFS_ROWS(offA, (N0_) + (r_ < 224 ? r_ - 192 : r_ - 224 + 160), a.N, a.ldb) }        \
    else { NTD_OFFS_ROWS(offA, (M0_) + r_, a.M, a.lda) }
#define NTD_OFFS_H1(N0_) { NTD_OFFS_ROWS(offB, (N0_) + (r_ < 128 ? r_ + 32 : r_ - 128 + 192), a.N, a.ldb) }
#define NTD_RS(P, NREC) __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P), 0, (NREC), 0x00020000)
#define NTD_DMA_A(E, SLOT) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(smem + (SLOT) * NTD_SLOT + (wave * 8 + (E)) * 1024), 16, offA[E], ka * 2, 0, 0);
#define NTD_DMA_B(E, SLOT) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(smem + (SLOT) * NTD_SLOT + (wave * 8 + (E)) * 1024), 16, offB[E], kb * 2, 0, 0);
    const unsigned short* Ab = reinterpret_cast<const unsigned short*>(a.A);
    const unsigned short* Bb = reinterpret_cast<const unsigned short*>(a.B);
    rsA = h0_is_b ? NTD_RS(Bb, 0xffffffffu) : NTD_RS(Ab, 0xffffffffu);
    rsB = NTD_RS(Bb, 0xffffffffu);
    NTD_OFFS_H0(m0, n0)
    NTD_OFFS_H1(n0)
#pragma unroll
    for (int e = 0; e < 8; ++e) NTD_DMA_A(e, 0)
#pragma unroll
    for (int e = 0; e < 8; ++e) NTD_DMA_B(e, 1)
    ka = BK; kb = BK;
#pragma unroll
    for (int e = 0; e < 8; ++e) NTD_DMA_A(e, 2)
    ka = 2 * BK;
    int sa = 0;
    int nv = v, nm0 = m0, nn0 = n0;
    bool have_next = false;
#define NTD_SWITCH_A()                                                                                       \
    {                                                                                                        \
        nv = v + gridDim.x;                                                                                  \
        have_next = nv < total;                                                                              \
        if (have_next) {                                                                                     \
            int tm_, tn_;                                                                                    \
            ntd_tile_coords(ntm, ntn, nv, tm_, tn_);                                                         \
            nm0 = tm_ * BMT; nn0 = tn_ * BNT;                                                                \
            NTD_OFFS_H0(nm0, nn0)                                                                            \
            rsA = h0_is_b ? NTD_RS(Bb, 0xffffffffu) : NTD_RS(Ab, 0xffffffffu);                               \
        } else {                                                                                             \
            rsA = NTD_RS(Ab, 0u);                                                                            \
        }                                                                                                    \
        ka = 0;                                                                                              \
    }
#define NTD_SWITCH_B()                                                                                       \
    {                                                                                                        \
        if (have_next) { NTD_OFFS_H1(nn0) rsB = NTD_RS(Bb, 0xffffffffu); }                                   \
        else { rsB = NTD_RS(Bb, 0u); }                                                                       \
        kb = 0;                                                                                              \
    }

    // ---- the held tile and what the deferred epilogue needs of it
    // blocks (n block i < 4, m block j) are held: vector H<j>, dwords 8 i + 2 g + {0, 1} (quad g = columns 8 g + 4 hh .. + 3 of the block);
    // the three blocks of n block 4 are finished at once when the tile ends (120 held registers do not fit beside the k-loop's)
    u32x32_t H0, H1, H2;
#pragma unroll
    for (int e = 0; e < 32; ++e) { H0[e] = 0u; H1[e] = 0u; H2[e] = 0u; }
    bool p_valid = false;                             // a held tile exists
    const int ml = lane & 31, hh = lane >> 5;
    const unsigned OOB = 0x80000000u;
#if NTD_DIAG == 4 || NTD_DIAG == 5
#define NTD_ROK(X) false      /* diagnostic: the stores of the immediate blocks and of the flush are dropped (5: deferred steps live) */
#else
#define NTD_ROK(X) (X)
#endif
    // (32-bit scalar arithmetic: the host checks that every byte offset fits; a 64-bit product would be computed on the VALU and the
    // descriptor word would live in a VGPR -- every store then becomes a readfirstlane waterfall loop)
    const unsigned nrecC = (unsigned)__builtin_amdgcn_readfirstlane(((a.M - 1) * (int)a.ldc + a.N) * ESZ);
    const unsigned nrecR = RES_K ? (unsigned)__builtin_amdgcn_readfirstlane(((a.M - 1) * (int)a.ldr + a.N) * 4) : 0u;
    const unsigned nrecX = DAUX ? (unsigned)__builtin_amdgcn_readfirstlane(((a.M - 1) * (int)a.ldaux + a.N) * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.C), 0, nrecC, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.aux), 0, nrecX, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bias), 0, BIAS_K ? (unsigned)(a.N * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual), 0, nrecR, 0x00020000);
    int prow = 0;                                     // row of this lane in block j = 0 of the held tile
    unsigned pvoC = OOB, pvoX = OOB, pvoR = OOB;      // lane byte offset of element (row ml, column 4 hh) of the HELD tile's wave quadrant
    // address state of the block a deferred step works on (cur) and of the one before it (prv): scalar byte offset of the block in
    // the wave quadrant + the lane offset (out of range when there is nothing to do)
    unsigned cur_voC = OOB, cur_voX = OOB, cur_voR = OOB, prv_voC = OOB, prv_voX = OOB;
    int cur_soC = 0, cur_soX = 0, cur_soR = 0, prv_soC = 0, prv_soX = 0;
    // GELU: outputs carried from the step that computed them to the first half of the next one.  Residual: the block's held
    // dwords and its residual quads (requested one step ahead)
    u32x2_t cst_o[4], cst_d[4];
    unsigned hcar[8];
    u32x4_t rres[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { cst_o[g] = u32x2_t{0u, 0u}; cst_d[g] = u32x2_t{0u, 0u}; rres[g] = u32x4_t{0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int e = 0; e < 8; ++e) hcar[e] = 0u;

    // quad G of the block at dword DIX_ of held vector HV: GELU of the four bf16 pre-activations -> packed outputs
#define NTD_QUAD(HV, DIX_, G, OUT_O, OUT_D)                                                                  \
    {                                                                                                        \
        const unsigned w0_ = (HV)[(DIX_) + 2 * (G)], w1_ = (HV)[(DIX_) + 2 * (G) + 1];                       \
        const float x0_ = __uint_as_float(w0_ << 16), x1_ = __uint_as_float(w0_ & 0xffff0000u);             \
        const float x2_ = __uint_as_float(w1_ << 16), x3_ = __uint_as_float(w1_ & 0xffff0000u);             \
        float c0_, c1_, c2_, c3_, p0_, p1_, p2_, p3_;                                                        \
        gelu_cdf_pdf(x0_, c0_, p0_); gelu_cdf_pdf(x1_, c1_, p1_);                                  \
        gelu_cdf_pdf(x2_, c2_, p2_); gelu_cdf_pdf(x3_, c3_, p3_);                                  \
        (OUT_O) = u32x2_t{pack_bf16x2(x0_ * c0_, x1_ * c1_), pack_bf16x2(x2_ * c2_, x3_ * c3_)};            \
        if (DAUX) (OUT_D) = u32x2_t{pack_bf16x2(fmaf(x0_, p0_, c0_), fmaf(x1_, p1_, c1_)),                   \
                                    pack_bf16x2(fmaf(x2_, p2_, c2_), fmaf(x3_, p3_, c3_))};                  \
    }
#define NTD_STORE(G)                                                                                         \
    {                                                                                                        \
        __builtin_amdgcn_raw_buffer_store_b64(cst_o[G], rsC, prv_voC, prv_soC + (G) * 16, NTD_C16_NT);       \
        if (DAUX) __builtin_amdgcn_raw_buffer_store_b64(cst_d[G], rsX, prv_voX, prv_soX + (G) * 16, NTD_X_NT); \
    }
    // light epilogues.  bf16 output: quad G of block (HV, DIX_) straight from the held registers to its place (cur)
#define NTD_LSTORE(HV, DIX_, G)                                                                              \
    {                                                                                                        \
        const u32x2_t o_ = {(HV)[(DIX_) + 2 * (G)], (HV)[(DIX_) + 2 * (G) + 1]};                             \
        __builtin_amdgcn_raw_buffer_store_b64(o_, rsC, cur_voC, cur_soC + (G) * 16, NTD_C16_NT);             \
    }
    // residual: quad G of the PREVIOUS block = its held bf16 values + the residual quad requested one step ago -> fp32 (prv)
#define NTD_RSTORE(G)                                                                                        \
    {                                                                                                        \
        const unsigned w0_ = hcar[2 * (G)], w1_ = hcar[2 * (G) + 1];                                         \
        const u32x4_t o_ = {__float_as_uint(__uint_as_float(w0_ << 16) + __uint_as_float(rres[G][0])),       \
                            __float_as_uint(__uint_as_float(w0_ & 0xffff0000u) + __uint_as_float(rres[G][1])), \
                            __float_as_uint(__uint_as_float(w1_ << 16) + __uint_as_float(rres[G][2])),       \
                            __float_as_uint(__uint_as_float(w1_ & 0xffff0000u) + __uint_as_float(rres[G][3]))}; \
        __builtin_amdgcn_raw_buffer_store_b128(o_, rsC, prv_voC, prv_soC + (G) * 32, NTD_C32_NT);            \
    }
    // ... and this step's block: its held dwords and its residual quads (cur) for the next step
#define NTD_RFETCH(HV, DIX_)                                                                                 \
    {                                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) hcar[e] = (HV)[(DIX_) + e];                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) rres[g] = __builtin_amdgcn_raw_buffer_load_b128(rsR, cur_voR, cur_soR + g * 32, NTD_RES_NT); \
    }

    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A0, B0 landed; A1 in flight
    while (true) {
        f32x16_t acc[NI][NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8_t wf0[NI], xf0[NJ], wf1[NI], xf1[NJ];
#define LDFRAG(WF, XF, KK)                                                                                   \
    {                                                                                                        \
        const int c_ = (KK) * 2 + (lane >> 5);                                                               \
        WF[0] = ntd_frag(sA, 192 + wn * 32 + (lane & 31), c_);                                               \
        _Pragma("unroll") for (int i = 1; i < NI; ++i) WF[i] = ntd_frag(sB, wn * 128 + (i - 1) * 32 + (lane & 31), c_); \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) XF[j] = ntd_frag(sA, wm * WMR + j * 32 + (lane & 31), c_); \
    }
#define DOMFMA(WF, XF)                                                                                       \
    { _Pragma("unroll") for (int j = 0; j < NJ; ++j) _Pragma("unroll") for (int i = 0; i < NI; ++i)          \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[i], XF[j], acc[i][j], 0, 0, 0); }
    // one slice: 15 MFMAs with NR LDS reads and ND DMA instructions threaded between them; DEF > 0: NV VALU / transcendental
    // instructions and NS stores of the deferred epilogue behind every MFMA as well
#define SCHED(NR, ND)                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < (NR); ++s_) {                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
    _Pragma("unroll") for (int s_ = 0; s_ < (ND); ++s_) {                                                    \
        if ((NR) + s_ < NJ * NI) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }                                                 \
    if (NJ * NI - (NR) - (ND) > 0) __builtin_amdgcn_sched_group_barrier(0x008, NJ * NI - (NR) - (ND), 0);
#define SCHED_D(NR, ND, NST)                                                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NJ * NI; ++s_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
        if (s_ < (NR)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        else if (s_ < (NR) + (ND)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        \
        else if (s_ < (NR) + (ND) + (NST)) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                \
        __builtin_amdgcn_sched_group_barrier(NTD_VALU_MASK, NTD_VALU_PER_GAP, 0);                            \
    }
    // light deferred slices: NST stores beside the first fragment reads, NLD extra loads behind the DMA requests, NV VALU per gap
#define SCHED_L(NR, ND, NST, NLD, NV)                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NJ * NI; ++s_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   \
        if (s_ < (NR)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
        else if (s_ < (NR) + (ND)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                        \
        else if (s_ < (NR) + (ND) + ((NLD) + 1) / 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      \
        if (s_ < (NST)) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                   \
        if ((NV) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);                                  \
    }
    // MODE_: 0 plain step, 1 deferred step on block (HV, DIX_), 2 the step after the last deferred one.
    //   GELU: a deferred step stores the quads carried from the step before (slices 0 / 1) and computes quad g of its block in slice g;
    //   bf16 light: stores its own block, two quads in slice 0, two in slice 1;
    //   residual: finishes the block before (held dwords + residual quads fetched one step ago), then fetches its own.
#define KSTEP(FIRST_, WAIT_, MODE_, HV, DIX_)                                                                \
    {                                                                                                        \
        const int sb_ = sa + 1 >= 5 ? sa - 4 : sa + 1;                                                       \
        const int db_ = sa + 3 >= 5 ? sa - 2 : sa + 3;                                                       \
        const int da_ = sa + 4 >= 5 ? sa - 1 : sa + 4;                                                       \
        char* sA = smem + sa * NTD_SLOT;                                                                     \
        char* sB = smem + sb_ * NTD_SLOT;                                                                    \
        u32x2_t no_[4], nd_[4];                                                                              \
        asm volatile(WAIT_ ::: "memory");                                                                    \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        LDFRAG(wf0, xf0, 0)                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) NTD_DMA_B(e, db_)                                      \
        if (GELU_K) { if ((MODE_) >= 1) { NTD_STORE(0) NTD_STORE(1) } if ((MODE_) == 1) NTD_QUAD(HV, DIX_, 0, no_[0], nd_[0]) } \
        else if (RES_K) { if ((MODE_) >= 1) { NTD_RSTORE(0) NTD_RSTORE(1) } }                                \
        else { if ((MODE_) == 1) { NTD_LSTORE(HV, DIX_, 0) NTD_LSTORE(HV, DIX_, 1) } }                       \
        if (!(FIRST_)) { DOMFMA(wf1, xf1)                                                                    \
            if (GELU_K && (MODE_) >= 1) { SCHED_D(8, 4, (DAUX ? 4 : 2)) }                                    \
            else if (RES_K && (MODE_) >= 1) { SCHED_L(8, 4, 2, 0, 1) }                                       \
            else if (!GELU_K && !RES_K && (MODE_) == 1) { SCHED_L(8, 4, 2, 0, 0) }                           \
            else { SCHED(8, 4) } }                                                                           \
        LDFRAG(wf1, xf1, 1)                                                                                  \
        _Pragma("unroll") for (int e = 4; e < 8; ++e) NTD_DMA_B(e, db_)                                      \
        if (GELU_K) { if ((MODE_) >= 1) { NTD_STORE(2) NTD_STORE(3) } if ((MODE_) == 1) NTD_QUAD(HV, DIX_, 1, no_[1], nd_[1]) } \
        else if (RES_K) { if ((MODE_) >= 1) { NTD_RSTORE(2) NTD_RSTORE(3) } if ((MODE_) == 1) NTD_RFETCH(HV, DIX_) } \
        else { if ((MODE_) == 1) { NTD_LSTORE(HV, DIX_, 2) NTD_LSTORE(HV, DIX_, 3) } }                       \
        DOMFMA(wf0, xf0)                                                                                     \
        if (GELU_K && (MODE_) >= 1) { SCHED_D(8, 4, (DAUX ? 4 : 2)) }                                        \
        else if (RES_K && (MODE_) >= 1) { SCHED_L(8, 4, 2, ((MODE_) == 1 ? 4 : 0), 2) }                      \
        else if (!GELU_K && !RES_K && (MODE_) == 1) { SCHED_L(8, 4, 2, 0, 0) }                               \
        else { SCHED(8, 4) }                                                                                 \
        LDFRAG(wf0, xf0, 2)                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) NTD_DMA_A(e, da_)                                      \
        if (GELU_K && (MODE_) == 1) NTD_QUAD(HV, DIX_, 2, no_[2], nd_[2])                                    \
        DOMFMA(wf1, xf1) if (GELU_K && (MODE_) == 1) { SCHED_D(8, 4, 0) } else { SCHED(8, 4) }               \
        LDFRAG(wf1, xf1, 3)                                                                                  \
        _Pragma("unroll") for (int e = 4; e < 8; ++e) NTD_DMA_A(e, da_)                                      \
        if (GELU_K && (MODE_) == 1) NTD_QUAD(HV, DIX_, 3, no_[3], nd_[3])                                    \
        DOMFMA(wf0, xf0) if (GELU_K && (MODE_) == 1) { SCHED_D(8, 4, 0) } else { SCHED(8, 4) }               \
        if (GELU_K && (MODE_) == 1) { _Pragma("unroll") for (int g = 0; g < 4; ++g) { cst_o[g] = no_[g]; if (DAUX) cst_d[g] = nd_[g]; } } \
        sa = sa + 2 >= 5 ? sa - 3 : sa + 2;                                                                  \
        ka += BK; kb += BK;                                                                                  \
    }
        // the address state moves on to block (I_, J_) of the held tile (called BEFORE the step that works on it); without a held
        // tile, past the last block or past row M: out-of-range lane offsets (loads return zeros, stores are dropped)
#define NTD_ADDR(I_, J_, LIVE_)                                                                              \
    {                                                                                                        \
        prv_soC = cur_soC; prv_voC = cur_voC; prv_soX = cur_soX; prv_voX = cur_voX;                          \
        const bool ok_ = NTD_DROP_DEFERRED ? false : (p_valid && (LIVE_) && prow + 32 * (J_) < a.M);           \
        cur_soC = ((J_) * 32 * (int)a.ldc + (I_) * 32) * ESZ;                                                \
        cur_voC = ok_ ? pvoC : OOB;                                                                          \
        if (DAUX) { cur_soX = ((J_) * 32 * (int)a.ldaux + (I_) * 32) * 2; cur_voX = ok_ ? pvoX : OOB; }      \
        if (RES_K) { cur_soR = ((J_) * 32 * (int)a.ldr + (I_) * 32) * 4; cur_voR = ok_ ? pvoR : OOB; }       \
    }
#define NTD_STEP_WAIT "s_waitcnt vmcnt(8) lgkmcnt(0)"
        KSTEP(true, "s_waitcnt lgkmcnt(0)", 0, H0, 0)
        cur_voC = OOB; cur_voX = OOB; cur_voR = OOB;  // nothing is pending when the deferred steps start
#pragma nounroll
        for (int r = 0; r < 4; ++r) {
            const int ri = __builtin_amdgcn_readfirstlane(r);
            const int dix = ri * 8;
#if NTD_DIAG >= 1 && NTD_DIAG != 5
            (void)dix; KSTEP(false, NTD_STEP_WAIT, 0, H0, 0) KSTEP(false, NTD_STEP_WAIT, 0, H0, 0) KSTEP(false, NTD_STEP_WAIT, 0, H0, 0)
#else
            NTD_ADDR(ri, 0, true) KSTEP(false, NTD_STEP_WAIT, 1, H0, dix)
            NTD_ADDR(ri, 1, true) KSTEP(false, NTD_STEP_WAIT, 1, H1, dix)
            NTD_ADDR(ri, 2, true) KSTEP(false, NTD_STEP_WAIT, 1, H2, dix)
#endif
        }
        NTD_ADDR(0, 0, false)
#if NTD_DIAG >= 1 && NTD_DIAG != 5
        KSTEP(false, NTD_STEP_WAIT, 0, H0, 0)
#else
        KSTEP(false, NTD_STEP_WAIT, 2, H0, 0)         // step 13: what the last deferred step left pending
#endif
        for (int t = 14; t < nk - 2; ++t) KSTEP(false, NTD_STEP_WAIT, 0, H0, 0)
        NTD_SWITCH_A()
        KSTEP(false, NTD_STEP_WAIT, 0, H0, 0)
        NTD_SWITCH_B()
        // the bias quads of this tile's 20 column quads (column 32 i + 8 g + 4 hh of the wave's 160): requested before the last
        // step (they are older than its DMA requests: the counted wait behind the loop covers them); the held registers are dead here
        float4 bq[NI][4];
        {
            const int en0_ = n0 + wn * WNC;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (BIAS_K) {
                        const auto b_ = __builtin_amdgcn_raw_buffer_load_b128(rsBi, (unsigned)((en0_ + 32 * i + 8 * g + 4 * hh) * 4), 0, 0);
                        bq[i][g] = make_float4(__uint_as_float(b_[0]), __uint_as_float(b_[1]), __uint_as_float(b_[2]), __uint_as_float(b_[3]));
                    } else {
                        bq[i][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
        }
        KSTEP(false, NTD_STEP_WAIT, 0, H0, 0)
        DOMFMA(wf1, xf1)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+a"(acc[i][j]));
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // bias quads + the B half of the next tile's step 0
#undef KSTEP
#undef SCHED
#undef SCHED_D
#undef LDFRAG
#undef DOMFMA
#if NTD_DIAG < 2 || NTD_DIAG >= 4
        // ---- hold: H <- bf16(acc + bias [, x scale on the q columns]), the AMP Linear output; lane (ml, hh) of block (i, j): row 32 j + ml,
        // columns 32 i + 8 g + 4 hh + e
        {
            const int em0 = m0 + wm * WMR, en0 = n0 + wn * WNC;
            const int nq0 = en0 + 4 * hh;             // (x 1.0f is exact: the same bits as the ring kernel's conditional multiply)
            // the quad (I_, J_, g) as two packed dwords
#define NTD_PACKQ(I_, J_, G_, W0, W1)                                                                        \
    {                                                                                                        \
        float v0_ = acc[I_][J_][4 * (G_)], v1_ = acc[I_][J_][4 * (G_) + 1], v2_ = acc[I_][J_][4 * (G_) + 2], v3_ = acc[I_][J_][4 * (G_) + 3]; \
        if (BIAS_K) { v0_ += bq[I_][G_].x; v1_ += bq[I_][G_].y; v2_ += bq[I_][G_].z; v3_ += bq[I_][G_].w; } \
        if (SCALE_K) { const float sc_ = (nq0 + 32 * (I_) + 8 * (G_) < a.scale_ncols) ? a.scale : 1.0f; v0_ *= sc_; v1_ *= sc_; v2_ *= sc_; v3_ *= sc_; } \
        (W0) = pack_bf16x2(v0_, v1_); (W1) = pack_bf16x2(v2_, v3_);                                          \
    }
#define NTD_HOLD(HV, I_, J_)                                                                                 \
    { _Pragma("unroll") for (int g = 0; g < 4; ++g) { unsigned w0_, w1_; NTD_PACKQ(I_, J_, g, w0_, w1_) (HV)[8 * (I_) + 2 * g] = w0_; (HV)[8 * (I_) + 2 * g + 1] = w1_; } }
            pvoC = (unsigned)((((int64_t)(em0 + ml)) * a.ldc + en0 + 4 * hh) * ESZ);
            if (DAUX) pvoX = (unsigned)((((int64_t)(em0 + ml)) * a.ldaux + en0 + 4 * hh) * 2);
            if (RES_K) pvoR = (unsigned)((((int64_t)(em0 + ml)) * a.ldr + en0 + 4 * hh) * 4);
            prow = em0 + ml;
            p_valid = true;
            // n block 4 of the three row blocks is finished here (not held); its residual quads are requested first
            u32x4_t ir[RES_K ? NJ : 1][RES_K ? 4 : 1];
            if (RES_K) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        ir[j][g] = __builtin_amdgcn_raw_buffer_load_b128(rsR, prow + 32 * j < a.M ? pvoR : OOB, (j * 32 * (int)a.ldr + 128) * 4 + g * 32, NTD_RES_NT);
            }
            NTD_HOLD(H0, 0, 0) NTD_HOLD(H1, 0, 1) NTD_HOLD(H2, 0, 2) NTD_HOLD(H0, 1, 0) NTD_HOLD(H1, 1, 1) NTD_HOLD(H2, 1, 2)
            NTD_HOLD(H0, 2, 0) NTD_HOLD(H1, 2, 1) NTD_HOLD(H2, 2, 2) NTD_HOLD(H0, 3, 0) NTD_HOLD(H1, 3, 1) NTD_HOLD(H2, 3, 2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bool rok = NTD_ROK(prow + 32 * j < a.M);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned w0, w1;
                    NTD_PACKQ(4, j, g, w0, w1)
                    if (GELU_K) {
                        const float x0 = __uint_as_float(w0 << 16), x1 = __uint_as_float(w0 & 0xffff0000u);
                        const float x2 = __uint_as_float(w1 << 16), x3 = __uint_as_float(w1 & 0xffff0000u);
                        float c0, c1, c2, c3, p0, p1, p2, p3;
                        gelu_cdf_pdf(x0, c0, p0); gelu_cdf_pdf(x1, c1, p1); gelu_cdf_pdf(x2, c2, p2); gelu_cdf_pdf(x3, c3, p3);
                        const u32x2_t o = {pack_bf16x2(x0 * c0, x1 * c1), pack_bf16x2(x2 * c2, x3 * c3)};
                        __builtin_amdgcn_raw_buffer_store_b64(o, rsC, rok ? pvoC : OOB, (j * 32 * (int)a.ldc + 128) * 2 + g * 16, NTD_C16_NT);
                        if (DAUX) {
                            const u32x2_t d = {pack_bf16x2(fmaf(x0, p0, c0), fmaf(x1, p1, c1)), pack_bf16x2(fmaf(x2, p2, c2), fmaf(x3, p3, c3))};
                            __builtin_amdgcn_raw_buffer_store_b64(d, rsX, rok ? pvoX : OOB, (j * 32 * (int)a.ldaux + 128) * 2 + g * 16, NTD_X_NT);
                        }
                    } else if (RES_K) {
                        const u32x4_t o = {__float_as_uint(__uint_as_float(w0 << 16) + __uint_as_float(ir[j][g][0])),
                                           __float_as_uint(__uint_as_float(w0 & 0xffff0000u) + __uint_as_float(ir[j][g][1])),
                                           __float_as_uint(__uint_as_float(w1 << 16) + __uint_as_float(ir[j][g][2])),
                                           __float_as_uint(__uint_as_float(w1 & 0xffff0000u) + __uint_as_float(ir[j][g][3]))};
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, rok ? pvoC : OOB, (j * 32 * (int)a.ldc + 128) * 4 + g * 32, NTD_C32_NT);
                    } else {
                        const u32x2_t o = {w0, w1};
                        __builtin_amdgcn_raw_buffer_store_b64(o, rsC, rok ? pvoC : OOB, (j * 32 * (int)a.ldc + 128) * 2 + g * 16, NTD_C16_NT);
                    }
                }
            }
#undef NTD_HOLD
#undef NTD_PACKQ
        }
#endif
        const bool more_tiles = have_next;
        v = nv; m0 = nm0; n0 = nn0;
        if (!more_tiles) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (empty-descriptor) DMA instructions
#if NTD_DIAG < 3 || NTD_DIAG >= 4
    // ---- flush: the last tile's held blocks (static register indices)
    {
        const int ldc2 = (int)a.ldc * ESZ, ldx2 = DAUX ? (int)a.ldaux * 2 : 0, ldr4 = RES_K ? (int)a.ldr * 4 : 0;
#define NTD_FLUSH(HV, I_, J_)                                                                                \
    {                                                                                                        \
        const bool rok_ = NTD_ROK(prow + 32 * (J_) < a.M);                                                   \
        u32x4_t fr_[RES_K ? 4 : 1];                                                                          \
        if (RES_K) { _Pragma("unroll") for (int g = 0; g < 4; ++g)                                           \
            fr_[g] = __builtin_amdgcn_raw_buffer_load_b128(rsR, rok_ ? pvoR : OOB, (J_) * 32 * ldr4 + (I_) * 128 + g * 32, NTD_RES_NT); } \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                      \
            if (GELU_K) {                                                                                    \
                u32x2_t o_, dd_ = u32x2_t{0u, 0u};                                                           \
                NTD_QUAD(HV, 8 * (I_), g, o_, dd_)                                                           \
                __builtin_amdgcn_raw_buffer_store_b64(o_, rsC, rok_ ? pvoC : OOB, (J_) * 32 * ldc2 + (I_) * 64 + g * 16, NTD_C16_NT); \
                if (DAUX) __builtin_amdgcn_raw_buffer_store_b64(dd_, rsX, rok_ ? pvoX : OOB, (J_) * 32 * ldx2 + (I_) * 64 + g * 16, NTD_X_NT); \
            } else if (RES_K) {                                                                              \
                const unsigned w0_ = (HV)[8 * (I_) + 2 * g], w1_ = (HV)[8 * (I_) + 2 * g + 1];               \
                const u32x4_t o_ = {__float_as_uint(__uint_as_float(w0_ << 16) + __uint_as_float(fr_[g][0])), \
                                    __float_as_uint(__uint_as_float(w0_ & 0xffff0000u) + __uint_as_float(fr_[g][1])), \
                                    __float_as_uint(__uint_as_float(w1_ << 16) + __uint_as_float(fr_[g][2])), \
                                    __float_as_uint(__uint_as_float(w1_ & 0xffff0000u) + __uint_as_float(fr_[g][3]))}; \
                __builtin_amdgcn_raw_buffer_store_b128(o_, rsC, rok_ ? pvoC : OOB, (J_) * 32 * ldc2 + (I_) * 128 + g * 32, NTD_C32_NT); \
            } else {                                                                                         \
                const u32x2_t o_ = {(HV)[8 * (I_) + 2 * g], (HV)[8 * (I_) + 2 * g + 1]};                     \
                __builtin_amdgcn_raw_buffer_store_b64(o_, rsC, rok_ ? pvoC : OOB, (J_) * 32 * ldc2 + (I_) * 64 + g * 16, NTD_C16_NT); \
            }                                                                                                \
        }                                                                                                    \
    }
        NTD_FLUSH(H0, 0, 0) NTD_FLUSH(H1, 0, 1) NTD_FLUSH(H2, 0, 2) NTD_FLUSH(H0, 1, 0) NTD_FLUSH(H1, 1, 1) NTD_FLUSH(H2, 1, 2)
        NTD_FLUSH(H0, 2, 0) NTD_FLUSH(H1, 2, 1) NTD_FLUSH(H2, 2, 2) NTD_FLUSH(H0, 3, 0) NTD_FLUSH(H1, 3, 1) NTD_FLUSH(H2, 3, 2)
#undef NTD_FLUSH
    }
#endif
}

// ---- host side (called by gemm_nt_impl in gemm.hip; not part of the C ABI)
#include <stdlib.h>
// which epilogues take this kernel (bits: 1 inference fc1, 2 training fc1, 4 plain / bias / q-scale, 8 fp32 residual): DICOW_NT_DEFER
extern "C" __attribute__((visibility("hidden"))) int dicow_ntd_mode_(int dflt) {
    const char* e = getenv("DICOW_NT_DEFER");
    return e ? atoi(e) : dflt;
}
#define NTD_FOR_FLAGS(X)                                                                                     \
    X(DICOW_EPI_BIAS | DICOW_EPI_GELU) X(DICOW_EPI_BIAS | DICOW_EPI_GELU | DICOW_EPI_GELU_DAUX)             \
    X(0) X(DICOW_EPI_BIAS) X(DICOW_EPI_BIAS | DICOW_EPI_SCALE_N) X(DICOW_EPI_BIAS | DICOW_EPI_RESIDUAL | DICOW_EPI_OUT_F32)
extern "C" __attribute__((visibility("hidden"))) int dicow_ntd_launch_(const dicow_gemm_args* a, int grid, void* stream) {
    static bool once = false;
    if (!once) {
#define X(F) (void)hipFuncSetAttribute((const void*)gemm_ntd_kernel<(F)>, hipFuncAttributeMaxDynamicSharedMemorySize, NTD_LDS);
        NTD_FOR_FLAGS(X)
#undef X
        once = true;
    }
    switch (a->flags) {
#define X(F) case (F): hipLaunchKernelGGL((gemm_ntd_kernel<(F)>), dim3(grid), dim3(256), NTD_LDS, (hipStream_t)stream, *a); return 0;
        NTD_FOR_FLAGS(X)
#undef X
        default: return -1;
    }
}
