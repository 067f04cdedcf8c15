// Fused FDDT (+pos-emb) + LayerNorm row kernels, forward and backward.  HBM-bound.
//
// Replaces (reference, /root/reference): src/models/dicow/FDDT.py:41-63, src/models/dicow/layers.py:73-77,
// encoder.py:173-180,205-206 and the LayerNorm calls of HF WhisperEncoderLayer/DecoderLayer.
//
// Work decomposition ("column owner"): a block has D/4 threads (rounded up to a wave); thread `tid` owns the
// four columns [4*tid, 4*tid+4) for EVERY row the block visits, so the 8 FDDT vectors + LayerNorm affine live
// in registers (loaded once) and the parameter-gradient column sums accumulate in registers across rows.
// Rows are visited R at a time (R independent 16-byte loads in flight per thread); row statistics use one
// wave-shuffle + LDS reduction per R rows.  Algorithmic HBM bytes per row: fwd  D*(4 in + 4 out + 2 ln-out),
// bwd  D*(4 h_in + 2 d_y + 4 g_res + 4 g_out + 2 g_out_bf16).
#include <stdlib.h>
#include "common.h"

#define MAX_WAVES 16
#ifndef ROWS_GATHER
#define ROWS_GATHER 2      // staged forward: partials read back two waves at a time (78 -> 62 us; all at once: 88 us, register cliff)
#endif
#ifndef ROWS_G_FWD
#define ROWS_G_FWD 2       // LayerNorm-only forward: two at a time (40 -> 30.5 us; all at once 33 us)
#endif
#ifndef ROWS_G_BWD
#define ROWS_G_BWD 0       // LayerNorm-only / generic backward: the plain loop
#endif
#ifndef ROWS_G_BWDS
#define ROWS_G_BWDS 16     // staged backward: all at once
#endif


struct f4 { float x, y, z, w; };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#ifndef ROWS_G_NT
#define ROWS_G_NT 1        // generic forward body: nontemporal fp32 row loads (LayerNorm 2 reads the residual stream once: encoder forward -0.25 %)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4n_t;
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    const f32x4n_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#ifndef ROWS_GB_NT
#define ROWS_GB_NT 0       // generic backward body: nontemporal h_in / d_y / g_res loads
#endif
typedef __attribute__((ext_vector_type(2))) unsigned u32x2r_t;
__device__ __forceinline__ float4 ld4_bf16_nt(const void* base, int64_t idx) {
    const u32x2r_t u = __builtin_nontemporal_load(reinterpret_cast<const u32x2r_t*>(reinterpret_cast<const unsigned short*>(base) + idx));
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4_bf16(const void* base, int64_t idx) {
    uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4_bf16(void* base, int64_t idx, float4 v) {
    uint2 u = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = u;
}

// Reference evaluation order, separate fp32 multiplies/adds (bit-exact vs the fp32 reference).
__device__ __forceinline__ float fddt_diag_elem(float h, float w0, float b0, float w1, float b1, float w2, float b2,
                                                float w3, float b3, float m0, float m1, float m2, float m3) {
#pragma clang fp contract(off)
    float t0 = (h * w0 + b0) * m0;
    float t1 = (h * w1 + b1) * m1;
    float t2 = (h * w2 + b2) * m2;
    float t3 = (h * w3 + b3) * m3;
    return ((t0 + t1) + t2) + t3;
}
__device__ __forceinline__ float fddt_bias_elem(float h, float b0, float b1, float b2, float b3, float m0, float m1,
                                                float m2, float m3, int use) {
#pragma clang fp contract(off)
    if (use & 1) h = h + m0 * b0;
    if (use & 2) h = h + m1 * b1;
    if (use & 4) h = h + m2 * b2;
    if (use & 8) h = h + m3 * b3;
    return h;
}

// The same evaluation order on a PAIR of columns: v_pk_mul_f32 / v_pk_add_f32 are the IEEE operations of the scalar form
// (no contraction), two columns per issue slot.  The staged row kernels are VALU-bound without it: ~500 VALU instructions
// per thread and 4-row trip against ~8800 cycles per trip and CU.
typedef __attribute__((ext_vector_type(2))) float f32x2r_t;
__device__ __forceinline__ f32x2r_t fddt_diag_pair(f32x2r_t h, f32x2r_t w0, f32x2r_t b0, f32x2r_t w1, f32x2r_t b1, f32x2r_t w2,
                                                   f32x2r_t b2, f32x2r_t w3, f32x2r_t b3, float m0, float m1, float m2, float m3) {
#pragma clang fp contract(off)
    const f32x2r_t t0 = (h * w0 + b0) * f32x2r_t{m0, m0};
    const f32x2r_t t1 = (h * w1 + b1) * f32x2r_t{m1, m1};
    const f32x2r_t t2 = (h * w2 + b2) * f32x2r_t{m2, m2};
    const f32x2r_t t3 = (h * w3 + b3) * f32x2r_t{m3, m3};
    return ((t0 + t1) + t2) + t3;
}

#define F4_APPLY(dst, expr) do { dst.x = expr(x); dst.y = expr(y); dst.z = expr(z); dst.w = expr(w); } while (0)

// block-wide sum of NV values per thread; result broadcast to all threads.  `red` is [MAX_WAVES][NV].
// The read-back of the per-wave partials is written out per wave count: with the count in a loop variable every partial was
// a dependent LDS round trip of its own (5 waves x 2 reductions per trip = 30 % of the staged forward's time, ablation builds
// in profiles/r02_ablations.txt); unrolled, the reads of a reduction are issued together and waited for once.  Same order of
// additions as the loop.
template <int NV, int NW, int CHUNK = NW>      // CHUNK partials in flight at a time (registers: CHUNK x NV)
__device__ __forceinline__ void block_sum_gather(float (&v)[NV], const float* red) {
    float s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = 0.f;
#pragma unroll
    for (int w0 = 0; w0 < NW; w0 += CHUNK) {
        float p[CHUNK][NV];
#pragma unroll
        for (int w = 0; w < CHUNK; ++w)
#pragma unroll
            for (int i = 0; i < NV; ++i) p[w][i] = (w0 + w < NW) ? red[(w0 + w) * NV + i] : 0.f;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int w = 0; w < CHUNK; ++w)
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (w0 + w < NW) s[i] += p[w][i];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = s[i];
}
template <int NV, bool UNROLLED = false, int CHUNK = 16>   // UNROLLED costs min(NW, CHUNK) x NV registers: only where it measured faster
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red, int nwaves) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_dpp(v[i]);
    if (nwaves == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wv * NV + i] = v[i];
    }
    __syncthreads();
    if (UNROLLED) switch (nwaves) {
        case 2: block_sum_gather<NV, 2, (CHUNK < 2 ? CHUNK : 2)>(v, red); return;
        case 3: block_sum_gather<NV, 3, (CHUNK < 3 ? CHUNK : 3)>(v, red); return;
        case 4: block_sum_gather<NV, 4, (CHUNK < 4 ? CHUNK : 4)>(v, red); return;
        case 5: block_sum_gather<NV, 5, (CHUNK < 5 ? CHUNK : 5)>(v, red); return;          // D = 1280 (whisper-large-v3-turbo)
        case 6: block_sum_gather<NV, 6, (CHUNK < 6 ? CHUNK : 6)>(v, red); return;          // D = 1536
        case 8: block_sum_gather<NV, 8, (CHUNK < 8 ? CHUNK : 8)>(v, red); return;          // D = 2048
        default: break;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int w = 0; w < nwaves; ++w) s += red[w * NV + i];
        v[i] = s;
    }
}

// Chan/Welford combination of (count, mean, M2) so that mean and variance need ONE block-wide reduction (one barrier)
// without the E[x^2]-mean^2 cancellation.
__device__ __forceinline__ void wf_combine(float& n, float& mu, float& m2, float nb, float mub, float m2b) {
    const float nt = n + nb;
    if (nt > 0.f) {
        const float d = mub - mu, f = nb / nt;
        mu += d * f;
        m2 += m2b + d * d * n * f;
        n = nt;
    }
}
template <int R>
__device__ __forceinline__ void block_welford(float (&n)[R], float (&mu)[R], float (&m2)[R], float* red, int nwaves) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float nb = __shfl_xor(n[r], o, 64), mb = __shfl_xor(mu[r], o, 64), qb = __shfl_xor(m2[r], o, 64);
            wf_combine(n[r], mu[r], m2[r], nb, mb, qb);
        }
    }
    if (nwaves == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { red[(wv * R + r) * 3] = n[r]; red[(wv * R + r) * 3 + 1] = mu[r]; red[(wv * R + r) * 3 + 2] = m2[r]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float tn = 0.f, tm = 0.f, tq = 0.f;
        for (int w = 0; w < nwaves; ++w)
            wf_combine(tn, tm, tq, red[(w * R + r) * 3], red[(w * R + r) * 3 + 1], red[(w * R + r) * 3 + 2]);
        n[r] = tn; mu[r] = tm; m2[r] = tq;
    }
}

template <int R, int MAXT, int MODE_T = -1>     // MODE_T: compile-time mode (see fddt_ln_bwd_kernel), -1 = runtime
__global__ void __launch_bounds__(MAXT) fddt_ln_fwd_kernel(const dicow_fddt_ln_fwd_args a) {
    const int mode = MODE_T >= 0 ? MODE_T : a.mode;
    __shared__ __attribute__((aligned(16))) float red[2][MAX_WAVES * R * 3];
    const int tid = threadIdx.x, col = tid * 4, D = a.D;
    const bool act = col < D;
    const int nwaves = blockDim.x >> 6;
    const float4 one = make_float4(1, 1, 1, 1), zero = make_float4(0, 0, 0, 0);
    float4 w[4], b[4], lnw = one, lnb = zero;
    int use = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        w[c] = (mode == 1 && a.w[c] && act) ? ld4(a.w[c] + col) : one;
        b[c] = (mode != 0 && a.b[c] && act) ? ld4(a.b[c] + col) : zero;
        if (a.b[c]) use |= 1 << c;
    }
    const bool do_ln = a.ln_w != nullptr;
    if (do_ln && act) { lnw = ld4(a.ln_w + col); lnb = ld4(a.ln_b + col); }
    const float inv_d = 1.0f / (float)D;

    for (int row0 = blockIdx.x * R; row0 < a.rows; row0 += gridDim.x * R) {
        float4 x[R];
        float m[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            const bool ok = act && row < a.rows;
            const int64_t off = (int64_t)row * D + col;
            x[r] = zero;
            if (ok) x[r] = a.in_bf16 ? ld4_bf16(a.h_in, off) : (ROWS_G_NT ? ld4_nt(reinterpret_cast<const float*>(a.h_in) + off) : ld4(reinterpret_cast<const float*>(a.h_in) + off));
            if (mode != 0 && row < a.rows) {
                const int bi = row / a.T, t = row - bi * a.T;
                const float* mp = a.stno + (int64_t)bi * a.stno_bstride + t;
#pragma unroll
                for (int c = 0; c < 4; ++c)      // row-indexed, identical in every lane: keep it in a scalar register
                    m[r][c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[(int64_t)c * a.T])));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) m[r][c] = 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (mode == 1) {
#define FD(e) fddt_diag_elem(x[r].e, w[0].e, b[0].e, w[1].e, b[1].e, w[2].e, b[2].e, w[3].e, b[3].e, m[r][0], m[r][1], m[r][2], m[r][3])
                F4_APPLY(x[r], FD);
#undef FD
            } else if (mode == 2) {
#define FB(e) fddt_bias_elem(x[r].e, b[0].e, b[1].e, b[2].e, b[3].e, m[r][0], m[r][1], m[r][2], m[r][3], use)
                F4_APPLY(x[r], FB);
#undef FB
            }
            if (a.pos && act && row < a.rows) {
                const int t = row % a.T;
                const float4 p = ld4(a.pos + (int64_t)t * D + col);
                x[r].x += p.x; x[r].y += p.y; x[r].z += p.z; x[r].w += p.w;
            }
            if (a.h_out && act && row < a.rows) st4(a.h_out + (int64_t)row * D + col, x[r]);
        }
        if (!do_ln) continue;
        float sm[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sm[r] = act ? (x[r].x + x[r].y) + (x[r].z + x[r].w) : 0.f;
        // (unrolled read-back of the partials: LayerNorm-only body 40 -> 32 us; the FDDT bodies lose occupancy to its registers)
        block_sum<R, (MODE_T == 0 && ROWS_G_FWD != 0), (ROWS_G_FWD ? ROWS_G_FWD : 16)>(sm, red[0], nwaves);
        float mu[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            mu[r] = sm[r] * inv_d;
            const float dx = x[r].x - mu[r], dy = x[r].y - mu[r], dz = x[r].z - mu[r], dw = x[r].w - mu[r];
            q[r] = act ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f;
        }
        block_sum<R, (MODE_T == 0 && ROWS_G_FWD != 0), (ROWS_G_FWD ? ROWS_G_FWD : 16)>(q, red[1], nwaves);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (row >= a.rows) continue;
            const float rs = rsqrtf(q[r] * inv_d + a.eps);
            if (tid == 0) {
                if (a.mean) a.mean[row] = mu[r];
                if (a.rstd) a.rstd[row] = rs;
            }
            if (!act) continue;
            float4 y;
            y.x = (x[r].x - mu[r]) * rs * lnw.x + lnb.x;
            y.y = (x[r].y - mu[r]) * rs * lnw.y + lnb.y;
            y.z = (x[r].z - mu[r]) * rs * lnw.z + lnb.z;
            y.w = (x[r].w - mu[r]) * rs * lnw.w + lnb.w;
            const int64_t off = (int64_t)row * D + col;
            if (a.y_bf16) st4_bf16(a.y_bf16, off, y);
            if (a.y_f32) st4(a.y_f32 + off, y);
        }
        // (the two barriers inside block_sum order the LDS reuse across iterations)
    }
}

// ------------------------------------------------------------------------------------------------ forward, LayerNorm only, a wave per row
// The LayerNorm that follows every attention block (and the encoder's final one): fp32 rows in, bf16 (and optionally fp32) rows,
// mean, rstd out.  The column-owner body above pays two block-wide reductions (two barriers) per trip of R rows and runs at
// 4.7 TB/s where its own copy mode reaches 6; the forward has no column sums to keep, so here a WAVE owns whole rows: lane l holds
// the float4 at columns 4 l + 256 k, k < NC (D = 256 NC: 1280 -> 5), both statistics are DPP reductions inside the wave -- no LDS,
// no barrier, nothing shared between the waves of a workgroup -- and the affine vectors live in registers (2 x 4 NC).  R rows are
// requested together.  Same two-pass arithmetic (mean, then centred sum of squares) as the block form.
#ifndef LNW_ON
#define LNW_ON 1
#endif
#ifndef LNW_R
#define LNW_R 2
#endif
#ifndef LNW_WAVES
#define LNW_WAVES 8       // (4 or 8 waves per workgroup, 1 / 2 / 4 rows per request group: equal within 0.1 % of the encoder forward)
#endif
template <int NC>
__global__ void __launch_bounds__(LNW_WAVES * 64) ln_fwd_wave_kernel(const dicow_fddt_ln_fwd_args a) {
    constexpr int R = LNW_R;
    const int lane = threadIdx.x & 63;
    const int D = a.D;
    const int wave = blockIdx.x * LNW_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * LNW_WAVES;
    float4 lw[NC], lb[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { lw[k] = ld4(a.ln_w + 4 * lane + 256 * k); lb[k] = ld4(a.ln_b + 4 * lane + 256 * k); }
    const float inv_d = 1.0f / (float)D;
    const float* H = reinterpret_cast<const float*>(a.h_in);
    for (int row0 = wave * R; row0 < a.rows; row0 += nwaves * R) {
        float4 x[R][NC];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < a.rows ? row0 + r : a.rows - 1;
#pragma unroll
            for (int k = 0; k < NC; ++k) x[r][k] = ROWS_G_NT ? ld4_nt(H + (int64_t)row * D + 4 * lane + 256 * k) : ld4(H + (int64_t)row * D + 4 * lane + 256 * k);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) sm += (x[r][k].x + x[r][k].y) + (x[r][k].z + x[r][k].w);
            const float mu = wave_sum_dpp(sm) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float dx = x[r][k].x - mu, dy = x[r][k].y - mu, dz = x[r][k].z - mu, dw = x[r][k].w - mu;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            const float rs = rsqrtf(wave_sum_dpp(q) * inv_d + a.eps);
            if (row >= a.rows) continue;                      // (wave-uniform)
            if (lane == 0) {
                if (a.mean) a.mean[row] = mu;
                if (a.rstd) a.rstd[row] = rs;
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                float4 y;
                y.x = (x[r][k].x - mu) * rs * lw[k].x + lb[k].x;
                y.y = (x[r][k].y - mu) * rs * lw[k].y + lb[k].y;
                y.z = (x[r][k].z - mu) * rs * lw[k].z + lb[k].z;
                y.w = (x[r][k].w - mu) * rs * lw[k].w + lb[k].w;
                const int64_t off = (int64_t)row * D + 4 * lane + 256 * k;
                if (a.y_bf16) st4_bf16(a.y_bf16, off, y);
                if (a.y_f32) st4(a.y_f32 + off, y);
            }
        }
    }
}

// FDDT(diag) + LayerNorm forward of every encoder layer in the same wave-per-row form: the eight FDDT vectors (4 classes x weight,
// bias) and the two LayerNorm affine vectors (10 x 4 D bytes = 50 KiB at D = 1280) live in LDS, written once per workgroup -- the only barrier of the kernel -- and read
// back as conflict-free 16-byte fragments (10 x NC ds_read_b128 per row and lane against NC 16-byte HBM loads and 3 NC stores).
// The FDDT arithmetic is the reference's evaluation order (fddt_diag_pair: packed
// IEEE multiplies / adds, no contraction), bit-exact like the other bodies.  fp32 rows leave in 8-byte stores (see the note on
// 16-byte stores in the staged kernel).
#ifndef FLW_R
#define FLW_R 1
#endif
#ifndef FLW_WAVES
#define FLW_WAVES 8
#endif
#ifndef FLW_ON
#define FLW_ON 0        // measured equal to the LDS-staged column-owner kernel (62-68 against 64.5 us in isolation, encoder forward +-0.1 ms over
#endif                  // 1 / 2 rows x 4 / 8 / 16 waves: profiles/r03_rows_wave.txt): that kernel is not bound by its barriers; kept for A/B builds
#ifndef FLW_INIT_ON
#define FLW_INIT_ON 1
#endif
#ifndef FLW_MINWG
#define FLW_MINWG 2       // resident workgroups per CU the register budget is cut for (8 waves each: 128 VGPRs)
#endif
template <int NC, bool INIT = false>      // INIT: the encoder's initial FDDT -- bf16 rows in, + positions, fp32 rows out, no LayerNorm
__global__ void __launch_bounds__(FLW_WAVES * 64, FLW_MINWG) fddt_ln_fwd_wave_kernel(const dicow_fddt_ln_fwd_args a) {
    constexpr int R = FLW_R;
    extern __shared__ __attribute__((aligned(16))) char prm[];        // [10 vectors][D] fp32: w0 b0 w1 b1 w2 b2 w3 b3 ln_w ln_b
    const int lane = threadIdx.x & 63, D = a.D;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x * 4; i < (INIT ? 8 : 10) * D; i += FLW_WAVES * 64 * 4) {
        const int v = i / D, c = i - v * D;
        const float* src = v == 8 ? a.ln_w : v == 9 ? a.ln_b : (v & 1) ? a.b[v >> 1] : a.w[v >> 1];
        *reinterpret_cast<float4*>(prm + (int64_t)i * 4) = ld4(src + c);
    }
    __syncthreads();
    const float inv_d = 1.0f / (float)D;
    const float* H = reinterpret_cast<const float*>(a.h_in);
    const int wave = blockIdx.x * FLW_WAVES + wv, nwaves = gridDim.x * FLW_WAVES;
    for (int row0 = wave * R; row0 < a.rows; row0 += nwaves * R) {
        float4 xi[R][NC];
        float m[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < a.rows ? row0 + r : a.rows - 1;
#pragma unroll
            for (int k = 0; k < NC; ++k) xi[r][k] = INIT ? ld4_bf16(a.h_in, (int64_t)row * D + 4 * lane + 256 * k)
                                                         : ROWS_G_NT ? ld4_nt(H + (int64_t)row * D + 4 * lane + 256 * k) : ld4(H + (int64_t)row * D + 4 * lane + 256 * k);
            const int bi = row / a.T, t = row - bi * a.T;
            const float* mp = a.stno + (int64_t)bi * a.stno_bstride + t;
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = mp[(int64_t)c * a.T];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            float m0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[r][0]))), m1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[r][1])));
            float m2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[r][2]))), m3 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[r][3])));
            f32x2r_t xl[NC], xh[NC];
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                float4 pv[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) pv[v] = *reinterpret_cast<const float4*>(prm + ((int64_t)v * D + 4 * lane + 256 * k) * 4);
#define FLW_P(v, lo, hi) f32x2r_t{pv[v].lo, pv[v].hi}
                xl[k] = fddt_diag_pair(f32x2r_t{xi[r][k].x, xi[r][k].y}, FLW_P(0, x, y), FLW_P(1, x, y), FLW_P(2, x, y), FLW_P(3, x, y),
                                       FLW_P(4, x, y), FLW_P(5, x, y), FLW_P(6, x, y), FLW_P(7, x, y), m0, m1, m2, m3);
                xh[k] = fddt_diag_pair(f32x2r_t{xi[r][k].z, xi[r][k].w}, FLW_P(0, z, w), FLW_P(1, z, w), FLW_P(2, z, w), FLW_P(3, z, w),
                                       FLW_P(4, z, w), FLW_P(5, z, w), FLW_P(6, z, w), FLW_P(7, z, w), m0, m1, m2, m3);
#undef FLW_P
                if constexpr (INIT) {                          // + positions, straight out (separate IEEE adds, as the column-owner body)
#pragma clang fp contract(off)
                    const int t = row - (row / a.T) * a.T;
                    const float4 pz = ld4(a.pos + (int64_t)t * D + 4 * lane + 256 * k);
                    xl[k] = xl[k] + f32x2r_t{pz.x, pz.y};
                    xh[k] = xh[k] + f32x2r_t{pz.z, pz.w};
                    if (row0 + r < a.rows) {
                        const int64_t off = (int64_t)row * D + 4 * lane + 256 * k;
                        *reinterpret_cast<u32x2r_t*>(a.h_out + off) = u32x2r_t{__float_as_uint(xl[k].x), __float_as_uint(xl[k].y)};
                        *reinterpret_cast<u32x2r_t*>(a.h_out + off + 2) = u32x2r_t{__float_as_uint(xh[k].x), __float_as_uint(xh[k].y)};
                    }
                }
                sm += (xl[k].x + xl[k].y) + (xh[k].x + xh[k].y);
                asm volatile("" ::: "memory");                // (keeps the parameter fragments of the NC chunks from being read all at once: 32 registers each)
            }
            if constexpr (INIT) continue;
            const float mu = wave_sum_dpp(sm) * inv_d;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
#pragma clang fp contract(off)
                const f32x2r_t mu2 = {mu, mu};
                const f32x2r_t dl = xl[k] - mu2, dh = xh[k] - mu2;
                const f32x2r_t ql = dl * dl, qh = dh * dh;
                q += (ql.x + ql.y) + (qh.x + qh.y);
            }
            const float rs = rsqrtf(wave_sum_dpp(q) * inv_d + a.eps);
            if (row >= a.rows) continue;                      // (wave-uniform)
            if (lane == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
#pragma clang fp contract(off)
                const int64_t off = (int64_t)row * D + 4 * lane + 256 * k;
                *reinterpret_cast<u32x2r_t*>(a.h_out + off) = u32x2r_t{__float_as_uint(xl[k].x), __float_as_uint(xl[k].y)};
                *reinterpret_cast<u32x2r_t*>(a.h_out + off + 2) = u32x2r_t{__float_as_uint(xh[k].x), __float_as_uint(xh[k].y)};
                const float4 lw = *reinterpret_cast<const float4*>(prm + ((int64_t)8 * D + 4 * lane + 256 * k) * 4);
                const float4 lb = *reinterpret_cast<const float4*>(prm + ((int64_t)9 * D + 4 * lane + 256 * k) * 4);
                const f32x2r_t mu2 = {mu, mu}, rs2 = {rs, rs};
                const f32x2r_t yl = (xl[k] - mu2) * rs2 * f32x2r_t{lw.x, lw.y} + f32x2r_t{lb.x, lb.y};
                const f32x2r_t yh = (xh[k] - mu2) * rs2 * f32x2r_t{lw.z, lw.w} + f32x2r_t{lb.z, lb.w};
                *reinterpret_cast<u32x2r_t*>(reinterpret_cast<unsigned short*>(a.y_bf16) + off) = u32x2r_t{pack_bf16x2(yl.x, yl.y), pack_bf16x2(yh.x, yh.y)};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, LDS-staged rows
// The FDDT(diag) + LayerNorm forward of every encoder layer (fp32 in, fp32 h_out, bf16 y, mean / rstd), with the rows of the
// next trip fetched by LDS-DMA while the current trip is computed -- see fddt_ln_bwd_staged_kernel for the scheme.  The FDDT
// arithmetic is the reference's evaluation order (fddt_diag_elem), bit-exact like the generic body.
typedef __attribute__((address_space(3))) void lds_void_f_t;
#ifndef ROWS_BWD_H_NT
#define ROWS_BWD_H_NT 0    // staged backward: cache-policy bits of the h_in / g_res / d_y row requests (each is the last use)
#endif
#ifndef ROWS_BWD_G_NT
#define ROWS_BWD_G_NT 0
#endif
#ifndef ROWS_BWD_Y_NT
#define ROWS_BWD_Y_NT 0
#endif
#ifndef ROWS_FWD_ST_NT
#define ROWS_FWD_ST_NT 0   // ... of its fp32 row stores / its bf16 LayerNorm-output stores
#endif
#ifndef ROWS_FWD_Y_NT
#define ROWS_FWD_Y_NT 0
#endif
#ifndef ROWS_FWD_NT
#define ROWS_FWD_NT 2      // cache-policy bits of the staged forward's row requests (2 = nt: read once; encoder forward -0.25 %)
#endif
#ifndef ROWS_ABL
#define ROWS_ABL 0     // diagnostic builds (tools/build_rows_variants.sh): 1 no fp32 row store, 2 no bf16 / stats stores, 4 no block reductions, 8 no DMA wait
#endif
template <int R>
__global__ void __launch_bounds__(512) fddt_ln_fwd_staged_kernel(const dicow_fddt_ln_fwd_args a) {
    extern __shared__ __attribute__((aligned(16))) char stg[];       // [2 stages][R rows][4*D bytes]
    __shared__ __attribute__((aligned(16))) float red[2][MAX_WAVES * R * 3];
    const int tid = threadIdx.x, col = tid * 4, D = a.D;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    float4 w[4], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { w[c] = ld4(a.w[c] + col); b[c] = ld4(a.b[c] + col); }
    const float4 lnw = ld4(a.ln_w + col), lnb = ld4(a.ln_b + col);
    const float inv_d = 1.0f / (float)D;
    const unsigned nb32 = (unsigned)((int64_t)a.rows * D * 4), nb16 = (unsigned)((int64_t)a.rows * D * 2);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.h_in), 0, nb32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.h_out, 0, nb32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y_bf16, 0, nb16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(a.mean, 0, (unsigned)(a.rows * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(a.rstd, 0, (unsigned)(a.rows * 4), 0x00020000);
    const unsigned vo32 = (unsigned)(col * 4), vo16 = (unsigned)(col * 2);
    const unsigned voStat = tid == 0 ? 0u : 0x80000000u;             // every wave issues the store; only thread 0's lands
    const int row_lds = 4 * D;
    auto stage_rows = [&](int row0, int s) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            dicow_dma16<ROWS_FWD_NT>((unsigned)(uintptr_t)(stg + (s * R + r) * row_lds + wave * 1024), rsH, vo32, (unsigned)((row0 + r) * D * 4));     // (assembly-issued: common.h)
    };
    float m_n[R][4];
    auto load_masks = [&](int row0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = row0 + r; row = row < a.rows ? row : a.rows - 1;
            const int bi = row / a.T, t = row - bi * a.T;
            const float* mp = a.stno + (int64_t)bi * a.stno_bstride + t;
#pragma unroll
            for (int c = 0; c < 4; ++c) m_n[r][c] = mp[(int64_t)c * a.T];
        }
    };
    const int stride = gridDim.x * R;
    int row0 = blockIdx.x * R, it = 0;
    if (row0 < a.rows) { load_masks(row0); stage_rows(row0, 0); }
    for (; row0 < a.rows; row0 += stride, ++it) {
        const int s = it & 1;
        float m[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) m[r][c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m_n[r][c])));
        load_masks(row0 + stride);
        stage_rows(row0 + stride, s ^ 1);
        // younger than this trip's DMA: the previous trip's 4R stores, the next trip's 4R mask loads and R DMA instructions
        if (ROWS_ABL & 8) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
        else if (ROWS_ABL & 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the store counts changed: wait for everything)
        else if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(5 * R) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(9 * R) : "memory");
        f32x2r_t xl[R], xh[R];                       // columns (0,1) and (2,3) of this thread's quad
        float sm[R];
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        u32x4_t ov[R];                               // store data: kept live until after the reductions (see below)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 xi = *reinterpret_cast<const float4*>(stg + (s * R + r) * row_lds + tid * 16);
#define FDP(lo, hi, src) fddt_diag_pair(src, f32x2r_t{w[0].lo, w[0].hi}, f32x2r_t{b[0].lo, b[0].hi}, f32x2r_t{w[1].lo, w[1].hi}, \
                                        f32x2r_t{b[1].lo, b[1].hi}, f32x2r_t{w[2].lo, w[2].hi}, f32x2r_t{b[2].lo, b[2].hi},       \
                                        f32x2r_t{w[3].lo, w[3].hi}, f32x2r_t{b[3].lo, b[3].hi}, m[r][0], m[r][1], m[r][2], m[r][3])
            xl[r] = FDP(x, y, (f32x2r_t{xi.x, xi.y}));
            xh[r] = FDP(z, w, (f32x2r_t{xi.z, xi.w}));
#undef FDP
            ov[r] = u32x4_t{__float_as_uint(xl[r].x), __float_as_uint(xl[r].y), __float_as_uint(xh[r].x), __float_as_uint(xh[r].y)};
            if (!(ROWS_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(ov[r], rsO, vo32, (row0 + r) * D * 4, ROWS_FWD_ST_NT);
            sm[r] = (xl[r].x + xl[r].y) + (xh[r].x + xh[r].y);
        }
        if (!(ROWS_ABL & 4)) block_sum<R, ROWS_GATHER != 0, (ROWS_GATHER ? ROWS_GATHER : 16)>(sm, red[0], nwaves);
        float mu[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma clang fp contract(off)
            mu[r] = sm[r] * inv_d;
            const f32x2r_t mu2 = {mu[r], mu[r]};
            const f32x2r_t dl = xl[r] - mu2, dh = xh[r] - mu2;
            const f32x2r_t ql = dl * dl, qh = dh * dh;
            q[r] = (ql.x + ql.y) + (qh.x + qh.y);
        }
        if (!(ROWS_ABL & 4)) block_sum<R, ROWS_GATHER != 0, (ROWS_GATHER ? ROWS_GATHER : 16)>(q, red[1], nwaves);
        // (one block-wide Chan/Welford reduction instead of these two measured SLOWER here: 108 vs 87 us -- the combination
        // steps cost more than the second barrier)
        // The registers a 16-byte store reads its data from must not be rewritten while the store may still be queued: with
        // the copy the compiler made for the last row (v_mov into a temporary quad that the DPP reduction right after the
        // store reused) ~1.5 % of h_out came out wrong at D = 1280 -- lanes 12..15 of every 16, second dword -- while the
        // values computed from the same registers (y, mean, rstd) were right.  Keeping the quads live across both block
        // reductions removes the reuse.
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" :: "v"(ov[r]));
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma clang fp contract(off)
            const float rs = rsqrtf(q[r] * inv_d + a.eps);
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
            if (!(ROWS_ABL & 2)) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu[r]), rsM, voStat, (row0 + r) * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rs), rsS, voStat, (row0 + r) * 4, 0);
            }
            const f32x2r_t mu2 = {mu[r], mu[r]}, rs2 = {rs, rs};
            const f32x2r_t yl = (xl[r] - mu2) * rs2 * f32x2r_t{lnw.x, lnw.y} + f32x2r_t{lnb.x, lnb.y};
            const f32x2r_t yh = (xh[r] - mu2) * rs2 * f32x2r_t{lnw.z, lnw.w} + f32x2r_t{lnb.z, lnb.w};
            const u32x2_t yv = {pack_bf16x2(yl.x, yl.y), pack_bf16x2(yh.x, yh.y)};
            if (!(ROWS_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b64(yv, rsY, vo16, (row0 + r) * D * 2, ROWS_FWD_Y_NT);
            else if (yv.x == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b64(yv, rsY, vo16, (row0 + r) * D * 2, 0);
        }
    }
}

// resident workgroups per CU for a row kernel (occupancy API, cached): the grid is sized to exactly fill the chip
// once and every workgroup strides over rows, so no partial tail wave runs at low occupancy.
template <typename K>
static int resident_grid(K kernel, int block, int* cache) {
    int nb = __atomic_load_n(cache, __ATOMIC_RELAXED);      // (any host thread may get here first: the query is idempotent)
    if (nb == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 1;
        __atomic_store_n(cache, nb, __ATOMIC_RELAXED);
    }
    return 256 * nb;
}

static int pick_block(int D) {
    int t = ((D / 4) + 63) / 64 * 64;
    return t;
}

extern "C" int dicow_fddt_ln_fwd(const dicow_fddt_ln_fwd_args* a, void* stream) {
    DICOW_REQUIRE(a && a->h_in && a->rows > 0 && a->D > 0, "fddt_ln_fwd: null/empty input");
    DICOW_REQUIRE(a->D % 4 == 0 && a->D <= 4096, "fddt_ln_fwd: D=%d must be a multiple of 4 and <= 4096", a->D);
    DICOW_REQUIRE(a->mode >= 0 && a->mode <= 2, "fddt_ln_fwd: bad mode %d", a->mode);
    DICOW_REQUIRE(a->mode == 0 || (a->stno && a->T > 0), "fddt_ln_fwd: mode %d needs stno and T", a->mode);
    DICOW_REQUIRE(a->pos == nullptr || a->T > 0, "fddt_ln_fwd: pos needs T");
    DICOW_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr), "fddt_ln_fwd: ln_w/ln_b must both be given");
    DICOW_REQUIRE(a->ln_w || a->h_out, "fddt_ln_fwd: nothing to write");
    const int R = 4;
    const int block = pick_block(a->D);
    int grid = dicow_cdiv(a->rows, R);
#ifdef DICOW_ABLATIONS
    static const int fwd_env = getenv("DICOW_ROW_FWD") ? atoi(getenv("DICOW_ROW_FWD")) : 0;      // 9: generic body (diagnostic builds only)
#else
    constexpr int fwd_env = 0;
#endif
    const bool staged = fwd_env != 9 && block <= 512 && block * 4 == a->D && a->mode == 1 && a->ln_w && !a->in_bf16 && a->h_out &&
                        a->y_bf16 && !a->y_f32 && a->mean && a->rstd && !a->pos && a->w[0] && a->w[1] && a->w[2] && a->w[3] &&
                        a->b[0] && a->b[1] && a->b[2] && a->b[3] && (int64_t)a->rows * a->D * 4 < (1ll << 31);
    if (staged && FLW_ON && fwd_env == 0 && a->D % 256 == 0 && a->D >= 512 && a->D <= 1280) {
        const int nc = a->D / 256;
        const int lds = 10 * a->D * 4;
        const void* fn = nc == 5 ? (const void*)fddt_ln_fwd_wave_kernel<5> : nc == 4 ? (const void*)fddt_ln_fwd_wave_kernel<4>
                       : nc == 3 ? (const void*)fddt_ln_fwd_wave_kernel<3> : (const void*)fddt_ln_fwd_wave_kernel<2>;
        static int occf[6] = {0};
        int per_cu = __atomic_load_n(&occf[nc], __ATOMIC_RELAXED);
        if (per_cu == 0) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FLW_WAVES * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            __atomic_store_n(&occf[nc], per_cu, __ATOMIC_RELAXED);
        }
        int gw = dicow_cdiv(a->rows, FLW_R * FLW_WAVES);
        if (gw > 256 * per_cu) gw = 256 * per_cu;
        switch (nc) {
            case 5: hipLaunchKernelGGL(fddt_ln_fwd_wave_kernel<5>, dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            case 4: hipLaunchKernelGGL(fddt_ln_fwd_wave_kernel<4>, dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            case 3: hipLaunchKernelGGL(fddt_ln_fwd_wave_kernel<3>, dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            default: hipLaunchKernelGGL(fddt_ln_fwd_wave_kernel<2>, dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
        }
        DICOW_CHECK_LAUNCH("fddt_ln_fwd_wave");
        return DICOW_OK;
    }
    if (staged) {
        static const bool attr = [] {
            (void)hipFuncSetAttribute((const void*)fddt_ln_fwd_staged_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 4 * 2048);
#ifdef DICOW_ABLATIONS
            (void)hipFuncSetAttribute((const void*)fddt_ln_fwd_staged_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 4 * 2048);
#endif
            return true; }();
        (void)attr;
        int cap = 256 * 2;                            // two resident workgroups per CU (a third one measured SLOWER: 90 vs 82 us)
#ifdef DICOW_ABLATIONS
        const int Rv = getenv("DICOW_ROW_R") ? atoi(getenv("DICOW_ROW_R")) : 4;
        if (getenv("DICOW_ROW_CAP")) cap = 256 * atoi(getenv("DICOW_ROW_CAP"));
        if (Rv != 4) {
            grid = dicow_cdiv(a->rows, Rv); if (grid > cap) grid = cap;
            hipLaunchKernelGGL((fddt_ln_fwd_staged_kernel<2>), dim3(grid), dim3(block), 2 * 2 * 4 * a->D, (hipStream_t)stream, *a);
            DICOW_CHECK_LAUNCH("fddt_ln_fwd_staged");
            return DICOW_OK;
        }
#endif
        if (grid > cap) grid = cap;
        hipLaunchKernelGGL((fddt_ln_fwd_staged_kernel<4>), dim3(grid), dim3(block), 2 * R * 4 * a->D, (hipStream_t)stream, *a);
        DICOW_CHECK_LAUNCH("fddt_ln_fwd_staged");
        return DICOW_OK;
    }
    // the encoder's initial FDDT(diag) + positions (bf16 rows in, fp32 rows out, no LayerNorm): the wave-per-row form with the eight
    // vectors in LDS (the column-owner body runs this shape at 1.8 TB/s: scalar FDDT arithmetic at two workgroups per CU)
    if (FLW_INIT_ON && a->mode == 1 && !a->ln_w && a->in_bf16 && a->h_out && a->pos && !a->y_bf16 && !a->y_f32 && a->D % 256 == 0 &&
        a->D >= 512 && a->D <= 1280 && a->w[0] && a->w[1] && a->w[2] && a->w[3] && a->b[0] && a->b[1] && a->b[2] && a->b[3]) {
        const int nc = a->D / 256;
        const int lds = 8 * a->D * 4;
        const void* fn = nc == 5 ? (const void*)fddt_ln_fwd_wave_kernel<5, true> : nc == 4 ? (const void*)fddt_ln_fwd_wave_kernel<4, true>
                       : nc == 3 ? (const void*)fddt_ln_fwd_wave_kernel<3, true> : (const void*)fddt_ln_fwd_wave_kernel<2, true>;
        static int occi[6] = {0};
        int per_cu = __atomic_load_n(&occi[nc], __ATOMIC_RELAXED);
        if (per_cu == 0) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, FLW_WAVES * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            __atomic_store_n(&occi[nc], per_cu, __ATOMIC_RELAXED);
        }
        int gw = dicow_cdiv(a->rows, FLW_R * FLW_WAVES);
        if (gw > 256 * per_cu) gw = 256 * per_cu;
        switch (nc) {
            case 5: hipLaunchKernelGGL((fddt_ln_fwd_wave_kernel<5, true>), dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            case 4: hipLaunchKernelGGL((fddt_ln_fwd_wave_kernel<4, true>), dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            case 3: hipLaunchKernelGGL((fddt_ln_fwd_wave_kernel<3, true>), dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
            default: hipLaunchKernelGGL((fddt_ln_fwd_wave_kernel<2, true>), dim3(gw), dim3(FLW_WAVES * 64), lds, (hipStream_t)stream, *a); break;
        }
        DICOW_CHECK_LAUNCH("fddt_fwd_wave_init");
        return DICOW_OK;
    }
    // LayerNorm only, fp32 rows of 256 NC columns: a wave per row (no barrier, no LDS)
    if (LNW_ON && a->mode == 0 && a->ln_w && !a->in_bf16 && !a->h_out && !a->pos && a->D % 256 == 0 && a->D >= 512 && a->D <= 1280 &&
        (a->y_bf16 || a->y_f32)) {
        static int occw[6] = {0};
        const int nc = a->D / 256;
        int per_cu = __atomic_load_n(&occw[nc], __ATOMIC_RELAXED);
        if (per_cu == 0) {
            const void* fn = nc == 5 ? (const void*)ln_fwd_wave_kernel<5> : nc == 4 ? (const void*)ln_fwd_wave_kernel<4>
                           : nc == 3 ? (const void*)ln_fwd_wave_kernel<3> : (const void*)ln_fwd_wave_kernel<2>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, LNW_WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            __atomic_store_n(&occw[nc], per_cu, __ATOMIC_RELAXED);
        }
        int gw = dicow_cdiv(a->rows, LNW_R * LNW_WAVES);
        const int capw = 256 * per_cu;
        if (gw > capw) gw = capw;
        switch (nc) {
            case 5: hipLaunchKernelGGL(ln_fwd_wave_kernel<5>, dim3(gw), dim3(LNW_WAVES * 64), 0, (hipStream_t)stream, *a); break;
            case 4: hipLaunchKernelGGL(ln_fwd_wave_kernel<4>, dim3(gw), dim3(LNW_WAVES * 64), 0, (hipStream_t)stream, *a); break;
            case 3: hipLaunchKernelGGL(ln_fwd_wave_kernel<3>, dim3(gw), dim3(LNW_WAVES * 64), 0, (hipStream_t)stream, *a); break;
            default: hipLaunchKernelGGL(ln_fwd_wave_kernel<2>, dim3(gw), dim3(LNW_WAVES * 64), 0, (hipStream_t)stream, *a); break;
        }
        DICOW_CHECK_LAUNCH("ln_fwd_wave");
        return DICOW_OK;
    }
    static int occ[4][17] = {{0}};
    const int variant = block > 512 ? 2 : (a->mode == 0 ? 1 : a->mode == 1 ? 3 : 0);
    const int cap = variant == 2 ? resident_grid(fddt_ln_fwd_kernel<R, 1024>, block, &occ[2][block / 64])
                  : variant == 1 ? resident_grid(fddt_ln_fwd_kernel<R, 512, 0>, block, &occ[1][block / 64])
                  : variant == 3 ? resident_grid(fddt_ln_fwd_kernel<R, 512, 1>, block, &occ[3][block / 64])
                                 : resident_grid(fddt_ln_fwd_kernel<R, 512>, block, &occ[0][block / 64]);
    if (grid > cap) grid = cap;
    if (variant == 2)
        hipLaunchKernelGGL((fddt_ln_fwd_kernel<R, 1024>), dim3(grid), dim3(block), 0, (hipStream_t)stream, *a);
    else if (variant == 1)
        hipLaunchKernelGGL((fddt_ln_fwd_kernel<R, 512, 0>), dim3(grid), dim3(block), 0, (hipStream_t)stream, *a);
    else if (variant == 3)
        hipLaunchKernelGGL((fddt_ln_fwd_kernel<R, 512, 1>), dim3(grid), dim3(block), 0, (hipStream_t)stream, *a);
    else
        hipLaunchKernelGGL((fddt_ln_fwd_kernel<R, 512>), dim3(grid), dim3(block), 0, (hipStream_t)stream, *a);
    DICOW_CHECK_LAUNCH("fddt_ln_fwd");
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ backward
#define F4_FMA(acc, a_, b_) do { acc.x += (a_).x * (b_).x; acc.y += (a_).y * (b_).y; acc.z += (a_).z * (b_).z; acc.w += (a_).w * (b_).w; } while (0)
#define F4_ADD(acc, a_) do { acc.x += (a_).x; acc.y += (a_).y; acc.z += (a_).z; acc.w += (a_).w; } while (0)

// MODE_T / LN_T: compile-time copies of mode / (a.ln_w != NULL) for the two shapes the encoder layers use (-1 = take
// them from the arguments).  The LayerNorm-only body drops the FDDT vectors and their 32 gradient accumulators (76 instead
// of 156 VGPRs): 4 instead of 2 resident workgroups per CU, and this kernel is latency-bound -- bytes in flight / memory
// latency -- so that doubled its rate (2.3 -> 4.7 TB/s).  The FDDT(diag)+LN shape stays on the generic body: moving its
// vectors or accumulators to LDS to gain a third workgroup measured slower (LDS read-modify-write traffic, spills).
template <int R, int MAXT, int MODE_T = -1, int LN_T = -1>
__global__ void __launch_bounds__(MAXT) fddt_ln_bwd_kernel(const dicow_fddt_ln_bwd_args a) {
    const int mode = MODE_T >= 0 ? MODE_T : a.mode;
    __shared__ float red[2][MAX_WAVES * 2 * R];
    const int tid = threadIdx.x, col = tid * 4, D = a.D;
    const bool act = col < D;
    const int nwaves = blockDim.x >> 6;
    const float4 one = make_float4(1, 1, 1, 1), zero = make_float4(0, 0, 0, 0);
    float4 w[4], b[4], lnw = one;
    int use = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        w[c] = (mode == 1 && a.w[c] && act) ? ld4(a.w[c] + col) : one;
        b[c] = (mode != 0 && a.b[c] && act) ? ld4(a.b[c] + col) : zero;
        if (a.b[c]) use |= 1 << c;
    }
    const bool do_ln = LN_T >= 0 ? (LN_T != 0) : (a.ln_w != nullptr);
    if (do_ln && act) lnw = ld4(a.ln_w + col);
    const float inv_d = 1.0f / (float)D;
    float4 acc_lnw = zero, acc_lnb = zero, acc_cs = zero, acc_dw[4], acc_db[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc_dw[c] = zero; acc_db[c] = zero; }

    int it = 0;
    for (int row0 = blockIdx.x * R; row0 < a.rows; row0 += gridDim.x * R, ++it) {
        float4 hin[R], xh[R], dy[R], gr[R], scw[R];
        float m[R][4], rs[R];
        float sums[2 * R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            const bool ok = act && row < a.rows;
            const int64_t off = (int64_t)row * D + col;
            hin[r] = zero; dy[r] = zero; rs[r] = 0.f;
            float mu = 0.f;
            if (ok) hin[r] = a.in_bf16 ? ld4_bf16(a.h_in, off) : ROWS_GB_NT ? ld4_nt(reinterpret_cast<const float*>(a.h_in) + off) : ld4(reinterpret_cast<const float*>(a.h_in) + off);
            if (ok && do_ln) dy[r] = a.dy_f32 ? ld4(reinterpret_cast<const float*>(a.d_y) + off) : ROWS_GB_NT ? ld4_bf16_nt(a.d_y, off) : ld4_bf16(a.d_y, off);
            gr[r] = (ok && a.g_res) ? (ROWS_GB_NT ? ld4_nt(a.g_res + off) : ld4(a.g_res + off)) : zero;      // issued with the other loads, ahead of the reduction
            if (do_ln && row < a.rows) { mu = a.mean[row]; rs[r] = a.rstd[row]; }
            if (mode != 0 && row < a.rows) {
                const int bi = row / a.T, t = row - bi * a.T;
                const float* mp = a.stno + (int64_t)bi * a.stno_bstride + t;
#pragma unroll
                for (int c = 0; c < 4; ++c)      // row-indexed, identical in every lane: keep it in a scalar register
                    m[r][c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp[(int64_t)c * a.T])));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) m[r][c] = 0.f;
            }
            // recompute x = FDDT(h_in) + pos.  Diagonal mode uses the collapsed form h * (sum_c m_c w_c) + sum_c m_c b_c
            // (20 flops per element instead of the forward's 60; it differs from the forward's evaluation order by an ulp
            // or two, which only perturbs x-hat inside the gradient formulas), and keeps sum_c m_c w_c for g0 below.
            float4 x = hin[r];
            if (mode == 1) {
                float4 sw = zero, sb = zero;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float mc = m[r][c];
                    const float4 wc = w[c], bc = b[c];
                    sw.x += mc * wc.x; sw.y += mc * wc.y; sw.z += mc * wc.z; sw.w += mc * wc.w;
                    sb.x += mc * bc.x; sb.y += mc * bc.y; sb.z += mc * bc.z; sb.w += mc * bc.w;
                }
                x = make_float4(x.x * sw.x + sb.x, x.y * sw.y + sb.y, x.z * sw.z + sb.z, x.w * sw.w + sb.w);
                scw[r] = sw;
            } else if (mode == 2) {
                const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
#define FB(e) fddt_bias_elem(x.e, b0.e, b1.e, b2.e, b3.e, m[r][0], m[r][1], m[r][2], m[r][3], use)
                F4_APPLY(x, FB);
#undef FB
            }
            if (a.pos && ok) {
                const int t = row % a.T;
                const float4 p = ld4(a.pos + (int64_t)t * D + col);
                x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
            }
            xh[r].x = (x.x - mu) * rs[r]; xh[r].y = (x.y - mu) * rs[r];
            xh[r].z = (x.z - mu) * rs[r]; xh[r].w = (x.w - mu) * rs[r];
            // dxhat = dy * gamma
            const float4 dxh = make_float4(dy[r].x * lnw.x, dy[r].y * lnw.y, dy[r].z * lnw.z, dy[r].w * lnw.w);
            sums[2 * r] = ok ? (dxh.x + dxh.y) + (dxh.z + dxh.w) : 0.f;
            sums[2 * r + 1] = ok ? (dxh.x * xh[r].x + dxh.y * xh[r].y) + (dxh.z * xh[r].z + dxh.w * xh[r].w) : 0.f;
        }
        if (do_ln) block_sum<2 * R, ROWS_G_BWD != 0, (ROWS_G_BWD ? ROWS_G_BWD : 16)>(sums, red[it & 1], nwaves);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            if (!(act && row < a.rows)) continue;
            const int64_t off = (int64_t)row * D + col;
            float4 g = gr[r];
            if (do_ln) {
                const float c1 = sums[2 * r] * inv_d, c2 = sums[2 * r + 1] * inv_d;
                g.x += rs[r] * (dy[r].x * lnw.x - c1 - xh[r].x * c2);
                g.y += rs[r] * (dy[r].y * lnw.y - c1 - xh[r].y * c2);
                g.z += rs[r] * (dy[r].z * lnw.z - c1 - xh[r].z * c2);
                g.w += rs[r] * (dy[r].w * lnw.w - c1 - xh[r].w * c2);
                F4_FMA(acc_lnw, dy[r], xh[r]);
                F4_ADD(acc_lnb, dy[r]);
            }
            float4 g0 = g;
            if (mode == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float mc = m[r][c];
                    const float4 mg = make_float4(mc * g.x, mc * g.y, mc * g.z, mc * g.w);
                    F4_FMA(acc_dw[c], mg, hin[r]);
                    F4_ADD(acc_db[c], mg);
                }
                g0 = make_float4(g.x * scw[r].x, g.y * scw[r].y, g.z * scw[r].z, g.w * scw[r].w);
            } else if (mode == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float mc = m[r][c];
                    const float4 mg = make_float4(mc * g.x, mc * g.y, mc * g.z, mc * g.w);
                    F4_ADD(acc_db[c], mg);
                }
            }
            F4_ADD(acc_cs, g0);
            if (a.g_out) st4(a.g_out + off, g0);
            if (a.g_out_bf16) st4_bf16(a.g_out_bf16, off, g0);
        }
    }
    if (!act) return;
    // per-workgroup partial column sums -> workspace [block][11][D]; reduced by dicow_launch_reduce_parts (no atomics)
    float* part = reinterpret_cast<float*>(a.ws) + (int64_t)blockIdx.x * 11 * D + col;
    if (do_ln && a.dln_w) st4(part + 0 * D, acc_lnw);
    if (do_ln && a.dln_b) st4(part + 1 * D, acc_lnb);
    if (a.colsum_out) st4(part + 2 * D, acc_cs);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (mode == 1 && a.dw[c]) st4(part + (3 + c) * D, acc_dw[c]);
        if (mode != 0 && a.db[c]) st4(part + (7 + c) * D, acc_db[c]);
    }
}

// ------------------------------------------------------------------------------------------------ backward, LDS-staged rows
// The FDDT(diag)+LayerNorm backward of every encoder layer (fp32 h_in, bf16 d_y, fp32 g_res; mode 1, no pos).  Same column-
// owner arithmetic as fddt_ln_bwd_kernel, but the rows of trip t+1 are fetched by LDS-DMA (buffer_load ... lds) while trip
// t is computed: the bytes in flight no longer live in VGPRs, which is what capped the generic body at two workgroups of
// 2 rows per CU (2.0 TB/s, latency-bound).  Every lane reads back exactly the LDS bytes its own wave's DMA wrote (h_in /
// g_res: its own 16 B; d_y: the wave's 512 B segment, fetched by lanes 0-31), so the wave's counted vmcnt wait is the
// only synchronisation the staging needs -- no extra barrier.  Stores go through a buffer descriptor too (rows past the
// end are dropped by num_records), so the number of VMEM operations per trip is fixed and the wait can be counted.
typedef __attribute__((address_space(3))) void lds_void_t;
template <int R, bool OUT_BF16>
__global__ void __launch_bounds__(512) fddt_ln_bwd_staged_kernel(const dicow_fddt_ln_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) char stg[];       // [2 or 3 stages][R rows][10*D bytes]: h_in | g_res | d_y
    __shared__ float red[2][MAX_WAVES * 2 * R];
    const int tid = threadIdx.x, lane = tid & 63, col = tid * 4, D = a.D;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const float4 zero = make_float4(0, 0, 0, 0);
    float4 w[4], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { w[c] = ld4(a.w[c] + col); b[c] = ld4(a.b[c] + col); }
    const float4 lnw = ld4(a.ln_w + col);
    const float inv_d = 1.0f / (float)D;
    float4 acc_lnw = zero, acc_lnb = zero, acc_cs = zero, acc_dw[4], acc_db[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc_dw[c] = zero; acc_db[c] = zero; }

    const unsigned nb32 = (unsigned)((int64_t)a.rows * D * 4), nb16 = (unsigned)((int64_t)a.rows * D * 2);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.h_in), 0, nb32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g_res), 0, nb32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.d_y), 0, nb16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(a.g_out, 0, nb32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(OUT_BF16 ? a.g_out_bf16 : (void*)a.g_out, 0, OUT_BF16 ? nb16 : nb32, 0x00020000);
    const unsigned vo32 = (unsigned)(col * 4);                       // this lane's 16 B of an fp32 row
    const unsigned voY = (unsigned)(wave * 512 + (lane & 31) * 16);  // lanes 0-31: the wave's 512 B of a bf16 row
    const int row_lds = 10 * D;
    auto stage_rows = [&](int row0, int s) {                         // 3 DMA instructions per row and wave
#pragma unroll
        for (int r = 0; r < R; ++r) {
            char* base = stg + (s * R + r) * row_lds;
            const int so32 = (row0 + r) * D * 4, so16 = (row0 + r) * D * 2;
            // (assembly-issued, common.h dicow_dma16: the compiler's own LDS-DMA made it wait for the NEWEST request at every trip's first LDS read)
            dicow_dma16<ROWS_BWD_H_NT>((unsigned)(uintptr_t)(base + wave * 1024), rsH, vo32, (unsigned)so32);
            dicow_dma16<ROWS_BWD_G_NT>((unsigned)(uintptr_t)(base + 4 * D + wave * 1024), rsG, vo32, (unsigned)so32);
            if (lane < 32 && (int)voY < 2 * D)       // (a last, partly filled wave: only the lanes inside the row)
                dicow_dma16<ROWS_BWD_Y_NT>((unsigned)(uintptr_t)(base + 8 * D + wave * 512), rsY, voY, (unsigned)so16);
        }
    };
    // per-row scalars (mean, rstd, 4 STNO masks) of a trip are ordinary loads: they are requested one trip ahead, BEFORE
    // that trip's DMA, so that waiting for them never drags a younger DMA along (vmcnt retires in order)
    float sc_n[R][6];
    auto load_scalars = [&](int row0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = row0 + r; row = row < a.rows ? row : a.rows - 1;
            const int bi = row / a.T, t = row - bi * a.T;
            const float* mp = a.stno + (int64_t)bi * a.stno_bstride + t;
            sc_n[r][0] = a.mean[row]; sc_n[r][1] = a.rstd[row];
#pragma unroll
            for (int c = 0; c < 4; ++c) sc_n[r][2 + c] = mp[(int64_t)c * a.T];
        }
    };
    const int stride = gridDim.x * R;
    int row0 = blockIdx.x * R, it = 0;
#if BWD_DEPTH == 2
    // Three LDS stages, the row DMA runs TWO trips ahead (the per-row scalars one): with one trip in flight a CU holds 38 KB
    // of requests, ~19 GB/s per CU at the loaded HBM latency -- 4.9 TB/s chip-wide at best, 3.9 measured; two trips hold 77 KB.
    if (row0 < a.rows) { load_scalars(row0); stage_rows(row0, 0); stage_rows(row0 + stride, 1); }
    int sg = 0;                                                      // LDS stage of the trip being computed
#else
    if (row0 < a.rows) { load_scalars(row0); stage_rows(row0, 0); }
#endif
    // Store data lives in these quads until the NEXT trip's reduction is over: rewriting the registers of a queued 16-byte store
    // (the compiler reused its temporaries at once) corrupted ~1 % of g_out at D = 1280, exactly as in the staged forward.
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    u32x4_t ov[R];
    u32x2_t bv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ov[r] = u32x4_t{0, 0, 0, 0}; bv[r] = u32x2_t{0, 0}; }
    for (; row0 < a.rows; row0 += stride, ++it) {
#if BWD_DEPTH == 2
        const int s = sg;
        const int s2 = sg == 0 ? 2 : sg - 1;                         // (sg + 2) % 3: the stage trip t-1 has finished reading
        sg = sg == 2 ? 0 : sg + 1;
#else
        const int s = it & 1;
#endif
        float sc[R][6];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) sc[r][c] = sc_n[r][c];
        load_scalars(row0 + stride);
#if BWD_DEPTH == 2
        stage_rows(row0 + 2 * stride, s2);                           // (past the end: out-of-range rows read as zero)
        // issue order: ... S(t) D(t+1) | stores(t-1) | S(t+1) D(t+2) | this wait.  Needed: D(t) (older) and S(t); younger than
        // S(t): D(t+1) 3R, the previous trip's stores, S(t+1) 6R, D(t+2) 3R.  First trip: S(0) D(0) D(1) | S(1) D(2).
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(12 * R) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"((OUT_BF16 ? 14 : 13) * R) : "memory");
#else
        stage_rows(row0 + stride, s ^ 1);                            // (past the end: out-of-range rows read as zero)
        // younger than this trip's DMA: the previous trip's stores, the scalars and the DMA just issued -- fixed counts
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(9 * R) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "i"((OUT_BF16 ? 11 : 10) * R) : "memory");
#endif
        float4 hin[R], xh[R], dy[R], scw[R];
        float m[R][4], rs[R], sums[2 * R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            const bool rok = row < a.rows;
            const char* base = stg + (s * R + r) * row_lds;
            hin[r] = *reinterpret_cast<const float4*>(base + tid * 16);
            const uint2 uy = *reinterpret_cast<const uint2*>(base + 8 * D + tid * 8);
            dy[r] = make_float4(__uint_as_float(uy.x << 16), __uint_as_float(uy.x & 0xffff0000u),
                                __uint_as_float(uy.y << 16), __uint_as_float(uy.y & 0xffff0000u));
            const float mu = rok ? sc[r][0] : 0.f;
            rs[r] = rok ? sc[r][1] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)      // row-indexed, identical in every lane: scalar registers
                m[r][c] = rok ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sc[r][2 + c]))) : 0.f;
            float4 sw = zero, sb = zero;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float mc = m[r][c];
                sw.x += mc * w[c].x; sw.y += mc * w[c].y; sw.z += mc * w[c].z; sw.w += mc * w[c].w;
                sb.x += mc * b[c].x; sb.y += mc * b[c].y; sb.z += mc * b[c].z; sb.w += mc * b[c].w;
            }
            scw[r] = sw;
            const float4 x = make_float4(hin[r].x * sw.x + sb.x, hin[r].y * sw.y + sb.y, hin[r].z * sw.z + sb.z, hin[r].w * sw.w + sb.w);
            xh[r] = make_float4((x.x - mu) * rs[r], (x.y - mu) * rs[r], (x.z - mu) * rs[r], (x.w - mu) * rs[r]);
            const float4 dxh = make_float4(dy[r].x * lnw.x, dy[r].y * lnw.y, dy[r].z * lnw.z, dy[r].w * lnw.w);
            sums[2 * r] = (dxh.x + dxh.y) + (dxh.z + dxh.w);
            sums[2 * r + 1] = (dxh.x * xh[r].x + dxh.y * xh[r].y) + (dxh.z * xh[r].z + dxh.w * xh[r].w);
        }
        if (!(ROWS_ABL & 16)) block_sum<2 * R, ROWS_G_BWDS != 0, (ROWS_G_BWDS ? ROWS_G_BWDS : 16)>(sums, red[it & 1], nwaves);    // (160 -> 134 us)
#pragma unroll
        for (int r = 0; r < R; ++r) { asm volatile("" :: "v"(ov[r])); if (OUT_BF16) asm volatile("" :: "v"(bv[r])); }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const char* base = stg + (s * R + r) * row_lds;
            float4 g = *reinterpret_cast<const float4*>(base + 4 * D + tid * 16);
            const float c1 = sums[2 * r] * inv_d, c2 = sums[2 * r + 1] * inv_d;
            g.x += rs[r] * (dy[r].x * lnw.x - c1 - xh[r].x * c2);
            g.y += rs[r] * (dy[r].y * lnw.y - c1 - xh[r].y * c2);
            g.z += rs[r] * (dy[r].z * lnw.z - c1 - xh[r].z * c2);
            g.w += rs[r] * (dy[r].w * lnw.w - c1 - xh[r].w * c2);
            F4_FMA(acc_lnw, dy[r], xh[r]);
            F4_ADD(acc_lnb, dy[r]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float mc = m[r][c];
                const float4 mg = make_float4(mc * g.x, mc * g.y, mc * g.z, mc * g.w);
                F4_FMA(acc_dw[c], mg, hin[r]);
                F4_ADD(acc_db[c], mg);
            }
            const float4 g0 = make_float4(g.x * scw[r].x, g.y * scw[r].y, g.z * scw[r].z, g.w * scw[r].w);
            F4_ADD(acc_cs, g0);
            const int so32 = (row0 + r) * D * 4;
            ov[r] = u32x4_t{__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g0.z), __float_as_uint(g0.w)};
            __builtin_amdgcn_raw_buffer_store_b128(ov[r], rsO, vo32, so32, 0);
            if (OUT_BF16) {
                bv[r] = u32x2_t{pack_bf16x2(g0.x, g0.y), pack_bf16x2(g0.z, g0.w)};
                __builtin_amdgcn_raw_buffer_store_b64(bv[r], rsB, (unsigned)(col * 2), (row0 + r) * D * 2, 0);
            }
        }
    }
    // per-workgroup partial column sums -> workspace [block][11][D]
    float* part = reinterpret_cast<float*>(a.ws) + (int64_t)blockIdx.x * 11 * D + col;
    if (a.dln_w) st4(part + 0 * D, acc_lnw);
    if (a.dln_b) st4(part + 1 * D, acc_lnb);
    if (a.colsum_out) st4(part + 2 * D, acc_cs);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a.dw[c]) st4(part + (3 + c) * D, acc_dw[c]);
        if (a.db[c]) st4(part + (7 + c) * D, acc_db[c]);
    }
}

// ------------------------------------------------------------------------------------------------ backward, LayerNorm only, a wave per row
// LayerNorm backward + residual gradient of LN2 / the final LayerNorm / the decoder's LayerNorms (mode 0): like ln_fwd_wave_kernel a
// WAVE owns whole rows (lane l: the float4 at columns 4 l + 256 k), the two row sums are DPP reductions -- no barrier per trip -- and
// the three column sums (d ln_w, d ln_b, the bias gradient of the producing Linear) accumulate in 3 x 4 NC registers per lane.  They
// meet once, at the end, in LDS (wave after wave, in order: deterministic) and leave as this workgroup's row of the same
// [workgroup][11][D] workspace the column-owner bodies write.  Same arithmetic per element as fddt_ln_bwd_kernel.
#ifndef LBW_ON
#define LBW_ON 1
#endif
#ifndef LBW_WAVES
#define LBW_WAVES 8
#endif
#ifndef LBW_MINWG
#define LBW_MINWG 1
#endif
template <int NC>
__global__ void __launch_bounds__(LBW_WAVES * 64, LBW_MINWG) ln_bwd_wave_kernel(const dicow_fddt_ln_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) char csum[];       // [3][D] fp32
    const int lane = threadIdx.x & 63, D = a.D;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4 zero = make_float4(0, 0, 0, 0);
    float4 lw[NC], acc_w[NC], acc_b[NC], acc_c[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { lw[k] = ld4(a.ln_w + 4 * lane + 256 * k); acc_w[k] = zero; acc_b[k] = zero; acc_c[k] = zero; }
    const float inv_d = 1.0f / (float)D;
    const float* H = reinterpret_cast<const float*>(a.h_in);
    const int wave = blockIdx.x * LBW_WAVES + wv, nwaves = gridDim.x * LBW_WAVES;
    for (int row = wave; row < a.rows; row += nwaves) {
        float4 x[NC], dy[NC], g[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int64_t off = (int64_t)row * D + 4 * lane + 256 * k;
            x[k] = ld4(H + off);
            dy[k] = a.dy_f32 ? ld4(reinterpret_cast<const float*>(a.d_y) + off) : ld4_bf16(a.d_y, off);
            g[k] = a.g_res ? ld4(a.g_res + off) : zero;
        }
        const float mu = a.mean[row], rs = a.rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            x[k] = make_float4((x[k].x - mu) * rs, (x[k].y - mu) * rs, (x[k].z - mu) * rs, (x[k].w - mu) * rs);      // x-hat
            const float4 dxh = make_float4(dy[k].x * lw[k].x, dy[k].y * lw[k].y, dy[k].z * lw[k].z, dy[k].w * lw[k].w);
            s1 += (dxh.x + dxh.y) + (dxh.z + dxh.w);
            s2 += (dxh.x * x[k].x + dxh.y * x[k].y) + (dxh.z * x[k].z + dxh.w * x[k].w);
        }
        const float c1 = wave_sum_dpp(s1) * inv_d, c2 = wave_sum_dpp(s2) * inv_d;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            g[k].x += rs * (dy[k].x * lw[k].x - c1 - x[k].x * c2);
            g[k].y += rs * (dy[k].y * lw[k].y - c1 - x[k].y * c2);
            g[k].z += rs * (dy[k].z * lw[k].z - c1 - x[k].z * c2);
            g[k].w += rs * (dy[k].w * lw[k].w - c1 - x[k].w * c2);
            F4_FMA(acc_w[k], dy[k], x[k]);
            F4_ADD(acc_b[k], dy[k]);
            F4_ADD(acc_c[k], g[k]);
            const int64_t off = (int64_t)row * D + 4 * lane + 256 * k;
            if (a.g_out) {                                     // (8-byte stores: see the note on 16-byte stores in the staged kernels)
                *reinterpret_cast<u32x2r_t*>(a.g_out + off) = u32x2r_t{__float_as_uint(g[k].x), __float_as_uint(g[k].y)};
                *reinterpret_cast<u32x2r_t*>(a.g_out + off + 2) = u32x2r_t{__float_as_uint(g[k].z), __float_as_uint(g[k].w)};
            }
            if (a.g_out_bf16) st4_bf16(a.g_out_bf16, off, g[k]);
        }
    }
    // column sums: wave after wave into LDS, then this workgroup's row of the workspace
    float* cs = reinterpret_cast<float*>(csum);
    for (int w = 0; w < LBW_WAVES; ++w) {
        if (wv == w) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c = 4 * lane + 256 * k;
                float4 t0 = acc_w[k], t1 = acc_b[k], t2 = acc_c[k];
                if (w > 0) {
                    const float4 o0 = *reinterpret_cast<const float4*>(cs + c), o1 = *reinterpret_cast<const float4*>(cs + D + c),
                                 o2 = *reinterpret_cast<const float4*>(cs + 2 * D + c);
                    t0 = make_float4(o0.x + t0.x, o0.y + t0.y, o0.z + t0.z, o0.w + t0.w);
                    t1 = make_float4(o1.x + t1.x, o1.y + t1.y, o1.z + t1.z, o1.w + t1.w);
                    t2 = make_float4(o2.x + t2.x, o2.y + t2.y, o2.z + t2.z, o2.w + t2.w);
                }
                *reinterpret_cast<float4*>(cs + c) = t0; *reinterpret_cast<float4*>(cs + D + c) = t1; *reinterpret_cast<float4*>(cs + 2 * D + c) = t2;
            }
        }
        __syncthreads();
    }
    float* part = reinterpret_cast<float*>(a.ws) + (int64_t)blockIdx.x * 11 * D;
    for (int i = threadIdx.x * 4; i < 3 * D; i += LBW_WAVES * 64 * 4) {
        const int v = i / D, c = i - v * D;
        if ((v == 0 && a.dln_w) || (v == 1 && a.dln_b) || (v == 2 && a.colsum_out)) st4(part + (int64_t)v * D + c, *reinterpret_cast<const float4*>(cs + i));
    }
}

#ifndef BWD_DEPTH
#define BWD_DEPTH 1       // row DMA trips in flight in the staged backward (1: two LDS stages, 2: three -- measured equal: 124-126 us
                          // either way, profiles/r03_rows_depth.txt; the barrier of the per-trip block reduction is the limit, not the DMA latency)
#endif
#ifndef BWD_R0
#define BWD_R0 2          // rows per trip and resident workgroups per CU of the LayerNorm-only (mode 0) body
#endif
#ifndef BWD_CU0
#define BWD_CU0 4
#endif
#ifndef BWD_RS
#define BWD_RS 3          // rows per trip / resident workgroups per CU of the LDS-staged FDDT(diag)+LN body: 3 rows x 1 workgroup
#endif                    // measured 126 us, 2 x 2: 137, 4 x 1: 128, 3 x 2: 131, 2 x 1: 129 (one block reduction per trip: 32 us of it)
#ifndef BWD_CUS
#define BWD_CUS 1
#endif
static int bwd_grid(int rows, int D, int per_cu) {
    const int block = ((D / 4) + 63) / 64 * 64;
    int grid = (rows + 3) / 4;
    const int cap = 256 * (block <= 256 ? 4 : per_cu);
    return grid > cap ? cap : grid;
}

extern "C" int64_t dicow_fddt_ln_bwd_ws_bytes(int rows, int D) { return (int64_t)bwd_grid(rows, D, BWD_CU0 > BWD_CUS ? BWD_CU0 : BWD_CUS) * 11 * D * 4; }

extern "C" int dicow_fddt_ln_bwd(const dicow_fddt_ln_bwd_args* a, void* stream) {
    DICOW_REQUIRE(a && a->h_in && a->rows > 0 && a->D > 0, "fddt_ln_bwd: null/empty input");
    DICOW_REQUIRE(a->D % 4 == 0 && a->D <= 4096, "fddt_ln_bwd: D=%d must be a multiple of 4 and <= 4096", a->D);
    DICOW_REQUIRE(a->mode >= 0 && a->mode <= 2, "fddt_ln_bwd: bad mode %d", a->mode);
    DICOW_REQUIRE(a->mode == 0 || (a->stno && a->T > 0), "fddt_ln_bwd: mode %d needs stno and T", a->mode);
    DICOW_REQUIRE(a->ln_w == nullptr || (a->mean && a->rstd && a->d_y), "fddt_ln_bwd: LayerNorm needs mean/rstd/d_y");
    DICOW_REQUIRE(a->ln_w || a->g_res, "fddt_ln_bwd: no incoming gradient");
    const int block = pick_block(a->D);
#ifdef DICOW_ABLATIONS
    static const int r_env = getenv("DICOW_ROW_R") ? atoi(getenv("DICOW_ROW_R")) : 0;       // diagnostic builds only: 0 = auto, 9 = generic body
#else
    constexpr int r_env = 0;
#endif
    const bool ln0 = block <= 512 && r_env != 9 && a->mode == 0 && a->ln_w;
    // LDS-staged body: the encoder-layer shape (every FDDT vector present, fp32 in, bf16 d_y, residual gradient, no pos)
    const bool staged = block <= 512 && block * 4 == a->D && a->D % 8 == 0 && r_env != 9 && r_env != 8 && a->mode == 1 && a->ln_w && !a->in_bf16 &&
                        !a->dy_f32 && a->g_res && a->g_out && !a->pos && a->w[0] && a->w[1] && a->w[2] && a->w[3] &&
                        a->b[0] && a->b[1] && a->b[2] && a->b[3] && (int64_t)a->rows * a->D * 4 < (1ll << 31) &&
                        (BWD_DEPTH + 1) * BWD_RS * 10 * a->D <= 150 * 1024;       // the row stages must fit the CU's 160 KiB of LDS
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)fddt_ln_bwd_staged_kernel<BWD_RS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)fddt_ln_bwd_staged_kernel<BWD_RS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        return true; }();
    (void)attr_set;
    const int grid = bwd_grid(a->rows, a->D, ln0 ? BWD_CU0 : staged ? BWD_CUS : 2);
    const int D = a->D;
    float* outs[11] = {a->ln_w ? a->dln_w : nullptr, a->ln_w ? a->dln_b : nullptr, a->colsum_out,
                       a->mode == 1 ? a->dw[0] : nullptr, a->mode == 1 ? a->dw[1] : nullptr, a->mode == 1 ? a->dw[2] : nullptr,
                       a->mode == 1 ? a->dw[3] : nullptr, a->mode != 0 ? a->db[0] : nullptr, a->mode != 0 ? a->db[1] : nullptr,
                       a->mode != 0 ? a->db[2] : nullptr, a->mode != 0 ? a->db[3] : nullptr};
    bool any = false;
    for (int k = 0; k < 11; ++k) any = any || outs[k];
    DICOW_REQUIRE(!any || (a->ws && a->ws_bytes >= (int64_t)grid * 11 * D * 4),
                  "fddt_ln_bwd: workspace too small (need %ld bytes)", (long)grid * 11 * D * 4);
    hipStream_t st = (hipStream_t)stream;
    if (LBW_ON && ln0 && r_env == 0 && !a->in_bf16 && !a->pos && a->D % 256 == 0 && a->D >= 512 && a->D <= 1280 && (a->g_out || a->g_out_bf16)) {
        const int nc = a->D / 256;
        const void* fn = nc == 5 ? (const void*)ln_bwd_wave_kernel<5> : nc == 4 ? (const void*)ln_bwd_wave_kernel<4>
                       : nc == 3 ? (const void*)ln_bwd_wave_kernel<3> : (const void*)ln_bwd_wave_kernel<2>;
        static int occb[6] = {0};
        int per_cu = __atomic_load_n(&occb[nc], __ATOMIC_RELAXED);
        const int lds = 3 * D * 4;
        if (per_cu == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, LBW_WAVES * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            __atomic_store_n(&occb[nc], per_cu, __ATOMIC_RELAXED);
        }
        int gw = dicow_cdiv(a->rows, LBW_WAVES);
        if (gw > 256 * per_cu) gw = 256 * per_cu;
        if (gw > grid) gw = grid;                                // (the workspace is sized for `grid` partial rows)
        switch (nc) {
            case 5: hipLaunchKernelGGL(ln_bwd_wave_kernel<5>, dim3(gw), dim3(LBW_WAVES * 64), lds, st, *a); break;
            case 4: hipLaunchKernelGGL(ln_bwd_wave_kernel<4>, dim3(gw), dim3(LBW_WAVES * 64), lds, st, *a); break;
            case 3: hipLaunchKernelGGL(ln_bwd_wave_kernel<3>, dim3(gw), dim3(LBW_WAVES * 64), lds, st, *a); break;
            default: hipLaunchKernelGGL(ln_bwd_wave_kernel<2>, dim3(gw), dim3(LBW_WAVES * 64), lds, st, *a); break;
        }
        DICOW_CHECK_LAUNCH("ln_bwd_wave");
        if (any) return dicow_launch_reduce_multi(reinterpret_cast<const float*>(a->ws), gw, (int64_t)11 * D, D, outs, 11, D, st);
        return DICOW_OK;
    }
    if (block > 512)
        hipLaunchKernelGGL((fddt_ln_bwd_kernel<2, 1024>), dim3(grid), dim3(block), 0, st, *a);
    else if (r_env == 9)                                                                    // generic body (ablation)
        hipLaunchKernelGGL((fddt_ln_bwd_kernel<2, 512>), dim3(grid), dim3(block), 0, st, *a);
    else if (staged && a->g_out_bf16)
        hipLaunchKernelGGL((fddt_ln_bwd_staged_kernel<BWD_RS, true>), dim3(grid), dim3(block), (BWD_DEPTH + 1) * BWD_RS * 10 * a->D, st, *a);
    else if (staged)
        hipLaunchKernelGGL((fddt_ln_bwd_staged_kernel<BWD_RS, false>), dim3(grid), dim3(block), (BWD_DEPTH + 1) * BWD_RS * 10 * a->D, st, *a);
    else if (ln0)
        hipLaunchKernelGGL((fddt_ln_bwd_kernel<BWD_R0, 512, 0, 1>), dim3(grid), dim3(block), 0, st, *a);
    else
        hipLaunchKernelGGL((fddt_ln_bwd_kernel<2, 512>), dim3(grid), dim3(block), 0, st, *a);
    DICOW_CHECK_LAUNCH("fddt_ln_bwd");
    if (any) return dicow_launch_reduce_multi(reinterpret_cast<const float*>(a->ws), grid, (int64_t)11 * D, D, outs, 11, D,
                                              (hipStream_t)stream);
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ full (DxD) FDDT combine
__global__ void fddt_full_combine_fwd_kernel(const unsigned short* y4, const void* h_in, int in_bf16, const float* stno,
                                             int64_t bstride, int use_mask, float* h_out, int rows, int T, int D) {
    const int64_t n4 = (int64_t)rows * D / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const int row = (int)(e / D), col = (int)(e - (int64_t)row * D);
        const int bi = row / T, t = row - bi * T;
        const float* mp = stno + (int64_t)bi * bstride + t;
        const float4 h = in_bf16 ? ld4_bf16(h_in, e) : ld4(reinterpret_cast<const float*>(h_in) + e);
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float mc = mp[(int64_t)c * T];
            const float4 v = (use_mask >> c) & 1 ? ld4_bf16(y4, (int64_t)row * 4 * D + (int64_t)c * D + col) : h;
            acc.x += v.x * mc; acc.y += v.y * mc; acc.z += v.z * mc; acc.w += v.w * mc;
        }
        st4(h_out + e, acc);
    }
}

__global__ void fddt_full_combine_bwd_kernel(const float* g, const float* stno, int64_t bstride, int use_mask,
                                             unsigned short* d_y4, float* dh_direct, int rows, int T, int D) {
    const int64_t n4 = (int64_t)rows * D / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4;
        const int row = (int)(e / D), col = (int)(e - (int64_t)row * D);
        const int bi = row / T, t = row - bi * T;
        const float* mp = stno + (int64_t)bi * bstride + t;
        const float4 gv = ld4(g + e);
        float md = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float mc = mp[(int64_t)c * T];
            if ((use_mask >> c) & 1) {
                st4_bf16(d_y4, (int64_t)row * 4 * D + (int64_t)c * D + col,
                         make_float4(gv.x * mc, gv.y * mc, gv.z * mc, gv.w * mc));
            } else {
                md += mc;
            }
        }
        if (dh_direct) st4(dh_direct + e, make_float4(gv.x * md, gv.y * md, gv.z * md, gv.w * md));
    }
}

extern "C" int dicow_fddt_full_combine_fwd(const void* y4, const void* h_in, int in_bf16, const float* stno,
                                           int64_t stno_bstride, int use_mask, float* h_out, int rows, int T, int D,
                                           void* stream) {
    DICOW_REQUIRE(y4 && h_in && stno && h_out && rows > 0 && D % 4 == 0 && T > 0, "fddt_full_combine_fwd: bad args");
    const int64_t n4 = (int64_t)rows * D / 4;
    int grid = (int)((n4 + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fddt_full_combine_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)y4, h_in, in_bf16, stno, stno_bstride, use_mask, h_out, rows, T, D);
    DICOW_CHECK_LAUNCH("fddt_full_combine_fwd");
    return DICOW_OK;
}

extern "C" int dicow_fddt_full_combine_bwd(const float* g, const float* stno, int64_t stno_bstride, int use_mask,
                                           void* d_y4, float* dh_direct, int rows, int T, int D, void* stream) {
    DICOW_REQUIRE(g && stno && d_y4 && rows > 0 && D % 4 == 0 && T > 0, "fddt_full_combine_bwd: bad args");
    const int64_t n4 = (int64_t)rows * D / 4;
    int grid = (int)((n4 + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fddt_full_combine_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, stno,
                       stno_bstride, use_mask, (unsigned short*)d_y4, dh_direct, rows, T, D);
    DICOW_CHECK_LAUNCH("fddt_full_combine_bwd");
    return DICOW_OK;
}
