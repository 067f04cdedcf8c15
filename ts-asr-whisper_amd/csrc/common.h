// Shared device/host helpers for libdicow_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define DICOW_EXPERIMENTAL_ABI 1      // the sources name the experimental flags / struct tail; the DEFINITIONS are under DICOW_EXPERIMENTS
#include "../../include/dicow_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA operand (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define DICOW_WAVE 64

// ---- error plumbing (no exceptions across the C ABI)
void dicow_set_error(const char* fmt, ...);
#define DICOW_FAIL(code, ...) do { dicow_set_error(__VA_ARGS__); return (code); } while (0)
#define DICOW_REQUIRE(cond, ...) do { if (!(cond)) DICOW_FAIL(DICOW_ERR_INVALID, __VA_ARGS__); } while (0)
#define DICOW_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) DICOW_FAIL(DICOW_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); } while (0)

static inline int dicow_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 (round-to-nearest-even; hipcc lowers the casts to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }
__device__ __forceinline__ float bfbits2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bfbits(float x) {
    bf16_t b = (bf16_t)x;
    return *reinterpret_cast<unsigned short*>(&b);
}
// ONE v_cvt_pk_bf16_f32 (two scalar casts + shift + or compile to four instructions, and plain VALU instructions are paid
// in matrix-pipe time: they share the SIMD's issue port with the MFMAs)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, b);
}

// ---- LDS-DMA the compiler must not see (round 6).  hipcc tracks every `buffer_load ... lds` IT emits and puts s_waitcnt vmcnt(0) in front of the
// next typed LDS load it cannot tell apart from the DMA's destination -- in practice every one: a kernel that requests rows / tiles
// several trips ahead and orders them with its own counted waits then waits for ALL of them, the newest included, at the first LDS
// read of every trip (seen in the ISA of attn_bwd_dkv_kernel and of both staged row kernels: their prefetch never ran ahead).  Issued
// from inline assembly the DMA is invisible to that pass; the kernel's own s_waitcnt vmcnt(N) + barriers order it.  m0 = wave-uniform
// LDS byte address (nothing else in these kernels uses m0).  AUX: the builtin's cache-policy bits (0 default, 2 = nt, 17 = sc0 sc1).
template <int AUX>
__device__ __forceinline__ void dicow_dma16(unsigned lds_addr, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    static_assert(AUX == 0 || AUX == 2 || AUX == 17, "dicow_dma16: cache policy 0, 2 (nt) or 17 (sc0 sc1)");
    if constexpr (AUX == 17) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else if constexpr (AUX == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// exact (erf) GELU and its derivative (Whisper activation_function="gelu").  Every kernel evaluates the SAME expression (this
// scalar form and the packed stage-major gelu_cdf_pdf_p below are operation-for-operation identical), so the training
// forward, the inference forward and the decoder step produce identical activations.
//   Round 3 form (DICOW_GELU_V 1): Phi(x) = 1 / (1 + 2^(x q(x^2))), q = degree-6 fit of -log2(e) logit(Phi(x)) / x -- see
//   gelu_cdf_pdf_p.  |d gelu| < 5e-7 over all bf16 inputs.
//   Round 1/2 form (DICOW_GELU_V 0): Abramowitz-Stegun 7.1.26 erfc, 0.5 erfc(|x|/sqrt2) = t poly4(t) exp(-x^2/2) / 2,
//   t = 1/(1 + p|x|/sqrt2); |error| < 5e-7.
#define GELU_Q0 -2.30220720357529451e+00f
#define GELU_Q1 -1.04838581318361504e-01f
#define GELU_Q2 9.55929831375757871e-05f
#define GELU_Q3 1.59389397366708184e-04f
#define GELU_Q4 -1.14524280114105203e-05f
#define GELU_Q5 3.85044882878601587e-07f
#define GELU_Q6 -5.20990630162781774e-09f
#ifndef DICOW_GELU_V
#define DICOW_GELU_V 1
#endif
__device__ __forceinline__ void gelu_cdf_pdf(float x, float& cdf, float& pdf) {
#if DICOW_GELU_V == 1
    const float s = x * x;
    float q = fmaf(s, GELU_Q6, GELU_Q5);
    q = fmaf(q, s, GELU_Q4);
    q = fmaf(q, s, GELU_Q3);
    q = fmaf(q, s, GELU_Q2);
    q = fmaf(q, s, GELU_Q1);
    q = fmaf(q, s, GELU_Q0);
    cdf = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(q * x) + 1.0f);
    pdf = __builtin_amdgcn_exp2f(s * -0.72134752044448170f) * 0.3989422804014327f;
#else
    // z = |x| sqrt(log2(e)/2): exp(-x^2/2) = 2^(-z^2) needs no further scaling, t = 1/(1 + p |x|/sqrt2) = 1/(1 + p' z), and the
    // 0.5 of 0.5*erfc is folded into the polynomial coefficients (two multiplies fewer per element than the textbook form)
    const float z = fabsf(x) * 0.84932180028801907f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.27273748087922245f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-(z * z));                                // exp(-x^2/2)
    float p = fmaf(0.5307027145f, t, -0.7265760135f);
    p = fmaf(p, t, 0.7107068705f);
    p = fmaf(p, t, -0.142248368f);
    p = fmaf(p, t, 0.127414796f);
    const float h = p * t * e;                                                       // Phi(-|x|)
    cdf = x >= 0.f ? 1.0f - h : h;
    pdf = e * 0.3989422804014327f;
#endif
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, pdf;
    gelu_cdf_pdf(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, pdf;
    gelu_cdf_pdf(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& dg) {
    float cdf, pdf;
    gelu_cdf_pdf(x, cdf, pdf);
    g = x * cdf;
    dg = fmaf(x, pdf, cdf);
}

// N elements at once, STAGE-MAJOR and PACKED.  The GEMM epilogues run ONE wave per SIMD, and a SIMD gets one issue slot every
// four cycles: a lone wave retires at most one instruction per 4 cycles whatever its type, so the epilogue's cost is its
// INSTRUCTION COUNT x 4 cycles (measured: 26 scalar VALU instructions per element x 256 elements per lane = 15.6 us per
// 256 x 256 tile at 1.7 GHz, against 3.6 us for the same epilogue without the GELU).  Hence
//   * packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two elements per issue slot) wherever the formula
//     allows it -- the transcendental, sign and conversion steps stay per element;
//   * every step of the formula applied to all N values before the next step starts, so that N independent dependency
//     chains sit side by side (element by element, hipcc emits one ~12-deep chain of dependent operations after the other
//     and the wave also idles for the VALU latency between them).  Empty asm statements naming a stage's values as in/out
//     operands pin that order: left alone hipcc re-serialises the chains to save registers, and
//     __builtin_amdgcn_sched_barrier does not hold pure arithmetic in place.
template <int NP>
__device__ __forceinline__ void stage_fence2(f32x2_t (&a)[NP]) {
    static_assert(NP % 4 == 0, "stage_fence2: multiples of 4 pairs");
#pragma unroll
    for (int i = 0; i < NP; i += 4) asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]));
}
// x: NP pairs -> cdf = Phi(x), pdf = phi(x)
#if DICOW_GELU_V == 0
// (round-1/2 form: same erfc expression and coefficients as the scalar A-S 7.1.26 version; kept for A/B builds)
template <int NP, bool WANT_PDF>
__device__ __forceinline__ void gelu_cdf_pdf_p(const f32x2_t (&x)[NP], f32x2_t (&cdf)[NP], f32x2_t (&pdf)[NP]) {
    f32x2_t z[NP], t[NP], e[NP], p[NP];
    const f32x2_t one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
#pragma unroll
    for (int i = 0; i < NP; ++i) z[i] = __builtin_elementwise_abs(x[i]) * 0.84932180028801907f;
    stage_fence2(z);
#pragma unroll
    for (int i = 0; i < NP; ++i) { t[i] = __builtin_elementwise_fma(z[i], f32x2_t{0.27273748087922245f, 0.27273748087922245f}, one); e[i] = -(z[i] * z[i]); }
    stage_fence2(t); stage_fence2(e);
#pragma unroll
    for (int i = 0; i < NP; ++i) { t[i].x = __builtin_amdgcn_rcpf(t[i].x); t[i].y = __builtin_amdgcn_rcpf(t[i].y);
                                    e[i].x = __builtin_amdgcn_exp2f(e[i].x); e[i].y = __builtin_amdgcn_exp2f(e[i].y); }
    stage_fence2(t); stage_fence2(e);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(t[i], f32x2_t{0.5307027145f, 0.5307027145f}, f32x2_t{-0.7265760135f, -0.7265760135f});
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2_t{0.7107068705f, 0.7107068705f});
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2_t{-0.142248368f, -0.142248368f});
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2_t{0.127414796f, 0.127414796f});
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = p[i] * t[i];
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = half - p[i] * e[i];                          // 0.5 - Phi(-|x|) >= 0
    stage_fence2(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) cdf[i] = half + __builtin_elementwise_copysign(p[i], x[i]);    // Phi(x) = 0.5 + sign(x) (0.5 - Phi(-|x|))
    stage_fence2(cdf);
    if (WANT_PDF) {
#pragma unroll
        for (int i = 0; i < NP; ++i) pdf[i] = e[i] * 0.3989422804014327f;
        stage_fence2(pdf);
    }
}
#elif DICOW_GELU_V == 1
// Round 3: Phi(x) = 1 / (1 + 2^(x q(x^2))), q = the degree-6 minimax fit (in x^2, over |x| <= 6) of -log2(e) logit(Phi(x)) / x,
// weighted by the gelu error it causes.  The logit of the normal cdf is odd, smooth and nearly cubic, so seven coefficients give
// |d gelu| < 5e-7 over EVERY bf16 input and a RELATIVE error below 3e-4 down to gelu(-4) = -1.3e-4 (tools/gelu_fit.py:
// exhaustive over the 65280 finite bf16 values in fp32 emulation; the bf16-rounded activation equals the rounded float64 one
// wherever |gelu| > 1e-4), and the leading coefficient is negative, so the form saturates by itself: x q -> -+inf,
// 2^. -> 0 / inf, 1/(1 + .) -> 1 / 0 -- no clamp, no sign handling, no NaN for any finite or infinite input.  Per pair:
// 9 packed full-rate instructions + 2 v_exp + 2 v_rcp (erfc form: ~15 + 4), all plain dependency chains that the
// stage-major order interleaves.
template <int NP, bool WANT_PDF>
__device__ __forceinline__ void gelu_cdf_pdf_p(const f32x2_t (&x)[NP], f32x2_t (&cdf)[NP], f32x2_t (&pdf)[NP]) {
    f32x2_t s[NP], q[NP];
    const f32x2_t one = {1.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < NP; ++i) s[i] = x[i] * x[i];
    stage_fence2(s);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(s[i], f32x2_t{GELU_Q6, GELU_Q6}, f32x2_t{GELU_Q5, GELU_Q5});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{GELU_Q4, GELU_Q4});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{GELU_Q3, GELU_Q3});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{GELU_Q2, GELU_Q2});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{GELU_Q1, GELU_Q1});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{GELU_Q0, GELU_Q0});
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] * x[i];
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) { q[i].x = __builtin_amdgcn_exp2f(q[i].x); q[i].y = __builtin_amdgcn_exp2f(q[i].y); }
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = q[i] + one;
    stage_fence2(q);
#pragma unroll
    for (int i = 0; i < NP; ++i) { cdf[i].x = __builtin_amdgcn_rcpf(q[i].x); cdf[i].y = __builtin_amdgcn_rcpf(q[i].y); }
    stage_fence2(cdf);
    if (WANT_PDF) {
#pragma unroll
        for (int i = 0; i < NP; ++i) s[i] = s[i] * -0.72134752044448170f;             // -x^2/2 * log2(e)
        stage_fence2(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) { s[i].x = __builtin_amdgcn_exp2f(s[i].x); s[i].y = __builtin_amdgcn_exp2f(s[i].y); }
        stage_fence2(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) pdf[i] = s[i] * 0.3989422804014327f;
        stage_fence2(pdf);
    }
}
#elif DICOW_GELU_V == 2
// A/B candidate without transcendentals: Phi(x) = 0.5 + xc P(xc^2), xc = clamp(x, -4, 4), P of degree 8 through P(16) = 1/8
template <int NP, bool WANT_PDF>
__device__ __forceinline__ void gelu_cdf_pdf_p(const f32x2_t (&x)[NP], f32x2_t (&cdf)[NP], f32x2_t (&pdf)[NP]) {
    f32x2_t s[NP], q[NP], xc[NP];
    const f32x2_t half = {0.5f, 0.5f};
#pragma unroll
    for (int i = 0; i < NP; ++i) { xc[i].x = __builtin_amdgcn_fmed3f(x[i].x, -4.0f, 4.0f); xc[i].y = __builtin_amdgcn_fmed3f(x[i].y, -4.0f, 4.0f); }
    stage_fence2(xc);
#pragma unroll
    for (int i = 0; i < NP; ++i) s[i] = xc[i] * xc[i];
    stage_fence2(s);
    const float c_[9] = {3.9907026948e-01f, -6.6837845791e-02f, 1.0242059735e-02f, -1.2744738650e-03f, 1.2776073518e-04f,
                         -9.6645953779e-06f, 4.9620894472e-07f, -1.4959220843e-08f, 1.9681844066e-10f};
#pragma unroll
    for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(s[i], f32x2_t{c_[8], c_[8]}, f32x2_t{c_[7], c_[7]});
    stage_fence2(q);
#pragma unroll
    for (int k = 6; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], s[i], f32x2_t{c_[k], c_[k]});
        stage_fence2(q);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) cdf[i] = __builtin_elementwise_fma(q[i], xc[i], half);
    stage_fence2(cdf);
    if (WANT_PDF) {
#pragma unroll
        for (int i = 0; i < NP; ++i) s[i] = (x[i] * x[i]) * -0.72134752044448170f;
        stage_fence2(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) { s[i].x = __builtin_amdgcn_exp2f(s[i].x); s[i].y = __builtin_amdgcn_exp2f(s[i].y); }
        stage_fence2(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) pdf[i] = s[i] * 0.3989422804014327f;
        stage_fence2(pdf);
    }
}
#else
// ablation: no activation arithmetic at all (cost of everything else in a GELU epilogue)
template <int NP, bool WANT_PDF>
__device__ __forceinline__ void gelu_cdf_pdf_p(const f32x2_t (&x)[NP], f32x2_t (&cdf)[NP], f32x2_t (&pdf)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) { cdf[i] = f32x2_t{0.5f, 0.5f}; pdf[i] = f32x2_t{0.25f, 0.25f}; }
}
#endif

// ---- wave / block reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sum with DPP (VALU lane shuffles, no LDS traffic): quad swaps, half-row / row mirrors, then the
// row-broadcast steps; the total lands in lane 63 and is returned wave-uniform through v_readlane.
// (__shfl_xor lowers to ds_bpermute_b32: ~6 dependent LDS round trips per reduction, measured 2x slower kernels.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    int x;
#define DPP_ADD(ctrl, rmask) \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false); v += __int_as_float(x);
    DPP_ADD(0xB1, 0xf)    // quad_perm [1,0,3,2]
    DPP_ADD(0x4E, 0xf)    // quad_perm [2,3,0,1]
    DPP_ADD(0x141, 0xf)   // row_half_mirror
    DPP_ADD(0x140, 0xf)   // row_mirror        -> every lane holds its 16-lane row sum
    DPP_ADD(0x142, 0xa)   // row_bcast15 into rows 1,3
    DPP_ADD(0x143, 0xc)   // row_bcast31 into rows 2,3 -> lane 63 holds the wave sum
#undef DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- two-stage (workspace) reductions instead of fp32 atomics: L2 float atomics sustain only ~130 GB/s on gfx950
// (measured: every extra split of a 26 MB weight gradient cost ~0.2 ms), a plain partial write + reduce pass is 20x cheaper.
// out[j] += sum_p part[p*stride + j], j < n          (defined in elementwise.hip)
int dicow_launch_reduce_parts(const float* part, int nparts, int64_t stride, float* out, int64_t n, hipStream_t st);
// out[k][j] += sum_p part[p*stride + k*kstride + j] for k < nout (<= 11; NULL outputs skipped), ONE launch
int dicow_launch_reduce_multi(const float* part, int nparts, int64_t stride, int64_t kstride, float* const* outs, int nout,
                              int64_t n, hipStream_t st);
