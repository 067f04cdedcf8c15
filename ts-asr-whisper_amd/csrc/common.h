// Shared device/host helpers for libdicow_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

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
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f2bfbits(lo) | ((unsigned)f2bfbits(hi) << 16);
}

// exact-erf GELU and its derivative (Whisper activation_function="gelu")
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- wave / block reductions (wave = 64 lanes)
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
