// Error plumbing + small HBM-bound helper kernels (casts, layout packs, column sums).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dicow_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dicow_abi_version(void) { return DICOW_ABI_VERSION; }
extern "C" const char* dicow_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ casts
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = f2bfbits(src[i]);
}

extern "C" int dicow_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    DICOW_REQUIRE(src && dst && n > 0, "cast_f32_to_bf16: bad args");
    int grid = (int)((n / 4 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n);
    DICOW_CHECK_LAUNCH("cast_f32_to_bf16");
    return DICOW_OK;
}

// [R,C] fp32 -> bf16 [R,C] and bf16 [C,R]; 64x64 tiles through LDS so that both stores are coalesced.
__global__ void cast_transpose_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                      unsigned short* __restrict__ dst_t, int R, int C) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        unsigned short v = 0;
        if (r < R && c < C) {
            v = f2bfbits(src[(int64_t)r * C + c]);
            if (dst) dst[(int64_t)r * C + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (!dst_t) return;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dst_t[(int64_t)c * R + r] = tile[tx][i];
    }
}

extern "C" int dicow_cast_transpose_f32_to_bf16(const float* src, void* dst, void* dst_t, int R, int C, void* stream) {
    DICOW_REQUIRE(src && (dst || dst_t) && R > 0 && C > 0, "cast_transpose: bad args");
    dim3 grid(dicow_cdiv(C, 64), dicow_cdiv(R, 64));
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst,
                       (unsigned short*)dst_t, R, C);
    DICOW_CHECK_LAUNCH("cast_transpose");
    return DICOW_OK;
}

// Conv1d weight [O,C,3] -> [O,Kpad] with k = tap*C + c
__global__ void conv_weight_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int O, int C, int Kpad) {
    const int64_t n = (int64_t)O * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / Kpad), k = (int)(i - (int64_t)o * Kpad);
        float v = 0.f;
        if (k < 3 * C) {
            const int tap = k / C, c = k - tap * C;
            v = w[((int64_t)o * C + c) * 3 + tap];
        }
        dst[i] = f2bfbits(v);
    }
}

extern "C" int dicow_conv_weight_pack(const float* w, void* dst, int O, int C, int Kpad, void* stream) {
    DICOW_REQUIRE(w && dst && O > 0 && C > 0 && Kpad >= 3 * C, "conv_weight_pack: bad args");
    const int64_t n = (int64_t)O * Kpad;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, O, C, Kpad);
    DICOW_CHECK_LAUNCH("conv_weight_pack");
    return DICOW_OK;
}

__global__ void conv_weight_unpack_grad_kernel(const float* __restrict__ gp, float* __restrict__ gw, int O, int C, int Kpad) {
    const int64_t n = (int64_t)O * C * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 3);
        const int64_t oc = i / 3;
        const int c = (int)(oc % C), o = (int)(oc / C);
        gw[i] += gp[(int64_t)o * Kpad + tap * C + c];
    }
}

extern "C" int dicow_conv_weight_unpack_grad(const float* g_packed, float* g_w, int O, int C, int Kpad, void* stream) {
    DICOW_REQUIRE(g_packed && g_w && O > 0 && C > 0 && Kpad >= 3 * C, "conv_weight_unpack_grad: bad args");
    const int64_t n = (int64_t)O * C * 3;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv_weight_unpack_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_packed, g_w, O, C, Kpad);
    DICOW_CHECK_LAUNCH("conv_weight_unpack_grad");
    return DICOW_OK;
}

// mel [B,M,Tin] fp32 -> [B,Tin+2,M] bf16 with zero first/last rows (LDS-tiled transpose, coalesced both sides)
__global__ void mel_to_timemajor_kernel(const float* __restrict__ mel, unsigned short* __restrict__ dst, int M, int Tin) {
    __shared__ unsigned short tile[64][66];
    const int b = blockIdx.z, m0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* src = mel + (int64_t)b * M * Tin;
    unsigned short* out = dst + (int64_t)b * (Tin + 2) * M;
    for (int i = ty; i < 64; i += 4) {
        const int m = m0 + i, t = t0 + tx;
        tile[i][tx] = (m < M && t < Tin) ? f2bfbits(src[(int64_t)m * Tin + t]) : 0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int t = t0 + i, m = m0 + tx;
        if (t < Tin && m < M) out[(int64_t)(t + 1) * M + m] = tile[tx][i];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int m = threadIdx.x; m < M; m += blockDim.x) { out[m] = 0; out[(int64_t)(Tin + 1) * M + m] = 0; }
    }
}

extern "C" int dicow_mel_to_timemajor(const float* mel, void* dst, int B, int M, int Tin, void* stream) {
    DICOW_REQUIRE(mel && dst && B > 0 && M > 0 && Tin > 0, "mel_to_timemajor: bad args");
    dim3 grid(dicow_cdiv(Tin, 64), dicow_cdiv(M, 64), B);
    hipLaunchKernelGGL(mel_to_timemajor_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel, (unsigned short*)dst, M, Tin);
    DICOW_CHECK_LAUNCH("mel_to_timemajor");
    return DICOW_OK;
}

// column sums of bf16 [rows,N] (ld) += into fp32 out[N]; block = 256 threads owning 256*2 columns, rows strided by grid.y
__global__ void colsum_bf16_kernel(const unsigned short* __restrict__ x, int64_t ld, float* __restrict__ out, int rows, int N) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (c >= N) return;
    float s0 = 0.f, s1 = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const unsigned u = *reinterpret_cast<const unsigned*>(x + (int64_t)r * ld + c);
        s0 += __uint_as_float(u << 16);
        s1 += __uint_as_float(u & 0xffff0000u);
    }
    atomicAdd(out + c, s0);
    if (c + 1 < N) atomicAdd(out + c + 1, s1);
}

extern "C" int dicow_colsum_bf16(const void* x, int64_t ld, float* out, int rows, int N, void* stream) {
    DICOW_REQUIRE(x && out && rows > 0 && N > 0 && N % 2 == 0 && ld % 2 == 0, "colsum_bf16: bad args (N, ld must be even)");
    const int gx = dicow_cdiv(N, 512);
    int gy = 2048 / gx; if (gy < 1) gy = 1; if (gy > rows) gy = rows;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ld, out, rows, N);
    DICOW_CHECK_LAUNCH("colsum_bf16");
    return DICOW_OK;
}

__global__ void sum_over_batch_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int64_t TD) {
    const int64_t n4 = TD >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<float4*>(out)[i];
        for (int b = 0; b < B; ++b) {
            const float4 v = reinterpret_cast<const float4*>(g + (int64_t)b * TD)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

extern "C" int dicow_sum_over_batch(const float* g, float* out, int B, int64_t TD, void* stream) {
    DICOW_REQUIRE(g && out && B > 0 && TD > 0 && TD % 4 == 0, "sum_over_batch: bad args");
    int grid = (int)((TD / 4 + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, out, B, TD);
    DICOW_CHECK_LAUNCH("sum_over_batch");
    return DICOW_OK;
}
