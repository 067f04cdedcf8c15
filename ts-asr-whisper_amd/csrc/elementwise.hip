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
                                      unsigned short* __restrict__ dst_t, int R, int C, int64_t ld, int64_t ld_t) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        unsigned short v = 0;
        if (r < R && c < C) {
            v = f2bfbits(src[(int64_t)r * C + c]);
            if (dst) dst[(int64_t)r * ld + c] = v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (!dst_t) return;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) dst_t[(int64_t)c * ld_t + r] = tile[tx][i];
    }
}

// Same, 4 elements per thread: 16-byte loads, 8-byte stores in both orientations (the scalar form above moves 2 bytes per
// lane and store instruction and ran at a fifth of the streaming rate; it stays for shapes that are not multiples of 4).
__global__ void cast_transpose4_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                       unsigned short* __restrict__ dst_t, int R, int C, int64_t ld, int64_t ld_t) {
    __shared__ unsigned short tile[64][68];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;  // 256 threads: 16 rows x 16 column quads per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = k * 16 + tr, r = r0 + i, c = c0 + tq * 4;
        uint2 u = make_uint2(0, 0);
        if (r < R && c < C) {
            const float4 f = *reinterpret_cast<const float4*>(src + (int64_t)r * C + c);
            u = make_uint2(pack_bf16x2(f.x, f.y), pack_bf16x2(f.z, f.w));
            if (dst) *reinterpret_cast<uint2*>(dst + (int64_t)r * ld + c) = u;
        }
        *reinterpret_cast<uint2*>(&tile[i][tq * 4]) = u;
    }
    __syncthreads();
    if (!dst_t) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = k * 16 + tr, c = c0 + i, r = r0 + tq * 4;      // row c of the transposed copy, 4 consecutive r
        if (c < C && r < R) {
            const unsigned lo = (unsigned)tile[tq * 4][i] | ((unsigned)tile[tq * 4 + 1][i] << 16);
            const unsigned hi = (unsigned)tile[tq * 4 + 2][i] | ((unsigned)tile[tq * 4 + 3][i] << 16);
            *reinterpret_cast<uint2*>(dst_t + (int64_t)c * ld_t + r) = make_uint2(lo, hi);
        }
    }
}

extern "C" int dicow_cast_transpose_f32_to_bf16(const float* src, void* dst, int64_t ld, void* dst_t, int64_t ld_t, int R,
                                                int C, void* stream) {
    DICOW_REQUIRE(src && (dst || dst_t) && R > 0 && C > 0, "cast_transpose: bad args");
    DICOW_REQUIRE((!dst || ld >= C) && (!dst_t || ld_t >= R), "cast_transpose: leading dimensions too small");
    dim3 grid(dicow_cdiv(C, 64), dicow_cdiv(R, 64));
    const bool vec = (R % 4 == 0) && (C % 4 == 0) && (ld % 4 == 0) && (ld_t % 4 == 0) &&
                     ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0) && ((uintptr_t)dst_t % 8 == 0);
    if (vec)
        hipLaunchKernelGGL(cast_transpose4_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst,
                           (unsigned short*)dst_t, R, C, ld, ld_t);
    else
        hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst,
                           (unsigned short*)dst_t, R, C, ld, ld_t);
    DICOW_CHECK_LAUNCH("cast_transpose");
    return DICOW_OK;
}

// Several matrices in ONE launch (the six weight matrices of an encoder layer, re-cast after every optimizer step): one
// launch per matrix moved 13-52 MB each at 1.5-3 TB/s (8.4 us average, 215 launches per step); pooled, a layer's 157 MB go at the
// streaming rate.  Block b belongs to the problem whose tile range contains it.
struct cast_group_kargs_t { dicow_cast_problem p[DICOW_CAST_GROUP_MAX]; int tile0[DICOW_CAST_GROUP_MAX + 1]; int n; };
__global__ void cast_transpose4_group_kernel(const cast_group_kargs_t g) {
    __shared__ unsigned short tile[64][68];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < DICOW_CAST_GROUP_MAX; ++i) if (i < g.n && (int)blockIdx.x >= g.tile0[i]) pi = i;
    const dicow_cast_problem& q = g.p[pi];
    const int R = q.R, C = q.C;
    const int64_t ld = q.ld, ld_t = q.ld_t;
    const float* __restrict__ src = q.src;
    unsigned short* __restrict__ dst = reinterpret_cast<unsigned short*>(q.dst);
    unsigned short* __restrict__ dst_t = reinterpret_cast<unsigned short*>(q.dst_t);
    const int lb = (int)blockIdx.x - g.tile0[pi], ntc = (C + 63) / 64;
    const int r0 = (lb / ntc) * 64, c0 = (lb % ntc) * 64;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = k * 16 + tr, r = r0 + i, c = c0 + tq * 4;
        uint2 u = make_uint2(0, 0);
        if (r < R && c < C) {
            const float4 f = *reinterpret_cast<const float4*>(src + (int64_t)r * C + c);
            u = make_uint2(pack_bf16x2(f.x, f.y), pack_bf16x2(f.z, f.w));
            if (dst) *reinterpret_cast<uint2*>(dst + (int64_t)r * ld + c) = u;
        }
        *reinterpret_cast<uint2*>(&tile[i][tq * 4]) = u;
    }
    __syncthreads();
    if (!dst_t) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = k * 16 + tr, c = c0 + i, r = r0 + tq * 4;
        if (c < C && r < R) {
            const unsigned lo = (unsigned)tile[tq * 4][i] | ((unsigned)tile[tq * 4 + 1][i] << 16);
            const unsigned hi = (unsigned)tile[tq * 4 + 2][i] | ((unsigned)tile[tq * 4 + 3][i] << 16);
            *reinterpret_cast<uint2*>(dst_t + (int64_t)c * ld_t + r) = make_uint2(lo, hi);
        }
    }
}

extern "C" int dicow_cast_transpose_group(const dicow_cast_problem* p, int n, void* stream) {
    DICOW_REQUIRE(p && n >= 1 && n <= DICOW_CAST_GROUP_MAX, "cast_transpose_group: 1..%d problems", DICOW_CAST_GROUP_MAX);
    cast_group_kargs_t k;
    memset(&k, 0, sizeof(k));
    int tiles = 0;
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        const dicow_cast_problem& q = p[i];
        DICOW_REQUIRE(q.src && (q.dst || q.dst_t) && q.R > 0 && q.C > 0, "cast_transpose_group: bad problem %d", i);
        DICOW_REQUIRE((!q.dst || q.ld >= q.C) && (!q.dst_t || q.ld_t >= q.R), "cast_transpose_group: leading dimensions too small");
        vec = vec && (q.R % 4 == 0) && (q.C % 4 == 0) && (q.ld % 4 == 0) && (q.ld_t % 4 == 0) && ((uintptr_t)q.src % 16 == 0) &&
              ((uintptr_t)q.dst % 8 == 0) && ((uintptr_t)q.dst_t % 8 == 0);
        k.p[i] = q; k.tile0[i] = tiles;
        tiles += dicow_cdiv(q.R, 64) * dicow_cdiv(q.C, 64);
    }
    for (int i = n; i <= DICOW_CAST_GROUP_MAX; ++i) k.tile0[i] = tiles;
    k.n = n;
    if (!vec) {                                      // odd shapes / alignments: one by one through the general entry point
        for (int i = 0; i < n; ++i) {
            const int rc = dicow_cast_transpose_f32_to_bf16(p[i].src, p[i].dst, p[i].ld, p[i].dst_t, p[i].ld_t, p[i].R, p[i].C, stream);
            if (rc != DICOW_OK) return rc;
        }
        return DICOW_OK;
    }
    hipLaunchKernelGGL(cast_transpose4_group_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, k);
    DICOW_CHECK_LAUNCH("cast_transpose_group");
    return DICOW_OK;
}

// Conv1d weight [O,C,3] -> [O,Kpad] with k = tap*C + c
__global__ void conv_weight_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                        unsigned short* __restrict__ dst_t, int O, int C, int Kpad) {
    const int64_t n = (int64_t)O * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / Kpad), k = (int)(i - (int64_t)o * Kpad);
        float v = 0.f;
        if (k < 3 * C) {
            const int tap = k / C, c = k - tap * C;
            v = w[((int64_t)o * C + c) * 3 + tap];
        }
        if (dst) dst[i] = f2bfbits(v);
        if (dst_t) dst_t[(int64_t)k * O + o] = f2bfbits(v);
    }
}

extern "C" int dicow_conv_weight_pack(const float* w, void* dst, void* dst_t, int O, int C, int Kpad, void* stream) {
    DICOW_REQUIRE(w && (dst || dst_t) && O > 0 && C > 0 && Kpad >= 3 * C, "conv_weight_pack: bad args");
    const int64_t n = (int64_t)O * Kpad;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst,
                       (unsigned short*)dst_t, O, C, Kpad);
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

// ---- partial-sum reductions --------------------------------------------------------------------------------------
// few, long partials (split weight gradients): one float4 column per thread
__global__ void reduce_parts_wide_kernel(const float* __restrict__ part, int nparts, int64_t stride, float* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s = reinterpret_cast<float4*>(out)[i];
        for (int p = 0; p < nparts; ++p) {
            const float4 v = reinterpret_cast<const float4*>(part + (int64_t)p * stride)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}
// many, short partials (column sums over hundreds of workgroups): 16 columns x 16 partial-lanes per block;
// blockIdx.y selects one of up to 11 outputs whose partials are interleaved [part][k][n] (fddt_ln_bwd)
struct reduce_multi_args { float* out[11]; };
__global__ void reduce_parts_tall_kernel(const float* __restrict__ part, int nparts, int64_t stride, int64_t kstride,
                                         reduce_multi_args outs, int64_t n) {
    __shared__ float red[16][17];
    float* out = outs.out[blockIdx.y];
    if (!out) return;
    part += (int64_t)blockIdx.y * kstride;
    const int c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + c;
    float s = 0.f;
    if (j < n) {
        int p = pl;
        for (; p + 48 < nparts; p += 64) {
            const float a0 = part[(int64_t)p * stride + j], a1 = part[(int64_t)(p + 16) * stride + j];
            const float a2 = part[(int64_t)(p + 32) * stride + j], a3 = part[(int64_t)(p + 48) * stride + j];
            s += (a0 + a1) + (a2 + a3);
        }
        for (; p < nparts; p += 16) s += part[(int64_t)p * stride + j];
    }
    red[pl][c] = s;
    __syncthreads();
    if (pl == 0 && j < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][c];
        out[j] += t;
    }
}

int dicow_launch_reduce_multi(const float* part, int nparts, int64_t stride, int64_t kstride, float* const* outs, int nout,
                              int64_t n, hipStream_t st) {
#ifdef DICOW_SKIP_REDUCE_MULTI      // ablation builds only (tools/build_var.sh): timing without the small reduction launches -- results are WRONG
    return DICOW_OK;
#endif
    reduce_multi_args ra;
    for (int k = 0; k < 11; ++k) ra.out[k] = k < nout ? outs[k] : nullptr;
    hipLaunchKernelGGL(reduce_parts_tall_kernel, dim3((unsigned)((n + 15) / 16), nout), dim3(256), 0, st, part, nparts, stride,
                       kstride, ra, n);
    DICOW_CHECK_LAUNCH("reduce_parts_multi");
    return DICOW_OK;
}

int dicow_launch_reduce_parts(const float* part, int nparts, int64_t stride, float* out, int64_t n, hipStream_t st) {
    if (nparts <= 16 && n % 4 == 0 && stride % 4 == 0) {
        int grid = (int)((n / 4 + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
        hipLaunchKernelGGL(reduce_parts_wide_kernel, dim3(grid), dim3(256), 0, st, part, nparts, stride, out, n / 4);
    } else {
        float* outs[1] = {out};
        return dicow_launch_reduce_multi(part, nparts, stride, 0, outs, 1, n, st);
    }
    DICOW_CHECK_LAUNCH("reduce_parts");
    return DICOW_OK;
}

// column sums of bf16 [rows,N] (ld) += into fp32 out[N]: stage 1 = per-(column pair, row slice) partials, stage 2 = reduce
__global__ void colsum_bf16_kernel(const unsigned short* __restrict__ x, int64_t ld, float* __restrict__ part, int rows, int N) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (c >= N) return;
    float s0 = 0.f, s1 = 0.f;
    const int r1 = (int)(((int64_t)rows * (blockIdx.y + 1)) / gridDim.y);
    for (int r = (int)(((int64_t)rows * blockIdx.y) / gridDim.y); r < r1; ++r) {
        const unsigned u = *reinterpret_cast<const unsigned*>(x + (int64_t)r * ld + c);
        s0 += __uint_as_float(u << 16);
        s1 += __uint_as_float(u & 0xffff0000u);
    }
    *reinterpret_cast<float2*>(part + (int64_t)blockIdx.y * N + c) = make_float2(s0, s1);
}

static int colsum_slices(int rows, int N) {
    const int gx = dicow_cdiv(N, 512);
    int gy = 2048 / gx; if (gy < 1) gy = 1; if (gy > rows) gy = rows; if (gy > 512) gy = 512;
    return gy;
}

extern "C" int64_t dicow_colsum_ws_bytes(int rows, int N) { return (int64_t)colsum_slices(rows, N) * N * 4; }

extern "C" int dicow_colsum_bf16(const void* x, int64_t ld, float* out, int rows, int N, void* ws, int64_t ws_bytes, void* stream) {
    DICOW_REQUIRE(x && out && rows > 0 && N > 0 && N % 2 == 0 && ld % 2 == 0, "colsum_bf16: bad args (N, ld must be even)");
    const int gx = dicow_cdiv(N, 512), gy = colsum_slices(rows, N);
    DICOW_REQUIRE(ws && ws_bytes >= (int64_t)gy * N * 4, "colsum_bf16: workspace too small (need %ld bytes)", (long)gy * N * 4);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, ld, (float*)ws, rows, N);
    DICOW_CHECK_LAUNCH("colsum_bf16");
    return dicow_launch_reduce_parts((const float*)ws, gy, N, out, N, (hipStream_t)stream);
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

// ------------------------------------------------------------------------------------------------ decoder embedding
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                 float* __restrict__ out, int Lq, int D) {
    const int row = blockIdx.x;                   // b*Lq + l
    const int l = row % Lq;
    const int64_t id = ids[row];
    const float4* t = reinterpret_cast<const float4*>(tok + id * D);
    const float4* p = reinterpret_cast<const float4*>(pos + (int64_t)l * D);
    float4* o = reinterpret_cast<float4*>(out + (int64_t)row * D);
    for (int i = threadIdx.x; i < D / 4; i += blockDim.x) {
        const float4 a = t[i], b = p[i];
        o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

extern "C" int dicow_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* out, int B, int Lq, int D, void* stream) {
    DICOW_REQUIRE(ids && tok && pos && out && B > 0 && Lq > 0 && D % 4 == 0, "embed_fwd: bad args");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(B * Lq), dim3(256), 0, (hipStream_t)stream, ids, tok, pos, out, Lq, D);
    DICOW_CHECK_LAUNCH("embed_fwd");
    return DICOW_OK;
}

__global__ void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ g, float* __restrict__ d_tok,
                                 float* __restrict__ d_pos, int Lq, int D) {
    const int row = blockIdx.x;
    const int l = row % Lq;
    const int64_t id = ids[row];
    const float* gr = g + (int64_t)row * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float v = gr[i];
        if (d_tok) atomicAdd(d_tok + id * D + i, v);
        if (d_pos) atomicAdd(d_pos + (int64_t)l * D + i, v);
    }
}

extern "C" int dicow_embed_bwd(const int64_t* ids, const float* g, float* d_tok, float* d_pos, int B, int Lq, int D, void* stream) {
    DICOW_REQUIRE(ids && g && B > 0 && Lq > 0 && D > 0, "embed_bwd: bad args");
    if (!d_tok && !d_pos) return DICOW_OK;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * Lq), dim3(256), 0, (hipStream_t)stream, ids, g, d_tok, d_pos, Lq, D);
    DICOW_CHECK_LAUNCH("embed_bwd");
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ conv-stem backward helpers
__global__ void gelu_bwd_bf16_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ pre,
                                     unsigned short* __restrict__ out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 a = reinterpret_cast<const uint2*>(g)[i], b = reinterpret_cast<const uint2*>(pre)[i];
        const float r0 = __uint_as_float(a.x << 16) * gelu_erf_grad(__uint_as_float(b.x << 16));
        const float r1 = __uint_as_float(a.x & 0xffff0000u) * gelu_erf_grad(__uint_as_float(b.x & 0xffff0000u));
        const float r2 = __uint_as_float(a.y << 16) * gelu_erf_grad(__uint_as_float(b.y << 16));
        const float r3 = __uint_as_float(a.y & 0xffff0000u) * gelu_erf_grad(__uint_as_float(b.y & 0xffff0000u));
        reinterpret_cast<uint2*>(out)[i] = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
    }
}

extern "C" int dicow_gelu_bwd_bf16(const void* g, const void* pre, void* out, int64_t n, void* stream) {
    DICOW_REQUIRE(g && pre && out && n > 0 && n % 4 == 0, "gelu_bwd_bf16: bad args (n %% 4)");
    int grid = (int)((n / 4 + 255) / 256); if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(gelu_bwd_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)g,
                       (const unsigned short*)pre, (unsigned short*)out, n);
    DICOW_CHECK_LAUNCH("gelu_bwd_bf16");
    return DICOW_OK;
}

// padded time index p = s + 1 (s = conv1 output frame); im2col row t of conv2 covers padded rows 2t, 2t+1, 2t+2 (taps 0,1,2)
__global__ void conv2_col2im_gelu_bwd_kernel(const unsigned short* __restrict__ dA2, const unsigned short* __restrict__ pre1,
                                             unsigned short* __restrict__ d_pre1, int T2, int C) {
    const int b = blockIdx.y, s = blockIdx.x;                      // s in [0, 2*T2)
    const int p = s + 1;
    const unsigned short* base = dA2 + (int64_t)b * T2 * 3 * C;
    const unsigned short* src0 = nullptr; const unsigned short* src1 = nullptr;
    if (p & 1) {
        const int t = (p - 1) >> 1;
        if (t < T2) src0 = base + (int64_t)t * 3 * C + C;          // tap 1
    } else {
        const int t = p >> 1;
        if (t < T2) src0 = base + (int64_t)t * 3 * C;              // tap 0
        if (t - 1 >= 0) src1 = base + (int64_t)(t - 1) * 3 * C + 2 * C;   // tap 2
    }
    const int64_t off = ((int64_t)b * 2 * T2 + s) * C;
    for (int c = threadIdx.x * 2; c < C; c += blockDim.x * 2) {
        float g0 = 0.f, g1 = 0.f;
        if (src0) { const unsigned u = *reinterpret_cast<const unsigned*>(src0 + c); g0 += __uint_as_float(u << 16); g1 += __uint_as_float(u & 0xffff0000u); }
        if (src1) { const unsigned u = *reinterpret_cast<const unsigned*>(src1 + c); g0 += __uint_as_float(u << 16); g1 += __uint_as_float(u & 0xffff0000u); }
        if (pre1) {
            const unsigned pu = *reinterpret_cast<const unsigned*>(pre1 + off + c);
            g0 *= gelu_erf_grad(__uint_as_float(pu << 16));
            g1 *= gelu_erf_grad(__uint_as_float(pu & 0xffff0000u));
        }
        *reinterpret_cast<unsigned*>(d_pre1 + off + c) = pack_bf16x2(g0, g1);
    }
}

extern "C" int dicow_conv2_col2im_gelu_bwd(const void* dA2, const void* pre1, void* d_pre1, int B, int T2, int C, void* stream) {
    DICOW_REQUIRE(dA2 && d_pre1 && B > 0 && T2 > 0 && C % 2 == 0, "conv2_col2im_gelu_bwd: bad args");
    hipLaunchKernelGGL(conv2_col2im_gelu_bwd_kernel, dim3(2 * T2, B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)dA2, (const unsigned short*)pre1, (unsigned short*)d_pre1, T2, C);
    DICOW_CHECK_LAUNCH("conv2_col2im_gelu_bwd");
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ optimizer
// Deterministic: the clip coefficient derived from this sum multiplies every gradient, and data-parallel replicas must apply
// bit-identical updates (nothing ever re-synchronises their parameters).  fp32 atomics would add the workgroup partials in
// arrival order; instead every workgroup stores its partial in a fixed slot and the last one to arrive (ticket counter) adds
// the slots in index order.  The slots are module-level device storage (one sumsq in flight per device: calls are
// stream-ordered on the training stream).
#define SUMSQ_MAX_BLOCKS 2048
__device__ float g_sumsq_part[SUMSQ_MAX_BLOCKS];
__device__ unsigned g_sumsq_ticket = 0;

__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ bool last;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        g_sumsq_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(&g_sumsq_ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += __builtin_nontemporal_load(&g_sumsq_part[i]);
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] += (red[0] + red[1]) + (red[2] + red[3]);
        g_sumsq_ticket = 0;
    }
}

// ---- "fabric emulator" (measurement aid, round 6): what an 8-rank RCCL all-reduce of a gradient bucket costs the REST of the GPU
// while it runs beside the backward pass -- a few workgroups parked on CUs, the bucket read and written through HBM twice
// (reduce-scatter + all-gather), for as long as the links would take -- reproduced on ONE GPU: `workgroups` blocks stream over the
// bucket in place (every value rewritten with itself) and pace themselves against the 100 MHz wall clock so that the bucket takes
// bytes / (gbps GB/s).  trainer.GradReducer runs it on the side stream behind the (one-rank) all-reduce of each bucket
// (DICOW_EMULATE_FABRIC_GBPS); bench.py --emulate-fabric-gbps reports step time and exposed wait per rate.
__global__ void __launch_bounds__(256) fabric_emulate_kernel(f32x4_t* buf, int64_t n16, double ticks_per_f4_per_block, int passes) {
    const int64_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    const long long t0 = wall_clock64();
    int64_t done = 0;
    for (int p = 0; p < passes; ++p)
        for (int64_t i = lo; i < hi; i += 256 * 16) {           // 64 KB per workgroup in flight (16 workgroups: up to ~250 GB/s of algorithm bandwidth)
            f32x4_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int64_t j = i + k * 256 + threadIdx.x; if (j < hi) v[k] = __builtin_nontemporal_load(buf + j); }
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int64_t j = i + k * 256 + threadIdx.x; if (j < hi) __builtin_nontemporal_store(v[k], buf + j); }
            done += (hi - i < 256 * 16 ? hi - i : 256 * 16);
            const long long due = t0 + (long long)((double)done * ticks_per_f4_per_block / passes);
            while (wall_clock64() < due) __builtin_amdgcn_s_sleep(32);
        }
}

extern "C" int dicow_fabric_emulate(void* buf, int64_t bytes, double gbps, int workgroups, int passes, void* stream) {
    DICOW_REQUIRE(buf && bytes > 0 && bytes % 16 == 0 && gbps > 0.0 && workgroups > 0 && workgroups <= 1024 && passes > 0 && passes <= 8,
                  "fabric_emulate: bad arguments");
    const int64_t n16 = bytes / 16;
    // the whole bucket in bytes / (gbps 1e9) seconds = that many 1e8-per-second ticks; each block owns 1 / workgroups of it
    const double ticks_total = (double)bytes / (gbps * 1e9) * 1e8;
    const double per_block_f4 = (double)((n16 + workgroups - 1) / workgroups);
    hipLaunchKernelGGL(fabric_emulate_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4_t*>(buf), n16,
                       ticks_total / per_block_f4, passes);
    DICOW_CHECK_LAUNCH("fabric_emulate");
    return DICOW_OK;
}

extern "C" int dicow_sumsq_f32(const float* x, int64_t n, float* out, void* stream) {
    DICOW_REQUIRE(x && out && n > 0, "sumsq_f32: bad args");
    int grid = (int)((n / 4 + 255) / 256); if (grid < 1) grid = 1; if (grid > SUMSQ_MAX_BLOCKS) grid = SUMSQ_MAX_BLOCKS;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, out);
    DICOW_CHECK_LAUNCH("sumsq_f32");
    return DICOW_OK;
}

#ifndef ADAMW_NT
#define ADAMW_NT 0         // 1: nontemporal loads, 2: nontemporal stores (everything here is touched once per step)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4a_t;
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ gnorm_sq, float max_norm, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }      // step-dependent scalars read from the device (graph replay)
    float clip = 1.f;
    if (gnorm_sq && max_norm > 0.f) {
        const float c = max_norm / (sqrtf(gnorm_sq[0]) + 1e-6f);
        clip = c < 1.f ? c : 1.f;
    }
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp, gg, mm, vv;
        if (ADAMW_NT & 1) {
            const f32x4a_t p_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4a_t*>(p) + i), g_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4a_t*>(g) + i);
            const f32x4a_t m_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4a_t*>(m) + i), v_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4a_t*>(v) + i);
            pp = make_float4(p_.x, p_.y, p_.z, p_.w); gg = make_float4(g_.x, g_.y, g_.z, g_.w);
            mm = make_float4(m_.x, m_.y, m_.z, m_.w); vv = make_float4(v_.x, v_.y, v_.z, v_.w);
        } else {
            pp = reinterpret_cast<float4*>(p)[i]; gg = reinterpret_cast<const float4*>(g)[i];
            mm = reinterpret_cast<float4*>(m)[i]; vv = reinterpret_cast<float4*>(v)[i];
        }
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = ga[e] * clip;
            pa[e] *= 1.f - lr * wd;
            ma[e] = b1 * ma[e] + (1.f - b1) * gr;
            va[e] = b2 * va[e] + (1.f - b2) * gr * gr;
            const float denom = sqrtf(va[e]) / sqrtf(bc2) + eps;
            pa[e] -= (lr / bc1) * (ma[e] / denom);
        }
        if (ADAMW_NT & 2) {
            __builtin_nontemporal_store(f32x4a_t{pp.x, pp.y, pp.z, pp.w}, reinterpret_cast<f32x4a_t*>(p) + i);
            __builtin_nontemporal_store(f32x4a_t{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f32x4a_t*>(m) + i);
            __builtin_nontemporal_store(f32x4a_t{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4a_t*>(v) + i);
        } else {
            reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
        }
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * clip;
        float pv = p[i] * (1.f - lr * wd);
        const float mv = b1 * m[i] + (1.f - b1) * gr, vv2 = b2 * v[i] + (1.f - b2) * gr * gr;
        pv -= (lr / bc1) * (mv / (sqrtf(vv2) / sqrtf(bc2) + eps));
        p[i] = pv; m[i] = mv; v[i] = vv2;
    }
}

extern "C" int dicow_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, const float* gnorm_sq, float max_norm, void* stream) {
    DICOW_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_f32: bad args");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    int grid = (int)((n / 4 + 255) / 256); if (grid < 1) grid = 1; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2, gnorm_sq, max_norm, (const float*)nullptr);
    DICOW_CHECK_LAUNCH("adamw_f32");
    return DICOW_OK;
}

// Step counters and schedule ON THE DEVICE: counters[0] = optimizer steps taken (HF state.global_step), counters[1 + i] =
// updates run i has received.  One launch advances them and writes hyper[i] = {lr_i, 1 - beta1^t_i, 1 - beta2^t_i} for every
// active run: lr_i = base lr at scheduler step k - 1 (LambdaLR steps after the optimizer; linear warm-up, then cosine to zero
// at max_steps or constant) x the group multiplier for preheat runs.  Both the launch-by-launch step and its captured
// hipGraph go through this kernel, so they apply bit-identical updates.
__global__ void adamw_hyper_kernel(int* __restrict__ counters, float* __restrict__ hyper, const int* __restrict__ is_pre, int n_runs,
                                   int preheat_only, double lr, double mult, int warmup, int max_steps, int cosine, double b1, double b2) {
    // The schedule and the bias corrections are evaluated in DOUBLE, as torch does (LambdaLR and AdamW's 1 - beta^t are Python
    // floats): 1 - powf(0.999f, t) is off by ~1e-5 relative for small t -- the float nearest to 0.999 is not 0.999 -- and the
    // logged learning rate (FusedAdamW.lr_at, Python doubles) then equals the applied one to the last bit of its float value.
    // One thread per run, any number of runs.
    const int k = counters[0] + 1;
    __syncthreads();
    if (threadIdx.x == 0) counters[0] = k;
    const int ss = k - 1;
    double l = lr;
    if (ss < warmup) l = lr * (double)ss / (double)(warmup > 1 ? warmup : 1);
    else if (cosine && max_steps > 0) {
        const double prog = (double)(ss - warmup) / (double)((max_steps - warmup) > 1 ? (max_steps - warmup) : 1);
        const double c = 0.5 * (1.0 + cos(3.14159265358979323846 * prog));
        l = lr * (c > 0.0 ? c : 0.0);
    }
    for (int i = threadIdx.x; i < n_runs; i += blockDim.x) {
        const int pre = is_pre[i];
        if (preheat_only && !pre) continue;
        const int t = counters[1 + i] + 1;
        counters[1 + i] = t;
        hyper[3 * i] = (float)(l * (pre ? mult : 1.0));
        hyper[3 * i + 1] = (float)(1.0 - pow(b1, (double)t));
        hyper[3 * i + 2] = (float)(1.0 - pow(b2, (double)t));
    }
}

extern "C" int dicow_adamw_hyper(int* counters, float* hyper, const int* is_pre, int n_runs, int preheat_only, double lr, double mult,
                                 int warmup_steps, int max_steps, int cosine, double beta1, double beta2, void* stream) {
    DICOW_REQUIRE(counters && hyper && is_pre && n_runs > 0, "adamw_hyper: bad args");
    const int block = n_runs >= 1024 ? 1024 : (n_runs + 63) / 64 * 64;
    hipLaunchKernelGGL(adamw_hyper_kernel, dim3(1), dim3(block), 0, (hipStream_t)stream, counters, hyper, is_pre, n_runs, preheat_only,
                       lr, mult, warmup_steps, max_steps, cosine, beta1, beta2);
    DICOW_CHECK_LAUNCH("adamw_hyper");
    return DICOW_OK;
}

// The same update with the step-dependent scalars (learning rate after the schedule and the group multiplier, the two bias
// corrections 1 - beta^t) read from DEVICE memory: hyper = {lr, 1 - beta1^t, 1 - beta2^t}.  A training step captured in a
// hipGraph replays with fixed kernel arguments; the host rewrites these three floats before each replay instead.
extern "C" int dicow_adamw_f32_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                                   float beta2, float eps, float weight_decay, const float* gnorm_sq, float max_norm, void* stream) {
    DICOW_REQUIRE(p && g && m && v && n > 0 && hyper, "adamw_f32_dev: bad args");
    int grid = (int)((n / 4 + 255) / 256); if (grid < 1) grid = 1; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, beta1, beta2, eps,
                       weight_decay, 1.f, 1.f, gnorm_sq, max_norm, hyper);
    DICOW_CHECK_LAUNCH("adamw_f32_dev");
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ speaker-communication block glue
// The element-wise steps around the SE-DiCoW enrollment cross-attention (reference layers.py:145-193) on the interleaved row
// layout [Bp][2][T][D] (slot 0 = mixture, slot 1 = enrollment), one pass each instead of cast + strided copies + torch ops:
//   split      hf fp32 -> q_in bf16 (mixture rows), kv_in bf16 (enrollment rows), right half of `cat` = q_in   (layers.py:161)
//   merge_fwd  out = hf;  out[mixture] += tanh(gate) * upd
//   gate_bwd   d_upd = bf16(g[mixture] * tanh(gate));  d gate += (1 - tanh^2) * sum(g[mixture] * upd)   (fixed-order sum)
//   merge_bwd  gin = g;  gin[mixture] += d_qin + d_cat[:, D:];  gin[enrollment] += d_kvin
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__global__ void scb_split_kernel(const float* __restrict__ hf, unsigned short* __restrict__ q_in, unsigned short* __restrict__ kv_in,
                                 unsigned short* __restrict__ cat, int64_t td, int D, int64_t ldcat, int Bp) {
    const int64_t n4 = (int64_t)Bp * td / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4, b = e / td, off = e - b * td;
        const float4 m = *reinterpret_cast<const float4*>(hf + (2 * b) * td + off);
        const float4 r = *reinterpret_cast<const float4*>(hf + (2 * b + 1) * td + off);
        const uint2 mb = make_uint2(pack_bf16x2(m.x, m.y), pack_bf16x2(m.z, m.w));
        *reinterpret_cast<uint2*>(q_in + e) = mb;
        *reinterpret_cast<uint2*>(kv_in + e) = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
        const int64_t row = e / D; const int col = (int)(e - row * D);
        *reinterpret_cast<uint2*>(cat + row * ldcat + D + col) = mb;
    }
}
__global__ void scb_merge_fwd_kernel(const float* __restrict__ hf, const unsigned short* __restrict__ upd, const float* __restrict__ gate,
                                     float* __restrict__ out, int64_t td, int Bp) {
    const float tg = tanhf(gate[0]);
    const int64_t n4 = (int64_t)2 * Bp * td / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4, slot = e / td, off = e - slot * td;
        float4 v = *reinterpret_cast<const float4*>(hf + e);
        if ((slot & 1) == 0) {
            const float4 u = bf4_to_f4(*reinterpret_cast<const uint2*>(upd + (slot >> 1) * td + off));
            v.x += u.x * tg; v.y += u.y * tg; v.z += u.z * tg; v.w += u.w * tg;
        }
        *reinterpret_cast<float4*>(out + e) = v;
    }
}
__global__ void __launch_bounds__(256) scb_gate_bwd_kernel(const float* __restrict__ g, const unsigned short* __restrict__ upd,
                                                           const float* __restrict__ gate, unsigned short* __restrict__ d_upd,
                                                           float* __restrict__ partial, int64_t td, int Bp) {
    __shared__ float red[4];
    const float tg = tanhf(gate[0]);
    const int64_t n4 = (int64_t)Bp * td / 4;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4, b = e / td, off = e - b * td;
        const float4 gq = *reinterpret_cast<const float4*>(g + (2 * b) * td + off);
        const float4 u = bf4_to_f4(*reinterpret_cast<const uint2*>(upd + e));
        acc += (gq.x * u.x + gq.y * u.y) + (gq.z * u.z + gq.w * u.w);
        *reinterpret_cast<uint2*>(d_upd + e) = make_uint2(pack_bf16x2(gq.x * tg, gq.y * tg), pack_bf16x2(gq.z * tg, gq.w * tg));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void scb_gate_finish_kernel(const float* __restrict__ partial, int n, const float* __restrict__ gate, float* __restrict__ d_gate) {
    __shared__ float red[256];
    float t = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) t += partial[i];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { const float tg = tanhf(gate[0]); d_gate[0] += red[0] * (1.f - tg * tg); }
}
__global__ void scb_merge_bwd_kernel(const float* __restrict__ g, const float* __restrict__ d_qin, const unsigned short* __restrict__ d_cat,
                                     int64_t ldcat, const float* __restrict__ d_kvin, float* __restrict__ gin, int64_t td, int D, int Bp) {
    const int64_t n4 = (int64_t)2 * Bp * td / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i * 4, slot = e / td, off = e - slot * td, b = slot >> 1, pe = b * td + off;
        float4 v = *reinterpret_cast<const float4*>(g + e);
        if ((slot & 1) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(d_qin + pe);
            const int64_t row = pe / D; const int col = (int)(pe - row * D);
            const float4 c = bf4_to_f4(*reinterpret_cast<const uint2*>(d_cat + row * ldcat + D + col));
            v.x += a.x + c.x; v.y += a.y + c.y; v.z += a.z + c.z; v.w += a.w + c.w;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(d_kvin + pe);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *reinterpret_cast<float4*>(gin + e) = v;
    }
}
static int scb_grid(int64_t n4) { int64_t g = (n4 + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }
#define SCB_CHECK(name) DICOW_REQUIRE(Bp > 0 && T > 0 && D > 0 && D % 4 == 0, name ": bad shape")
extern "C" int dicow_scb_split(const float* hf, void* q_in, void* kv_in, void* cat, int64_t ldcat, int Bp, int T, int D, void* stream) {
    DICOW_REQUIRE(hf && q_in && kv_in && cat && ldcat >= 2 * D && ldcat % 4 == 0, "scb_split: bad args");
    SCB_CHECK("scb_split");
    const int64_t td = (int64_t)T * D;
    hipLaunchKernelGGL(scb_split_kernel, dim3(scb_grid(Bp * td / 4)), dim3(256), 0, (hipStream_t)stream, hf, (unsigned short*)q_in,
                       (unsigned short*)kv_in, (unsigned short*)cat, td, D, ldcat, Bp);
    DICOW_CHECK_LAUNCH("scb_split");
    return DICOW_OK;
}
extern "C" int dicow_scb_merge_fwd(const float* hf, const void* upd, const float* gate, float* out, int Bp, int T, int D, void* stream) {
    DICOW_REQUIRE(hf && upd && gate && out, "scb_merge_fwd: null operand");
    SCB_CHECK("scb_merge_fwd");
    const int64_t td = (int64_t)T * D;
    hipLaunchKernelGGL(scb_merge_fwd_kernel, dim3(scb_grid(2 * Bp * td / 4)), dim3(256), 0, (hipStream_t)stream, hf,
                       (const unsigned short*)upd, gate, out, td, Bp);
    DICOW_CHECK_LAUNCH("scb_merge_fwd");
    return DICOW_OK;
}
extern "C" int64_t dicow_scb_gate_bwd_ws_bytes(void) { return 1024 * 4; }
extern "C" int dicow_scb_gate_bwd(const float* g, const void* upd, const float* gate, void* d_upd, float* d_gate, int Bp, int T, int D,
                                  void* ws, int64_t ws_bytes, void* stream) {
    DICOW_REQUIRE(g && upd && gate && d_upd && ws && ws_bytes >= 1024 * 4, "scb_gate_bwd: bad args (ws: dicow_scb_gate_bwd_ws_bytes)");
    SCB_CHECK("scb_gate_bwd");
    const int64_t td = (int64_t)T * D;
    int grid = scb_grid(Bp * td / 4); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(scb_gate_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, (const unsigned short*)upd, gate,
                       (unsigned short*)d_upd, (float*)ws, td, Bp);
    DICOW_CHECK_LAUNCH("scb_gate_bwd");
    if (d_gate) {
        hipLaunchKernelGGL(scb_gate_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, grid, gate, d_gate);
        DICOW_CHECK_LAUNCH("scb_gate_finish");
    }
    return DICOW_OK;
}
extern "C" int dicow_scb_merge_bwd(const float* g, const float* d_qin, const void* d_cat, int64_t ldcat, const float* d_kvin, float* gin,
                                   int Bp, int T, int D, void* stream) {
    DICOW_REQUIRE(g && d_qin && d_cat && d_kvin && gin && ldcat >= 2 * D && ldcat % 4 == 0, "scb_merge_bwd: bad args");
    SCB_CHECK("scb_merge_bwd");
    const int64_t td = (int64_t)T * D;
    hipLaunchKernelGGL(scb_merge_bwd_kernel, dim3(scb_grid(2 * Bp * td / 4)), dim3(256), 0, (hipStream_t)stream, g, d_qin,
                       (const unsigned short*)d_cat, ldcat, d_kvin, gin, td, D, Bp);
    DICOW_CHECK_LAUNCH("scb_merge_bwd");
    return DICOW_OK;
}

#ifdef DICOW_EXPERIMENTS
// ------------------------------------------------------------------------------------------------ LayerNorm fold: weight preparation
// (include/dicow_hip.h: dicow_lnfold_prep).  One workgroup per output row n; fixed-order block sums (bit-reproducible).
__global__ void __launch_bounds__(256) lnfold_prep_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ bias,
                                                          unsigned short* __restrict__ Wf, int64_t ldw, float* __restrict__ c,
                                                          float* __restrict__ bf, int K) {
    __shared__ float red[2][4];
    const int n = blockIdx.x;
    const float* w = W + (int64_t)n * K;
    unsigned short* o = Wf + (int64_t)n * ldw;
    float sc = 0.f, sb = 0.f;
    for (int k = threadIdx.x * 2; k < K; k += 512) {          // (K even: host-checked)
        const float w0 = w[k], w1 = w[k + 1];
        const unsigned pf = pack_bf16x2(w0 * gamma[k], w1 * gamma[k + 1]);
        *reinterpret_cast<unsigned*>(o + k) = pf;
        sc += __uint_as_float(pf << 16) + __uint_as_float(pf & 0xffff0000u);
        const unsigned pw = pack_bf16x2(w0, w1);              // what the forward's plain bf16 copy of W holds
        sb = fmaf(beta[k], __uint_as_float(pw << 16), sb);
        sb = fmaf(beta[k + 1], __uint_as_float(pw & 0xffff0000u), sb);
    }
    sc = wave_sum_dpp(sc); sb = wave_sum_dpp(sb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sc; red[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        bf[n] = (bias ? bias[n] : 0.f) + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

extern "C" int dicow_lnfold_prep(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, int64_t ldw, float* c,
                                 float* bf, int N, int K, void* stream) {
    DICOW_REQUIRE(W && gamma && beta && Wf && c && bf, "lnfold_prep: null operand");
    DICOW_REQUIRE(N > 0 && K > 0 && K % 2 == 0 && ldw >= K && ldw % 2 == 0, "lnfold_prep: need K %% 2 == 0 and ldw >= K (N=%d K=%d)", N, K);
    hipLaunchKernelGGL(lnfold_prep_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, gamma, beta, bias,
                       reinterpret_cast<unsigned short*>(Wf), ldw, c, bf, K);
    DICOW_CHECK_LAUNCH("lnfold_prep");
    return DICOW_OK;
}
#endif  // DICOW_EXPERIMENTS
