// Training-time batch augmentations on the GPU.
//
// The reference applies these on the CPU inside its collator (src/data/collators.py:189-214), per batch, in dataloader
// workers.  Here the features already live in HBM (csrc/logmel.hip), so the augmentations run there too.  Random
// *decisions* stay on the host: ts-asr-whisper_amd/augment.py draws them from the torch CPU generator with the same
// calls in the same order as the reference, which makes a fixed seed reproduce the reference exactly, and uploads
// a small plan; the kernels below are the deterministic arithmetic on the batch.
//
//   stno_noise_rescale_kernel   collators.py:50-77    x = stno + noise*sd; x -= min(min_c x, 0); x /= sum_c x
//   stno_segment_kernel         collators.py:79-138   one wave per changed segment: dominant class = argmax of the
//                               segment means, blend towards the one-hot of another class, renormalise
//   specaug_joint_kernel        collators.py:209-214 + augmentations.py:23-120,363-379: one pass that reads
//                               [mel ; STNO repeated x sub], resamples time piecewise with the Keys bicubic kernel
//                               (time warp), zeroes the frequency / time masks on features < 128 and averages the
//                               STNO rows back to encoder rate.
//
// All three are pure HBM streams (read once, write once; the four bicubic taps of neighbouring outputs hit the same
// cache lines).  Floating point is evaluated in the reference's operation order; contraction is switched off for the
// file and the places where ATen's own build contracts (index and weight polynomials, see oracle/augment.py) use
// explicit fmaf.
#include "common.h"

#pragma clang fp contract(off)

#define AUG_C 4            // S, T, N, O

__global__ void __launch_bounds__(256) stno_noise_rescale_kernel(float* __restrict__ stno, const int* __restrict__ rows,
                                                                 const float* __restrict__ noise, float sd, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (t >= T) return;
    float* p = stno + (int64_t)rows[i] * AUG_C * T + t;
    const float* nz = noise + (int64_t)i * AUG_C * T + t;
    float x[AUG_C];
#pragma unroll
    for (int c = 0; c < AUG_C; ++c) x[c] = p[(int64_t)c * T] + nz[(int64_t)c * T] * sd;
    float lo = x[0];
#pragma unroll
    for (int c = 1; c < AUG_C; ++c) lo = fminf(lo, x[c]);
    lo = fminf(lo, 0.f);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < AUG_C; ++c) { x[c] -= lo; s = c == 0 ? x[0] : s + x[c]; }
#pragma unroll
    for (int c = 0; c < AUG_C; ++c) p[(int64_t)c * T] = x[c] / s;
}

// segs int32 [n][4] = (batch row, start, end, index among the non-dominant classes); coef fp32 [n][2] = (1-soft, soft)
__global__ void __launch_bounds__(256) stno_segment_kernel(float* __restrict__ stno, const int* __restrict__ segs,
                                                           const float* __restrict__ coef, int n_seg, int T) {
    const int sid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sid >= n_seg) return;
    const int b = segs[sid * 4], s0 = segs[sid * 4 + 1], s1 = segs[sid * 4 + 2], pick = segs[sid * 4 + 3];
    const float keep = coef[sid * 2], soft = coef[sid * 2 + 1];
    float* base = stno + (int64_t)b * AUG_C * T;
    float sum[AUG_C];
#pragma unroll
    for (int c = 0; c < AUG_C; ++c) {
        float a = 0.f;
        for (int t = s0 + lane; t < s1; t += 64) a += base[(int64_t)c * T + t];
        sum[c] = wave_sum(a) / (float)(s1 - s0);
    }
    int dom = 0;
#pragma unroll
    for (int c = 1; c < AUG_C; ++c) if (sum[c] > sum[dom]) dom = c;         // first maximum, like argmax
    const int target = pick < dom ? pick : pick + 1;
    for (int t = s0 + lane; t < s1; t += 64) {
        float x[AUG_C], s = 0.f;
#pragma unroll
        for (int c = 0; c < AUG_C; ++c) {
            x[c] = keep * base[(int64_t)c * T + t] + soft * (c == target ? 1.f : 0.f);
            s = c == 0 ? x[0] : s + x[c];
        }
#pragma unroll
        for (int c = 0; c < AUG_C; ++c) base[(int64_t)c * T + t] = x[c] / s;
    }
}

struct warp_plan_t { int T, center, warped; };

// Keys cubic convolution weights with A = -0.75 (ATen UpSample.h get_cubic_upsample_coefficients), fused like ATen's build
__device__ __forceinline__ float cubic_near(float x) { return fmaf(fmaf(1.25f, x, -2.25f) * x, x, 1.f); }
__device__ __forceinline__ float cubic_far(float x) { return fmaf(fmaf(fmaf(-0.75f, x, 3.75f), x, -6.f), x, 3.f); }

// value of the time-warped row at output frame t; `at(k)` returns the unwarped row at frame k
template <typename F>
__device__ __forceinline__ float warp_sample(const warp_plan_t& wp, int t, F at) {
    if (wp.warped < 0) return at(t);
    const bool left = t < wp.warped;
    const int n_in = left ? wp.center : wp.T - wp.center, n_out = left ? wp.warped : wp.T - wp.warped;
    const int base = left ? 0 : wp.center, i = left ? t : t - wp.warped;
    const float scale = (float)n_in / (float)n_out;
    const float real = fmaf(scale, (float)i + 0.5f, -0.5f);
    int idx = (int)floorf(real);
    idx = idx < n_in - 1 ? idx : n_in - 1;
    const float lam = fminf(fmaxf(real - (float)idx, 0.f), 1.f), ml = 1.f - lam;
    const float w0 = cubic_far(lam + 1.f), w1 = cubic_near(lam), w2 = cubic_near(ml), w3 = cubic_far(ml + 1.f);
    auto cl = [&](int k) { return base + (k < 0 ? 0 : (k > n_in - 1 ? n_in - 1 : k)); };
    float y = w0 * at(cl(idx - 1));
    y = fmaf(w1, at(cl(idx)), y);
    y = fmaf(w2, at(cl(idx + 1)), y);
    y = fmaf(w3, at(cl(idx + 2)), y);
    return y;
}

__device__ __forceinline__ bool in_masks(const int* __restrict__ m, int n, int p) {
    bool hit = false;
    for (int k = 0; k < n; ++k) hit |= (m[2 * k] <= p) && (p < m[2 * k] + m[2 * k + 1]);
    return hit;
}

// grid (ceil(T/256), M + AUG_C, B).  Feature rows f < M are mel rows; the rest are STNO rows, whose threads each own one
// encoder frame (sub feature frames).
__global__ void __launch_bounds__(256) specaug_joint_kernel(const float* __restrict__ mel, const float* __restrict__ stno,
                                                            float* __restrict__ mel_out, float* __restrict__ stno_out,
                                                            int M, warp_plan_t wp, int sub, const int* __restrict__ fmask,
                                                            int n_fmask, const int* __restrict__ tmask, int n_tmask,
                                                            int n_maskable) {
    const int f = blockIdx.y, b = blockIdx.z, T = wp.T, Te = T / sub;
    const int* fm = fmask + (int64_t)b * n_fmask * 2;
    const int* tm = tmask + (int64_t)b * n_tmask * 2;
    const bool maskable = f < n_maskable;
    const bool fzero = maskable && in_masks(fm, n_fmask, f);
    if (f < M) {
        const int t = blockIdx.x * 256 + threadIdx.x;
        if (t >= T) return;
        const float* row = mel + ((int64_t)b * M + f) * T;
        float y = 0.f;
        if (!(fzero || (maskable && in_masks(tm, n_tmask, t)))) y = warp_sample(wp, t, [&](int k) { return row[k]; });
        mel_out[((int64_t)b * M + f) * T + t] = y;
    } else {
        const int j = blockIdx.x * 256 + threadIdx.x, c = f - M;
        if (j >= Te) return;
        const float* row = stno + ((int64_t)b * AUG_C + c) * Te;
        float acc = 0.f;
        for (int k = 0; k < sub; ++k) {
            const int t = j * sub + k;
            float y = 0.f;
            if (!(fzero || (maskable && in_masks(tm, n_tmask, t))))
                y = warp_sample(wp, t, [&](int q) { return row[q / sub]; });
            acc = k == 0 ? y : acc + y;
        }
        stno_out[((int64_t)b * AUG_C + c) * Te + j] = acc / (float)sub;
    }
}

extern "C" int dicow_stno_noise_rescale(float* stno, const int* rows, const float* noise, int n_rows, int C, int T,
                                        float sd, void* stream) {
    DICOW_REQUIRE(C == AUG_C, "dicow_stno_noise_rescale: C=%d, the STNO mask has %d classes", C, AUG_C);
    DICOW_REQUIRE(n_rows >= 0 && T > 0, "dicow_stno_noise_rescale: bad sizes n_rows=%d T=%d", n_rows, T);
    if (n_rows == 0) return DICOW_OK;
    DICOW_REQUIRE(stno && rows && noise, "dicow_stno_noise_rescale: null pointer");
    stno_noise_rescale_kernel<<<dim3((T + 255) / 256, n_rows), 256, 0, (hipStream_t)stream>>>(stno, rows, noise, sd, T);
    DICOW_CHECK_LAUNCH("stno_noise_rescale_kernel");
    return DICOW_OK;
}

extern "C" int dicow_stno_segment_augment(float* stno, const int* segs, const float* coef, int n_seg, int C, int T,
                                          void* stream) {
    DICOW_REQUIRE(C == AUG_C, "dicow_stno_segment_augment: C=%d, the STNO mask has %d classes", C, AUG_C);
    DICOW_REQUIRE(n_seg >= 0 && T > 0, "dicow_stno_segment_augment: bad sizes n_seg=%d T=%d", n_seg, T);
    if (n_seg == 0) return DICOW_OK;
    DICOW_REQUIRE(stno && segs && coef, "dicow_stno_segment_augment: null pointer");
    stno_segment_kernel<<<(n_seg + 3) / 4, 256, 0, (hipStream_t)stream>>>(stno, segs, coef, n_seg, T);
    DICOW_CHECK_LAUNCH("stno_segment_kernel");
    return DICOW_OK;
}

extern "C" int dicow_specaug_joint(const float* mel, const float* stno, float* mel_out, float* stno_out, int B, int M,
                                   int T, int sub, int center, int warped, const int* fmask, int n_fmask,
                                   const int* tmask, int n_tmask, int n_maskable, void* stream) {
    DICOW_REQUIRE(B > 0 && M > 0 && T > 0 && sub > 0 && T % sub == 0, "dicow_specaug_joint: bad sizes B=%d M=%d T=%d sub=%d",
                  B, M, T, sub);
    DICOW_REQUIRE(mel && stno && mel_out && stno_out, "dicow_specaug_joint: null pointer");
    DICOW_REQUIRE(mel != mel_out && stno != stno_out, "dicow_specaug_joint: the resampling is not in-place");
    DICOW_REQUIRE(warped < 0 || (center > 0 && center < T && warped > 0 && warped < T),
                  "dicow_specaug_joint: warp centre %d -> %d outside (0, %d)", center, warped, T);
    DICOW_REQUIRE(n_fmask >= 0 && n_tmask >= 0 && (n_fmask == 0 || fmask) && (n_tmask == 0 || tmask),
                  "dicow_specaug_joint: mask tables missing");
    DICOW_REQUIRE(n_maskable >= 0 && n_maskable <= M + AUG_C, "dicow_specaug_joint: n_maskable=%d > %d features", n_maskable,
                  M + AUG_C);
    warp_plan_t wp{T, center, warped};
    specaug_joint_kernel<<<dim3((T + 255) / 256, M + AUG_C, B), 256, 0, (hipStream_t)stream>>>(
        mel, stno, mel_out, stno_out, M, wp, sub, fmask, n_fmask, tmask, n_tmask, n_maskable);
    DICOW_CHECK_LAUNCH("specaug_joint_kernel");
    return DICOW_OK;
}
