// Whisper log-mel front end on the GPU.
//
// Replaces the CPU feature extraction the reference reaches at src/data/local_datasets.py:208-214 through
// transformers' WhisperFeatureExtractor (HF feature_extraction_whisper.py:135-165): centred reflect-padded STFT
// (hann-400 periodic window, hop 160, last frame dropped) -> |.|^2 -> slaney mel projection -> log10(clamp 1e-10)
// -> max(x, clip_max - 8) -> (x + 4) / 4, output [B, M, n_frames] fp32.
//
// Kernel 1 (one workgroup = 32 frames of one clip): the 5360 samples the frames touch are staged in LDS; thread k
// owns frequency bin k and runs the direct DFT for all 32 frames against window-folded twiddle tables
// tw_cos/tw_sin [400, 201] (coalesced over k, L2-resident), 64 fp32 accumulators in registers; power spectrum goes
// to LDS; the mel projection is a [32 x 201] x [201 x M] product from LDS; log10 and the workgroup max are written.
// Kernel 2: clip max over the workgroup maxima, then the dynamic-range clamp and affine normalisation.
// ~15 GFLOP fp32 per 16 clips: far below both rooflines; exact-ish fp32 (no FFT reordering error).
#include "common.h"

#define LM_FRAMES 32
#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_SPAN (LM_HOP * (LM_FRAMES - 1) + LM_NFFT)      // 5360 samples

__global__ void __launch_bounds__(256) logmel_kernel(const float* __restrict__ wave, int n_samples, int n_frames,
                                                     const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                     const float* __restrict__ fb, int M, float* __restrict__ out,
                                                     float* __restrict__ blockmax) {
    __shared__ __attribute__((aligned(16))) float xs[LM_SPAN];
    __shared__ float pw[LM_FRAMES][LM_BINS + 1];
    __shared__ float red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * LM_FRAMES, tid = threadIdx.x;
    const float* w = wave + (int64_t)b * n_samples;
    for (int i = tid; i < LM_SPAN; i += 256) {
        int p = t0 * LM_HOP + i - LM_NFFT / 2;                       // centred frames, reflect padding
        if (p < 0) p = -p;
        if (p >= n_samples) p = 2 * (n_samples - 1) - p;
        p = p < 0 ? 0 : (p >= n_samples ? n_samples - 1 : p);        // frames beyond the clip (masked below)
        xs[i] = w[p];
    }
    __syncthreads();
    if (tid < LM_BINS) {
        float re[LM_FRAMES], im[LM_FRAMES];
#pragma unroll
        for (int f = 0; f < LM_FRAMES; ++f) { re[f] = 0.f; im[f] = 0.f; }
        for (int n = 0; n < LM_NFFT; n += 4) {
            float c[4], s[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { c[e] = tw_cos[(n + e) * LM_BINS + tid]; s[e] = tw_sin[(n + e) * LM_BINS + tid]; }
#pragma unroll
            for (int f = 0; f < LM_FRAMES; ++f) {
                const float4 x = *reinterpret_cast<const float4*>(&xs[f * LM_HOP + n]);
                re[f] += x.x * c[0] + x.y * c[1] + x.z * c[2] + x.w * c[3];
                im[f] += x.x * s[0] + x.y * s[1] + x.z * s[2] + x.w * s[3];
            }
        }
#pragma unroll
        for (int f = 0; f < LM_FRAMES; ++f) pw[f][tid] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();
    const int f = tid & 31, mg = tid >> 5;
    float lmax = -INFINITY;
    const bool fvalid = t0 + f < n_frames;
    for (int m = mg; m < M; m += 8) {
        float acc = 0.f;
        for (int k = 0; k < LM_BINS; ++k) acc += pw[f][k] * fb[k * M + m];
        const float v = log10f(fmaxf(acc, 1e-10f));
        if (fvalid) {
            out[((int64_t)b * M + m) * n_frames + t0 + f] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) blockmax[(int64_t)b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void logmel_finalize_kernel(float* __restrict__ out, const float* __restrict__ blockmax, int nblocks, int64_t per_clip) {
    __shared__ float cm;
    const int b = blockIdx.y;
    if (threadIdx.x < 64) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < nblocks; i += 64) m = fmaxf(m, blockmax[(int64_t)b * nblocks + i]);
        m = wave_max(m);
        if (threadIdx.x == 0) cm = m;
    }
    __syncthreads();
    const float floor_v = cm - 8.0f;
    float* o = out + (int64_t)b * per_clip;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = (fmaxf(o[i], floor_v) + 4.0f) * 0.25f;
}

extern "C" int64_t dicow_logmel_ws_bytes(int B, int n_samples) {
    const int n_frames = n_samples / LM_HOP;
    return (int64_t)B * dicow_cdiv(n_frames, LM_FRAMES) * 4;
}

extern "C" int dicow_logmel(const float* wave, int B, int n_samples, const float* tw_cos, const float* tw_sin, const float* fb,
                            int M, float* out, void* ws, int64_t ws_bytes, void* stream) {
    DICOW_REQUIRE(wave && tw_cos && tw_sin && fb && out && B > 0 && M > 0, "logmel: null/empty argument");
    DICOW_REQUIRE(n_samples >= LM_NFFT && n_samples % LM_HOP == 0, "logmel: n_samples=%d must be a multiple of %d (pad to 30 s)", n_samples, LM_HOP);
    const int n_frames = n_samples / LM_HOP, nb = dicow_cdiv(n_frames, LM_FRAMES);
    DICOW_REQUIRE(ws && ws_bytes >= (int64_t)B * nb * 4, "logmel: workspace too small (need %ld bytes)", (long)B * nb * 4);
    hipLaunchKernelGGL(logmel_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, wave, n_samples, n_frames, tw_cos, tw_sin,
                       fb, M, out, (float*)ws);
    DICOW_CHECK_LAUNCH("logmel");
    const int64_t per_clip = (int64_t)M * n_frames;
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(64, B), dim3(256), 0, (hipStream_t)stream, out, (const float*)ws, nb, per_clip);
    DICOW_CHECK_LAUNCH("logmel_finalize");
    return DICOW_OK;
}
