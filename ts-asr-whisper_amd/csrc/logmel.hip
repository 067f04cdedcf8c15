// Whisper log-mel front end on the GPU.
//
// Replaces the CPU feature extraction the reference reaches at src/data/local_datasets.py:208-214 through
// transformers' WhisperFeatureExtractor (HF feature_extraction_whisper.py:135-165): centred reflect-padded STFT
// (hann-400 periodic window, hop 160, last frame dropped) -> |.|^2 -> slaney mel projection -> log10(clamp 1e-10)
// -> max(x, clip_max - 8) -> (x + 4) / 4, output [B, M, n_frames] fp32.
//
// Round 4: the 400-point real DFT of a block of frames IS a matrix product -- [frames x 400 samples] . [400 x (201 cos | 201 sin)]
// against the window-folded twiddle tables -- and runs on the matrix pipe in EXACT fp32 (v_mfma_f32_32x32x2_f32: 64 flop / clock /
// SIMD = the fp32 vector rate, but one instruction per 64 cycles instead of 32 v_fma per lane, with every frame / bin pair reusing
// its operands from registers).  logmel_mfma_kernel, one workgroup = 64 frames of one clip, 4 waves = 2 frame blocks x {re, im}:
//   * the 10480 samples the frames touch are staged in LDS once, one pad word every 160 samples (hop 160 = 0 mod 32 banks: the
//     A fragment -- lane = frame -- would otherwise be a 32-way bank conflict);
//   * the product is folded about sample 200 (the periodic Hann window, the cosines and -- with a sign -- the sines are symmetric
//     there): re = C (x[n] + x[400 - n]), im = S (x[n] - x[400 - n]) over 204 table rows instead of 400;
//   * B fragments (2 samples x 32 bins of the cos or sin table, rows padded to 224 bins) come straight from L2 (the folded table pair
//     is 366 KB, shared by every workgroup), two k-steps ahead in registers; 7 bin blocks x 102 k-steps = 714 MFMAs per wave;
//   * re^2 + im^2 meet in LDS ([64][201] fp32 over the sample stage), the slaney projection walks only each filter's non-zero bins
//     (~400 multiply-adds per frame instead of 201 x M), log10, workgroup maximum.
// 104 VGPRs + 57 KB of LDS: three workgroups per CU, the 750 workgroups of 16 clips are resident at once.
// Kernel 2: clip max over the workgroup maxima, then the dynamic-range clamp and affine normalisation.
// logmel_kernel (round 1-3: the direct DFT on the vector pipe, thread = frequency bin) is kept for -DLOGMEL_DIRECT A/B builds.
#include "common.h"

#define LM_FRAMES 32
#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_SPAN (LM_HOP * (LM_FRAMES - 1) + LM_NFFT)      // 5360 samples

__global__ void __launch_bounds__(256) logmel_kernel(const float* __restrict__ wave, int n_samples, int n_frames,
                                                     const float* __restrict__ tw_cos, const float* __restrict__ tw_sin, int ldt,
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
            for (int e = 0; e < 4; ++e) { c[e] = tw_cos[(n + e) * ldt + tid]; s[e] = tw_sin[(n + e) * ldt + tid]; }
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

// ---- round 4: the DFT on the matrix pipe (see the header)
#define LMM_FRAMES 64
#define LMM_SPAN (LM_HOP * (LMM_FRAMES - 1) + LM_NFFT)                 // 10480 samples
#define LMM_XS (LMM_SPAN + LMM_SPAN / LM_HOP + 1)                       // + one pad word per 160 samples
#define LMM_LD 224                                                      // table row length (bins, zero-padded): 7 blocks of 32
#define LMM_PW 201                                                      // row stride of the power image (odd: conflict-free columns)
#define LMM_NK 204                                                      // rows of the folded tables (201 / 200 live ones, zero-padded to whole pairs of k-steps)
typedef __attribute__((ext_vector_type(16))) float lm_f32x16_t;

__global__ void __launch_bounds__(256, 3) logmel_mfma_kernel(const float* __restrict__ wave, int n_samples, int n_frames,
                                                             const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                             const float* __restrict__ fb, const int* __restrict__ mel_range, int M,
                                                             float* __restrict__ out, float* __restrict__ blockmax) {
    __shared__ __attribute__((aligned(16))) float sm[LMM_FRAMES * LMM_PW > LMM_XS ? LMM_FRAMES * LMM_PW : LMM_XS];
    __shared__ float red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * LMM_FRAMES, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* w = wave + (int64_t)b * n_samples;
    const int p0 = t0 * LM_HOP - LM_NFFT / 2;
    if (p0 >= 0 && p0 + LMM_SPAN <= n_samples) {                      // interior block: 16-byte loads (p0 and the span are multiples of 4)
        const float4* w4 = reinterpret_cast<const float4*>(w + p0);
        for (int i4 = tid; i4 < LMM_SPAN / 4; i4 += 256) {
            const float4 x = w4[i4];
            float* d = &sm[4 * i4 + (4 * i4) / LM_HOP];               // (a quad never straddles a 160-sample block)
            d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
        }
    } else {
        for (int i = tid; i < LMM_SPAN; i += 256) {
            int p = p0 + i;                                           // centred frames, reflect padding
            if (p < 0) p = -p;
            if (p >= n_samples) p = 2 * (n_samples - 1) - p;
            p = p < 0 ? 0 : (p >= n_samples ? n_samples - 1 : p);    // frames beyond the clip (masked below)
            sm[i + i / LM_HOP] = w[p];
        }
    }
    if (tid == 0) sm[LMM_SPAN + LMM_SPAN / LM_HOP] = 0.f;             // sample 400 of the last frame: meets a zero table row, must be finite
    __syncthreads();
    const int rb = wv & 1, half = wv >> 1;                            // frame block, {0: cos -> re, 1: sin -> im}
    const int f = lane & 31, kk = lane >> 5;
    // The window-folded tables are symmetric (cos) / antisymmetric (sin) about sample 200 (periodic Hann: w[n] = w[400 - n], w[0] = 0):
    //   re_k = sum_{n=0..200} C[n][k] (x[n] + x[400 - n]),   im_k = sum_{n=0..199} S[n][k] (x[n] - x[400 - n])
    // (host tables: C[200] holds HALF the centre term -- x[200] is added to itself --, rows >= 201 / >= 200 are zero, LMM_NK rows in
    // all): half the multiply-adds of the plain product; the A fragment costs a second LDS read and an add per k-step of 7 MFMAs.
    const float* tab = (half ? tw_sin : tw_cos) + (LM_NFFT + kk) * LMM_LD + f;   // folded rows follow the 400 plain ones; B fragment of k-step n: tab[n * LMM_LD + 32 cb]
    const float* xf = sm + (rb * 32 + f) * (LM_HOP + 1);               // sample i of this lane's frame: xf[i + i / 160]
    const float sgn = half ? -1.0f : 1.0f;
    lm_f32x16_t acc[7];
#pragma unroll
    for (int cb = 0; cb < 7; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    // B fragments TWO k-steps ahead in registers (L2 latency ~ two k-steps of 7 MFMAs when three waves share the SIMD); the asm
    // statement pins the order "request step n + 4, then compute step n" -- left alone hipcc sinks the loads next to their use
    float b0[7], b1[7], bc[7];
#pragma unroll
    for (int cb = 0; cb < 7; ++cb) { b0[cb] = tab[cb * 32]; b1[cb] = tab[2 * LMM_LD + cb * 32]; }
    // folded sample idx = n + kk of a k-step: x[idx] at pad offset (idx >= 160), x[400 - idx] at pad offset 2 if idx <= 80 else 1
#define LMM_AFRAG(N_) ({ const int i_ = (N_) + kk; xf[i_ + (i_ >= LM_HOP ? 1 : 0)] + sgn * xf[LM_NFFT - i_ + (i_ <= 80 ? 2 : 1)]; })
#pragma unroll 2
    for (int n = 0; n < LMM_NK; n += 4) {
        const float a0 = LMM_AFRAG(n), a1 = LMM_AFRAG(n + 2);
        const int na = n + 4 < LMM_NK ? n + 4 : n, nb = n + 6 < LMM_NK ? n + 6 : n;           // (past the end: re-read, no branch)
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) bc[cb] = b0[cb];
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) b0[cb] = tab[na * LMM_LD + cb * 32];
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bc[cb], acc[cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) bc[cb] = b1[cb];
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) b1[cb] = tab[nb * LMM_LD + cb * 32];
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bc[cb], acc[cb], 0, 0, 0);
    }
#undef LMM_AFRAG
    __syncthreads();                                                  // every wave is done with the sample stage
    // power image pw[frame][bin] over the same LDS: the re waves write re^2, then the im waves add im^2
    // C layout of the 32 x 32 product: lane -> column (bin) lane & 31, register r -> row (frame) 8 (r / 4) + 4 (lane / 32) + r % 4
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (half == ph) {
#pragma unroll
            for (int cb = 0; cb < 7; ++cb) {
                const int bin = cb * 32 + f;
                if (bin < LM_BINS) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* p = &sm[(rb * 32 + 8 * (r >> 2) + 4 * kk + (r & 3)) * LMM_PW + bin];
                        const float v = acc[cb][r] * acc[cb][r];
                        *p = ph ? *p + v : v;
                    }
                }
            }
        }
        __syncthreads();
    }
    // slaney projection over each filter's non-zero bins [lo, hi) only, log10, outputs, workgroup maximum
    const int fr = tid & 63, mg = tid >> 6;
    const bool fvalid = t0 + fr < n_frames;
    float lmax = -INFINITY;
    for (int m = mg; m < M; m += 4) {
        const int lo = mel_range[2 * m], hi = mel_range[2 * m + 1];
        float a = 0.f;
        for (int k = lo; k < hi; ++k) a += sm[fr * LMM_PW + k] * fb[k * M + m];
        const float v = __builtin_amdgcn_logf(fmaxf(a, 1e-10f)) * 0.30102999566398120f;      // v_log_f32 (base 2, ~1 ulp) x log10(2)
        if (fvalid) {
            out[((int64_t)b * M + m) * n_frames + t0 + fr] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wv] = lmax;
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
    return (int64_t)B * dicow_cdiv(n_frames, LM_FRAMES) * 4;          // (sized for the smaller of the two frame blocks: covers both kernels)
}

extern "C" int dicow_logmel(const float* wave, int B, int n_samples, const float* tw_cos, const float* tw_sin, const float* fb,
                            const int* mel_range, int M, float* out, void* ws, int64_t ws_bytes, void* stream) {
    DICOW_REQUIRE(wave && tw_cos && tw_sin && fb && mel_range && out && B > 0 && M > 0, "logmel: null/empty argument");
    DICOW_REQUIRE(n_samples >= LM_NFFT && n_samples % LM_HOP == 0, "logmel: n_samples=%d must be a multiple of %d (pad to 30 s)", n_samples, LM_HOP);
    const int n_frames = n_samples / LM_HOP;
#ifdef LOGMEL_DIRECT
    const int nb = dicow_cdiv(n_frames, LM_FRAMES);
#else
    const int nb = dicow_cdiv(n_frames, LMM_FRAMES);
#endif
    DICOW_REQUIRE(ws && ws_bytes >= (int64_t)B * nb * 4, "logmel: workspace too small (need %ld bytes)", (long)B * nb * 4);
#ifdef LOGMEL_DIRECT
    hipLaunchKernelGGL(logmel_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, wave, n_samples, n_frames, tw_cos, tw_sin,
                       LMM_LD, fb, M, out, (float*)ws);
#else
    hipLaunchKernelGGL(logmel_mfma_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, wave, n_samples, n_frames, tw_cos, tw_sin,
                       fb, mel_range, M, out, (float*)ws);
#endif
    DICOW_CHECK_LAUNCH("logmel");
    const int64_t per_clip = (int64_t)M * n_frames;
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(64, B), dim3(256), 0, (hipStream_t)stream, out, (const float*)ws, nb, per_clip);
    DICOW_CHECK_LAUNCH("logmel_finalize");
    return DICOW_OK;
}
