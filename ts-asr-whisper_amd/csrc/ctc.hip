// CTC auxiliary loss of the DiCoW encoder (SURVEY.md section 8 "next" row f2), forward and backward.
//
// Replaces torch.nn.functional.ctc_loss as called at reference src/models/dicow/encoder.py:108-135 (blank = last class,
// zero_infinity=True, reduction "mean" = mean_b(nll_b / max(target_len_b, 1))) on the logits of
// get_enc_logits (modeling_dicow.py:242-246): log_softmax over V+1 classes in fp32, alpha/beta recursions over the
// blank-extended label sequence, gradient  d logit[t,c] = softmax[t,c] - posterior[t,c].
//
//   ctc_lse_kernel        one workgroup per (b,t) row: log-sum-exp of the bf16 logits (streams V+1 values once)
//   ctc_alpha_beta_kernel one workgroup per utterance: both recursions with the 2L+1 states spread over the threads,
//                         state vectors ping-pong in LDS, log alpha / log beta kept in the caller's workspace
//   ctc_grad_kernel       one workgroup per (b,t) row: label posteriors accumulated in an LDS hash table (repeated
//                         labels / the L+1 blanks collide by design), then one dense pass writes the bf16 gradient
#include "common.h"

#define CTC_NEG (-1e30f)
#define CTC_THREADS 1024

__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    return m <= CTC_NEG ? CTC_NEG : m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m <= CTC_NEG ? CTC_NEG : m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__global__ void __launch_bounds__(256) ctc_lse_kernel(const unsigned short* __restrict__ logits, int64_t ld, int C, float* __restrict__ lse) {
    __shared__ float red[4];
    const unsigned short* row = logits + (int64_t)blockIdx.x * ld;
    float m = -INFINITY, s = 0.f;
    for (int v = threadIdx.x; v < C; v += 256) {
        const float x = bfbits2f(row[v]);
        const float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    float gm = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
    __syncthreads();
    gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float gs = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - gm));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs;
    __syncthreads();
    if (threadIdx.x == 0) lse[blockIdx.x] = gm + __logf(red[0] + red[1] + red[2] + red[3]);
}

__global__ void __launch_bounds__(CTC_THREADS) ctc_alpha_beta_kernel(const dicow_ctc_args a) {
    __shared__ float st[2][CTC_THREADS + 2];
    __shared__ int tl_s;
    const int b = blockIdx.x, s = threadIdx.x;
    const int64_t* lab = a.labels + (int64_t)b * a.Lc;
    if (s == 0) {
        int n = 0;
        for (int i = 0; i < a.Lc; ++i) n += lab[i] >= 0;
        tl_s = n;
    }
    __syncthreads();
    const int tl = tl_s, S = 2 * tl + 1, Smax = a.Smax;
    const bool act = s < S;
    const int cls = act ? ((s & 1) ? (int)lab[s >> 1] : a.blank) : a.blank;
    const bool skip_f = act && (s & 1) && s >= 2 && lab[s >> 1] != lab[(s >> 1) - 1];                  // alpha: s-2 -> s
    const bool skip_b = act && (s & 1) && s + 2 < S && lab[s >> 1] != lab[(s >> 1) + 1];               // beta:  s+2 -> s
    const unsigned short* lg = reinterpret_cast<const unsigned short*>(a.logits) + (int64_t)b * a.Tn * a.ld;
    const float* lse = a.lse + (int64_t)b * a.Tn;
    float* la = a.alpha + (int64_t)b * a.Tn * Smax;
    float* lb = a.beta + (int64_t)b * a.Tn * Smax;
    // ---- alpha
    float cur = CTC_NEG;
    if (act && s < 2) cur = bfbits2f(lg[cls]) - lse[0];
    st[0][s + 2] = cur;
    if (s < 2) { st[0][s] = CTC_NEG; st[1][s] = CTC_NEG; }
    if (s < Smax) la[s] = cur;
    __syncthreads();
    for (int t = 1; t < a.Tn; ++t) {
        const float* p = st[(t - 1) & 1];
        float v = CTC_NEG;
        if (act) {
            const float lp = bfbits2f(lg[(int64_t)t * a.ld + cls]) - lse[t];
            v = lse3(p[s + 2], p[s + 1], skip_f ? p[s] : CTC_NEG);
            v = v <= CTC_NEG ? CTC_NEG : v + lp;
        }
        st[t & 1][s + 2] = v;
        if (s < Smax) la[(int64_t)t * Smax + s] = v;
        __syncthreads();
    }
    const float* pl = st[(a.Tn - 1) & 1];
    float nll = 0.f;
    if (s == 0) {
        const float tot = lse2(pl[S - 1 + 2], tl > 0 ? pl[S - 2 + 2] : CTC_NEG);
        nll = tot <= CTC_NEG ? INFINITY : -tot;
        a.nll[b] = nll;
        a.tlen[b] = (float)tl;
        if (isfinite(nll)) atomicAdd(a.loss_sum, nll / (float)(tl > 0 ? tl : 1));        // zero_infinity: inf -> 0
    }
    __syncthreads();
    // ---- beta (state vector shifted by 0, two guard cells on the right)
    cur = CTC_NEG;
    if (act && s >= S - 2) cur = bfbits2f(lg[(int64_t)(a.Tn - 1) * a.ld + cls]) - lse[a.Tn - 1];
    st[0][s] = cur;
    if (s < 2) { st[0][CTC_THREADS + s] = CTC_NEG; st[1][CTC_THREADS + s] = CTC_NEG; }
    if (s < Smax) lb[(int64_t)(a.Tn - 1) * Smax + s] = cur;
    __syncthreads();
    for (int t = a.Tn - 2, it = 1; t >= 0; --t, ++it) {
        const float* p = st[(it - 1) & 1];
        float v = CTC_NEG;
        if (act) {
            const float lp = bfbits2f(lg[(int64_t)t * a.ld + cls]) - lse[t];
            v = lse3(p[s], p[s + 1], skip_b ? p[s + 2] : CTC_NEG);
            v = v <= CTC_NEG ? CTC_NEG : v + lp;
        }
        st[it & 1][s] = v;
        if (s < Smax) lb[(int64_t)t * Smax + s] = v;
        __syncthreads();
    }
}

#define HT 2048
__global__ void __launch_bounds__(256) ctc_grad_kernel(const dicow_ctc_args a, const float* grad_scale) {
    __shared__ int keys[HT];
    __shared__ float vals[HT];
    const int t = blockIdx.x, b = blockIdx.y;
    const int64_t rowi = (int64_t)b * a.Tn + t;
    const unsigned short* row = reinterpret_cast<const unsigned short*>(a.logits) + rowi * a.ld;
    unsigned short* drow = reinterpret_cast<unsigned short*>(a.d_logits) + rowi * a.ld;
    const float nll = a.nll[b], tl = a.tlen[b], lse = a.lse[rowi];
    const bool dead = !isfinite(nll);
    const float sc = dead ? 0.f : grad_scale[0] / ((float)a.B * fmaxf(tl, 1.f));
    for (int i = threadIdx.x; i < HT; i += 256) { keys[i] = -1; vals[i] = 0.f; }
    __syncthreads();
    if (!dead) {
        const int S = 2 * (int)tl + 1;
        const int64_t* lab = a.labels + (int64_t)b * a.Lc;
        const float* la = a.alpha + rowi * a.Smax;
        const float* lb = a.beta + rowi * a.Smax;
        for (int s = threadIdx.x; s < S; s += 256) {
            const int cls = (s & 1) ? (int)lab[s >> 1] : a.blank;
            const float ab = la[s] + lb[s];
            if (ab <= CTC_NEG) continue;
            const float lp = bfbits2f(row[cls]) - lse;
            const float post = __expf(ab - lp + nll);
            unsigned h = ((unsigned)cls * 2654435761u) >> 21;
            for (;;) {
                const int old = atomicCAS(&keys[h], -1, cls);
                if (old == -1 || old == cls) { atomicAdd(&vals[h], post); break; }
                h = (h + 1) & (HT - 1);
            }
        }
    }
    __syncthreads();
    for (int v = threadIdx.x * 2; v < a.ld; v += 512) {
        float g[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = v + e;
            float gv = 0.f;
            if (c < a.C && !dead) {
                float post = 0.f;
                unsigned h = ((unsigned)c * 2654435761u) >> 21;
                for (;;) {
                    const int k = keys[h];
                    if (k == c) { post = vals[h]; break; }
                    if (k == -1) break;
                    h = (h + 1) & (HT - 1);
                }
                gv = sc * (__expf(bfbits2f(row[c]) - lse) - post);
            }
            g[e] = gv;
        }
        *reinterpret_cast<unsigned*>(drow + v) = pack_bf16x2(g[0], g[1]);
    }
}

extern "C" int64_t dicow_ctc_ws_bytes(int B, int Tn, int Lc) { return (int64_t)2 * B * Tn * (2 * Lc + 1) * 4; }

static int ctc_check(const dicow_ctc_args* a) {
    DICOW_REQUIRE(a && a->logits && a->labels && a->lse && a->alpha && a->beta && a->nll && a->tlen && a->loss_sum, "ctc: null argument");
    DICOW_REQUIRE(a->B > 0 && a->Tn > 0 && a->C > 1 && a->Lc > 0 && a->ld >= a->C && a->ld % 2 == 0, "ctc: bad shape");
    DICOW_REQUIRE(a->Smax == 2 * a->Lc + 1 && a->Smax <= CTC_THREADS, "ctc: at most %d labels per utterance", (CTC_THREADS - 1) / 2);
    DICOW_REQUIRE(a->blank >= 0 && a->blank < a->C, "ctc: blank out of range");
    return DICOW_OK;
}

extern "C" int dicow_ctc_loss_fwd(const dicow_ctc_args* a, void* stream) {
    if (int rc = ctc_check(a)) return rc;
    hipLaunchKernelGGL(ctc_lse_kernel, dim3(a->B * a->Tn), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)a->logits,
                       a->ld, a->C, a->lse);
    DICOW_CHECK_LAUNCH("ctc_lse");
    hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(a->B), dim3(CTC_THREADS), 0, (hipStream_t)stream, *a);
    DICOW_CHECK_LAUNCH("ctc_alpha_beta");
    return DICOW_OK;
}

extern "C" int dicow_ctc_loss_bwd(const dicow_ctc_args* a, const float* grad_scale, void* stream) {
    if (int rc = ctc_check(a)) return rc;
    DICOW_REQUIRE(a->d_logits && grad_scale, "ctc_loss_bwd: null d_logits / grad_scale");
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(a->Tn, a->B), dim3(256), 0, (hipStream_t)stream, *a, grad_scale);
    DICOW_CHECK_LAUNCH("ctc_grad");
    return DICOW_OK;
}
