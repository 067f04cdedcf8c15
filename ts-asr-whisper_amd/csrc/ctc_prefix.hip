// CTC prefix scoring for joint CTC / attention decoding.
//
// Replaces CTCPrefixScore of the reference (src/models/dicow/decoding.py:8-163, derived from ESPnet; Watanabe et al. 2017
// algorithm 2, vectorised over hypotheses and candidate labels as in Seki et al. 2019).  The reference walks the T encoder
// frames in a Python loop of five torch kernels per frame -- ~2000 launches per decoded token at T = 375 -- over
// [n_hyp, T, n_cand] tensors.  Here one thread owns one (hypothesis, candidate label) pair and runs the frame recursion
//     r_n[t] = logaddexp(r_n[t-1], phi[t-1]) + x[t, c]          phi = r_b(g) if c repeats the prefix's last label, else r_n(g) (+) r_b(g)
//     r_b[t] = logaddexp(r_n[t-1], r_b[t-1]) + x[t, blank]
//     psi    = logaddexp(r_n[start-1], logsumexp_{t >= max(d,1)} (phi[t-1] + x[t, c]))
// in registers; the per-hypothesis rows (phi, r_n (+) r_b, blank and normaliser columns) are staged once in LDS, the
// candidate's column of the frame-major log-probabilities is prefetched CPS_PF frames ahead (scattered 2/4-byte reads, one
// HBM sector each: the recursion is bound by that latency, hence the deep prefetch and the small 64-thread workgroups that
// spread a step's ~100 workgroups over the chip), and the new states stream out coalesced over the candidate index.
// Log-probabilities are not materialised: x[t, c] = logit[t, alias[c]] - lse[t] with the per-frame normaliser from
// ctc_frame_lse_kernel (the reference's log_softmax + upper-case aliasing, decoding.py:183-186).
// fp32 throughout; logzero = -1e10 exactly as the reference (adding a log-probability leaves it unchanged in fp32).
#include "common.h"

#define CPS_LOGZERO (-1e10f)
#define CPS_PF 16           // frames of look-ahead on the gathered column
#define CPS_BLOCK 64

__device__ __forceinline__ float lae(float a, float b) {          // torch.logaddexp
    const float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}

// The frame recursion is one dependent chain per thread, so its speed is the latency of two logaddexp per frame: the libm
// forms (~100 instructions each) made a T = 375 call 440 us.  Hardware exp2 / log2 (1 ulp of the result's log2) bring the
// chain to ~10 instructions; the absolute error per evaluation is ~1e-7 and the states stay within 1e-4 of the reference.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float lae_fast(float a, float b) {
    const float m = fmaxf(a, b);
    return m + __builtin_amdgcn_logf(1.f + fast_exp(-fabsf(a - b))) * 0.693147180559945309f;
}

template <int BF>
__device__ __forceinline__ float ld_logit_t(const void* p, int64_t idx) {
    if (BF) {
        const unsigned short u = reinterpret_cast<const unsigned short*>(p)[idx];
        return __uint_as_float((unsigned)u << 16);
    }
    return reinterpret_cast<const float*>(p)[idx];
}

__device__ __forceinline__ float ld_logit(const void* p, int in_bf16, int64_t idx) {
    if (in_bf16) {
        const unsigned short u = reinterpret_cast<const unsigned short*>(p)[idx];
        return __uint_as_float((unsigned)u << 16);
    }
    return reinterpret_cast<const float*>(p)[idx];
}

// one workgroup per frame row: lse[row] = log sum_v exp(logit[row, v]), v < V1
__global__ void __launch_bounds__(256) ctc_frame_lse_kernel(const void* __restrict__ logits, int in_bf16, int V1, int64_t ld,
                                                            float* __restrict__ lse) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    float m = -INFINITY;
    for (int v = tid; v < V1; v += 256) m = fmaxf(m, ld_logit(logits, in_bf16, row * ld + v));
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int v = tid; v < V1; v += 256) s += expf(ld_logit(logits, in_bf16, row * ld + v) - m);
    s = wave_sum(s);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
    __syncthreads();
    if (tid == 0) lse[row] = m + logf(red[4] + red[5] + red[6] + red[7]);
}

// state of the empty prefix (decoding.py:36-43): r[b, t, 0] = logzero, r[b, t, 1] = sum_{u <= t} x[b, u, blank]
__global__ void ctc_prefix_init_kernel(const void* __restrict__ logits, int in_bf16, int64_t ld, const float* __restrict__ lse,
                                       int T, int blank_col, float* __restrict__ r0) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const int64_t row = (int64_t)b * T + t;
        acc += ld_logit(logits, in_bf16, row * ld + blank_col) - lse[row];
        r0[row * 2] = CPS_LOGZERO;
        r0[row * 2 + 1] = acc;
    }
}

template <int BF>
__global__ void __launch_bounds__(CPS_BLOCK) ctc_prefix_score_kernel(const dicow_ctc_prefix_args a) {
    extern __shared__ float sm[];
    const int T = a.T, C = a.C, i = blockIdx.y, tid = threadIdx.x;
    float* phi_rep = sm;                  // r_b(g)            : label repeats the last one
    float* phi_any = sm + T;              // r_n(g) (+) r_b(g) : any other label
    float* xb = sm + 2 * T;               // x[t, blank]
    float* nrm = sm + 3 * T;              // per-frame normaliser
    const int64_t frame0 = (int64_t)a.rows[i] * T;
    const int blank_col = a.alias ? a.alias[a.blank] : a.blank;
    for (int t = tid; t < T; t += CPS_BLOCK) {
        const float rn = a.r_prev[((int64_t)i * T + t) * 2], rb = a.r_prev[((int64_t)i * T + t) * 2 + 1];
        const float z = a.lse[frame0 + t];
        phi_rep[t] = rb;
        phi_any[t] = lae(rn, rb);
        nrm[t] = z;
        xb[t] = ld_logit(a.logits, a.in_bf16, (frame0 + t) * a.ld + blank_col) - z;
    }
    __syncthreads();
    const int c = blockIdx.x * CPS_BLOCK + tid;
    if (c >= C) return;
    const int lab = a.cs[(int64_t)i * C + c];
    const int col = a.alias ? a.alias[lab] : lab;
    const int d = a.decoded_len[i];
    const float* phi = (d > 0 && lab == a.last[i]) ? phi_rep : phi_any;
    const int64_t base = frame0 * a.ld + col;
    float* rout = a.r + (int64_t)i * T * 2 * C + c;

    float rn = d == 0 ? ld_logit(a.logits, a.in_bf16, base) - nrm[0] : CPS_LOGZERO, rb = CPS_LOGZERO;
    const float psi0 = rn;                                   // r_n[start - 1] before the recursion (decoding.py:88-91)
    rout[0] = rn;
    rout[C] = rb;
    float mx = -INFINITY, sum = 0.f;                         // running logsumexp of phi[t-1] + x[t, c], t >= d
    // one frame of the recursion
#define CPS_STEP(TT, RAW)                                                                                    \
    {                                                                                                        \
        const int t_ = (TT);                                                                                 \
        const float xs = (RAW) - nrm[t_], ph = phi[t_ - 1];                                                  \
        const float term = t_ >= d ? ph + xs : CPS_LOGZERO;                                                  \
        const float m2_ = fmaxf(mx, term);                                                                   \
        sum = sum * fast_exp(mx - m2_) + fast_exp(term - m2_);                                               \
        mx = m2_;                                                                                            \
        const float rn2 = lae_fast(rn, ph) + xs, rb2 = lae_fast(rn, rb) + xb[t_];                            \
        rn = rn2;                                                                                            \
        rb = rb2;                                                                                            \
        rout[(int64_t)t_ * 2 * C] = rn;                                                                      \
        rout[(int64_t)t_ * 2 * C + C] = rb;                                                                  \
    }
    // The gathered column is read a whole chunk of CPS_PF frames ahead (clamped at the end): the loads of chunk k+1 are in
    // flight while chunk k is consumed, so the counted vmcnt waits never drain the queue (loads and stores retire in order).
    float cur[CPS_PF], nxt[CPS_PF];
    const int64_t last_frame = (int64_t)(T - 1) * a.ld;
    int t = 1;
#pragma unroll
    for (int u = 0; u < CPS_PF; ++u) cur[u] = ld_logit_t<BF>(a.logits, base + min((int64_t)(t + u) * a.ld, last_frame));
    for (; t + CPS_PF <= T; t += CPS_PF) {
#pragma unroll
        for (int u = 0; u < CPS_PF; ++u) nxt[u] = ld_logit_t<BF>(a.logits, base + min((int64_t)(t + CPS_PF + u) * a.ld, last_frame));
#pragma unroll
        for (int u = 0; u < CPS_PF; ++u) CPS_STEP(t + u, cur[u])
#pragma unroll
        for (int u = 0; u < CPS_PF; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int u = 0; u < CPS_PF; ++u)
        if (t + u < T) CPS_STEP(t + u, cur[u])
#undef CPS_STEP
    float psi = T > 1 ? lae(psi0, mx + logf(sum)) : psi0;
    if (lab == a.eos) psi = phi_any[T - 1];                   // P(prefix ends here), decoding.py:111-114
    else if (lab == a.blank && a.eos != a.blank) psi = CPS_LOGZERO;
    a.psi[(int64_t)i * C + c] = psi;
}

extern "C" int dicow_ctc_frame_lse(const void* logits, int in_bf16, int64_t rows, int V1, int64_t ld, float* lse, void* stream) {
    DICOW_REQUIRE(logits && lse && rows > 0 && V1 > 0 && ld >= V1 && rows < (1ll << 31), "ctc_frame_lse: bad args rows=%lld V1=%d ld=%lld",
                  (long long)rows, V1, (long long)ld);
    ctc_frame_lse_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(logits, in_bf16, V1, ld, lse);
    DICOW_CHECK_LAUNCH("ctc_frame_lse_kernel");
    return DICOW_OK;
}

extern "C" int dicow_ctc_prefix_init(const void* logits, int in_bf16, int64_t ld, const float* lse, int B, int T, int blank_col,
                                     float* r0, void* stream) {
    DICOW_REQUIRE(logits && lse && r0 && B > 0 && T > 0 && blank_col >= 0 && blank_col < ld, "ctc_prefix_init: bad args");
    ctc_prefix_init_kernel<<<B, 64, 0, (hipStream_t)stream>>>(logits, in_bf16, ld, lse, T, blank_col, r0);
    DICOW_CHECK_LAUNCH("ctc_prefix_init_kernel");
    return DICOW_OK;
}

extern "C" int dicow_ctc_prefix_score(const dicow_ctc_prefix_args* a, void* stream) {
    DICOW_REQUIRE(a, "ctc_prefix_score: null argument block");
    DICOW_REQUIRE(a->n >= 0 && a->C > 0 && a->T > 0 && a->T <= 8192, "ctc_prefix_score: bad sizes n=%d C=%d T=%d", a->n, a->C, a->T);
    if (a->n == 0) return DICOW_OK;                       // no active hypothesis: nothing to score (empty tensors have no storage)
    DICOW_REQUIRE(a->logits && a->lse && a->rows && a->cs && a->decoded_len && a->last && a->r_prev && a->psi && a->r,
                  "ctc_prefix_score: null pointer");
    DICOW_REQUIRE(a->blank >= 0 && a->blank < a->ld && a->eos >= 0, "ctc_prefix_score: blank %d / eos %d out of range", a->blank, a->eos);
    const size_t lds = (size_t)a->T * 4 * sizeof(float);
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)ctc_prefix_score_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
        (void)hipFuncSetAttribute((const void*)ctc_prefix_score_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
        return true;
    }();
    (void)attr;
    const dim3 grid((a->C + CPS_BLOCK - 1) / CPS_BLOCK, a->n);
    if (a->in_bf16) ctc_prefix_score_kernel<1><<<grid, CPS_BLOCK, lds, (hipStream_t)stream>>>(*a);
    else ctc_prefix_score_kernel<0><<<grid, CPS_BLOCK, lds, (hipStream_t)stream>>>(*a);
    DICOW_CHECK_LAUNCH("ctc_prefix_score_kernel");
    return DICOW_OK;
}

// ------------------------------------------------------------------------------------------------ timestamp rules
// Whisper's timestamp constraints on one row of next-token scores (reference src/models/dicow/utils.py:5-14 over
// transformers' WhisperTimeStampLogitsProcessor): the reference loops over the batch on the host (`.tolist()` per row) around
// ~10 masked assignments and a log_softmax; here one workgroup per row derives the three facts it needs from the generated
// suffix (was the last / the one before a timestamp, the last timestamp emitted), evaluates the mask per label, reduces
// max(text) and logsumexp(timestamps) in one pass, and writes the row once.
#define TSR_BLOCK 1024
__global__ void __launch_bounds__(TSR_BLOCK) timestamp_rules_kernel(float* __restrict__ scores, int64_t ld, int V, const int64_t* __restrict__ ids,
                                                              int L, int begin, int ts0, int eos, int no_ts, int max_init, int detect) {
    __shared__ float red_m[TSR_BLOCK / 64], red_t[TSR_BLOCK / 64], red_s[TSR_BLOCK / 64];
    __shared__ int facts[3];
    const int k = blockIdx.x, tid = threadIdx.x;
    float* row = scores + (int64_t)k * ld;
    const int64_t* seq = ids + (int64_t)k * L;
    if (tid == 0) {
        const int n = L - begin;
        const int last_ts = n >= 1 && seq[L - 1] >= ts0;
        const int pen_ts = n < 2 || seq[L - 2] >= ts0;
        int upto = -1;                                    // labels in [ts0, upto) are forbidden (timestamps do not decrease)
        for (int j = L - 1; j >= begin; --j)
            if (seq[j] >= ts0) { upto = (int)seq[j] + ((last_ts && !pen_ts) ? 0 : 1); break; }
        facts[0] = last_ts; facts[1] = pen_ts; facts[2] = upto;
    }
    __syncthreads();
    const int last_ts = facts[0], pen_ts = facts[1], upto = facts[2];
    const bool first = L == begin;
    const float eos_score = (first && eos < V) ? row[eos] : 0.f;
    auto banned = [&](int v) {
        if (v == no_ts) return true;
        if (last_ts && (pen_ts ? v >= ts0 : v < eos)) return true;
        if (v >= ts0 && v < upto) return true;
        if (first && (v < ts0 || (max_init >= 0 && v > ts0 + max_init))) return true;
        return false;
    };
    // max over text labels, online logsumexp over timestamp labels (both after the rules above)
    float mt = -INFINITY, tm = -INFINITY, tsum = 0.f;
    for (int v = tid; v < V; v += TSR_BLOCK) {
        if (banned(v)) continue;
        const float s = row[v];
        if (v < ts0) mt = fmaxf(mt, s);
        else if (s > tm) { tsum = tsum * expf(tm - s) + 1.f; tm = s; }
        else tsum += expf(s - tm);
    }
    const float wm = wave_max(tm);
    tsum = wave_sum(tm == -INFINITY ? 0.f : tsum * expf(tm - wm));
    mt = wave_max(mt);
    if ((tid & 63) == 0) { red_m[tid >> 6] = mt; red_t[tid >> 6] = wm; red_s[tid >> 6] = tsum; }
    __syncthreads();
    mt = red_m[0];
    tm = red_t[0];
    for (int w = 1; w < TSR_BLOCK / 64; ++w) { mt = fmaxf(mt, red_m[w]); tm = fmaxf(tm, red_t[w]); }
    float tot = 0.f;
    for (int w = 0; w < TSR_BLOCK / 64; ++w) tot += red_t[w] == -INFINITY ? 0.f : red_s[w] * expf(red_t[w] - tm);
    const float ts_lse = tm == -INFINITY ? -INFINITY : tm + logf(tot);
    const bool only_ts = detect && ts_lse > mt;           // the normaliser of log_softmax cancels in the comparison
    for (int v = tid; v < V; v += TSR_BLOCK)
        if (banned(v) || (only_ts && v < ts0)) row[v] = -INFINITY;
    if (first && eos < V) {                               // utils.py:10-12: a silent window may end immediately
        __syncthreads();
        if (tid == 0) row[eos] = eos_score;
    }
}

extern "C" int dicow_whisper_timestamp_rules(float* scores, int64_t ld, int B, int V, const int64_t* input_ids, int L, int begin_index,
                                             int timestamp_begin, int eos, int no_timestamps, int max_initial_timestamp_index,
                                             int detect_from_logprob, void* stream) {
    DICOW_REQUIRE(scores && input_ids && B > 0 && V > 0 && ld >= V && L >= begin_index && begin_index >= 0,
                  "whisper_timestamp_rules: bad args B=%d V=%d L=%d begin=%d", B, V, L, begin_index);
    DICOW_REQUIRE(timestamp_begin > 0 && timestamp_begin <= V && no_timestamps >= 0 && eos >= 0, "whisper_timestamp_rules: bad token ids");
    timestamp_rules_kernel<<<B, TSR_BLOCK, 0, (hipStream_t)stream>>>(scores, ld, V, input_ids, L, begin_index, timestamp_begin, eos, no_timestamps,
                                                               max_initial_timestamp_index, detect_from_logprob);
    DICOW_CHECK_LAUNCH("timestamp_rules_kernel");
    return DICOW_OK;
}
