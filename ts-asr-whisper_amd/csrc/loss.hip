// Fused log-softmax + cross-entropy over the tied LM-head logits, forward and backward.  HBM-bound.
//
// Replaces (reference): hard-label fallback src/models/dicow/modeling_dicow.py:310-323 and the soft-label loss
// SoftLabelCreator.compute_loss :95-144 (Gaussian-smoothed timestamp rows :74-93 built by :35-72).  The reference
// materialises two dense [B*L, V] fp32 one-hot/soft-target matrices; here a row's target is implicit:
//   ordinary token  : one-hot(label)
//   timestamp token : ts_w[ts_index[label], :] scattered at ts_ids        (ts_index = -1 for non-timestamp ids)
// One workgroup per (b, l) row streams the V logits once (bf16, 16 B/lane):
//   fwd: lse, CE(lower), CE(upper), per-row min, argmin      bwd: d_logits = scale_row * (softmax - target_sel)
// Algorithmic bytes per row: fwd 2V, bwd 2V + 2V.
#include "common.h"

#define LOSS_THREADS 256

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) s = fmaxf(s, red[w]);
    return s;
}

__device__ __forceinline__ float ld_logit(const unsigned short* row, int v) { return bfbits2f(row[v]); }

// sum_v target(v) * logit(v) for one label
__device__ float target_dot(const unsigned short* row, int64_t label, const dicow_ce_args& a, float* red) {
    if (label < 0) return 0.f;
    const int ti = a.ts_index ? a.ts_index[label] : -1;
    if (ti < 0) return ld_logit(row, (int)label);            // uniform across the block
    float s = 0.f;
    const float* w = a.ts_w + (int64_t)ti * a.n_ts;
    for (int j = threadIdx.x; j < a.n_ts; j += LOSS_THREADS) s += w[j] * ld_logit(row, a.ts_ids[j]);
    return block_reduce_sum(s, red);
}

// The step's loss is the sum of the row losses IN ROW ORDER: ce_sum_kernel, a one-workgroup launch that follows ce_fwd_kernel on the
// same stream, adds row_loss[] up with a fixed assignment of rows to threads and a fixed tree, so the logged loss is
// bit-reproducible from run to run (a float atomicAdd per row made its last bits depend on the order the workgroups retired in).
// (Round 3 had the LAST workgroup of ce_fwd_kernel do this, elected through a __device__ global ticket: two launches in flight on
// one device -- two streams, a graph replay beside an eager step, a second model -- shared that counter and could drop or
// truncate the sum, and an aborted launch poisoned every later one.  Stream order needs no shared state.)
__global__ void __launch_bounds__(LOSS_THREADS) ce_sum_kernel(const float* row_loss, int rows, float* loss_sum) {
    __shared__ float tot[LOSS_THREADS];
    float t = 0.f;
    for (int i = threadIdx.x; i < rows; i += LOSS_THREADS) t += row_loss[i];
    tot[threadIdx.x] = t;
    __syncthreads();
    for (int o = LOSS_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) tot[threadIdx.x] += tot[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] += tot[0];
}

__global__ void __launch_bounds__(LOSS_THREADS) ce_fwd_kernel(const dicow_ce_args a) {
    __shared__ float red[LOSS_THREADS / 64];
    const int r = blockIdx.x;
    const unsigned short* row = reinterpret_cast<const unsigned short*>(a.logits) + (int64_t)r * a.ld;
    // online max / sum-exp, 8 bf16 per load
    float m = -INFINITY, s = 0.f;
    const int nv8 = a.V >> 3;
    for (int i = threadIdx.x; i < nv8; i += LOSS_THREADS) {
        const uint4 u = reinterpret_cast<const uint4*>(row)[i];
        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w4[e] << 16); x[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
        float mx = x[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) mx = fmaxf(mx, x[e]);
        const float mn = fmaxf(m, mx);
        float acc = s * __expf(m - mn);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += __expf(x[e] - mn);
        s = acc; m = mn;
    }
    for (int v = (nv8 << 3) + threadIdx.x; v < a.V; v += LOSS_THREADS) {
        const float x = ld_logit(row, v);
        const float mn = fmaxf(m, x);
        s = s * __expf(m - mn) + __expf(x - mn);
        m = mn;
    }
    const float gm = block_reduce_max(m, red);
    const float gs = block_reduce_sum(m == -INFINITY ? 0.f : s * __expf(m - gm), red);
    const float lse = gm + __logf(gs);

    int64_t lab = a.labels[r];
    int64_t upp = a.upp_labels ? a.upp_labels[r] : lab;
    // labels outside [0, V) other than the ignore index: torch's CrossEntropyLoss / one_hot raise; here the row loss is
    // poisoned (NaN -> the step's loss is NaN) instead of reading logits / timestamp tables out of bounds
    const bool bad = (lab != -100 && (lab < 0 || lab >= a.V)) || (upp != -100 && (upp < 0 || upp >= a.V));
    if (bad) { lab = -100; upp = -100; }
    const bool valid_lo = lab != -100;
    // soft path: the reference clamps an ignored label to token 0 and masks BOTH losses by the lower-case labels' padding
    // only (SoftLabelCreator._get_soft_distribution / compute_loss, modeling_dicow.py:79,130-136): a row whose upper-case
    // label alone is -100 competes with -log p(token 0)
    if (a.soft && valid_lo && upp == -100) upp = 0;
    const bool valid_up = upp != -100;
    float lo = valid_lo ? lse - target_dot(row, lab, a, red) : 0.f;
    float up = lo;
    if (a.upp_labels) up = valid_up ? lse - target_dot(row, upp, a, red) : 0.f;
    if (a.soft && !valid_lo) { lo = 0.f; up = 0.f; }
    if (bad) lo = up = __int_as_float(0x7fc00000);
    const int choice = (up < lo) ? 1 : 0;
    const float l = choice ? up : lo;
    if (threadIdx.x == 0) {
        a.lse[r] = lse;
        a.row_loss[r] = l;
        a.choice[r] = choice;
        if (valid_lo) atomicAdd(a.count, 1.0f);           // (a count of ones: exact in any order)
    }
}

__global__ void __launch_bounds__(LOSS_THREADS) ce_bwd_kernel(const dicow_ce_args a, const float* grad_scale) {
    const int r = blockIdx.x;
    const unsigned short* row = reinterpret_cast<const unsigned short*>(a.logits) + (int64_t)r * a.ld;
    unsigned short* drow = reinterpret_cast<unsigned short*>(a.d_logits) + (int64_t)r * a.ld;
    const int64_t lab = a.labels[r];
    int64_t upp = a.upp_labels ? a.upp_labels[r] : lab;
    if (a.soft && lab != -100 && upp == -100) upp = 0;          // as in the forward: the reference's clamp of an ignored label
    const int64_t sel = a.choice[r] ? upp : lab;
    bool active = sel != -100 && sel >= 0 && sel < a.V;         // out-of-range labels poisoned the loss in the forward
    if (a.soft && lab == -100) active = false;
    const float sc = active ? grad_scale[0] : 0.f;
    const float lse = a.lse[r];
    const int ti = (active && a.ts_index) ? a.ts_index[sel] : -1;
    const float* w = ti >= 0 ? a.ts_w + (int64_t)ti * a.n_ts : nullptr;
    for (int v = threadIdx.x * 2; v < a.ld; v += LOSS_THREADS * 2) {
        float g[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int vv = v + e;
            float gv = 0.f;
            if (vv < a.V && active) {
                const float p = __expf(ld_logit(row, vv) - lse);
                float t;
                if (w) { const int j = a.ts_index[vv]; t = j >= 0 ? w[j] : 0.f; }
                else t = (vv == (int)sel) ? 1.f : 0.f;
                gv = sc * (p - t);
            }
            g[e] = gv;
        }
        *reinterpret_cast<unsigned*>(drow + v) = pack_bf16x2(g[0], g[1]);
    }
}

extern "C" int dicow_ce_loss_fwd(const dicow_ce_args* a, void* stream) {
    DICOW_REQUIRE(a && a->logits && a->labels && a->lse && a->row_loss && a->choice && a->loss_sum && a->count,
                  "ce_loss_fwd: null operand");
    DICOW_REQUIRE(a->rows > 0 && a->V > 0 && a->ld >= a->V && a->ld % 8 == 0, "ce_loss_fwd: need ld >= V and ld %% 8 == 0");
    DICOW_REQUIRE(a->n_ts == 0 || (a->ts_index && a->ts_ids && a->ts_w), "ce_loss_fwd: timestamp tables missing");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(a->rows), dim3(LOSS_THREADS), 0, (hipStream_t)stream, *a);
    DICOW_CHECK_LAUNCH("ce_loss_fwd");
    hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, a->row_loss, a->rows, a->loss_sum);
    DICOW_CHECK_LAUNCH("ce_loss_fwd (ordered sum)");
    return DICOW_OK;
}

extern "C" int dicow_ce_loss_bwd(const dicow_ce_args* a, const float* grad_scale, void* stream) {
    DICOW_REQUIRE(a && a->logits && a->labels && a->lse && a->choice && a->d_logits && grad_scale, "ce_loss_bwd: null operand");
    DICOW_REQUIRE(a->rows > 0 && a->V > 0 && a->ld >= a->V && a->ld % 8 == 0, "ce_loss_bwd: need ld >= V and ld %% 8 == 0");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(a->rows), dim3(LOSS_THREADS), 0, (hipStream_t)stream, *a, grad_scale);
    DICOW_CHECK_LAUNCH("ce_loss_bwd");
    return DICOW_OK;
}
