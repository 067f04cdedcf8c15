/*
 * libdicow_hip.so -- C ABI of the MI355X-native (gfx950 / CDNA4) DiCoW / SE-DiCoW training-step hot path.
 *
 * The reference (BUTSpeechFIT/TS-ASR-Whisper) has NO native/FFI layer: its seam is the Python module surface
 * (SURVEY.md section 8b).  This header is therefore build-defined; every entry point cites the reference
 * arithmetic (file:line under /root/reference, or HF: = transformers' modeling_whisper.py that the reference
 * subclasses) that it replaces.  The Python host side (the .py files under ts-asr-whisper_amd/) binds these with ctypes and
 * keeps the reference's module names, constructor/forward signatures and state-dict keys.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch's caching
 *     allocator in practice).  The library never allocates, frees or retains device memory.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it; entry points are re-entrant and
 *     may be called from any host thread (autograd runs backward on its own thread).
 *   - return 0 on success, negative DICOW_ERR_* otherwise; dicow_last_error() gives a thread-local message.
 *   - bf16 tensors are passed as void* (raw 16-bit storage), row-major, rows 16-byte aligned.
 *   - "rows" = B*T flattened (batch-major); the fp32 residual stream is [rows, D].
 *   - STNO masks are fp32 [B,4,T] with channel order 0=silence 1=target 2=non-target 3=overlap
 *     (src/data/local_datasets.py:184-194); `stno_bstride` is the element stride between batch rows so that
 *     interleaved SE-DiCoW batches (encoder.py:152-154, 210-213) need no copy.
 */
#ifndef DICOW_HIP_H
#define DICOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DICOW_OK 0
#define DICOW_ERR_INVALID (-1)  /* bad argument / unsupported shape */
#define DICOW_ERR_LAUNCH (-2)   /* HIP launch failure */

#define DICOW_ABI_VERSION 7

int dicow_abi_version(void);
/* Number of CUs the persistent NT GEMM may occupy (0 = all, the default).  Its workgroups own a whole CU each for the
 * length of the launch; when a communication kernel (RCCL: one workgroup per channel) runs beside the step, leave it
 * that many CUs -- otherwise the GEMM workgroups that find their CU taken start only when another one has finished its
 * whole tile list.  Returns the previous value. */
int dicow_set_gemm_cus(int n);
/* Names (as a profiler prints them) and launch counts of the GEMM kernel instantiations this process has dispatched,
 * "name\tcount\n" per line; returns the number of bytes written (call with buf = NULL to size it).  Measurement support:
 * bench.py quotes a committed rocprofv3 counter summary only for kernels that occur in this list. */
int dicow_gemm_dispatch_log(char* buf, int cap);
const char* dicow_last_error(void);

/* ------------------------------------------------------------------------------------------------ casts
 * AMP weight preparation (configs/base.yaml:49 `bf16: true`): fp32 master -> bf16 compute copies.      */
int dicow_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* [R,C] fp32 -> bf16 [R,C] (dst, row stride ld; may be NULL) and bf16 [C,R] (dst_t, row stride ld_t; may be NULL):
 * the transposed copy is the dgrad operand of every Linear; the strides let q/k/v land in one fused [3D,D] / [D,3D]. */
int dicow_cast_transpose_f32_to_bf16(const float* src, void* dst, int64_t ld, void* dst_t, int64_t ld_t, int R, int C,
                                     void* stream);
/* The same for up to 8 matrices in ONE launch (AMP's re-cast of a layer's weights after every optimizer step). */
#define DICOW_CAST_GROUP_MAX 8
typedef struct { const float* src; void* dst; void* dst_t; int R, C; int64_t ld, ld_t; } dicow_cast_problem;
int dicow_cast_transpose_group(const dicow_cast_problem* p, int n, void* stream);
/* Conv1d weight [O,C,3] fp32 -> bf16 [O,Kpad] (dst) and its transpose [Kpad,O] (dst_t; either may be NULL),
 * k = tap*C + c (tap-major), zero padded to Kpad >= 3C: the GEMM view of conv1/conv2 (encoder.py:167-168). */
int dicow_conv_weight_pack(const float* w, void* dst, void* dst_t, int O, int C, int Kpad, void* stream);
/* inverse mapping for the weight gradient: [O,Kpad] fp32 (tap-major) accumulated into [O,C,3] fp32 */
int dicow_conv_weight_unpack_grad(const float* g_packed, float* g_w, int O, int C, int Kpad, void* stream);
/* input_features [B,M,Tin] fp32 -> time-major bf16 [B, Tin+2, M], rows 0 and Tin+1 zero (conv padding=1). */
int dicow_mel_to_timemajor(const float* mel, void* dst, int B, int M, int Tin, void* stream);
/* column sums of a bf16 [rows, N] matrix accumulated (+=) into fp32 out[N] (bias gradients).  Two-stage (partials in
 * the caller's workspace, then a reduce pass): fp32 L2 atomics are ~20x slower than that on gfx950. */
int64_t dicow_colsum_ws_bytes(int rows, int N);
int dicow_colsum_bf16(const void* x, int64_t ld, float* out, int rows, int N, void* ws, int64_t ws_bytes, void* stream);
/* out[t,:] += sum_b g[b,t,:]   (gradient of encoder.embed_positions, encoder.py:177-179) */
int dicow_sum_over_batch(const float* g, float* out, int B, int64_t TD, void* stream);

/* Element-wise glue of the SE-DiCoW speaker-communication block (reference src/models/dicow/layers.py:145-193) on the interleaved
 * row layout [Bp][2][T][D] (slot 0 = mixture, slot 1 = enrollment; encoder.py:152-154):
 *   split      hf fp32 -> q_in / kv_in bf16 [Bp*T, D] and the right half of cat bf16 [Bp*T, ldcat] (= q_in, layers.py:161)
 *   merge_fwd  out = hf; out[mixture] += tanh(gate[0]) * upd                                            (layers.py:186-191)
 *   gate_bwd   d_upd = bf16(g[mixture] * tanh(gate));  d_gate[0] += (1 - tanh^2) * sum(g[mixture] * upd)  (d_gate may be NULL)
 *   merge_bwd  gin = g; gin[mixture] += d_qin + d_cat[:, D:2D]; gin[enrollment] += d_kvin */
int dicow_scb_split(const float* hf, void* q_in, void* kv_in, void* cat, int64_t ldcat, int Bp, int T, int D, void* stream);
int dicow_scb_merge_fwd(const float* hf, const void* upd, const float* gate, float* out, int Bp, int T, int D, void* stream);
int64_t dicow_scb_gate_bwd_ws_bytes(void);
int dicow_scb_gate_bwd(const float* g, const void* upd, const float* gate, void* d_upd, float* d_gate, int Bp, int T, int D,
                       void* ws, int64_t ws_bytes, void* stream);
int dicow_scb_merge_bwd(const float* g, const float* d_qin, const void* d_cat, int64_t ldcat, const float* d_kvin, float* gin,
                        int Bp, int T, int D, void* stream);

/* ------------------------------------------------------------------------------------------------ FDDT + LayerNorm
 * Fused row kernel.  Replaces FDDT.forward (src/models/dicow/FDDT.py:41-63) with CustomDiagonalLinear
 * (layers.py:73-77), the optional position-embedding add (encoder.py:177-179) and the LayerNorm that follows
 * in WhisperEncoderLayer (HF:modeling_whisper.py:392,402).
 *   mode 0: no FDDT (plain LayerNorm / copy)
 *   mode 1: diagonal FDDT  h' = ((wS*h+bS)*mS + (wT*h+bT)*mT) + (wN*h+bN)*mN + (wO*h+bO)*mO   -- evaluated in
 *           exactly this order with separate fp32 mul/add (no FMA contraction) => bit-exact vs the fp32 reference;
 *           a NULL w[c] means the class is disabled (identity branch, FDDT.py:55-62)
 *   mode 2: bias-only FDDT h' = h + mS*bS + mT*bT + mN*bN + mO*bO (FDDT.py:43-51), NULL b[c] = disabled
 * Class order in w[]/b[] is S,T,N,O (the stno channel order).                                             */
typedef struct {
    const void* h_in;        /* [rows,D] fp32, or bf16 when in_bf16 != 0 */
    int in_bf16;
    int mode;
    const float* stno;       /* [B,4,T] fp32 (mode != 0) */
    int64_t stno_bstride;    /* elements between consecutive batch entries (4*T when dense) */
    const float* w[4];
    const float* b[4];
    const float* pos;        /* [T,D] fp32 added after the FDDT, or NULL */
    float* h_out;            /* [rows,D] fp32 post-FDDT(+pos) residual stream, or NULL */
    const float* ln_w;       /* LayerNorm affine; NULL => no LayerNorm (h_out only) */
    const float* ln_b;
    void* y_bf16;            /* [rows,D] bf16 LayerNorm output, or NULL */
    float* y_f32;            /* [rows,D] fp32 LayerNorm output, or NULL */
    float* mean;             /* [rows] saved statistics (may be NULL) */
    float* rstd;
    int rows, T, D;
    float eps;
} dicow_fddt_ln_fwd_args;
int dicow_fddt_ln_fwd(const dicow_fddt_ln_fwd_args* a, void* stream);

/* Backward of the same fused row op (+ residual-gradient add):
 *   g   = g_res + LayerNormBackward(d_y; x, mean, rstd, ln_w)          (x = FDDT(h_in)+pos, recomputed)
 *   g0  = FDDTBackward(g)  (diag: g * sum_c m_c w_c ; bias-only / mode 0: g)
 * and the column reductions  dln_w += sum_r d_y*xhat, dln_b += sum_r d_y,
 *   dw[c] += sum_r m_c*h_in*g, db[c] += sum_r m_c*g, colsum_out += sum_r g0  (bias grad of the producing Linear).
 * All parameter-gradient outputs are fp32 and ACCUMULATED (+=, NULL = not needed) by a deterministic two-stage
 * reduction through the caller's workspace (no atomics).                                                */
typedef struct {
    const void* h_in; int in_bf16; int mode;
    const float* stno; int64_t stno_bstride;
    const float* w[4]; const float* b[4];
    const float* pos;
    const float* ln_w;       /* NULL => no LayerNorm in this op (g = g_res) */
    const float* mean; const float* rstd;
    const void* d_y;         /* [rows,D] bf16 grad wrt LayerNorm output (or fp32 when dy_f32 != 0) */
    int dy_f32;
    const float* g_res;      /* [rows,D] fp32 residual-path gradient, or NULL */
    float* g_out;            /* [rows,D] fp32 g0, or NULL */
    void* g_out_bf16;        /* [rows,D] bf16 copy of g0 (dgrad/wgrad GEMM operand), or NULL */
    float* dln_w; float* dln_b;
    float* dw[4]; float* db[4];
    float* colsum_out;
    float* dpos_rows;        /* unused, reserved */
    int rows, T, D;
    void* ws; int64_t ws_bytes;   /* workspace for the per-workgroup partial column sums (dicow_fddt_ln_bwd_ws_bytes) */
} dicow_fddt_ln_bwd_args;
int64_t dicow_fddt_ln_bwd_ws_bytes(int rows, int D);
int dicow_fddt_ln_bwd(const dicow_fddt_ln_bwd_args* a, void* stream);

/* Full (D x D) FDDT combine: y4 = h @ [W_S;W_T;W_N;W_O]^T (bf16 [rows,4D], from dicow_gemm_nt with bias) ->
 * h' = sum_c m_c * y4[:, cD:(c+1)D]; disabled classes pass y4 block = h via use_mask.  (FDDT.py:13-16, 53-62) */
int dicow_fddt_full_combine_fwd(const void* y4, const void* h_in, int in_bf16, const float* stno, int64_t stno_bstride,
                                int use_mask, float* h_out, int rows, int T, int D, void* stream);
/* backward: d_y4[:, cD:(c+1)D] = m_c * g (bf16), dh_direct = g * sum_{c disabled} m_c (fp32) */
int dicow_fddt_full_combine_bwd(const float* g, const float* stno, int64_t stno_bstride, int use_mask,
                                void* d_y4, float* dh_direct, int rows, int T, int D, void* stream);

/* ------------------------------------------------------------------------------------------------ GEMM (MFMA bf16)
 * C[M,N] = epilogue( A[M,K] . B[N,K]^T ): every Linear / conv-as-GEMM forward and dgrad on the path
 * (HF:modeling_whisper.py:279-282,309,332-333,354,403-405; encoder.py:167-168; modeling_dicow.py:302).
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16, 128x128x64 LDS tiles.
 * Requirements: K % 64 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, N % 4 == 0; M, N tails handled.          */
#define DICOW_EPI_BIAS      1    /* + bias[n] */
#define DICOW_EPI_GELU      2    /* exact-erf GELU of the bf16-ROUNDED pre-activation (AMP: the Linear output is bf16); the pre-activation is stored to aux (bf16) when aux != NULL */
#define DICOW_EPI_RESIDUAL  4    /* + residual[m,n] (fp32, ld = ldr) */
#define DICOW_EPI_OUT_F32   8    /* C is fp32 (else bf16) */
#define DICOW_EPI_SCALE_N  16    /* columns n < scale_ncols multiplied by scale AFTER bias (q * head_dim^-0.5) */
#define DICOW_EPI_GELU_BWD 32    /* C = acc * gelu'(aux[m,n])  (aux = saved pre-activation, bf16) */
#define DICOW_EPI_ACCUM    64    /* C += result (fp32 C only) */
#define DICOW_EPI_GELU_DAUX 128  /* with GELU: aux receives gelu'(pre-activation) (bf16) instead of the pre-activation,  */
                                 /* so that the backward GEMM needs one multiply (MUL_AUX) and no transcendental    */
#define DICOW_EPI_MUL_AUX  256   /* C = acc * aux[m,n]  (aux = saved gelu' from GELU_DAUX, bf16)                      */
#define DICOW_EPI_COLSUM   512   /* colsum_out[n] += sum_m C[m,n] (bias gradient of the layer that produced the GEMM's */
                                 /* input gradient); needs colsum_ws of dicow_gemm_nt_colsum_ws_bytes(M, N) bytes     */
#define DICOW_EPI_FDDT    1024   /* with BIAS | RESIDUAL | OUT_F32: C = FDDT_next(bf16(acc + bias) + residual); persistent kernel only  */
typedef struct {
    const void* A; const void* B; void* C;
    const float* bias; const float* residual; void* aux;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr, ldaux;
    int batch; int64_t strideA, strideB, strideC, strideAux;   /* grid.z batches (conv stem: per-utterance strided views) */
    int flags; float scale; int scale_ncols;
    float* colsum_out; void* colsum_ws; int64_t colsum_ws_bytes;   /* DICOW_EPI_COLSUM only (else NULL / 0) */
    /* DICOW_EPI_FDDT only (ABI 3): the diagonal FDDT of the NEXT encoder layer applied to the fp32 result row by row
       (reference FDDT.py:41-63 in its evaluation order, bit-identical to dicow_fddt_ln_fwd's h_out):
       fddt_w[c] / fddt_b[c] = the [N] fp32 weight / bias vectors of class c (S, T, N, O), fddt_rowmask = [>= M rounded up to the
       tile height + 64 rows][4] fp32, row m = the four STNO class masks of output row m */
    const float* fddt_w[4]; const float* fddt_b[4]; const float* fddt_rowmask;
    /* EXPERIMENTAL (see the end of this header; zero in the stable ABI): DICOW_EPI_LNSTAT / LNFOLD only: lnstat = [M][DICOW_LN_SLOTS][2] fp32 row partials (LNSTAT writes its 4 * N / 320 slots,
       LNFOLD sums the first ln_nslots in slot order); ln_c = [N] fp32 column sums of the folded weight; ln_inv_dim = 1 / (normalised
       width), ln_eps = LayerNorm epsilon */
    float* lnstat; const float* ln_c; float ln_inv_dim; float ln_eps; int ln_nslots;
} dicow_gemm_args;
/* 1 when dicow_gemm_nt will run this problem on the persistent ring kernel with a compile-time epilogue (the only path
   that implements DICOW_EPI_FDDT), else 0 */
int dicow_gemm_nt_is_persistent(const dicow_gemm_args* a);
int dicow_gemm_nt(const dicow_gemm_args* a, void* stream);
int64_t dicow_gemm_nt_colsum_ws_bytes(int M, int N);
/* Deep contraction, small output (K >= 8192, fewer 128 x 128 tiles than workgroup slots, no epilogue: the tied LM head's dgrad,
 * modeling_dicow.py:302): when the caller passes this many bytes in colsum_ws / colsum_ws_bytes, the contraction is cut into
 * equal ranges that run as one batched launch and are added up in a fixed order.  0 = the problem is not split. */
int64_t dicow_gemm_nt_splitk_ws_bytes(const dicow_gemm_args* a);

/* C[N1,N2] (+)= sum_m A[m,N1] * B[m,N2]  (fp32 C): every weight gradient dW = dY^T X.  When the output has too few
 * tiles to fill the chip the contraction is split over grid.z; the splits write fp32 partials to the caller's
 * workspace (dicow_gemm_tn_ws_bytes) and a reduce pass adds them into C -- deterministic, no atomics.
 * Requirements: N1 % 8 == 0, N2 % 8 == 0, lda/ldb % 8 == 0.                                              */
typedef struct {
    const void* A; const void* B; float* C;
    int Mk, N1, N2;
    int64_t lda, ldb, ldc;
    int batch; int64_t strideA, strideB;            /* extra contraction batches (conv views) */
    int accumulate;                                 /* 1: C += result, 0: C = result */
    float* C_seg[2];                                /* optional row segments: rows [seg_rows, 2*seg_rows) of the logical C */
    int seg_rows;                                   /* go to C_seg[0], rows [2*seg_rows, ..) to C_seg[1] (q/k/v weight grads
                                                       from ONE GEMM over the fused d_qkv); 0 = single C */
    void* ws; int64_t ws_bytes;                     /* workspace for split partials */
} dicow_gemm_tn_args;
int64_t dicow_gemm_tn_ws_bytes(const dicow_gemm_tn_args* a);
int dicow_gemm_tn(const dicow_gemm_tn_args* a, void* stream);

/* Several weight gradients with the same contraction length as ONE persistent launch (the four dW of an encoder layer:
 * autograd's wgrad of q/k/v, out_proj, fc1, fc2 reached from encoder.py:216-221).  Pooled, the output tiles cover the chip
 * with whole contractions and only the remainder is split; one fix-up launch adds those partials in a fixed order.  Same
 * results as calling dicow_gemm_tn on p[0..n) in turn up to fp32 summation order (deterministic); problems that cannot be
 * pooled (N < 256, batches, different Mk, fewer pooled tiles than CUs) are run one by one.  ws: dicow_gemm_tn_group_ws_bytes.
 * Up to 24 problems: a small model (whisper-base: 48 tiles per layer) pools the weight gradients of ALL its layers. */
#define DICOW_TN_GROUP_MAX 24
typedef struct {
    int n;
    dicow_gemm_tn_args p[DICOW_TN_GROUP_MAX];       /* (their own ws / ws_bytes fields are ignored) */
    void* ws; int64_t ws_bytes;
} dicow_gemm_tn_group_args;
int64_t dicow_gemm_tn_group_ws_bytes(const dicow_gemm_tn_group_args* a);
int dicow_gemm_tn_group(const dicow_gemm_tn_group_args* a, void* stream);

/* ------------------------------------------------------------------------------------------------ attention
 * Flash-style softmax(Q K^T) V with head_dim 64, scaling 1.0 (q is pre-scaled by the projection epilogue,
 * HF:modeling_whisper.py:309,337-351).  Dense (encoder 1500x1500, SE enrollment cross-attention
 * layers.py:152-157), causal (decoder self-attention) and rectangular (decoder cross-attention) shapes.
 * q/k/v/o are bf16 with explicit strides (elements): element (b, t, h, d) at  base + b*bs + t*rs + h*64 + d. */
typedef struct {
    const void* q; const void* k; const void* v; void* o;
    float* lse;                                     /* [B,H,Lq] natural-log-sum-exp, saved for backward */
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
    int B, H, Lq, Lk; int causal;
    int q_log2;                                     /* 1: q carries a factor log2(e) (folded into the projection's q-scale, which the
                                                       caller then sets to head_dim^-0.5 * log2 e): the scores are base-2 exponents.
                                                       Same softmax, same lse (natural log) -- the kernel saves a multiply-add per score */
} dicow_attn_fwd_args;
int dicow_attn_fwd(const dicow_attn_fwd_args* a, void* stream);

typedef struct {
    const void* q; const void* k; const void* v; const void* o; const void* d_o;
    const float* lse; float* delta;                 /* delta: workspace [2,B,H,Lq] fp32 (-rowsum(dO*O), -lse) */
    void* dq; void* dk; void* dv;                   /* bf16, same addressing as q/k/v */
    int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs;
    int64_t dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs;
    int B, H, Lq, Lk; int causal;
    float dq_scale;                                 /* head_dim^-0.5 folded into dq (gradient of the pre-scale) */
    /* optional fused bias gradients: dq_colsum[h*64+d] += sum_{b,q} dq (as stored), dv_colsum likewise; NULL = off.
       cs_ws: dicow_attn_bwd_colsum_ws_bytes(B, H, Lq, Lk) bytes of scratch (per-wave partial rows, reduced afterwards) */
    float* dq_colsum; float* dv_colsum; void* cs_ws; int64_t cs_ws_bytes;
    int q_log2;                                     /* as in dicow_attn_fwd_args; dq keeps its meaning (gradient of the projection
                                                       output before ANY scaling when dq_scale = head_dim^-0.5), dk = ln 2 dS^T q */
    /* ABI 7: optional workspace of the FUSED backward (one kernel, 5 matrix passes instead of 7: dQ is summed over the key-block
       workgroups of a (batch, head) in a fixed order through the XCD's L2 -- no atomics, bit-reproducible).  NULL = the two-kernel
       form.  dicow_attn_bwd_fused_ws_bytes() bytes, 4096-byte aligned, contents irrelevant on entry; used for dense problems of
       at least 1024 key-block workgroups (the encoder's self-attention), ignored otherwise. */
    void* fused_ws; int64_t fused_ws_bytes;
    int fused_mode;                                 /* 0: the library decides (above); 1: the fused kernel for EVERY dense problem it can
                                                       address (B*H <= 1000), whatever its size -- how the tests reach its edge cases */
} dicow_attn_bwd_args;
int dicow_attn_bwd(const dicow_attn_bwd_args* a, void* stream);
int64_t dicow_attn_bwd_colsum_ws_bytes(int B, int H, int Lq, int Lk);
int64_t dicow_attn_bwd_fused_ws_bytes(int B, int H, int Lq, int Lk);
/* error bits the last fused launch left in its workspace (synchronise the stream first): 0 = fine, 1 = a hand-off wait timed
   out, 2 = a (batch, head) was spread over several XCDs; -1 = the copy failed.  dq is not to be trusted when non-zero. */
int dicow_attn_bwd_fused_status(const void* fused_ws);


/* ------------------------------------------------------------------------------------------------ loss
 * Fused log-softmax + cross-entropy on the LM-head logits (bf16 [rows, ld], V valid columns).
 * Hard-label fallback (src/models/dicow/modeling_dicow.py:310-323): CE(ignore_index=-100) for `labels` and
 * `upp_labels`, per-token min; caller divides loss_sum by rows (mean over ALL positions).
 * Soft-label loss (modeling_dicow.py:95-144, soft != 0): rows whose label is a timestamp token use the Gaussian-
 * smoothed target ts_w[ts_index[label], :] scattered at ts_ids (:35-93); both losses masked by labels != -100;
 * caller divides loss_sum by max(count, 1).
 * Backward: d_logits = grad_scale[0] * (softmax - target of the argmin branch) in bf16 (grad_scale on device so
 * that no host sync is needed for the 1/count normalisation).                                                  */
typedef struct {
    const void* logits; int64_t ld; int rows; int V;
    const int64_t* labels; const int64_t* upp_labels;   /* [rows]; upp_labels may be NULL */
    int soft;
    const int32_t* ts_index;                            /* [V] timestamp row of a token id, -1 if none (NULL: no smoothing) */
    const int32_t* ts_ids;                              /* [n_ts] sorted timestamp token ids */
    const float* ts_w;                                  /* [n_ts, n_ts] row-normalised Gaussian weights */
    int n_ts;
    float* lse; float* row_loss; int32_t* choice;       /* [rows] outputs of fwd, inputs of bwd */
    float* loss_sum; float* count;                      /* scalars, ACCUMULATED (zero them first) */
    void* d_logits;                                     /* bwd: bf16 [rows, ld] (pad columns written as 0) */
} dicow_ce_args;
int dicow_ce_loss_fwd(const dicow_ce_args* a, void* stream);
int dicow_ce_loss_bwd(const dicow_ce_args* a, const float* grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------ CTC auxiliary loss
 * torch.nn.functional.ctc_loss as called at src/models/dicow/encoder.py:108-135 on the encoder logits
 * (modeling_dicow.py:242-246, 326-336): bf16 logits [B, Tn, ld] with C = vocab+1 valid classes, blank = C-1,
 * labels [B, Lc] int64 with the valid targets as a prefix (-100 padded), zero_infinity, reduction "mean":
 * loss_sum accumulates sum_b nll_b / max(len_b, 1) (caller divides by B).  alpha/beta: workspace [B, Tn, 2*Lc+1] each.
 * Backward: d_logits = grad_scale[0] / (B * max(len_b,1)) * (softmax - label posterior), bf16, pad columns zero.   */
typedef struct {
    const void* logits; int64_t ld; int B, Tn, C;
    const int64_t* labels; int Lc; int blank;
    float* lse;                  /* [B*Tn] */
    float* alpha; float* beta;   /* [B, Tn, Smax] fp32 log-domain */
    int Smax;                    /* 2*Lc + 1 (<= 1024) */
    float* nll; float* tlen;     /* [B] */
    float* loss_sum;             /* scalar, ACCUMULATED */
    void* d_logits;              /* bwd */
} dicow_ctc_args;
int64_t dicow_ctc_ws_bytes(int B, int Tn, int Lc);
int dicow_ctc_loss_fwd(const dicow_ctc_args* a, void* stream);
int dicow_ctc_loss_bwd(const dicow_ctc_args* a, const float* grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------ decoder embedding
 * h[b,l,:] = embed_tokens[ids[b,l]] + embed_positions[l]   (HF:modeling_whisper.py:737,754-762), fp32.     */
int dicow_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* out, int B, int Lq, int D, void* stream);
/* d_tok[ids] += g (atomics), d_pos[l] += sum_b g[b,l]; either output may be NULL (frozen). */
int dicow_embed_bwd(const int64_t* ids, const float* g, float* d_tok, float* d_pos, int B, int Lq, int D, void* stream);

/* ------------------------------------------------------------------------------------------------ conv-stem backward helpers
 * out = g * gelu'(pre)   (bf16, elementwise): gradient through the GELU after conv2 (encoder.py:168).     */
int dicow_gelu_bwd_bf16(const void* g, const void* pre, void* out, int64_t n, void* stream);
/* col2im of the conv2 (k=3,s=2,p=1) input gradient + GELU' of conv1 (encoder.py:167):
 *   dA2 bf16 [B, T2, 3*C] (tap-major im2col gradient)  ->  d_pre1 bf16 [B, 2*T2, C] = gelu'(pre1) * scatter-add
 * pre1 == NULL: plain col2im (the activation-free CTC subsample convs, encoder.py:26-41). */
int dicow_conv2_col2im_gelu_bwd(const void* dA2, const void* pre1, void* d_pre1, int B, int T2, int C, void* stream);

/* ------------------------------------------------------------------------------------------------ log-mel front end
 * Whisper features on the GPU (reference call site src/data/local_datasets.py:208-214 -> HF feature_extraction_whisper.py
 * :135-165): wave fp32 [B, n_samples] (padded to a multiple of 30 s) -> out fp32 [B, M, n_samples/160].
 * tw_cos/tw_sin: [604, 224] hann-window-folded DFT tables (201 bins, rows zero-padded to 224 = 7 blocks of 32; the DFT runs as an
 * exact-fp32 matrix product on v_mfma_f32_32x32x2_f32).  Rows 0..399: T[n][k] = w[n] cos(2 pi k n / 400) resp. -w[n] sin(..)
 * (what a direct DFT reads).  Rows 400..603 (ABI 5): the SAME product folded about sample 200 -- the kernel pairs row 400 + n with
 * x[n] + x[400 - n] (cos) resp. x[n] - x[400 - n] (sin), n = 0..203: rows 400 + n = T[n] for n < 200, row 600 = T[200] / 2 in the cos
 * table (x[200] meets itself) and zero in the sin table, rows 601..603 zero.  fb: [201, M] slaney mel filterbank, mel_range: [M][2]
 * int32, the first and one-past-the-last bin where column m of fb is non-zero (host-built once, ts-asr-whisper_amd/features.py). */
int64_t dicow_logmel_ws_bytes(int B, int n_samples);
int dicow_logmel(const float* wave, int B, int n_samples, const float* tw_cos, const float* tw_sin, const float* fb,
                 const int* mel_range, int M, float* out, void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------ batch augmentation
 * The collator's training-time augmentations (reference src/data/collators.py:189-214), applied to the batch where it
 * already lives (HBM).  The random decisions are drawn on the host from the torch CPU generator in the reference's
 * own order (ts-asr-whisper_amd/augment.py) and arrive here as small plans, so a seed reproduces the reference.
 *
 * dicow_stno_noise_rescale (collators.py:50-77): for i < n_rows, r = rows[i]:
 *     x = stno[r] + noise[i] * sd;  x -= min(min_c x, 0);  stno[r] = x / sum_c x         (in place; rows distinct)
 *   stno fp32 [B, C=4, T], rows int32 [n_rows], noise fp32 [n_rows, C, T] (N(0,1) draws), sd = sqrt(variance).
 * dicow_stno_segment_augment (collators.py:79-138): for each changed segment s (segments are disjoint):
 *     dominant = argmax_c mean_t stno[b, c, start:end];  target = the pick-th class != dominant
 *     x = keep * stno[b, :, t] + soft * onehot(target);  stno[b, :, t] = x / sum_c x      (in place)
 *   segs int32 [n_seg, 4] = (b, start, end, pick), coef fp32 [n_seg, 2] = (keep = 1 - softness, soft = softness).
 * dicow_specaug_joint (collators.py:209-214; src/data/augmentations.py:85-120 time warp, :23-79 masks, :363-379
 *   masks limited to features [:128]): x = [mel ; stno repeated `sub` times along time]  (M + 4 feature rows, T frames)
 *     time warp: frames [0, center) are resampled to [0, warped) and [center, T) to [warped, T) with torch's bicubic
 *       kernel (align_corners = false, A = -0.75, border clamp per piece);  warped < 0 disables the warp
 *     frequency masks fmask int32 [B, n_fmask, 2] = (pos, len) and time masks tmask int32 [B, n_tmask, 2] zero
 *       x[b, pos:pos+len] on feature rows < n_maskable (= min(128, M + 4) in the reference)
 *     mel_out fp32 [B, M, T] = rows < M;  stno_out fp32 [B, 4, T/sub] = mean over each `sub` frames of rows >= M.
 *   Out of place (mel_out != mel, stno_out != stno). */
int dicow_stno_noise_rescale(float* stno, const int* rows, const float* noise, int n_rows, int C, int T, float sd,
                             void* stream);
int dicow_stno_segment_augment(float* stno, const int* segs, const float* coef, int n_seg, int C, int T, void* stream);
int dicow_specaug_joint(const float* mel, const float* stno, float* mel_out, float* stno_out, int B, int M, int T, int sub,
                        int center, int warped, const int* fmask, int n_fmask, const int* tmask, int n_tmask,
                        int n_maskable, void* stream);

/* ------------------------------------------------------------------------------------------------ CTC prefix scoring
 * Joint CTC / attention decoding (reference src/models/dicow/decoding.py; CTCPrefixScore :8-163 is the hot part: a Python
 * loop over the encoder frames per decoded token).  Log-probabilities are x[b, t, c] = logits[b, t, alias[c]] - lse[b, t]:
 *   dicow_ctc_frame_lse     lse[row] = logsumexp_v logits[row, v < V1]   (the log_softmax of decoding.py:183)
 *   alias int32 [V1] or NULL: column read for label c (the reference copies lower-cased columns over the upper-cased
 *                           ones, decoding.py:181-186)
 *   dicow_ctc_prefix_init   r0 fp32 [B, T, 2] = state of the empty prefix (CTCPrefixScore.initial_state, :36-43)
 *   dicow_ctc_prefix_score  CTCPrefixScore.__call__ (:122-163) for n hypotheses x C candidate labels:
 *       rows int32 [n]         batch row of each hypothesis (the reference's `samples_to_be_decoded` mask, as indices)
 *       cs int32 [n, C]        candidate next labels;  decoded_len int32 [n] labels already in the prefix;
 *       last int32 [n]         last label of the prefix;  r_prev fp32 [n, T, 2] its state (non-blank, blank)
 *       psi fp32 [n, C]        log prefix probability of prefix + candidate (eos: probability of ending, blank: logzero)
 *       r fp32 [n, T, 2, C]    the candidates' states (the caller keeps the chosen one as the next r_prev)
 *   logits fp32 or bf16 (in_bf16) [B, T, ld >= V1], frame-major. */
typedef struct {
    const void* logits; int in_bf16; int64_t ld; const float* lse; const int* alias;
    const int* rows; const int* cs; const int* decoded_len; const int* last; const float* r_prev;
    float* psi; float* r;
    int n, C, T, blank, eos;
} dicow_ctc_prefix_args;
int dicow_ctc_frame_lse(const void* logits, int in_bf16, int64_t rows, int V1, int64_t ld, float* lse, void* stream);
int dicow_ctc_prefix_init(const void* logits, int in_bf16, int64_t ld, const float* lse, int B, int T, int blank_col, float* r0,
                          void* stream);
int dicow_ctc_prefix_score(const dicow_ctc_prefix_args* a, void* stream);

/* Whisper's timestamp rules on next-token scores, in place (reference src/models/dicow/utils.py:5-14 =
 * transformers' WhisperTimeStampLogitsProcessor + eos allowed at the first generated position):
 * scores fp32 [B, ld >= V]; input_ids int64 [B, L] (prompt + generated so far), begin_index = prompt length;
 * timestamp_begin = no_timestamps + 1; max_initial_timestamp_index < 0: no cap; detect_from_logprob: if the total
 * probability of the timestamp labels exceeds every text label's, text labels are removed. */
int dicow_whisper_timestamp_rules(float* scores, int64_t ld, int B, int V, const int64_t* input_ids, int L, int begin_index,
                                  int timestamp_begin, int eos, int no_timestamps, int max_initial_timestamp_index,
                                  int detect_from_logprob, void* stream);

/* ------------------------------------------------------------------------------------------------ optimizer
 * Fused AdamW + global-norm clipping on flat fp32 regions (src/models/containers.py:100-114 two param groups;
 * HF Trainer max_grad_norm 1.0).  dicow_sumsq_f32 accumulates sum(x^2) into out[0], DETERMINISTICALLY for a given input
 * (fixed summation order: data-parallel replicas must derive the same clip coefficient); one call in flight per device;
 * dicow_adamw_f32 applies
 *   g' = g * min(1, max_norm / (sqrt(gnorm_sq[0]) + 1e-6));  decoupled weight decay; bias-corrected moments.   */
int dicow_sumsq_f32(const float* x, int64_t n, float* out, void* stream);
/* Measurement aid (ABI 7; no reference counterpart): the load an 8-rank RCCL all-reduce of `bytes` at `gbps` GB/s (algorithm bandwidth)
 * puts on THIS GPU while it runs beside the backward pass, reproduced on one GPU -- `workgroups` blocks rewrite the buffer in place
 * with its own values, `passes` times (2 = reduce-scatter + all-gather), pacing themselves so that the call takes bytes / gbps.
 * The data are unchanged.  trainer.GradReducer (DICOW_EMULATE_FABRIC_GBPS) / bench.py --emulate-fabric-gbps. */
int dicow_fabric_emulate(void* buf, int64_t bytes, double gbps, int workgroups, int passes, void* stream);
int dicow_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float* gnorm_sq, float max_norm, void* stream);
/* The same update with the step-dependent scalars read from device memory, hyper = {lr, 1 - beta1^t, 1 - beta2^t}: a training
 * step captured in a hipGraph (reference counterpart: HF Trainer.training_step, src/utils/trainers.py:116-139, re-run every
 * step) replays with fixed kernel arguments.  dicow_adamw_hyper keeps the counters on the device (counters[0] = optimizer steps
 * taken, counters[1 + i] = updates received by run i; torch keeps state['step'] per parameter), advances them and writes
 * hyper[3 i ..] for every active run: HF's LambdaLR indexing (the k-th step uses lambda(k - 1)), linear warm-up, cosine decay
 * to max_steps (configs/train/dicow_v3.yaml:66-68) or constant, x `mult` for preheat runs (containers.py:109-111);
 * preheat_only: the other runs are frozen (trainers.py:122-137) and neither counted nor written.  lr / mult / betas are doubles and
 * the schedule and 1 - beta^t are evaluated in double on the device, as torch evaluates them on the host. */
int dicow_adamw_hyper(int* counters, float* hyper, const int* is_pre, int n_runs, int preheat_only, double lr, double mult,
                      int warmup_steps, int max_steps, int cosine, double beta1, double beta2, void* stream);
int dicow_adamw_f32_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1, float beta2,
                        float eps, float weight_decay, const float* gnorm_sq, float max_norm, void* stream);


/* ------------------------------------------------------------------------------------------------ EXPERIMENTAL (not part of the stable ABI)
 * Declared only under -DDICOW_EXPERIMENTAL_ABI and exported only by a library built with -DDICOW_EXPERIMENTS (build.sh --exp -> libdicow_hip_exp.so).
 * Round 4's LayerNorm fold: built, parity-tested (tests/test_gpu_lnfold.py) and measured 3-4 % SLOWER than the two LayerNorm launches it
 * deletes (profiles/r04_lnfold.txt) -- kept for A/B builds, out of the stable ABI.  The tail fields of dicow_gemm_args (lnstat ... ln_nslots) belong to
 * it and must be zero for a stable-ABI library, which refuses the two flags. */
#ifdef DICOW_EXPERIMENTAL_ABI
/* LayerNorm folded into the GEMMs on either side of it (ABI 4; HF:modeling_whisper.py:392-405 via encoder.py:216-221: the
 * pre-LN encoder layer's self_attn_layer_norm -> q/k/v and final_layer_norm -> fc1).  With W' = bf16(gamma . W) (column scale),
 *   LN(h) W^T + b = rstd_r (bf16(h) W'^T) - rstd_r mean_r colsum(W')_n + (beta W^T + b)_n
 * so the GEMM that PRODUCES the residual stream h (out-proj / fc2 with the fp32 residual epilogue) also stores bf16(h) and per-row
 * partial (sum, sum of squares) of the fp32 h, and the GEMM that CONSUMES LN(h) reads bf16(h) as its A operand and applies the
 * row scale + rank-one correction in its epilogue: the LayerNorm launch between them disappears.  Rounding point: bf16 of the
 * un-normalised h instead of bf16 of LN(h) (one bf16 rounding per operand element either way). */
#define DICOW_EPI_LNSTAT  2048   /* producer: with BIAS | RESIDUAL | OUT_F32 (| FDDT): aux <- bf16 copy of the fp32 result (ld = ldaux), lnstat[m][slot] <- partial (sum, sum sq) */
#define DICOW_EPI_LNFOLD  4096   /* consumer: acc -> rstd_m acc - rstd_m mean_m ln_c[n] before bias / scale / GELU; A = the producer's bf16 copy, B = W', bias = beta W^T + b */
#define DICOW_LN_SLOTS 16        /* partial slots per row of lnstat: [M][16][2] fp32; slot = 4 * (column tile of 320) + 2 * wave column + {0: 128 main columns, 1: 32 tail columns} */
/* 1 when dicow_gemm_nt will run this DICOW_EPI_LNSTAT problem (flags BIAS | RESIDUAL | OUT_F32 [| FDDT] | LNSTAT) on the persistent
   ring kernel's 192 x 320 tiles -- the only producer of the row partials (needs N % 320 == 0, N <= 1280) -- else 0 */
int dicow_gemm_nt_lnstat_ok(const dicow_gemm_args* a);
/* Weights of a Linear that follows a LayerNorm, folded (once per optimizer step): W fp32 [N, K], gamma / beta fp32 [K], bias fp32 [N] or
   NULL -> Wf bf16 [N, ldw] = bf16(gamma_k W_nk), c fp32 [N] = sum_k float(Wf_nk), bf fp32 [N] = bias_n + sum_k beta_k float(bf16(W_nk)) */
int dicow_lnfold_prep(const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, int64_t ldw, float* c, float* bf,
                      int N, int K, void* stream);
#endif  /* DICOW_EXPERIMENTAL_ABI */

#ifdef __cplusplus
}
#endif
#endif /* DICOW_HIP_H */
