"""ctypes binding of libdicow_hip.so (include/dicow_hip.h).  Plain pointers and sizes only."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DICOW_HIP_LIB") or os.path.join(_HERE, "libdicow_hip.so")   # override: diagnostic builds
_lib = None

c_vp, c_i, c_i64, c_f, c_d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class FddtLnFwdArgs(C.Structure):
    _fields_ = [("h_in", c_vp), ("in_bf16", c_i), ("mode", c_i), ("stno", c_vp), ("stno_bstride", c_i64),
                ("w", c_vp * 4), ("b", c_vp * 4), ("pos", c_vp), ("h_out", c_vp), ("ln_w", c_vp), ("ln_b", c_vp),
                ("y_bf16", c_vp), ("y_f32", c_vp), ("mean", c_vp), ("rstd", c_vp),
                ("rows", c_i), ("T", c_i), ("D", c_i), ("eps", c_f)]


class FddtLnBwdArgs(C.Structure):
    _fields_ = [("h_in", c_vp), ("in_bf16", c_i), ("mode", c_i), ("stno", c_vp), ("stno_bstride", c_i64),
                ("w", c_vp * 4), ("b", c_vp * 4), ("pos", c_vp), ("ln_w", c_vp), ("mean", c_vp), ("rstd", c_vp),
                ("d_y", c_vp), ("dy_f32", c_i), ("g_res", c_vp), ("g_out", c_vp), ("g_out_bf16", c_vp),
                ("dln_w", c_vp), ("dln_b", c_vp), ("dw", c_vp * 4), ("db", c_vp * 4), ("colsum_out", c_vp),
                ("dpos_rows", c_vp), ("rows", c_i), ("T", c_i), ("D", c_i), ("ws", c_vp), ("ws_bytes", c_i64)]


class GemmArgs(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("bias", c_vp), ("residual", c_vp), ("aux", c_vp),
                ("M", c_i), ("N", c_i), ("K", c_i),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("ldr", c_i64), ("ldaux", c_i64),
                ("batch", c_i), ("strideA", c_i64), ("strideB", c_i64), ("strideC", c_i64), ("strideAux", c_i64),
                ("flags", c_i), ("scale", c_f), ("scale_ncols", c_i),
                ("colsum_out", c_vp), ("colsum_ws", c_vp), ("colsum_ws_bytes", c_i64),
                ("fddt_w", c_vp * 4), ("fddt_b", c_vp * 4), ("fddt_rowmask", c_vp),
                ("lnstat", c_vp), ("ln_c", c_vp), ("ln_inv_dim", c_f), ("ln_eps", c_f), ("ln_nslots", c_i)]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", c_vp), ("B", c_vp), ("C", c_vp), ("Mk", c_i), ("N1", c_i), ("N2", c_i),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
                ("batch", c_i), ("strideA", c_i64), ("strideB", c_i64), ("accumulate", c_i),
                ("C_seg", c_vp * 2), ("seg_rows", c_i), ("ws", c_vp), ("ws_bytes", c_i64)]


TN_GROUP_MAX = 24
CAST_GROUP_MAX = 8


class CastProblem(C.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("dst_t", c_vp), ("R", c_i), ("C", c_i), ("ld", c_i64), ("ld_t", c_i64)]


class GemmTnGroupArgs(C.Structure):
    _fields_ = [("n", c_i), ("p", GemmTnArgs * TN_GROUP_MAX), ("ws", c_vp), ("ws_bytes", c_i64)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("v", c_vp), ("o", c_vp), ("lse", c_vp),
                ("q_bs", c_i64), ("q_rs", c_i64), ("k_bs", c_i64), ("k_rs", c_i64),
                ("v_bs", c_i64), ("v_rs", c_i64), ("o_bs", c_i64), ("o_rs", c_i64),
                ("B", c_i), ("H", c_i), ("Lq", c_i), ("Lk", c_i), ("causal", c_i), ("q_log2", c_i)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("v", c_vp), ("o", c_vp), ("d_o", c_vp), ("lse", c_vp), ("delta", c_vp),
                ("dq", c_vp), ("dk", c_vp), ("dv", c_vp),
                ("q_bs", c_i64), ("q_rs", c_i64), ("k_bs", c_i64), ("k_rs", c_i64), ("v_bs", c_i64), ("v_rs", c_i64),
                ("o_bs", c_i64), ("o_rs", c_i64), ("do_bs", c_i64), ("do_rs", c_i64),
                ("dq_bs", c_i64), ("dq_rs", c_i64), ("dk_bs", c_i64), ("dk_rs", c_i64), ("dv_bs", c_i64), ("dv_rs", c_i64),
                ("B", c_i), ("H", c_i), ("Lq", c_i), ("Lk", c_i), ("causal", c_i), ("dq_scale", c_f),
                ("dq_colsum", c_vp), ("dv_colsum", c_vp), ("cs_ws", c_vp), ("cs_ws_bytes", c_i64), ("q_log2", c_i),
                ("fused_ws", c_vp), ("fused_ws_bytes", c_i64), ("fused_mode", c_i)]


class CeArgs(C.Structure):
    _fields_ = [("logits", c_vp), ("ld", c_i64), ("rows", c_i), ("V", c_i), ("labels", c_vp), ("upp_labels", c_vp),
                ("soft", c_i), ("ts_index", c_vp), ("ts_ids", c_vp), ("ts_w", c_vp), ("n_ts", c_i),
                ("lse", c_vp), ("row_loss", c_vp), ("choice", c_vp), ("loss_sum", c_vp), ("count", c_vp),
                ("d_logits", c_vp)]


class CtcArgs(C.Structure):
    _fields_ = [("logits", c_vp), ("ld", c_i64), ("B", c_i), ("Tn", c_i), ("C", c_i), ("labels", c_vp), ("Lc", c_i),
                ("blank", c_i), ("lse", c_vp), ("alpha", c_vp), ("beta", c_vp), ("Smax", c_i), ("nll", c_vp), ("tlen", c_vp),
                ("loss_sum", c_vp), ("d_logits", c_vp)]


class CtcPrefixArgs(C.Structure):
    _fields_ = [("logits", c_vp), ("in_bf16", c_i), ("ld", c_i64), ("lse", c_vp), ("alias", c_vp), ("rows", c_vp), ("cs", c_vp),
                ("decoded_len", c_vp), ("last", c_vp), ("r_prev", c_vp), ("psi", c_vp), ("r", c_vp), ("n", c_i), ("C", c_i),
                ("T", c_i), ("blank", c_i), ("eos", c_i)]


EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_OUT_F32, EPI_SCALE_N, EPI_GELU_BWD, EPI_ACCUM = 1, 2, 4, 8, 16, 32, 64
EPI_GELU_DAUX, EPI_MUL_AUX, EPI_COLSUM, EPI_FDDT = 128, 256, 512, 1024
EPI_LNSTAT, EPI_LNFOLD, LN_SLOTS = 2048, 4096, 16

# name -> argtypes ; every function returns int
_SIGS = {
    "dicow_set_gemm_cus": [c_i],
    "dicow_cast_transpose_group": [C.POINTER(CastProblem), c_i, c_vp],
    "dicow_gemm_dispatch_log": [c_vp, c_i],
    "dicow_cast_f32_to_bf16": [c_vp, c_vp, c_i64, c_vp],
    "dicow_cast_transpose_f32_to_bf16": [c_vp, c_vp, c_i64, c_vp, c_i64, c_i, c_i, c_vp],
    "dicow_conv_weight_pack": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_conv_weight_unpack_grad": [c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_mel_to_timemajor": [c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_colsum_bf16": [c_vp, c_i64, c_vp, c_i, c_i, c_vp, c_i64, c_vp],
    "dicow_sum_over_batch": [c_vp, c_vp, c_i, c_i64, c_vp],
    "dicow_scb_split": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i, c_i, c_i, c_vp],
    "dicow_scb_merge_fwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_scb_gate_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp, c_i64, c_vp],
    "dicow_scb_merge_bwd": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_fddt_ln_fwd": [C.POINTER(FddtLnFwdArgs), c_vp],
    "dicow_fddt_ln_bwd": [C.POINTER(FddtLnBwdArgs), c_vp],
    "dicow_fddt_full_combine_fwd": [c_vp, c_vp, c_i, c_vp, c_i64, c_i, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_fddt_full_combine_bwd": [c_vp, c_vp, c_i64, c_i, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_gemm_nt": [C.POINTER(GemmArgs), c_vp],
    "dicow_gemm_nt_is_persistent": [C.POINTER(GemmArgs)],
    "dicow_gemm_tn": [C.POINTER(GemmTnArgs), c_vp],
    "dicow_gemm_tn_group": [C.POINTER(GemmTnGroupArgs), c_vp],
    "dicow_attn_fwd": [C.POINTER(AttnFwdArgs), c_vp],
    "dicow_attn_bwd": [C.POINTER(AttnBwdArgs), c_vp],
    "dicow_attn_bwd_fused_status": [c_vp],
    "dicow_ce_loss_fwd": [C.POINTER(CeArgs), c_vp],
    "dicow_ce_loss_bwd": [C.POINTER(CeArgs), c_vp, c_vp],
    "dicow_ctc_loss_fwd": [C.POINTER(CtcArgs), c_vp],
    "dicow_ctc_loss_bwd": [C.POINTER(CtcArgs), c_vp, c_vp],
    "dicow_embed_fwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_embed_bwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_gelu_bwd_bf16": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "dicow_conv2_col2im_gelu_bwd": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_logmel": [c_vp, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_i64, c_vp],
    "dicow_stno_noise_rescale": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_f, c_vp],
    "dicow_stno_segment_augment": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "dicow_specaug_joint": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_i, c_vp, c_i, c_i, c_vp],
    "dicow_ctc_frame_lse": [c_vp, c_i, c_i64, c_i, c_i64, c_vp, c_vp],
    "dicow_ctc_prefix_init": [c_vp, c_i, c_i64, c_vp, c_i, c_i, c_i, c_vp, c_vp],
    "dicow_ctc_prefix_score": [C.POINTER(CtcPrefixArgs), c_vp],
    "dicow_whisper_timestamp_rules": [c_vp, c_i64, c_i, c_i, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "dicow_sumsq_f32": [c_vp, c_i64, c_vp, c_vp],
    "dicow_fabric_emulate": [c_vp, c_i64, c_d, c_i, c_i, c_vp],
    "dicow_adamw_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f, c_f, c_f, c_f, c_f, c_i, c_vp, c_f, c_vp],
    "dicow_adamw_hyper": [c_vp, c_vp, c_vp, c_i, c_i, c_d, c_d, c_i, c_i, c_i, c_d, c_d, c_vp],
    "dicow_adamw_f32_dev": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_f, c_f, c_f, c_f, c_vp, c_f, c_vp],
}


# EXPERIMENTAL entry points (include/dicow_hip.h, DICOW_EXPERIMENTAL_ABI): only a library built with -DDICOW_EXPERIMENTS exports them
# (round 4's LayerNorm fold: parity-tested, measured slower); bound when present, `has_experimental()` says whether they are
_SIGS_EXPERIMENTAL = {
    "dicow_gemm_nt_lnstat_ok": [C.POINTER(GemmArgs)],
    "dicow_lnfold_prep": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i, c_i, c_vp],
}


class DicowError(RuntimeError):
    pass


def lib():
    """Load the HIP library; fail loudly (no fallback) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DicowError(f"{LIB_PATH} is missing: build it with ts-asr-whisper_amd/csrc/build.sh "
                             "(or __graft_entry__.build()); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        l.dicow_abi_version.restype = c_i
        l.dicow_last_error.restype = C.c_char_p
        for name, at in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = at
            fn.restype = c_i
        for name, at in _SIGS64.items():
            fn = getattr(l, name)
            fn.argtypes = at
            fn.restype = c_i64
        l._dicow_experimental = True
        for name, at in _SIGS_EXPERIMENTAL.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                l._dicow_experimental = False
                continue
            fn.argtypes = at
            fn.restype = c_i
        _lib = l
    return _lib


def has_experimental():
    """True when the loaded library was built with -DDICOW_EXPERIMENTS (exports the LayerNorm-fold entry points)."""
    return bool(lib()._dicow_experimental)


_SIGS64 = {   # functions returning int64_t (workspace sizes)
    "dicow_colsum_ws_bytes": [c_i, c_i],
    "dicow_gemm_nt_colsum_ws_bytes": [c_i, c_i],
    "dicow_attn_bwd_colsum_ws_bytes": [c_i, c_i, c_i, c_i],
    "dicow_attn_bwd_fused_ws_bytes": [c_i, c_i, c_i, c_i],
    "dicow_fddt_ln_bwd_ws_bytes": [c_i, c_i],
    "dicow_gemm_tn_ws_bytes": [C.POINTER(GemmTnArgs)],
    "dicow_gemm_nt_splitk_ws_bytes": [C.POINTER(GemmArgs)],
    "dicow_gemm_tn_group_ws_bytes": [C.POINTER(GemmTnGroupArgs)],
    "dicow_logmel_ws_bytes": [c_i, c_i],
    "dicow_scb_gate_bwd_ws_bytes": [],
    "dicow_ctc_ws_bytes": [c_i, c_i, c_i],
}


def declared_symbols():
    return ["dicow_abi_version", "dicow_last_error"] + list(_SIGS) + list(_SIGS64)


def check(rc, what):
    if rc != 0:
        raise DicowError(f"{what} failed ({rc}): {lib().dicow_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def call_struct(name, st):
    check(getattr(lib(), name)(C.byref(st), stream()), name)
