"""ctypes binding of libadalog_hip.so (the C ABI declared in include/adalog_hip.h).

The library is built in-tree (adalog_amd/csrc, `make` or __graft_entry__.build()) so that it travels to the GPU box
and is visible as a loaded native extension.  There is NO fallback: if the library is missing, import of the product
path raises, and every call needs a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ADALOG_LIB: another build of the same library -- same-box A/B runs of a kernel change; the torch ops link the default one)
LIB_PATH = os.environ.get("ADALOG_LIB") or os.path.join(_HERE, "csrc", "libadalog_hip.so")

p, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); must list every function of include/adalog_hip.h (tests/test_abi.py checks that)
SIGNATURES = {
    "adalog_abi_version": (i32, []),
    "adalog_last_error": (C.c_char_p, []),
    "adalog_last_kernel": (C.c_char_p, []),
    "adalog_uniform_fake_quant_f32": (i32, [p, p, p, i64, p, p, i64, i64, i32, i32, p]),
    "adalog_log_fake_quant_f32": (i32, [p, p, p, i64, p, p, p, p, i32, p, i32, i32, p]),
    "adalog_pack_uniform": (i32, [p, i64, i64, i64, i64, i64, i64, p, p, i64, i64, i64, i64, i64, i32, i32, p, i64, p, i32, p]),
    "adalog_pack_adalog_bf16": (i32, [p, i64, i64, i64, i64, i64, i64, p, p, i64, i64, i64, i64, i32, p, p, i32, p, i64, i32, p]),
    "adalog_pack_raw_f32": (i32, [p, i64, i64, i64, i64, i64, i64, p, i64, p]),
    "adalog_pack_split3_bf16": (i32, [p, i64, i64, i64, i64, i64, i64, p, i64, p]),
    "adalog_gemm_score": (i32, [i32, p, p, i64, i64, i64, i64, i32, i32, i64, i64, i32, i32, i32, p, i64, i64, i64, i32,
                                p, i64, i64, f32, p, i64, i64, i64, p, i64, i64, i64, p, p, p, i64, p, i64, i64, i64, i32, i32, p]),
    "adalog_gemm_score_layout": (i64, [i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i32, p, p, p]),
    "adalog_log2_shift": (i32, [p, p, i64, f32, p]),
    "adalog_score_act_fused_ok": (i32, [i32, i64, i32, i64, i32, i32]),
    "adalog_score_act_fused_workspace_bytes": (i64, [i64, i64]),
    "adalog_score_act_fused": (i32, [p, i32, i64, p, p, i64, i32, p, p, p, p, p, i32, i32, p, f32, i32, f32, f64, p, i64, p, p]),
    "adalog_score_act_gen_ok": (i32, [i32, i32, i64, i32, i64, i32]),
    "adalog_score_act_gen_wgs": (i32, [i32, i32, i64, i32, i64, i32]),
    "adalog_score_act_gen_workspace_bytes": (i64, [i32, i32, i64, i32, i64, i32]),
    "adalog_score_act_gen": (i32, [i32, p, i32, i64, p, i64, i32, i64, p, p, i32, i32, p, p, p, f64, p, i64, p, p]),
    "adalog_score_w_gen_ok": (i32, [i32, i32, i32, i32, i64, i32]),
    "adalog_score_w_gen": (i32, [i32, p, i32, i64, p, i32, i32, i64, p, p, i32, i32, p, p, p, p, i64, p]),
    "adalog_gram_supported": (i32, [i32, i32, i32, i32, i32, i32]),
    "adalog_gram_ok": (i32, [i32, i32, i32, i32, i32, i32]),
    "adalog_gram_act_supported": (i32, [i32, i32, i32, i32, i32, i32]),
    "adalog_gram_act_ok": (i32, [i32, i32, i32, i32, i32, i32]),
    "adalog_gram_act_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "adalog_gram_act_sort_bytes": (i64, [i64]),
    "adalog_gram_act_splits": (i32, [i32, i32, i32, i32]),
    "adalog_gram_act_prepare": (i32, [p, i32, i32, i64, p, p, p, p, i64, p]),
    "adalog_gram_act_build": (i32, [p, i32, i32, p, p, i32, i64, p, p, i32, p, i32, p, i64, p]),
    "adalog_gram_act_score": (i32, [p, p, i32, i32, i32, p, p, i32, i32, p, f64, p, p, p]),
    "adalog_gram_workspace_bytes": (i64, [i32, i32, i32, i32]),
    "adalog_gram_limbs": (i32, [i32, i32]),
    "adalog_gram_build": (i32, [p, i32, i32, i64, p, p, i32, p, i32, p, p, i64, p]),
    "adalog_gram_score_w": (i32, [p, i32, i32, i64, p, p, i32, i32, p, i32, i32, p, f64, p, p]),
    "adalog_finish_scores": (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, f64, p, i64, p]),
    "adalog_finish_topk_next": (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, f64, p, i64, i32, p, p, p, i32, p, p, i32, f32,
                                      p, p, p, p]),
    "adalog_finish_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32]),
    "adalog_topk": (i32, [p, i32, i32, i32, p, p]),
    "adalog_topk_next": (i32, [p, i32, i32, i32, p, p, p, i32, p, p, i32, f32, p, p, p, p, p]),
    "adalog_topk_next_tail": (i32, [p, i32, i32, p, p, p]),
    "adalog_finish_topk_next_tail": (i32, [p, p, i32, i32, i32, i32, i32, i32, i32, i32, i32, f64, p, i64, p, p]),
    "adalog_gemm_out_gen_ex": (i32, [p, i64, i64, i32, p, p, i64, i32, p, i64, i32, i32, i64, i32, i32, p, i64, f32, p, i64, i64, p, i64, i64,
                                     p, p, i64, i64, p]),
    "adalog_gemm_out_ex": (i32, [i32, p, p, i64, i64, i32, i32, i64, i32, i32, p, i64, f32, p, i64, i64, p, i64, i64, p, p, i64, i64, i32,
                                 i64, p]),
    "adalog_pack_adalog_bf16_pre": (i32, [p, i64, i64, i64, i64, i64, i64, p, p, i64, i64, i64, i64, i32, p, p, i32, p, i64, i32, i32, p]),
    "adalog_log_fake_quant_f32_pre": (i32, [p, p, p, i64, p, p, p, p, i32, p, i32, i32, i32, p]),
    "adalog_log_fq_backward_pre": (i32, [p, p, p, p, i64, p, p, i32, p, i32, p, p, i32, p]),
    "adalog_brecq_prepare": (i32, [p, p, p, p, p, i64, i64, i64, p, p, i32, p]),
    "adalog_softmax_adalog_pack_bf16": (i32, [p, i64, i32, f32, p, p, i32, p, p, i64, p]),
    "adalog_attn_split_pack": (i32, [p, i32, i32, i32, p, p, i32, p, p, i32, p, p, i32, i32, p, p, p, i64, p]),
    "adalog_gemm_out_gen": (i32, [p, i64, i64, i32, p, p, i64, i32, p, i64, i32, i32, i64, i32, i32, p, i64, f32, p, i64, i64, p, i64, i64,
                            p, i64, i64, p]),
    "adalog_sort_workspace_bytes": (i64, [i64, i64, i32]),
    "adalog_sort_f32": (i32, [p, i64, i64, p, p, p, i64, p]),
    "adalog_gram_amax": (i32, [p, i32, i32, p, p, p]),
    "adalog_gram_build_sums": (i32, [p, i32, i32, i64, p, p, i32, p, i32, p, p, p, p, p, p, i64, p]),
    "adalog_gram_build_from_sums": (i32, [p, p, p, p, i32, i32, i32, i32, p, i64, p]),
    "adalog_gram_score_w_tail": (i32, [p, i32, i32, i64, p, p, i32, i32, p, i32, i32, p, f64, p, p, p]),
    "adalog_gram_act_score_tail": (i32, [p, p, i32, i32, i32, p, p, i32, i32, p, f64, p, p, p, p]),
    "adalog_score_self_sorted_tail": (i32, [p, p, i64, i64, p, p, i32, i32, f64, p, p, p]),
    "adalog_fpcs_next": (i32, [p, p, p, i32, p, i32, i32, p, p, i32, f32, p, p, p, p]),
    "adalog_candidate_grid": (i32, [p, i32, i32, i32, i32, i32, p, i32, f32, p, p, p, p]),
    "adalog_score_w_self": (i32, [p, i32, i32, p, p, i32, i32, p, p]),
    "adalog_score_a_self": (i32, [p, i64, i32, p, p, i32, i32, i32, f64, p, i64, p, p]),
    "adalog_score_a_self_partial_elems": (i64, [i64, i32, i32]),
    "adalog_sorted_prefix_workspace_bytes": (i64, [i64, i64]),
    "adalog_sorted_prefix_build": (i32, [p, i64, i64, p, p, p, i64, p]),
    "adalog_score_self_sorted": (i32, [p, p, i64, i64, p, p, i32, i32, f64, p, p]),
    "adalog_quantile_rows": (i32, [p, i64, i64, i32, p, p, i32, p, p, i64, p]),
    "adalog_positive_percentile_rows": (i32, [p, i64, i64, i32, p, p, p, i64, p]),
    "adalog_select_workspace_bytes": (i64, [i64, i32]),
    "adalog_select_init": (i32, [p, i64, i64, i32, p, p]),
    "adalog_select_hist": (i32, [p, i64, i64, i32, i32, i32, i64, i32, i32, i32, p, p]),
    "adalog_select_pick": (i32, [p, i64, i32, i32, p, i32, p]),
    "adalog_select_quantile_out": (i32, [p, i64, i32, p, i32, p, p]),
    "adalog_select_value_out": (i32, [p, i64, i32, p, p]),
    "adalog_uniform_int_f32": (i32, [p, p, i64, p, p, i32, p]),
    "adalog_permute_heads": (i32, [p, p, i64, i64, i32, i32, i32, i32, p]),
    "adalog_merge_heads": (i32, [p, p, p, p, p, i64, i64, i32, i32, i32, p]),
    "adalog_qkv_quant_chunks": (i32, [i64, i64, i32]),
    "adalog_qkv_split_quant": (i32, [p, p, p, p, i64, i64, i32, i32, p, p, p, p, p]),
    "adalog_qkv_merge_quant_backward": (i32, [p, p, p, p, p, i64, i64, i32, i32, p, p, p, p, p, p, p]),
    "adalog_scaled_softmax": (i32, [p, p, i64, i32, f32, p]),
    "adalog_scaled_softmax_backward": (i32, [p, p, p, i64, i32, f32, p]),
    "adalog_uniform_fq_backward_blocks": (i32, [i64, i64, i64]),
    "adalog_uniform_fq_backward": (i32, [p, p, p, i64, p, p, i64, i64, i32, i32, p, p, p, p]),
    "adalog_log_fq_backward": (i32, [p, p, p, p, i64, p, p, i32, p, i32, p, p, p]),
    "adalog_adaround": (i32, [p, p, p, p, i64, i64, p, p, i32, i32, i32, p]),
    "adalog_adaround_t": (i32, [p, p, p, p, i64, i64, p, p, i32, i32, p]),
    "adalog_alpha_step_multi": (i32, [p, p, p, p, p, p, p, p, p, p, i32, f32, p, f32, f32, f32, p, f32, p, f32, p, p, p]),
    "adalog_round_loss": (i32, [p, i64, f32, p, p, p, f32, p, i32, p, p]),
    "adalog_round_loss_multi_workspace": (i64, [p, i32]),
    "adalog_round_loss_multi": (i32, [p, p, p, i32, f32, p, f32, p, p, p, p]),
    "adalog_gemm_win_ok": (i32, [i32, i32, i32, i32, i32, i32, i64]),
    "adalog_gemm_score_gen": (i32, [i32, p, i64, i32, i32, i64, i64, i32, i32, p, i64, i64, p, i32, p, i64, i32, p, i64, i64, f32, p, i64,
                              i64, p, i64, p]),
    "adalog_gemm_score_gen_ok": (i32, [i32, i32, i32, i32, i32, i32, i64, i64]),
    "adalog_gemm_score_avq": (i32, [p, i64, i32, i32, i64, i64, i32, i32, p, i64, i64, p, p, i32, p, i64, i32, p, i64, i64, f32, p, i64, i64,
                              p, i64, p]),
    "adalog_gemm_score_avq_ok": (i32, [i32, i32, i32, i32, i32, i64, i64, i32]),
    "adalog_gemm_mixed_ok": (i32, [i32, i32, i32, i32, i32, i64]),
    "adalog_rec_loss": (i32, [p, p, i64, f32, p, p, p]),
    "adalog_rec_loss_backward": (i32, [p, p, i64, f32, p, p, p]),
    "adalog_brecq_init": (i32, []),
    "adalog_adam_multi": (i32, [p, p, p, p, p, i32, f32, p, f32, f32, f32, p, p]),
    "adalog_gemm_f32x3_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "adalog_gemm_f32x3_planes_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32]),
    "adalog_gemm_f32x3_planes": (i32, [p, i64, p, i64, p, i64, i32, i32, i32, i32, i64, i64, p, f32, p, i32, i32, p, p]),
    "adalog_gemm_f32x3": (i32, [p, i64, i32, p, i64, i32, p, i64, i32, i32, i32, i32, i64, i64, i64, p, f32, p, i32, i32, i32, p, p]),
    "adalog_gemm_f32x3_add": (i32, [p, i64, i32, p, i64, i32, p, i64, i32, i32, i32, i32, i64, i64, i64, p, f32, p, i32, i32, i32, p, p, p]),
    "adalog_gemm_f32x3_g2": (i32, [p, i64, i32, p, i64, i32, p, i64, i32, i32, i32, i32, i64, i64, i64, i32, i64, i64, i64, p, f32, p,
                             i32, i32, i32, p, p]),
    "adalog_shift_fold": (i32, [p, p, p, p, i32, i32, p, p]),
    "adalog_minmax_rows": (i32, [p, i32, i32, i32, p, p, p]),
    "adalog_absminmax_cols": (i32, [p, i64, i32, i32, p, p, p]),
}

class FpcsTail(C.Structure):
    """adalog_fpcs_tail of include/adalog_hip.h: the ranking + next grid / commit step of an FPCS step as arguments"""
    _fields_ = [("k", C.c_int32), ("new_cnt", C.c_int32), ("has_clamp", C.c_int32), ("clamp_min", C.c_float),
                ("scale", p), ("zp", p), ("third", p), ("lin", p), ("delta_in", p), ("delta_out", p),
                ("out_scale", p), ("out_zp", p), ("out_third", p)]


_lib = None


class AdalogHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdalogHipError(
                f"{LIB_PATH} not found: build the HIP kernels first (python -c 'import __graft_entry__ as g; g.build()' "
                "or `make -C adalog_amd/csrc`).  The AdaLog MI355X path has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().adalog_last_error().decode("utf-8", "replace")
        raise AdalogHipError(f"{what} failed (rc={rc}): {msg}")
