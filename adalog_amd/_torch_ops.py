"""torch.ops.adalog.* -- loads the TORCH_LIBRARY registration (adalog_amd/csrc/libadalog_torch.so, built in-tree by
__graft_entry__.build() / adalog_amd/csrc/build_torch_ops.py) and exposes the namespace.

This is INTEGRATION.md level 1 (PyTorch custom ops, HIP dispatch key only); the ctypes binding in _lib.py is level 2 and
is what everything falls back to when the torch extension has not been built.  Neither has a CPU kernel.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadalog_torch.so")
OPS = ("uniform_fake_quant", "log_fake_quant", "log2_shift", "score_act_fused", "topk", "pack_uniform", "pack_adalog", "gemm_score",
       "topk_next", "score_w_self", "score_a_self", "sorted_prefix", "score_self_sorted", "score_act_gen", "gemm_score_partial", "finish_topk_next", "score_act_gen_partial")
_state = None


def available() -> bool:
    """True when torch.ops.adalog is registered (loads the library on first call)."""
    global _state
    if _state is None:
        _state = False
        if os.path.exists(LIB_PATH) and os.environ.get("ADALOG_TORCH_OPS", "1") != "0":
            try:
                torch.ops.load_library(LIB_PATH)
                _state = all(hasattr(torch.ops.adalog, n) for n in OPS)
            except (OSError, RuntimeError):
                _state = False
    return _state


def ns():
    if not available():
        raise RuntimeError(f"{LIB_PATH} is not built: run python adalog_amd/csrc/build_torch_ops.py (or __graft_entry__.build())")
    return torch.ops.adalog
