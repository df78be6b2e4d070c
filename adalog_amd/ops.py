"""Tensor-level wrappers over the C ABI (include/adalog_hip.h): torch is used for device memory and streams only.

Every function takes CUDA(HIP) fp32 tensors, validates them, and enqueues the hand-written kernels on torch's current
stream.  No function here computes anything with torch ops, and none synchronises the host.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from . import _lib, _torch_ops

I8, BF16, F32, FP8 = 0, 1, 2, 3        # FP8: e4m3 bytes holding q - z of <= 4-bit layers exactly (same MFMA rate as int8)
BF16_FP8 = 4                           # gemm_score only: bf16 rows x fp8 candidate columns, converted in registers (gemm_mixed_ok)
# bench.py sets this to a list to time the scoring GEMM launches: (dtype, M, N, Kp, C, G, A.data_ptr(), start, end) with
# the two events recorded on the launch stream immediately around the adalog_gemm_score kernel (not the finish kernel)
GEMM_EVENTS = None
_ESZ = {I8: 1, BF16: 2, F32: 4, FP8: 1}
_TORCH_DT = {I8: torch.int8, BF16: torch.bfloat16, F32: torch.float32, FP8: torch.float8_e4m3fn}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _top(name, *args):
    """call torch.ops.adalog.<name>; C-ABI failures surface as AdalogHipError, like on the ctypes route"""
    try:
        return getattr(torch.ops.adalog, name)(*args)
    except RuntimeError as e:
        if "adalog::" in str(e):
            raise _lib.AdalogHipError(str(e)) from e
        raise


def _timed(meta, call):
    """bench.py's per-launch timing (GEMM_EVENTS is a list while it runs): events on the current stream around the launch,
    on whichever route the launch takes"""
    if GEMM_EVENTS is None:
        return call()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    r = call()
    ev1.record()
    GEMM_EVENTS.append(meta + (ev0, ev1, _lib.load().adalog_last_kernel().decode()))
    return r


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.AdalogHipError(f"{name}: expected a tensor on the HIP device, got {t.device} (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def pad_k(K: int, dtype: int, align: int = 128) -> int:
    """K rounded up to rows of a multiple of ``align`` bytes: 128 (one cache line; every kernel), or 64 for operands that
    only the streaming search kernel reads (its K-step) -- q.k^T with head_dim 64 then packs half the bytes."""
    per = align // _ESZ[dtype]
    return ((K + per - 1) // per) * per


def broadcast_layout(x_shape, p_shape) -> Tuple[int, int]:
    """(n_channels, inner) such that channel(i) = (i // inner) % n_channels reproduces torch broadcasting of a
    parameter of shape ``p_shape`` against ``x_shape`` (uniform.py:29-36 relies on plain broadcasting)."""
    xs = list(x_shape)
    ps = [1] * (len(xs) - len(p_shape)) + list(p_shape)
    if len(ps) != len(xs):
        raise ValueError(f"parameter shape {tuple(p_shape)} has more dims than input {tuple(x_shape)}")
    nz = [i for i, s in enumerate(ps) if s != 1]
    if not nz:
        return 1, int(math.prod(xs)) if xs else 1
    a, b = nz[0], nz[-1]
    for i in range(a, b + 1):
        if ps[i] != xs[i]:
            raise ValueError(f"unsupported broadcast of {tuple(p_shape)} against {tuple(x_shape)}")
    return int(math.prod(ps[a:b + 1])), int(math.prod(xs[b + 1:]))


# ------------------------------------------------------------------------------------------------ K1-K3
def uniform_fake_quant(x, scale, zero_point, n_bits: int, sym: bool = False, want_bins: bool = False,
                       want_y: bool = True):
    x = _f32c(x, "x")
    scale = _f32c(scale, "scale")
    zp = None if sym else _f32c(zero_point, "zero_point")
    n_ch, inner = broadcast_layout(x.shape, scale.shape)
    if want_y and not want_bins and _torch_ops.available():          # level 1: the registered PyTorch custom op
        return _top("uniform_fake_quant", x, scale, zp, n_ch, inner, int(n_bits), bool(sym))
    y = torch.empty_like(x) if want_y else None
    bins = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_bins else None
    lib = _lib.load()
    rc = lib.adalog_uniform_fake_quant_f32(_ptr(x), _ptr(y), _ptr(bins), x.numel(), _ptr(scale), _ptr(zp), n_ch, inner,
                                           int(n_bits), int(bool(sym)), _stream())
    _lib.check(rc, "adalog_uniform_fake_quant_f32")
    return (y, bins) if want_bins else y


def log_fake_quant(x, scale, q, table1, table2, n_bits: int, shift=None, sub_shift: bool = False,
                   train_form: bool = False, want_bins: bool = False, want_y: bool = True, pre_gelu: bool = False):
    """``pre_gelu``: the quantiser's input is GELU(x) (erf form), applied in the kernel (adalog_log_fake_quant_f32_pre)."""
    x = _f32c(x, "x")
    scale = _f32c(scale, "scale")
    if scale.numel() != 1:
        raise ValueError("AdaLog quantisers are per-tensor (scale must have one element)")
    if q.dtype != torch.int64 or not q.is_cuda:
        raise TypeError("q must be an int64 tensor on the device (the quantiser's buffer)")
    if want_y and not want_bins and not train_form and not pre_gelu and _torch_ops.available():
        return _top("log_fake_quant", x, scale, q, _f32c(table1, "table1"), _f32c(table2, "table2"), int(n_bits),
                    None if shift is None else _f32c(shift, "shift"), bool(sub_shift))
    y = torch.empty_like(x) if want_y else None
    bins = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_bins else None
    lib = _lib.load()
    rc = lib.adalog_log_fake_quant_f32_pre(_ptr(x), _ptr(y), _ptr(bins), x.numel(), _ptr(scale), _ptr(q),
                                           _ptr(None if train_form else _f32c(table1, "table1")),
                                           _ptr(None if train_form else _f32c(table2, "table2")), int(n_bits),
                                           _ptr(None if shift is None else _f32c(shift, "shift")), int(bool(sub_shift)),
                                           int(bool(train_form)), int(bool(pre_gelu)), _stream())
    _lib.check(rc, "adalog_log_fake_quant_f32")
    return (y, bins) if want_bins else y


# ------------------------------------------------------------------------------------------------ operand packing
def _view3(x3: torch.Tensor):
    if x3.dim() != 3 or x3.dtype != torch.float32 or not x3.is_cuda:
        raise ValueError("operand must be a 3-D float32 device view [G, R, K] (any strides)")
    return x3.shape[0], x3.shape[1], x3.shape[2], x3.stride(0), x3.stride(1), x3.stride(2)


def pack_uniform(x3, scale, zero_point, C: int, pc: int, gmod: int, pg: int, pr: int, n_bits: int, dtype: int = I8,
                 want_rowsum: bool = False, c_inner: bool = False, k_align: int = 128):
    """-> packed [C, G, R, Kp] (int8 / bf16 / fp32), or [1, G, R*C, Kp] with candidates innermost when ``c_inner``
    [+ int32 rowsum [C, G, R]]."""
    G, R, K, sg, sr, sk = _view3(x3)
    scale, zero_point = _f32c(scale, "scale"), _f32c(zero_point, "zero_point")
    Kp = pad_k(K, dtype, k_align)
    if _torch_ops.available():
        out, rowsum = _top("pack_uniform", x3, scale, zero_point, int(C), int(pc), int(gmod), int(pg), int(pr), int(n_bits),
                           int(dtype), int(Kp), bool(want_rowsum), bool(c_inner))
        out.k_valid = K
        return (out, rowsum) if want_rowsum else out
    shape = (1, G, R * C, Kp) if c_inner else (C, G, R, Kp)
    out = torch.empty(shape, dtype=_TORCH_DT[dtype], device=x3.device)
    rowsum = torch.empty((C, G, R), dtype=torch.int32, device=x3.device) if want_rowsum else None
    rc = _lib.load().adalog_pack_uniform(x3.data_ptr(), G, R, K, sg, sr, sk, _ptr(scale), _ptr(zero_point), C, pc, gmod,
                                        pg, pr, int(n_bits), dtype, out.data_ptr(), Kp, _ptr(rowsum), int(bool(c_inner)),
                                        _stream())
    _lib.check(rc, "adalog_pack_uniform")
    out.k_valid = K                                     # columns [K, Kp) are zero padding: gemm_score may skip them
    return (out, rowsum) if want_rowsum else out


def pack_adalog(x3, scale, qv, C: int, pc: int, gmod: int, pg: int, n_bits: int, mant37, shift=None,
                clamp_u: bool = True, c_inner: bool = False, k_align: int = 128, pre_gelu: bool = False):
    """``pre_gelu``: the operand is GELU(x3) (erf form) -- applied in the packer's loader (adalog_pack_adalog_bf16_pre)."""
    G, R, K, sg, sr, sk = _view3(x3)
    Kp = pad_k(K, BF16, k_align)
    if pre_gelu:
        out = torch.empty((1, G, R * C, Kp) if c_inner else (C, G, R, Kp), dtype=torch.bfloat16, device=x3.device)
        rc = _lib.load().adalog_pack_adalog_bf16_pre(x3.data_ptr(), G, R, K, sg, sr, sk, _ptr(_f32c(scale, "scale")),
                                                    _ptr(_f32c(qv, "qv")), C, pc, gmod, pg, int(n_bits), _ptr(_f32c(mant37, "mant37")),
                                                    _ptr(None if shift is None else _f32c(shift, "shift")), int(bool(clamp_u)),
                                                    out.data_ptr(), Kp, int(bool(c_inner)), 1, _stream())
        _lib.check(rc, "adalog_pack_adalog_bf16_pre")
        out.k_valid = K
        return out
    if _torch_ops.available():
        out = _top("pack_adalog", x3, _f32c(scale, "scale"), _f32c(qv, "qv"), int(C), int(pc), int(gmod), int(pg), int(n_bits),
                   _f32c(mant37, "mant37"), None if shift is None else _f32c(shift, "shift"), bool(clamp_u), int(Kp), bool(c_inner))
        out.k_valid = K
        return out
    out = torch.empty((1, G, R * C, Kp) if c_inner else (C, G, R, Kp), dtype=torch.bfloat16, device=x3.device)
    rc = _lib.load().adalog_pack_adalog_bf16(x3.data_ptr(), G, R, K, sg, sr, sk, _ptr(_f32c(scale, "scale")),
                                            _ptr(_f32c(qv, "qv")), C, pc, gmod, pg, int(n_bits),
                                            _ptr(_f32c(mant37, "mant37")),
                                            _ptr(None if shift is None else _f32c(shift, "shift")), int(bool(clamp_u)),
                                            out.data_ptr(), Kp, int(bool(c_inner)), _stream())
    _lib.check(rc, "adalog_pack_adalog_bf16")
    out.k_valid = K
    return out


def pack_raw(x3):
    G, R, K, sg, sr, sk = _view3(x3)
    Kp = pad_k(K, F32)
    out = torch.empty((1, G, R, Kp), dtype=torch.float32, device=x3.device)
    rc = _lib.load().adalog_pack_raw_f32(x3.data_ptr(), G, R, K, sg, sr, sk, out.data_ptr(), Kp, _stream())
    _lib.check(rc, "adalog_pack_raw_f32")
    out.k_valid = K
    return out


def pack_split3(x3, k_align: int = 128):
    """fp32 [G, R, K] -> bf16 [1, G, R, 3*Kt]: the three-term bf16 image hi | mid | lo of an unquantised operand (exact:
    hi + mid + lo == x), Kt = pad_k(K, BF16).  Pairs with a bf16 candidate operand repeated three times along K."""
    G, R, K, sg, sr, sk = _view3(x3)
    Kt = pad_k(K, BF16, k_align)
    out = torch.empty((1, G, R, 3 * Kt), dtype=torch.bfloat16, device=x3.device)
    rc = _lib.load().adalog_pack_split3_bf16(x3.data_ptr(), G, R, K, sg, sr, sk, out.data_ptr(), Kt, _stream())
    _lib.check(rc, "adalog_pack_split3_bf16")
    out.k_valid = 2 * Kt + K
    out.flops_k = K                                     # the ALGORITHM's K (flops of the product, not of its three-term decomposition)
    return out


# ------------------------------------------------------------------------------------------------ scoring GEMM
class Strided:
    """A device fp32 parameter with (candidate, head, column) element strides for the GEMM epilogue."""
    __slots__ = ("t", "c", "g", "n")

    def __init__(self, t, c=0, g=0, n=0):
        self.t, self.c, self.g, self.n = t, int(c), int(g), int(n)

    def checked(self):
        self.t = _f32c(self.t, "epilogue parameter")
        return self


def _layout(M, n_cols, c_grid, G, gmod, ref_div, reduce_cols, dtype, Kp, k_valid, ref_transposed):
    """-> (floats to allocate, MT, Npad, mode) of the partial buffer adalog_gemm_score will write for this launch;
    mode: 0 = [C][G][MT][Npad], 1 = candidate innermost, 2 = per-workgroup fp64 accumulators."""
    import ctypes
    mt, npad, mode = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    elems = _lib.load().adalog_gemm_score_layout(M, n_cols, c_grid, G, gmod, ref_div, reduce_cols, dtype, Kp, k_valid,
                                                 int(bool(ref_transposed)), ctypes.byref(mt), ctypes.byref(npad),
                                                 ctypes.byref(mode))
    return elems, mt.value, npad.value, mode.value


class FpcsTail:
    """The tail of one FPCS step (reference linear.py:483-523: top-k of the scores, then the next 16 x 8 grid around the survivors or
    the committed winner) as ARGUMENTS (adalog_fpcs_tail, csrc/fpcs_tail.h), so that a kernel producing final scores runs it in its own
    launch, or adalog_topk_next_tail / adalog_finish_topk_next_tail run it.  ``delta_in`` is read, ``delta_out`` written (they may be
    the same tensor): the memoised grid spacing is never modified.  ``out`` (commit step only): write the winner straight into these
    [cols] tensors -- the quantiser's own parameter storage -- instead of fresh ones."""
    __slots__ = ("c", "scale", "zp", "third", "o_s", "o_z", "o_t", "new_cnt", "keep")

    def __init__(self, scale, zp, third, k: int, new_cnt: int, lin, delta_in, delta_out, clamp_min, out=None):
        scale = _f32c(scale, "scale")
        cols = scale.shape[1]
        rows = k * new_cnt if new_cnt > 0 else 1
        dev = scale.device
        self.scale, self.zp, self.third, self.new_cnt = scale, zp, third, new_cnt
        if out is not None and new_cnt == 0:
            o_s, o_z, o_t = out
            for o_, src in ((o_s, scale), (o_z, zp), (o_t, third)):
                if (o_ is None) != (src is None) or (o_ is not None and not (o_.is_contiguous() and o_.dtype == torch.float32
                                                                             and o_.numel() == cols and o_.device == dev)):
                    raise ValueError("FpcsTail: commit targets must be contiguous fp32 [cols] tensors matching the grid's planes")
            self.o_s, self.o_z, self.o_t = o_s, o_z, o_t
        else:
            mk = lambda src: None if src is None else torch.empty((rows, cols), dtype=torch.float32, device=dev)
            self.o_s, self.o_z, self.o_t = mk(scale), mk(zp), mk(third)
        self.keep = (lin, delta_in, delta_out)
        self.c = _lib.FpcsTail(int(k), int(new_cnt), int(clamp_min is not None), float(clamp_min if clamp_min is not None else 0.0),
                               scale.data_ptr(), _ptr(zp), _ptr(third), _ptr(lin), _ptr(delta_in), _ptr(delta_out),
                               self.o_s.data_ptr(), _ptr(self.o_z), _ptr(self.o_t))

    def ref(self):
        import ctypes
        return ctypes.byref(self.c)

    def result(self):
        """what topk_next returns: the next grid [k * new_cnt, cols] planes, or the committed winner's [cols] planes"""
        if self.new_cnt == 0:
            f = lambda t: None if t is None else t.reshape(-1)
            return f(self.o_s), f(self.o_z), f(self.o_t)
        return self.o_s, self.o_z, self.o_t


class PendingScores:
    """The partial sums a scoring kernel left behind, not yet reduced to scores: what ``gemm_score(..., defer=True)`` and
    ``score_act_gen(..., defer=True)`` return.  ``finish()`` gives the [C, cols] scores; on one GPU the FPCS driver hands the
    object to ``finish_topk_next`` instead, which reduces, ranks and writes the next candidate grid in ONE launch."""
    __slots__ = ("partial", "MT", "n_last", "Npad", "C", "G", "gmod", "keep_h", "keep_n", "mode", "norm", "cols", "N")

    def __init__(self, partial, MT, n_last, Npad, C, G, gmod, keep_h, keep_n, mode, norm, N):
        self.partial, self.MT, self.n_last, self.Npad, self.C, self.G, self.gmod = partial, MT, n_last, Npad, C, G, gmod
        self.keep_h, self.keep_n, self.mode, self.norm, self.N = bool(keep_h), bool(keep_n), mode, float(norm), N
        self.cols = (gmod if keep_h else 1) * (N if keep_n else 1)

    def _ws(self):
        lib = _lib.load()
        nb = lib.adalog_finish_workspace_bytes(self.MT, self.n_last, self.C, self.G, int(self.keep_n), self.mode)
        return (torch.empty(nb // 8, dtype=torch.float64, device=self.partial.device) if nb else None), nb

    def finish(self):
        lib = _lib.load()
        scores = torch.empty((self.C, self.cols), dtype=torch.float32, device=self.partial.device)
        ws, nb = self._ws()
        rc = lib.adalog_finish_scores(self.partial.data_ptr(), scores.data_ptr(), self.MT, self.n_last, self.Npad, self.C, self.G,
                                      self.gmod, int(self.keep_h), int(self.keep_n), self.mode, self.norm, _ptr(ws), nb, _stream())
        _lib.check(rc, "adalog_finish_scores")
        return scores


def finish_topk_next(pend: PendingScores, scale, zp, third, k: int, new_cnt: int, lin, delta, clamp_min: Optional[float], tail=None):
    """finish(pend) followed by topk_next(...) -- one launch where the partial layout allows it (csrc/gemm_finish.inc).  Same return
    value as topk_next.  Single-GPU form: with several ranks the scores are all-reduced between the two steps.  ``tail``: a prepared
    FpcsTail (then scale .. clamp_min are ignored)."""
    if tail is None:
        tail = FpcsTail(scale, zp, third, k, new_cnt, lin, delta, delta, clamp_min)
    cols = pend.cols
    assert tail.scale.shape[0] == pend.C and tail.scale.shape[1] == cols
    scores = torch.empty((pend.C, cols), dtype=torch.float32, device=tail.scale.device)
    ws, nb = pend._ws()
    rc = _lib.load().adalog_finish_topk_next_tail(pend.partial.data_ptr(), scores.data_ptr(), pend.MT, pend.n_last, pend.Npad, pend.C,
                                                 pend.G, pend.gmod, int(pend.keep_h), int(pend.keep_n), pend.mode, pend.norm, _ptr(ws),
                                                 nb, tail.ref(), _stream())
    _lib.check(rc, "adalog_finish_topk_next")
    return tail.result()


def gemm_score(dtype: int, A, B, M: int, N: int, C: int, G: int, gmod: int, ref, sa: Strided, sb: Strided,
               bias: Optional[Strided], keep_h: bool, keep_n: bool, norm: float, sa_mul: float = 1.0,
               ref_div: int = 1, order: int = 1, ref_transposed: bool = False, row_scale=None, row_bias=None,
               defer: bool = False):
    """scores = finish(gemm_score(...)).  A: [C|1, G|1, M, Kp], B: [C|1, G|1, N, Kp]; ref: [G, M, N] fp32.

    With ``ref_div`` = P > 1 B is packed candidates-innermost ([1, G, N*P, Kp]), the GEMM runs over N*P columns and the
    P candidates of a column group share one reference column; scores come back as [P, ...] all the same.
    ``ref_transposed``: ref is stored [G, N, M].  ``row_scale`` / ``row_bias``: optional per-row factor / offset [M].
    Returns fp32 scores of shape [C, (gmod if keep_h) * (N if keep_n)].
    """
    lib = _lib.load()
    sa, sb = sa.checked(), sb.checked()
    bias = None if bias is None else bias.checked()
    Kp = A.shape[-1]
    n_cols = N * ref_div
    c_grid = 1 if ref_div > 1 else C
    assert B.shape[-1] == Kp and A.shape[-2] == M and B.shape[-2] == n_cols
    assert A.dtype == _TORCH_DT[BF16 if dtype == BF16_FP8 else dtype] and B.dtype == _TORCH_DT[FP8 if dtype == BF16_FP8 else dtype]
    assert A.is_contiguous() and B.is_contiguous()
    sAc = 0 if A.shape[0] == 1 else A.stride(0)
    sAg = 0 if A.shape[1] == 1 and G > 1 else A.stride(1)
    sBc = 0 if B.shape[0] == 1 else B.stride(0)
    sBg = 0 if B.shape[1] == 1 and G > 1 else B.stride(1)
    ref = _f32c(ref, "ref")
    if ref_transposed:                         # ref stored [G, N, M]: element (m, n) at n*M + m
        assert ref.shape[-1] == M and ref.shape[-2] == N
        ldr, ref_cs = 1, M
    else:
        ldr, ref_cs = ref.shape[-1], 1
    sRg = 0 if G == 1 else ref.shape[-1] * ref.shape[-2]
    reduce_cols = 0 if keep_n else 1                    # column axis not kept: the kernel may sum it (per tile / per workgroup)
    k_valid = min(getattr(A, "k_valid", Kp), getattr(B, "k_valid", Kp))
    def layout():
        n_part, MT, Npad, mode = _layout(M, n_cols, c_grid, G, gmod, ref_div, reduce_cols, dtype, Kp, k_valid, ref_transposed)
        # reduced column axis: one partial per n-tile (Npad = NT) / per workgroup
        return n_part, MT, Npad, mode, (Npad if (reduce_cols and mode != 1) else N)
    k_alg = min(getattr(A, "flops_k", k_valid), getattr(B, "flops_k", k_valid))   # (an operand split into exact terms counts once)
    meta = (dtype, M, N, k_alg, C, G)                                 # un-padded, algorithmic K
    if _torch_ops.available():
        args = (int(dtype), A, B, int(M), int(N), int(C), int(G), int(gmod), int(k_valid), ref, sa.t, sa.c, sa.g,
                float(sa_mul), sb.t, sb.c, sb.g, sb.n, None if bias is None else bias.t, 0 if bias is None else bias.c,
                0 if bias is None else bias.g, 0 if bias is None else bias.n, bool(keep_h), bool(keep_n), float(norm),
                int(ref_div), int(order), bool(ref_transposed),
                None if row_scale is None else _f32c(row_scale, "row_scale"),
                None if row_bias is None else _f32c(row_bias, "row_bias"))
        if not defer:
            return _timed(meta, lambda: _top("gemm_score", *args))
        n_part, MT, Npad, mode, n_last = layout()
        return PendingScores(_timed(meta, lambda: _top("gemm_score_partial", *args)), MT, n_last, Npad, C, G, gmod, keep_h, keep_n,
                             mode, norm, N)
    n_part, MT, Npad, mode, n_last = layout()
    partial = torch.empty((n_part + 1) // 2, dtype=torch.float64, device=A.device).view(torch.float32)   # 8-byte aligned
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_gemm_score(dtype, A.data_ptr(), B.data_ptr(), sAc, sAg, sBc, sBg, M, n_cols, Kp, k_valid, c_grid, G, gmod,
                               ref.data_ptr(), ldr, sRg, ref_cs, ref_div, sa.t.data_ptr(), sa.c, sa.g, float(sa_mul),
                               sb.t.data_ptr(), sb.c, sb.g, sb.n,
                               None if bias is None else bias.t.data_ptr(),
                               0 if bias is None else bias.c, 0 if bias is None else bias.g,
                               0 if bias is None else bias.n,
                               _ptr(None if row_scale is None else _f32c(row_scale, "row_scale")),
                               _ptr(None if row_bias is None else _f32c(row_bias, "row_bias")),
                               partial.data_ptr(), n_part, None, 0, 0, 0, int(order), reduce_cols, _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((dtype, M, N, k_alg, C, G, ev0, ev1, lib.adalog_last_kernel().decode()))     # un-padded, algorithmic K
    _lib.check(rc, "adalog_gemm_score")
    pend = PendingScores(partial, MT, n_last, Npad, C, G, gmod, keep_h, keep_n, mode, norm, N)
    return pend if defer else pend.finish()


def gemm_mixed_ok(M: int, N: int, G: int, gmod: int, ref_div: int, k_valid: int) -> bool:
    """True when gemm_score(BF16_FP8, ...) takes this shape (C = 1, candidates innermost, transposed reference, column axis
    summed): A bf16 [.., M, Kp], B fp8 [.., N * ref_div, Kp] with Kp = 64 (K <= 64: windows) or 256 (K = 193..256)."""
    return bool(_lib.load().adalog_gemm_mixed_ok(int(M), int(N) * int(ref_div), int(G), int(gmod), int(ref_div), int(k_valid)))


def gemm_win_ok(dtype: int, M: int, N: int, G: int, gmod: int, ref_div: int, k_valid: int) -> bool:
    """True when gemm_score (C = 1, candidates innermost, transposed reference, column axis summed) runs this shape on the
    window kernel: int8 / fp8 operands of K <= 32 may then be packed with 32-byte rows (``k_align=32``)."""
    return bool(_lib.load().adalog_gemm_win_ok(int(dtype), int(M), int(N) * int(ref_div), int(G), int(gmod), int(ref_div), int(k_valid)))


def gemm_score_gen_ok(dtype: int, M: int, N: int, G: int, gmod: int, ref_div: int, k_valid: int, Kp: int) -> bool:
    """True when gemm_score_gen takes this attention-search shape (N = source rows; ref_div candidates)."""
    return bool(_lib.load().adalog_gemm_score_gen_ok(int(dtype), int(M), int(N) * int(ref_div), int(G), int(gmod), int(ref_div),
                                                     int(k_valid), int(Kp)))


def gemm_score_gen(dtype: int, A, src3, zp, n_bits: int, M: int, N: int, P: int, G: int, gmod: int, ref, sa: Strided, sb: Strided,
                   keep_h: bool, norm: float, sa_mul: float = 1.0):
    """gemm_score(dtype, A, pack_uniform(src3, sb.t, zp, ...), M, N, P, ..., ref_div=P, ref_transposed=True, defer=True) without
    the packed candidate operand: src3 [G, N, K] fp32 is quantised inside the kernel (adalog_gemm_score_gen).  sb: the candidates'
    scales as the epilogue's column factors (Strided(scale [P, H], c=H, g=0|1)); zp: their zero points, same layout.
    -> PendingScores."""
    lib = _lib.load()
    sa, sb = sa.checked(), sb.checked()
    src3 = _f32c(src3, "src")
    zp = _f32c(zp, "zero_point")
    Gs, Ns, K = src3.shape
    Kp = A.shape[-1]
    assert (Gs, Ns) == (G, N) and A.shape[-2] == M and A.is_contiguous() and A.dtype == _TORCH_DT[dtype] and zp.shape == sb.t.shape
    ref = _f32c(ref, "ref")
    assert ref.shape[-1] == M and ref.shape[-2] == N
    sAg = 0 if A.shape[1] == 1 and G > 1 else A.stride(1)
    sRg = 0 if G == 1 else M * N
    n_part, MT, Npad, mode = _layout(M, N * P, 1, G, gmod, P, 1, dtype, Kp, K, True)
    n_last = Npad if mode != 1 else N
    partial = torch.empty((n_part + 1) // 2, dtype=torch.float64, device=A.device).view(torch.float32)
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_gemm_score_gen(int(dtype), A.data_ptr(), sAg, M, N * P, Kp, K, G, gmod, src3.data_ptr(), K, N * K, zp.data_ptr(),
                                   int(n_bits), ref.data_ptr(), sRg, P, sa.t.data_ptr(), sa.c, sa.g, float(sa_mul), sb.t.data_ptr(),
                                   sb.c, sb.g, partial.data_ptr(), n_part, _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((dtype, M, N, K, P, G, ev0, ev1, lib.adalog_last_kernel().decode()))
    _lib.check(rc, "adalog_gemm_score_gen")
    return PendingScores(partial, MT, n_last, Npad, P, G, gmod, keep_h, False, mode, norm, N)


def adalog_value_lut(q_all: torch.Tensor, n_bits: int, mant37: torch.Tensor) -> torch.Tensor:
    """[2^n_bits + 1, P] int32: entry [k][c] = bf16 bits of the value of bin k under base q_all[c] -- mant37[(k q) mod 37] * 2^-((k q) // 37),
    0 beyond 2^-100 -- exactly what pack_adalog writes for that bin; the last row (the masked code) is 0.  (Host-side table of the
    log-base kernel; reference logarithm.py:77-81.)"""
    nb = 1 << n_bits
    k = torch.arange(nb, device=q_all.device, dtype=torch.float32).view(nb, 1)
    kq = (k * q_all.view(1, -1).float()).long()
    t, j = kq // 37, kq % 37
    v = torch.ldexp(mant37.float()[j], (-t).to(torch.int32))
    v = torch.where(t > 100, torch.zeros_like(v), v)
    vb = v.to(torch.bfloat16)
    bits = vb.view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.cat([bits, torch.zeros(1, bits.shape[1], dtype=torch.int32, device=bits.device)], 0).contiguous()


def gemm_score_avq_ok(M: int, N: int, G: int, gmod: int, P: int, k_valid: int, Kp: int, n_bits: int) -> bool:
    return bool(_lib.load().adalog_gemm_score_avq_ok(int(M), int(N), int(G), int(gmod), int(P), int(k_valid), int(Kp), int(n_bits)))


def gemm_score_avq(A, src3, q_all, lut, n_bits: int, M: int, N: int, P: int, G: int, gmod: int, ref, sa: Strided, sb: Strided,
                   norm: float, sa_mul: float = 1.0):
    """The log-base search's scoring call (gemm_score(BF16, A, pack_adalog(src3, 1, q_all, ...), M, N, P, ..., keep_h=False)) without
    the packed candidate operand: src3 [G, N, K] fp32 probabilities are quantised inside the kernel (adalog_gemm_score_avq).
    -> scores [P, 1]."""
    lib = _lib.load()
    sa, sb = sa.checked(), sb.checked()
    src3, q_all, ref = _f32c(src3, "src"), _f32c(q_all, "q"), _f32c(ref, "ref")
    Gs, Ns, K = src3.shape
    Kp = A.shape[-1]
    assert (Gs, Ns) == (G, N) and A.shape[-2] == M and A.is_contiguous() and A.dtype == torch.bfloat16
    assert lut.dtype == torch.int32 and lut.is_contiguous() and lut.shape == ((1 << n_bits) + 1, P)
    assert ref.shape[-1] == M and ref.shape[-2] == N
    sAg = 0 if A.shape[1] == 1 and G > 1 else A.stride(1)
    sRg = 0 if G == 1 else M * N
    n_part, MT, Npad, mode = _layout(M, N * P, 1, G, gmod, P, 1, BF16, Kp, K, True)
    partial = torch.empty((n_part + 1) // 2, dtype=torch.float64, device=A.device).view(torch.float32)
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_gemm_score_avq(A.data_ptr(), sAg, M, N, Kp, K, G, gmod, src3.data_ptr(), K, N * K, q_all.data_ptr(), lut.data_ptr(),
                                   int(n_bits), ref.data_ptr(), sRg, P, sa.t.data_ptr(), sa.c, sa.g, float(sa_mul), sb.t.data_ptr(),
                                   sb.c, sb.g, partial.data_ptr(), n_part, _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((BF16, M, N, K, P, G, ev0, ev1, lib.adalog_last_kernel().decode()))
    _lib.check(rc, "adalog_gemm_score_avq")
    return PendingScores(partial, MT, Npad if mode != 1 else N, Npad, P, G, gmod, False, False, mode, norm, N).finish()


def gemm_out(dtype: int, A, B, M: int, N: int, G: int, gmod: int, sa: Strided, sb: Strided, bias: Optional[Strided],
             sa_mul: float = 1.0, addend=None, heads_last: int = 0):
    """Quantised forward: out[g] = (A[g] . B[g]^T) * sa * sb[n] + bias[n] (+ addend[g])   -> fp32 [G, M, N].
    ``addend``: fp32 [G, M, N] contiguous, added in the epilogue (the residual stream).  ``heads_last`` = H > 0 (G = B * H): the
    result is written as [B, M, H, N] storage -- returned in that shape -- so that the transpose(1, 2).reshape(B, M, H * N) after
    softmax . v is a view (adalog_gemm_out_ex: two-level output groups)."""
    lib = _lib.load()
    sa, sb = sa.checked(), sb.checked()
    bias = None if bias is None else bias.checked()
    Kp = A.shape[-1]
    assert B.shape[-1] == Kp and A.dtype == _TORCH_DT[dtype] and B.dtype == _TORCH_DT[dtype]
    sAg = 0 if A.shape[1] == 1 and G > 1 else A.stride(1)
    sBg = 0 if B.shape[1] == 1 and G > 1 else B.stride(1)
    if addend is not None or heads_last:
        assert dtype in (I8, BF16) and sa.c == 0 and sb.c == 0 and (bias is None or bias.c == 0)
        H = int(heads_last)
        if H:
            assert G % H == 0 and addend is None
            out = torch.empty((G // H, M, H, N), dtype=torch.float32, device=A.device)
            ldo, sOg, gi, sOo = H * N, N, H, M * H * N
        else:
            addend = _f32c(addend, "addend")
            assert addend.numel() == G * M * N
            out = torch.empty((G, M, N), dtype=torch.float32, device=A.device)
            ldo, sOg, gi, sOo = N, M * N, 0, 0
        rc = lib.adalog_gemm_out_ex(dtype, A.data_ptr(), B.data_ptr(), sAg, sBg, M, N, Kp, G, gmod, sa.t.data_ptr(), sa.g, float(sa_mul),
                                    sb.t.data_ptr(), sb.g, sb.n, None if bias is None else bias.t.data_ptr(),
                                    0 if bias is None else bias.g, 0 if bias is None else bias.n, _ptr(addend), out.data_ptr(),
                                    ldo, sOg, gi, sOo, _stream())
        _lib.check(rc, "adalog_gemm_out_ex")
        return out
    out = torch.empty((G, M, N), dtype=torch.float32, device=A.device)
    rc = lib.adalog_gemm_score(dtype, A.data_ptr(), B.data_ptr(), 0, sAg, 0, sBg, M, N, Kp, 0, 1, G, gmod, None, 0, 0, 1, 1,
                               sa.t.data_ptr(), sa.c, sa.g, float(sa_mul), sb.t.data_ptr(), sb.c, sb.g, sb.n,
                               None if bias is None else bias.t.data_ptr(),
                               0 if bias is None else bias.c, 0 if bias is None else bias.g,
                               0 if bias is None else bias.n, None, None,
                               None, 0, out.data_ptr(), N, 0, M * N, 0, 0, _stream())
    _lib.check(rc, "adalog_gemm_score(out)")
    return out


def gemm_out_gen_ok(x3, Kp: int, n_bits: int) -> bool:
    """gemm_out_gen takes this activation: fp32 [G, M, K], unit stride along K, rows and groups 16-byte aligned (a strided view such
    as the class token x[:, 0] qualifies), K a multiple of 16 covered by the packed operand's Kp"""
    return (x3.dim() == 3 and x3.dtype == torch.float32 and x3.is_cuda and x3.stride(2) == 1 and x3.stride(1) % 4 == 0
            and x3.stride(0) % 4 == 0 and x3.stride(1) >= x3.shape[2] and x3.shape[-1] % 16 == 0 and x3.shape[-1] <= Kp
            and 2 <= n_bits <= 7 and x3.data_ptr() % 16 == 0 and os.environ.get("ADALOG_QF_GEN", "1") != "0")


def gemm_out_gen(x3, a_scale, a_zp, n_bits: int, B, N: int, gmod: int, sa: Strided, sb: Strided, bias: Optional[Strided],
                 sa_mul: float = 1.0, addend=None):
    """Quantised forward with the A-side fake quantisation inside the GEMM's loader (adalog_gemm_out_gen, k_gemm_cand<GENA>):
    out[g] = (q_a(x3[g]) . B[g]^T) * sa * sb[n] + bias[n] -> fp32 [G, M, N], q_a the per-tensor (one (scale, zp)) or per-head
    ((scale, zp)[g % gmod]) uniform quantiser -- gemm_out(I8, pack_uniform(x3, ...), B, ...) without the pack launch and the int8
    image of the activation (reference linear.py:46-51, matmul.py:43-45)."""
    lib = _lib.load()
    sa, sb = sa.checked(), sb.checked()
    bias = None if bias is None else bias.checked()
    G, M, K = x3.shape
    Kp = B.shape[-1]
    a_scale, a_zp = _f32c(a_scale, "a_scale").reshape(-1), _f32c(a_zp, "a_zp").reshape(-1)
    assert a_scale.numel() == a_zp.numel() and a_scale.numel() in (1, gmod)
    assert B.dtype == torch.int8 and B.is_contiguous() and B.shape[-2] == N
    sBg = 0 if B.shape[1] == 1 and G > 1 else B.stride(1)
    out = torch.empty((G, M, N), dtype=torch.float32, device=x3.device)
    if addend is not None:                               # the residual stream, added in the epilogue
        addend = _f32c(addend, "addend")
        assert addend.numel() == G * M * N
    rc = lib.adalog_gemm_out_gen_ex(x3.data_ptr(), x3.stride(1), x3.stride(0), K, a_scale.data_ptr(), a_zp.data_ptr(),
                                    0 if a_scale.numel() == 1 else 1, int(n_bits), B.data_ptr(), sBg, M, N, Kp, G, gmod,
                                    sa.t.data_ptr(), sa.g, float(sa_mul), sb.t.data_ptr(), sb.g, sb.n,
                                    None if bias is None else bias.t.data_ptr(), 0 if bias is None else bias.g,
                                    0 if bias is None else bias.n, _ptr(addend), out.data_ptr(), N, M * N, _stream())
    _lib.check(rc, "adalog_gemm_out_gen_ex")
    return out


# Capability flag read by the layers (getattr(backend.get(), "QF_EXTRAS", False)): this backend has the quant_forward extras of round 6
# -- epilogue addend / heads-last store (gemm_out, gemm_out_gen), the GELU prologue of the AdaLog packer and of the training-form
# quantiser, softmax_adalog_pack, attn_split_pack.  The CPU specification backend of the tests does not: the layers then compose.
QF_EXTRAS = True


def softmax_adalog_pack(x3, mul: float, scale, qv, n_bits: int, mant37):
    """(x3 * mul).softmax(-1) through the post-softmax AdaLog quantiser, as the packed bf16 operand [1, G, R, Kp] of softmax . v
    (adalog_softmax_adalog_pack_bf16: one pass, the probabilities are never stored).  x3: fp32 [G, R, S] contiguous, S <= 256."""
    x3 = _f32c(x3, "scores")
    G, R, S = x3.shape
    Kp = pad_k(S, BF16)
    out = torch.empty((1, G, R, Kp), dtype=torch.bfloat16, device=x3.device)
    rc = _lib.load().adalog_softmax_adalog_pack_bf16(x3.data_ptr(), G * R, S, float(mul), _ptr(_f32c(scale, "scale")),
                                                    _ptr(_f32c(qv, "qv")), int(n_bits), _ptr(_f32c(mant37, "mant37")),
                                                    out.data_ptr(), Kp, _stream())
    _lib.check(rc, "adalog_softmax_adalog_pack_bf16")
    out.k_valid = S
    return out


def softmax_adalog_pack_ok(S: int) -> bool:
    return S <= 256 and pad_k(S, BF16) <= 256


def attn_split_pack(qkv, H: int, q_par, k_par, v_par, per_head: bool):
    """qkv fp32 [B, N, 3*H*64] (the qkv projection's output) -> (qp, kp, vp): the packed operands of q . k^T (int8 [1, B*H, N, 128])
    and of softmax . v's second operand (bf16 [1, B*H, 64, Np], v transposed) through the three uniform input quantisers
    ((scale, zero_point, n_bits) each; per head or per tensor) in one launch (adalog_attn_split_pack)."""
    qkv = _f32c(qkv, "qkv")
    B, N, C3 = qkv.shape
    assert C3 == 3 * H * 64
    Np = ((N + 63) // 64) * 64
    qp = torch.empty((1, B * H, N, 128), dtype=torch.int8, device=qkv.device)
    kp = torch.empty((1, B * H, N, 128), dtype=torch.int8, device=qkv.device)
    vp = torch.empty((1, B * H, 64, Np), dtype=torch.bfloat16, device=qkv.device)
    par = []
    for s_, z_, b_ in (q_par, k_par, v_par):
        s_, z_ = _f32c(s_, "scale").reshape(-1), _f32c(z_, "zero_point").reshape(-1)
        assert s_.numel() == z_.numel() == (H if per_head else 1)
        par += [s_, z_, int(b_)]
    rc = _lib.load().adalog_attn_split_pack(qkv.data_ptr(), B, N, H, par[0].data_ptr(), par[1].data_ptr(), par[2], par[3].data_ptr(),
                                           par[4].data_ptr(), par[5], par[6].data_ptr(), par[7].data_ptr(), par[8],
                                           1 if per_head else 0, qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), Np, _stream())
    _lib.check(rc, "adalog_attn_split_pack")
    qp.k_valid = kp.k_valid = 64
    vp.k_valid = N
    return qp, kp, vp


def log2_shift(x, shift: float):
    """log2(x + shift), correctly rounded, -inf where x + shift <= 0 (input of score_act_fused; once per layer)."""
    x = _f32c(x, "x")
    if _torch_ops.available():
        return _top("log2_shift", x, float(shift))
    out = torch.empty_like(x)
    rc = _lib.load().adalog_log2_shift(x.data_ptr(), out.data_ptr(), x.numel(), float(shift), _stream())
    _lib.check(rc, "adalog_log2_shift")
    return out


def score_act_fused_ok(M: int, T: int, K: int, Kp: int, P: int, n_bits: int) -> bool:
    return bool(_lib.load().adalog_score_act_fused_ok(int(M), int(T), int(K), int(Kp), int(P), int(n_bits)))


def score_act_fused(wp, x2, lx2, ref2, row_scale, row_bias, scale, qv, n_bits: int, mant37, shift: float, clamp_u: bool,
                    sa_mul: float, norm: float):
    """Post-GELU activation-candidate scores [P, 1] with the AdaLog quantisation fused into the GEMM's loader
    (gemm_fused.hip): wp = bf16 weight image [1, 1, M, Kp], x2 / lx2 = activation and log2_shift(activation) [T, K],
    ref2 = raw_out [T, M]; (scale, qv) = the P = 128 candidates."""
    lib = _lib.load()
    M, Kp = wp.shape[-2], wp.shape[-1]
    T, K = x2.shape
    P = scale.numel()
    assert wp.dtype == torch.bfloat16 and wp.is_contiguous() and ref2.shape == (T, M)
    x2, lx2, ref2 = _f32c(x2, "x"), _f32c(lx2, "log2 x"), _f32c(ref2, "ref")
    if _torch_ops.available():
        return _timed((BF16, M, T, K, P, 1), lambda: _top(
            "score_act_fused", wp, x2, lx2, ref2, _f32c(row_scale, "row_scale"),
            None if row_bias is None else _f32c(row_bias, "row_bias"), _f32c(scale, "scale"), _f32c(qv, "qv"),
            int(n_bits), _f32c(mant37, "mant37"), float(shift), bool(clamp_u), float(sa_mul), float(norm)))
    ws_bytes = lib.adalog_score_act_fused_workspace_bytes(T, Kp)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=x2.device)
    scores = torch.empty((P, 1), dtype=torch.float32, device=x2.device)
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_score_act_fused(wp.data_ptr(), M, Kp, x2.data_ptr(), lx2.data_ptr(), T, K, ref2.data_ptr(),
                                    _f32c(row_scale, "row_scale").data_ptr(),
                                    _ptr(None if row_bias is None else _f32c(row_bias, "row_bias")),
                                    _f32c(scale, "scale").data_ptr(), _f32c(qv, "qv").data_ptr(), P, int(n_bits),
                                    _f32c(mant37, "mant37").data_ptr(), float(shift), int(bool(clamp_u)), float(sa_mul),
                                    float(norm), ws.data_ptr(), ws_bytes, scores.data_ptr(), _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((BF16, M, T, K, P, 1, ev0, ev1, lib.adalog_last_kernel().decode()))
    _lib.check(rc, "adalog_score_act_fused")
    return scores


def score_act_gen_ok(dtype: int, M: int, T: int, K: int, Kp: int, P: int) -> bool:
    return bool(_lib.load().adalog_score_act_gen_ok(int(dtype), int(M), int(T), int(K), int(Kp), int(P)))


def score_act_gen(dtype: int, wp, x2, scale, zp, n_bits: int, ref2, row_scale, row_bias, norm: float, defer: bool = False):
    """Activation-candidate scores [P, 1] of a uniformly quantised Linear layer with the candidate operand generated inside
    the slab kernel (gemm_k_slab.inc, GEN form): wp = packed weight image [1, 1, M, Kp] (int8 / fp8), x2 = activation [T, K],
    ref2 = raw_out [T, M], (scale, zp) = the P per-tensor candidates; row_scale = weight scales [M], row_bias = bias [M]."""
    lib = _lib.load()
    M, Kp = wp.shape[-2], wp.shape[-1]
    x2, ref2 = _f32c(x2, "x"), _f32c(ref2, "ref")
    T, K = x2.shape
    scale, zp = _f32c(scale, "scale").reshape(-1), _f32c(zp, "zp").reshape(-1)
    P = scale.numel()
    assert wp.dtype == _TORCH_DT[dtype] and wp.is_contiguous() and ref2.shape == (T, M)
    row_scale = _f32c(row_scale, "row_scale")
    row_bias = None if row_bias is None else _f32c(row_bias, "row_bias")
    wgs = lib.adalog_score_act_gen_wgs(dtype, M, T, K, Kp, P)
    if wgs < 0:
        raise _lib.AdalogHipError("score_act_gen: shape not supported (score_act_gen_ok)")
    if _torch_ops.available():
        meta = (dtype, M, T, K, P, 1)
        if not defer:
            return _timed(meta, lambda: _top("score_act_gen", int(dtype), wp, x2, scale, zp, int(n_bits), ref2, row_scale, row_bias,
                                             float(norm)))
        ws = _timed(meta, lambda: _top("score_act_gen_partial", int(dtype), wp, x2, scale, zp, int(n_bits), ref2, row_scale, row_bias))
        return PendingScores(ws, wgs, 256, 256, P, 1, 1, False, False, 2, norm, T)
    wsb = lib.adalog_score_act_gen_workspace_bytes(dtype, M, T, K, Kp, P)
    ws = torch.empty((wsb + 15) // 16 * 2, dtype=torch.float64, device=x2.device)
    scores = None if defer else torch.empty((P, 1), dtype=torch.float32, device=x2.device)
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_score_act_gen(dtype, wp.data_ptr(), M, Kp, x2.data_ptr(), T, K, K, scale.data_ptr(), zp.data_ptr(), P,
                                  int(n_bits), ref2.data_ptr(), row_scale.data_ptr(), _ptr(row_bias), float(norm),
                                  ws.data_ptr(), ws.numel() * 8, _ptr(scores), _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((dtype, M, T, K, P, 1, ev0, ev1, lib.adalog_last_kernel().decode()))
    _lib.check(rc, "adalog_score_act_gen")
    # (the accumulators [wgs][1][256] fp64 sit at the start of the workspace: the per-workgroup layout of gemm_score)
    return PendingScores(ws, wgs, 256, 256, P, 1, 1, False, False, 2, norm, T) if defer else scores


_WGEN_LAYOUT = {}


def score_w_gen_ok(dtype: int, T: int, O: int, K: int, Kp: int, P: int) -> bool:
    return bool(_lib.load().adalog_score_w_gen_ok(int(dtype), int(T), int(O), int(K), int(Kp), int(P)))


def score_w_gen(dtype: int, xp, w2, scale, zp, n_bits: int, ref_t, sa, bias, norm: float, defer: bool = False):
    """Weight-candidate scores [P, O] of a uniformly quantised Linear layer (reference linear.py:355-392) with the candidate
    operand generated inside the slab kernel: xp = packed activation image [1, 1, T, Kp] (int8 / fp8), w2 = weight [O, K] fp32,
    (scale, zp) = the P candidates of every output row [P, O], ref_t = raw_out transposed [O, T], sa = activation scale [1]."""
    lib = _lib.load()
    T, Kp = xp.shape[-2], xp.shape[-1]
    w2, ref_t = _f32c(w2, "weight"), _f32c(ref_t, "ref")
    O, K = w2.shape
    scale, zp = _f32c(scale, "scale"), _f32c(zp, "zp")
    P = scale.shape[0]
    assert xp.dtype == _TORCH_DT[dtype] and xp.is_contiguous() and tuple(ref_t.shape[-2:]) == (O, T)
    assert scale.numel() == P * O and zp.numel() == P * O
    sa = _f32c(sa, "sa").reshape(-1)
    bias = None if bias is None else _f32c(bias, "bias")
    key = (T, O, P, dtype, Kp, K)
    if key not in _WGEN_LAYOUT:
        _WGEN_LAYOUT[key] = _layout(T, O * P, 1, 1, 1, P, 0, dtype, Kp, K, True)
    n_part, MT, Npad, mode = _WGEN_LAYOUT[key]
    partial = torch.empty((n_part + 1) // 2, dtype=torch.float64, device=w2.device).view(torch.float32)
    if GEMM_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.adalog_score_w_gen(dtype, xp.data_ptr(), T, Kp, w2.data_ptr(), O, K, K, scale.data_ptr(), zp.data_ptr(), P, int(n_bits),
                                ref_t.data_ptr(), sa.data_ptr(), _ptr(bias), partial.data_ptr(), n_part, _stream())
    if GEMM_EVENTS is not None:
        ev1.record()
        GEMM_EVENTS.append((dtype, T, O, K, P, 1, ev0, ev1, lib.adalog_last_kernel().decode()))
    _lib.check(rc, "adalog_score_w_gen")
    pend = PendingScores(partial, MT, O, Npad, P, 1, 1, False, True, mode, norm, O)
    return pend if defer else pend.finish()


# ------------------------------------------------------------------------------------------------ Gram-form weight search
def gram_ok(T: int, O: int, K: int, a_bits: int, w_bits: int, P: int) -> bool:
    """True when the Gram form scores this output-MSE weight search (csrc/gram.hip): the shape is supported and it pays
    (adalog_gram_ok).  ADALOG_GRAM_W=0: never; ADALOG_GRAM_W=2: wherever the shape is supported (tests drive the kernels at the
    golden traces' toy shapes with it)."""
    mode = os.environ.get("ADALOG_GRAM_W", "1")
    if mode == "0":
        return False
    fn = _lib.load().adalog_gram_supported if mode == "2" else _lib.load().adalog_gram_ok
    return bool(fn(int(T), int(O), int(K), int(a_bits), int(w_bits), int(P)))


class GramState:
    """G = X_int^T X_int, c = X_int^T (raw_out - bias), S0 = sum (raw_out - bias)^2 of one weight_fpcs call (csrc/gram.hip), built
    once from the captured activation, the activation quantiser and raw_out; ``score_w`` then scores an FPCS step from it.

    Image-sharded ranks (adalog_amd.parallel): x2 / ref_t hold this rank's tokens.  G, c and S0 are sums over tokens, so the build
    all-reduces them ONCE (adalog_gram_amax -> MAX, adalog_gram_build_sums -> SUM of int64 / fp64, adalog_gram_build_from_sums) and
    every rank then scores the same FINAL scores from the same state: ``global_scores`` -- the six steps of the search need no
    collective (the token-form kernels and the rank-local Gram state all-reduce [P, O] scores at every step)."""
    __slots__ = ("ws", "T", "O", "K", "a_bits", "sa", "build_ms", "global_scores")

    def __init__(self, x2, sa, za, a_bits: int, ref_t, bias):
        from . import parallel
        lib = _lib.load()
        x2, ref_t = _f32c(x2, "x"), _f32c(ref_t, "ref")
        t_local, self.K = x2.shape
        self.O = ref_t.shape[-2]
        assert ref_t.shape[-1] == t_local
        self.a_bits = int(a_bits)
        self.sa = _f32c(sa, "sa").reshape(-1)
        za = _f32c(za, "za").reshape(-1)
        assert self.sa.numel() == 1 and za.numel() == 1, "Gram form: per-tensor activation quantiser"
        bias = None if bias is None else _f32c(bias, "bias")
        self.global_scores = True
        dev = x2.device
        if parallel.is_dist():
            # (equal shards: adalog_amd.parallel.shard_slice deals the images evenly, so the global token count is ws x local)
            self.T = t_local * parallel.world_size()
            amax = torch.empty(self.O, dtype=torch.int32, device=dev)
            _lib.check(lib.adalog_gram_amax(ref_t.data_ptr(), t_local, self.O, _ptr(bias), amax.data_ptr(), _stream()), "adalog_gram_amax")
            parallel.all_reduce_max(amax)                     # bits of non-negative floats: integer order = float order
            nb_l = lib.adalog_gram_workspace_bytes(t_local, self.O, self.K, self.a_bits)
            nb = lib.adalog_gram_workspace_bytes(self.T, self.O, self.K, self.a_bits)
            if nb_l < 0 or nb < 0:
                raise _lib.AdalogHipError("gram_build: shape not supported (gram_ok)")
            tmp = _aligned_bytes(nb_l, dev)
            gsum = torch.empty((self.K, self.K), dtype=torch.int64, device=dev)
            csum = torch.empty((self.O, self.K), dtype=torch.int64, device=dev)
            s0 = torch.empty(self.O, dtype=torch.float64, device=dev)
            rc = lib.adalog_gram_build_sums(x2.data_ptr(), t_local, self.K, x2.stride(0), self.sa.data_ptr(), za.data_ptr(), self.a_bits,
                                            ref_t.data_ptr(), self.O, _ptr(bias), amax.data_ptr(), gsum.data_ptr(), csum.data_ptr(),
                                            s0.data_ptr(), tmp.data_ptr(), nb_l, _stream())
            _lib.check(rc, "adalog_gram_build_sums")
            parallel.all_reduce_sum(gsum)
            parallel.all_reduce_sum(csum)
            parallel.all_reduce_sum(s0)
            self.ws = _aligned_bytes(nb, dev)
            rc = lib.adalog_gram_build_from_sums(gsum.data_ptr(), csum.data_ptr(), s0.data_ptr(), amax.data_ptr(), self.T, self.O, self.K,
                                                 self.a_bits, self.ws.data_ptr(), nb, _stream())
            _lib.check(rc, "adalog_gram_build_from_sums")
            return
        # one rank: the sites an N-rank build would all-reduce (amax, G, c, S0)
        parallel.note_planned(4 * self.O + 8 * self.K * self.K + 8 * self.O * self.K + 8 * self.O, count=4)
        self.T = t_local
        nb = lib.adalog_gram_workspace_bytes(self.T, self.O, self.K, self.a_bits)
        if nb < 0:
            raise _lib.AdalogHipError("gram_build: shape not supported (gram_ok)")
        self.ws = torch.empty((nb + 255) // 8 + 32, dtype=torch.float64, device=x2.device)
        off = (-self.ws.data_ptr()) % 256                     # the kernels want a 256-byte aligned base
        self.ws = self.ws.view(torch.uint8)[off:off + nb]
        if GEMM_EVENTS is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.adalog_gram_build(x2.data_ptr(), self.T, self.K, x2.stride(0), self.sa.data_ptr(), za.data_ptr(), self.a_bits,
                                   ref_t.data_ptr(), self.O, _ptr(bias), self.ws.data_ptr(), nb, _stream())
        if GEMM_EVENTS is not None:
            ev1.record()
            # issued work: G (K x K, symmetry not used) + four reference limbs x (O x K), each over the T tokens
            GEMM_EVENTS.append((I8, self.T, self.K + 4 * self.O, self.K, 1, 1, ev0, ev1, "k_gram_build"))
        _lib.check(rc, "adalog_gram_build")

    def score_w(self, w2, scale, zp, w_bits: int, norm: float, tail=None):
        """scores [P, O] (final: no partial sums) for the candidates (scale, zp) [P, O] of every output row of w2 [O, K].
        ``tail`` (FpcsTail over this very grid): the kernel also ranks every row and writes its next grid / commits its winner."""
        lib = _lib.load()
        w2 = _f32c(w2, "weight")
        scale, zp = _f32c(scale, "scale"), _f32c(zp, "zp")
        P = scale.shape[0]
        assert tuple(w2.shape) == (self.O, self.K) and scale.numel() == P * self.O and zp.numel() == P * self.O
        scores = torch.empty((P, self.O), dtype=torch.float32, device=w2.device)
        if GEMM_EVENTS is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        assert tail is None or (tail.scale.data_ptr() == scale.data_ptr() and tail.zp is not None and tail.zp.data_ptr() == zp.data_ptr())
        rc = lib.adalog_gram_score_w_tail(w2.data_ptr(), self.O, self.K, w2.stride(0), scale.data_ptr(), zp.data_ptr(), P, int(w_bits),
                                          self.ws.data_ptr(), self.T, self.a_bits, self.sa.data_ptr(), float(norm), scores.data_ptr(),
                                          None if tail is None else tail.ref(), _stream())
        if GEMM_EVENTS is not None:
            ev1.record()
            # flops as ISSUED (limbs * K^2 + the 32-row w.c panel per candidate row), not the token form's 2 T K O P: a roofline
            # fraction of this kernel is a fraction of the work it does
            rows = lib.adalog_gram_limbs(self.T, self.a_bits) * self.K + 32
            GEMM_EVENTS.append((I8, rows, self.O, self.K, P, 1, ev0, ev1, lib.adalog_last_kernel().decode()))
        _lib.check(rc, "adalog_gram_score_w")
        return scores


# ------------------------------------------------------------------------------------------------ Gram-form activation search
def gram_act_ok(T: int, O: int, K: int, a_bits: int, w_bits: int, P: int) -> bool:
    """True when the Gram form scores this per-tensor output-MSE activation search (csrc/gram_act.hip).  ADALOG_GRAM_A=0: never;
    ADALOG_GRAM_A=2: wherever the shape is supported (tests drive the kernels at the golden traces' toy shapes with it)."""
    mode = os.environ.get("ADALOG_GRAM_A", "1")
    if mode == "0":
        return False
    fn = _lib.load().adalog_gram_act_supported if mode == "2" else _lib.load().adalog_gram_act_ok
    return bool(fn(int(T), int(O), int(K), int(a_bits), int(w_bits), int(P)))


def _aligned_bytes(nbytes: int, device):
    buf = torch.empty((nbytes + 255) // 8 + 32, dtype=torch.float64, device=device)
    off = (-buf.data_ptr()) % 256
    return buf.view(torch.uint8)[off:off + nbytes]


class GramActPrepared:
    """Per captured activation x [T, K]: its transposed fp32 image, the sorted values and the sorting permutation (memoised by the
    caller across the rounds of a module's search: the captured tensor does not change)."""
    __slots__ = ("xt", "sorted", "perm", "T", "K")

    def __init__(self, x2):
        lib = _lib.load()
        x2 = _f32c(x2, "x")
        self.T, self.K = x2.shape
        n = self.T * self.K
        Tp = (self.T + 127) // 128 * 128
        self.xt = torch.empty((self.K, Tp), dtype=torch.float32, device=x2.device)
        self.sorted = torch.empty(n, dtype=torch.float32, device=x2.device)
        self.perm = torch.empty(n, dtype=torch.int32, device=x2.device)
        nb = lib.adalog_gram_act_sort_bytes(n)
        if nb < 0:
            raise _lib.AdalogHipError("gram_act_prepare: tensor too large")
        ws = _aligned_bytes(nb, x2.device)
        rc = lib.adalog_gram_act_prepare(x2.data_ptr(), self.T, self.K, x2.stride(0), self.xt.data_ptr(), self.sorted.data_ptr(),
                                         self.perm.data_ptr(), ws.data_ptr(), nb, _stream())
        _lib.check(rc, "adalog_gram_act_prepare")


class GramActState:
    """One activation_fpcs call (reference linear.py:505-523) in the Gram form: H = Wq^T Wq, the prefix sums of C = r . Wq along the
    sorted activation and S0, built once from raw_out and the fixed weight quantiser; ``score`` then scores an FPCS step."""
    __slots__ = ("prep", "ws", "qpart", "T", "O", "K", "P", "a_bits")

    def __init__(self, prep: GramActPrepared, raw_out2, bias, w2, sw, zw, w_bits: int, a_bits: int, P: int):
        lib = _lib.load()
        raw_out2, w2 = _f32c(raw_out2, "raw_out"), _f32c(w2, "weight")
        self.prep, self.T, self.K, self.P, self.a_bits = prep, prep.T, prep.K, int(P), int(a_bits)
        self.O = w2.shape[0]
        assert tuple(raw_out2.shape) == (self.T, self.O) and w2.shape[1] == self.K
        sw, zw = _f32c(sw, "s_w").reshape(-1), _f32c(zw, "z_w").reshape(-1)
        assert sw.numel() == self.O and zw.numel() == self.O
        bias = None if bias is None else _f32c(bias, "bias")
        nb = lib.adalog_gram_act_workspace_bytes(self.T, self.O, self.K, self.P)
        if nb < 0:
            raise _lib.AdalogHipError("gram_act_build: shape not supported (gram_act_ok)")
        self.ws = _aligned_bytes(nb, w2.device)
        self.qpart = torch.empty(self.P * lib.adalog_gram_act_splits(self.T, self.O, self.K, self.P), dtype=torch.float64, device=w2.device)
        if GEMM_EVENTS is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.adalog_gram_act_build(raw_out2.data_ptr(), self.T, self.O, _ptr(bias), w2.data_ptr(), self.K, w2.stride(0), sw.data_ptr(),
                                       zw.data_ptr(), int(w_bits), prep.perm.data_ptr(), self.P, self.ws.data_ptr(), nb, _stream())
        if GEMM_EVENTS is not None:
            ev1.record()
            # issued matrix work: four reference limbs x (T x O x K)
            GEMM_EVENTS.append((I8, 4 * self.T, self.K, self.O, 1, 1, ev0, ev1, "k_gram_act_build"))
        _lib.check(rc, "adalog_gram_act_build")

    def score(self, scale, zp, norm: float, tail=None):
        """scores [P, 1] (final) for the per-tensor candidates (scale, zp) [P, 1].  ``tail`` (FpcsTail over this very grid): the
        finish kernel's last block also ranks the candidates and writes the next grid / commits the winner."""
        lib = _lib.load()
        scale, zp = _f32c(scale, "scale").reshape(-1), _f32c(zp, "zp").reshape(-1)
        assert tail is None or (tail.scale.data_ptr() == scale.data_ptr() and tail.zp is not None and tail.zp.data_ptr() == zp.data_ptr())
        P = scale.numel()
        assert P == self.P and zp.numel() == P
        scores = torch.empty((P, 1), dtype=torch.float32, device=scale.device)
        if GEMM_EVENTS is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        rc = lib.adalog_gram_act_score_tail(self.prep.xt.data_ptr(), self.prep.sorted.data_ptr(), self.T, self.O, self.K, scale.data_ptr(),
                                            zp.data_ptr(), P, self.a_bits, self.ws.data_ptr(), float(norm), self.qpart.data_ptr(),
                                            scores.data_ptr(), None if tail is None else tail.ref(), _stream())
        if GEMM_EVENTS is not None:
            ev1.record()
            # flops as ISSUED: the upper triangle of X_p^T X_p in 32 x 32 blocks, per candidate
            nj = self.K // 32
            GEMM_EVENTS.append((I8, self.T, 32 * nj * (nj + 1) // 2, 32, P, 1, ev0, ev1, lib.adalog_last_kernel().decode()))
        _lib.check(rc, "adalog_gram_act_score")
        return scores


# ------------------------------------------------------------------------------------------------ FPCS pieces
def topk(scores, k: int):
    scores = _f32c(scores, "scores")
    P, cols = scores.shape
    if _torch_ops.available():
        return _top("topk", scores, int(k))
    idx = torch.empty((k, cols), dtype=torch.int32, device=scores.device)
    rc = _lib.load().adalog_topk(scores.data_ptr(), P, cols, int(k), idx.data_ptr(), _stream())
    _lib.check(rc, "adalog_topk")
    return idx


def fpcs_next(scale, zp, third, idx, k: int, new_cnt: int, lin, delta, clamp_min: Optional[float]):
    """new_cnt > 0: returns the next grid (scale, zp, third) of k*new_cnt candidates and updates ``delta`` in place;
    new_cnt == 0: returns the committed winner, each of shape [cols]."""
    scale = _f32c(scale, "scale")
    cols = scale.shape[1]
    rows = k * new_cnt if new_cnt > 0 else 1
    mk = lambda src: None if src is None else torch.empty((rows, cols), dtype=torch.float32, device=scale.device)
    o_s, o_z, o_t = mk(scale), mk(zp), mk(third)
    rc = _lib.load().adalog_fpcs_next(scale.data_ptr(), _ptr(zp), _ptr(third), cols, idx.data_ptr(), int(k), int(new_cnt),
                                     _ptr(lin), _ptr(delta), int(clamp_min is not None),
                                     float(clamp_min if clamp_min is not None else 0.0), o_s.data_ptr(), _ptr(o_z),
                                     _ptr(o_t), _stream())
    _lib.check(rc, "adalog_fpcs_next")
    if new_cnt == 0:
        return o_s[0], (None if o_z is None else o_z[0]), (None if o_t is None else o_t[0])
    return o_s, o_z, o_t


def topk_next(scores, scale, zp, third, k: int, new_cnt: int, lin, delta, clamp_min: Optional[float], tail=None):
    """topk(scores, k) followed by fpcs_next(...) in one launch; same return value as fpcs_next.  ``tail``: a prepared FpcsTail."""
    scores = _f32c(scores, "scores")
    if tail is None:
        tail = FpcsTail(scale, zp, third, k, new_cnt, lin, delta, delta, clamp_min)
    P, cols = scores.shape
    assert tail.scale.shape == (P, cols)
    rc = _lib.load().adalog_topk_next_tail(scores.data_ptr(), P, cols, tail.ref(), None, _stream())
    _lib.check(rc, "adalog_topk_next")
    return tail.result()


def candidate_grid(quant4, num_scale: int, num_zp: int, zp_min: int, n_bits: int, lin, clamp_min: Optional[float]):
    quant4 = _f32c(quant4, "quant4")
    cols = quant4.shape[1]
    P = num_scale * num_zp
    scale = torch.empty((P, cols), dtype=torch.float32, device=quant4.device)
    zp = torch.empty_like(scale)
    delta = torch.empty(cols, dtype=torch.float32, device=quant4.device)
    rc = _lib.load().adalog_candidate_grid(quant4.data_ptr(), cols, num_scale, num_zp, zp_min, int(n_bits), lin.data_ptr(),
                                          int(clamp_min is not None), float(clamp_min or 0.0), scale.data_ptr(),
                                          zp.data_ptr(), delta.data_ptr(), _stream())
    _lib.check(rc, "adalog_candidate_grid")
    return scale, zp, delta


def score_w_self(w2, scale, zp, n_bits: int):
    w2 = _f32c(w2, "weight")
    rows, I = w2.shape
    P = scale.shape[0]
    if _torch_ops.available():
        return _top("score_w_self", w2, _f32c(scale, "scale"), _f32c(zp, "zp"), int(n_bits))
    scores = torch.empty((P, rows), dtype=torch.float32, device=w2.device)
    rc = _lib.load().adalog_score_w_self(w2.data_ptr(), rows, I, _f32c(scale, "scale").data_ptr(),
                                        _f32c(zp, "zp").data_ptr(), P, int(n_bits), scores.data_ptr(), _stream())
    _lib.check(rc, "adalog_score_w_self")
    return scores


def score_a_self(x2, scale, zp, channel_wise: bool, n_bits: int, norm: float):
    x2 = _f32c(x2, "x")
    rows, I = x2.shape
    P = scale.shape[0]
    if _torch_ops.available():
        return _top("score_a_self", x2, _f32c(scale, "scale"), _f32c(zp, "zp"), bool(channel_wise), int(n_bits), float(norm))
    lib = _lib.load()
    n_part = lib.adalog_score_a_self_partial_elems(rows, I, P)
    partial = torch.empty(n_part, dtype=torch.float32, device=x2.device)
    scores = torch.empty((P, I if channel_wise else 1), dtype=torch.float32, device=x2.device)
    rc = lib.adalog_score_a_self(x2.data_ptr(), rows, I, _f32c(scale, "scale").data_ptr(), _f32c(zp, "zp").data_ptr(), P,
                                 int(bool(channel_wise)), int(n_bits), float(norm), partial.data_ptr(), n_part,
                                 scores.data_ptr(), _stream())
    _lib.check(rc, "adalog_score_a_self")
    return scores


def sort_f32(x2, want_perm: bool = False):
    """x2 [S, n] fp32 -> (sorted [S, n] ascending per segment, perm [S, n] int32 | None): the hand-written stable LSD radix sort of
    csrc/radix_sort.hip (what SortedPrefix and GramActPrepared sort with).  perm[s, i] = index within segment s of its i-th smallest."""
    x2 = _f32c(x2, "x")
    if x2.dim() != 2:
        raise ValueError("sort_f32: expected [S, n]")
    S, n = x2.shape
    lib = _lib.load()
    nb = lib.adalog_sort_workspace_bytes(S, n, int(want_perm))
    if nb < 0:
        raise _lib.AdalogHipError("sort_f32: unsupported size")
    ws = _aligned_bytes(nb, x2.device)
    out = torch.empty_like(x2)
    perm = torch.empty((S, n), dtype=torch.int32, device=x2.device) if want_perm else None
    rc = lib.adalog_sort_f32(x2.data_ptr(), S, n, out.data_ptr(), _ptr(perm), ws.data_ptr(), nb, _stream())
    _lib.check(rc, "adalog_sort_f32")
    return out, perm


class SortedPrefix:
    """A [S, n] tensor sorted per segment with fp64 prefix sums of x and x^2 along the sorted order (csrc/sorted_score.hip):
    what the self-MSE searches score their candidates from.  Built once per captured tensor, shared by the FPCS steps."""
    __slots__ = ("sorted", "prefix", "S", "n")

    def __init__(self, x2):
        x2 = _f32c(x2, "x")
        if x2.dim() != 2:
            raise ValueError("SortedPrefix: expected [S, n]")
        self.S, self.n = x2.shape
        if _torch_ops.available():
            self.sorted, self.prefix = _top("sorted_prefix", x2)
            return
        lib = _lib.load()
        wsb = lib.adalog_sorted_prefix_workspace_bytes(self.S, self.n)
        if wsb < 0:
            raise _lib.AdalogHipError("sorted_prefix: unsupported size")
        ws = torch.empty((wsb + 15) // 16 * 2, dtype=torch.float64, device=x2.device)
        self.sorted = torch.empty_like(x2)
        self.prefix = torch.empty((self.S, self.n + 1, 2), dtype=torch.float64, device=x2.device)
        rc = lib.adalog_sorted_prefix_build(x2.data_ptr(), self.S, self.n, self.sorted.data_ptr(), self.prefix.data_ptr(),
                                            ws.data_ptr(), ws.numel() * 8, _stream())
        _lib.check(rc, "adalog_sorted_prefix_build")


def sorted_prefix(x2):
    return SortedPrefix(x2)


# the sorted copy + fp64 prefix sums + sort temporaries of a captured tensor cost ~6x its size for the length of a search: above
# this many bytes (ADALOG_SORTED_MAX_GIB, default 16 of the 288 GB) the self-MSE search falls back to the one-pass kernel
_SORTED_MAX_BYTES = int(float(__import__("os").environ.get("ADALOG_SORTED_MAX_GIB", "16")) * (1 << 30))


def sorted_prefix_ok(S: int, n: int, n_bits: int) -> bool:
    if not (1 <= n_bits <= 8 and S <= 65535):
        return False
    ws = _lib.load().adalog_sorted_prefix_workspace_bytes(int(S), int(n))
    return ws >= 0 and ws + 20 * int(S) * int(n) <= _SORTED_MAX_BYTES


def score_self_sorted(sp: SortedPrefix, scale, zp, n_bits: int, norm: float, tail=None):
    """scores [P, S] = -norm * sum over each segment of (x - fq_p(x))^2 for the P candidates scale / zp [P, S].  ``tail`` (FpcsTail
    over this very grid): the launch also ranks every segment's candidates and writes its next grid / commits its winner."""
    scale, zp = _f32c(scale, "scale"), _f32c(zp, "zp")
    P = scale.shape[0]
    if scale.numel() != P * sp.S or zp.numel() != P * sp.S:
        raise ValueError("score_self_sorted: scale / zp must be [P, S]")
    if tail is not None:
        assert tail.scale.data_ptr() == scale.data_ptr() and tail.zp is not None and tail.zp.data_ptr() == zp.data_ptr()
        scores = torch.empty((P, sp.S), dtype=torch.float32, device=scale.device)
        rc = _lib.load().adalog_score_self_sorted_tail(sp.sorted.data_ptr(), sp.prefix.data_ptr(), sp.S, sp.n, scale.data_ptr(),
                                                      zp.data_ptr(), P, int(n_bits), float(norm), scores.data_ptr(), tail.ref(), _stream())
        _lib.check(rc, "adalog_score_self_sorted")
        return scores
    if _torch_ops.available():
        return _top("score_self_sorted", sp.sorted, sp.prefix, scale, zp, int(n_bits), float(norm))
    scores = torch.empty((P, sp.S), dtype=torch.float32, device=scale.device)
    rc = _lib.load().adalog_score_self_sorted(sp.sorted.data_ptr(), sp.prefix.data_ptr(), sp.S, sp.n, scale.data_ptr(),
                                             zp.data_ptr(), P, int(n_bits), float(norm), scores.data_ptr(), _stream())
    _lib.check(rc, "adalog_score_self_sorted")
    return scores


# ------------------------------------------------------------------------------------------------ order statistics
def quantile_ranks(qs, n: int):
    """Host-side rank arithmetic exactly as ATen's quantile does it: pos = q * (n - 1) in fp32."""
    pos = torch.tensor(qs, dtype=torch.float32) * (n - 1)
    lo = pos.floor()
    hi = pos.ceil()
    lohi = torch.stack([lo, hi], dim=1).reshape(-1).to(torch.int64)
    return lohi, (pos - lo)


_RANKS_DEV = {}


def quantile_ranks_dev(qs, n: int, device):
    """quantile_ranks on the device, kept per (qs, n, device): the percentile grids ask for the same four ranks of the same
    row length every round of every module (2 host-to-device copies per call before: ~450 per deit_small calibration)"""
    key = (tuple(qs), int(n), str(device))
    hit = _RANKS_DEV.get(key)
    if hit is None:
        if len(_RANKS_DEV) > 512:
            _RANKS_DEV.clear()
        lohi, w = quantile_ranks(qs, n)
        hit = _RANKS_DEV[key] = (lohi.to(device), w.to(device))
    return hit


def quantile_rows(x2, qs, mbs: int = 1):
    """torch.quantile(x2, qs, dim=-1) followed by the mean over groups of ``mbs`` rows -> [len(qs), S/mbs]."""
    x2 = _f32c(x2, "x")
    S, n = x2.shape
    nq = len(qs)
    lohi, w = quantile_ranks_dev(qs, n, x2.device)
    lib = _lib.load()
    ws_bytes = lib.adalog_select_workspace_bytes(S, 2 * nq)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x2.device)
    out = torch.empty((nq, S // mbs), dtype=torch.float32, device=x2.device)
    rc = lib.adalog_quantile_rows(x2.data_ptr(), S, n, nq, lohi.data_ptr(), w.data_ptr(), int(mbs), out.data_ptr(),
                                  ws.data_ptr(), ws_bytes, _stream())
    _lib.check(rc, "adalog_quantile_rows")
    return out


def positive_percentile_rows(x2, qs):
    x2 = _f32c(x2, "x")
    S, n = x2.shape
    nq = len(qs)
    key = ("qf", tuple(qs), str(x2.device))
    qf = _RANKS_DEV.get(key)
    if qf is None:
        qf = _RANKS_DEV[key] = torch.tensor(qs, dtype=torch.float32).to(x2.device)
    lib = _lib.load()
    ws_bytes = lib.adalog_select_workspace_bytes(S, nq)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x2.device)
    out = torch.empty((nq, S), dtype=torch.float32, device=x2.device)
    rc = lib.adalog_positive_percentile_rows(x2.data_ptr(), S, n, nq, qf.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                             ws_bytes, _stream())
    _lib.check(rc, "adalog_positive_percentile_rows")
    return out


class ShardedSelect:
    """Radix select over segments whose elements are spread over the ranks (adalog_select_*): ``S`` global segments,
    ``R`` order statistics each.  Usage: ``hist_pass(p)`` on every rank, sum ``hist`` across ranks, ``pick(p)``, for the
    four passes; then ``quantiles()`` / ``values()``.  Local row s of ``x2`` belongs to global segment
    first + (s // inner) * outer + s % inner."""

    def __init__(self, x2, S: int, R: int, first: int, inner: int, outer: int, ranks=None, qfrac=None):
        self.x2 = _f32c(x2, "x")
        self.S, self.R, self.first, self.inner, self.outer = int(S), int(R), int(first), int(inner), int(outer)
        self.positive = qfrac is not None
        dev = self.x2.device
        self.qfrac = None if qfrac is None else torch.tensor(qfrac, dtype=torch.float32).to(dev)
        lib = _lib.load()
        self.ws_bytes = lib.adalog_select_workspace_bytes(self.S, self.R)
        self.ws = torch.empty((self.ws_bytes + 3) // 4, dtype=torch.int32, device=dev)
        self.hist = self.ws[: self.S * self.R * 256]          # the part ranks sum between counting and pick
        self._ranks = None if ranks is None else ranks.to(dev)
        rc = lib.adalog_select_init(self.ws.data_ptr(), self.ws_bytes, self.S, self.R, _ptr(self._ranks), _stream())
        _lib.check(rc, "adalog_select_init")

    def hist_pass(self, p: int):
        S_local, n_local = self.x2.shape
        rc = _lib.load().adalog_select_hist(self.x2.data_ptr(), S_local, n_local, self.first, self.inner, self.outer, self.S,
                                           self.R, p, int(self.positive), self.ws.data_ptr(), _stream())
        _lib.check(rc, "adalog_select_hist")

    def pick(self, p: int):
        rc = _lib.load().adalog_select_pick(self.ws.data_ptr(), self.S, self.R, p, _ptr(self.qfrac), int(self.positive), _stream())
        _lib.check(rc, "adalog_select_pick")

    def quantiles(self, weights, mbs: int):
        nq = self.R // 2
        out = torch.empty((nq, self.S // mbs), dtype=torch.float32, device=self.x2.device)
        w = weights.to(self.x2.device)
        rc = _lib.load().adalog_select_quantile_out(self.ws.data_ptr(), self.S, nq, w.data_ptr(), int(mbs), out.data_ptr(), _stream())
        _lib.check(rc, "adalog_select_quantile_out")
        return out

    def values(self):
        out = torch.empty((self.R, self.S), dtype=torch.float32, device=self.x2.device)
        rc = _lib.load().adalog_select_value_out(self.ws.data_ptr(), self.S, self.R, out.data_ptr(), _stream())
        _lib.check(rc, "adalog_select_value_out")
        return out


# ------------------------------------------------------------------------------------------------ small vector ops
def shift_fold(rowsum, w_scale, shift, bias):
    """fold[c][o] = bias[o] - shift * (w_scale[c][o] * rowsum[c][o])   (post-GELU shift folded into the bias)."""
    C, O = rowsum.shape
    w_scale = _f32c(w_scale, "w_scale")
    out = torch.empty((C, O), dtype=torch.float32, device=rowsum.device)
    rc = _lib.load().adalog_shift_fold(rowsum.data_ptr(), w_scale.data_ptr(), _f32c(shift, "shift").data_ptr(),
                                      _ptr(None if bias is None else _f32c(bias, "bias")), C, O, out.data_ptr(), _stream())
    _lib.check(rc, "adalog_shift_fold")
    return out


def minmax_rows(w2):
    """Per-row (min, max) of a [rows, I] matrix (K4; linear.py:267-273)."""
    w2 = _f32c(w2, "w")
    rows, I = w2.shape
    mn = torch.empty(rows, dtype=torch.float32, device=w2.device)
    mx = torch.empty_like(mn)
    rc = _lib.load().adalog_minmax_rows(w2.data_ptr(), rows, I, 0, mn.data_ptr(), mx.data_ptr(), _stream())
    _lib.check(rc, "adalog_minmax_rows")
    return mn, mx


def absminmax(x2, per_channel: bool):
    """(min |x|, max |x|) per tensor ([1]) or per column ([I]) of a [rows, I] matrix (K4; linear.py:282-287)."""
    x2 = _f32c(x2, "x")
    rows, I = x2.shape
    n = I if per_channel else 1
    mn = torch.empty(n, dtype=torch.float32, device=x2.device)
    mx = torch.empty_like(mn)
    rc = _lib.load().adalog_absminmax_cols(x2.data_ptr(), rows, I, int(bool(per_channel)), mn.data_ptr(), mx.data_ptr(),
                                          _stream())
    _lib.check(rc, "adalog_absminmax_cols")
    return mn, mx


# ------------------------------------------------------------------------------------------------ BRECQ (K17)
def uniform_fake_quant_backward(gy, x, scale, zero_point, n_bits: int, sym: bool, want_gscale: bool, want_gzp: bool):
    """-> (gx, gscale | None, gzp | None) of the straight-through training form."""
    gy, x, scale = _f32c(gy, "gy"), _f32c(x, "x"), _f32c(scale, "scale")
    zp = None if sym else _f32c(zero_point, "zero_point")
    n_ch, inner = broadcast_layout(x.shape, scale.shape)
    if inner == 1 and n_ch > 1:
        raise NotImplementedError("training gradients for per-last-dim-channel activations are not needed by BRECQ "
                                  "(channel-wise layers are re-parameterised to per-tensor before block reconstruction)")
    lib = _lib.load()
    gx = torch.empty_like(x)
    gs = torch.empty_like(scale) if want_gscale else None
    gz = torch.empty_like(zp) if (want_gzp and zp is not None) else None
    ws = None
    if gs is not None or gz is not None:
        nb = lib.adalog_uniform_fq_backward_blocks(x.numel(), n_ch, inner)
        ws = torch.empty(2 * (x.numel() // inner) * nb, dtype=torch.float32, device=x.device)
    rc = lib.adalog_uniform_fq_backward(gy.data_ptr(), x.data_ptr(), gx.data_ptr(), x.numel(), scale.data_ptr(), _ptr(zp),
                                        n_ch, inner, int(n_bits), int(bool(sym)), _ptr(gs), _ptr(gz), _ptr(ws), _stream())
    _lib.check(rc, "adalog_uniform_fq_backward")
    return gx, gs, gz


def log_fake_quant_backward(gy, x, y, scale, q, n_bits: int, shift, sub_shift: bool, pre_gelu: bool = False):
    """``pre_gelu``: backward of log_fake_quant(pre_gelu=True) -- x is the GELU's input, gx includes the GELU's derivative."""
    gy, x, y, scale = _f32c(gy, "gy"), _f32c(x, "x"), _f32c(y, "y"), _f32c(scale, "scale")
    gx = torch.empty_like(x)
    gs = torch.empty_like(scale)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    rc = _lib.load().adalog_log_fq_backward_pre(gy.data_ptr(), x.data_ptr(), y.data_ptr(), gx.data_ptr(), x.numel(),
                                               scale.data_ptr(), q.data_ptr(), int(n_bits),
                                               _ptr(None if shift is None else _f32c(shift, "shift")), int(bool(sub_shift)),
                                               gs.data_ptr(), ws.data_ptr(), int(bool(pre_gelu)), _stream())
    _lib.check(rc, "adalog_log_fq_backward")
    return gx, gs


def adaround(w2, alpha2, scale, zero_point, n_bits: int, soft: bool, gy=None):
    """Forward value (gy is None) or d/d alpha (gy given) of the AdaRound weight quantiser; w2/alpha2: [rows, inner]."""
    w2, alpha2 = _f32c(w2, "w"), _f32c(alpha2, "alpha")
    rows, inner = w2.shape
    out = torch.empty_like(w2)
    rc = _lib.load().adalog_adaround(w2.data_ptr(), alpha2.data_ptr(), _ptr(None if gy is None else _f32c(gy, "gy")),
                                    out.data_ptr(), rows, inner, _f32c(scale, "scale").data_ptr(),
                                    _f32c(zero_point, "zero_point").data_ptr(), int(n_bits), int(bool(soft)),
                                    int(gy is not None), _stream())
    _lib.check(rc, "adalog_adaround")
    return out


def adaround_t(w2, alpha2, scale, zero_point, n_bits: int, soft: bool):
    """Forward value of the AdaRound weight quantiser in both orientations: (out [rows, inner], out_t [inner, rows])."""
    w2, alpha2 = _f32c(w2, "w"), _f32c(alpha2, "alpha")
    rows, inner = w2.shape
    out = torch.empty_like(w2)
    out_t = torch.empty((inner, rows), dtype=torch.float32, device=w2.device)
    rc = _lib.load().adalog_adaround_t(w2.data_ptr(), alpha2.data_ptr(), out.data_ptr(), out_t.data_ptr(), rows, inner,
                                      _f32c(scale, "scale").data_ptr(), _f32c(zero_point, "zero_point").data_ptr(), int(n_bits),
                                      int(bool(soft)), _stream())
    _lib.check(rc, "adalog_adaround_t")
    return out, out_t


def round_loss(alpha, b, galpha=None, gscale: float = 1.0, want_loss: bool = True, gmul=None, overwrite: bool = False):
    """``b``: python float, or a device fp32 tensor of one element (read by the kernel: HIP-graph friendly).
    ``galpha`` (optional) receives gscale * gmul * d/d alpha: added to it, or written over it (``overwrite``);
    ``gmul``: optional one-element device tensor (the upstream gradient of the scalar loss)."""
    alpha = _f32c(alpha, "alpha")
    loss = torch.empty(1, dtype=torch.float32, device=alpha.device) if want_loss else None
    ws = torch.empty(1024, dtype=torch.float32, device=alpha.device) if want_loss else None
    b_dev = _f32c(b, "b") if torch.is_tensor(b) else None
    rc = _lib.load().adalog_round_loss(alpha.data_ptr(), alpha.numel(), 0.0 if b_dev is not None else float(b), _ptr(b_dev),
                                      _ptr(loss), _ptr(galpha), float(gscale),
                                      _ptr(None if gmul is None else _f32c(gmul, "gmul")), int(bool(overwrite)),
                                      _ptr(ws), _stream())
    _lib.check(rc, "adalog_round_loss")
    return loss


def rec_loss(pred, tgt, scale: float):
    """scale * sum (pred - tgt)^2 as a one-element tensor (LossFunction.lp_loss, p = 2; block_recon.py:186-199)."""
    pred, tgt = _f32c(pred, "pred"), _f32c(tgt, "tgt")
    if pred.shape != tgt.shape:
        raise _lib.AdalogHipError("rec_loss: pred and tgt differ in shape")
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    ws = torch.empty(2048, dtype=torch.float32, device=pred.device)
    rc = _lib.load().adalog_rec_loss(pred.data_ptr(), tgt.data_ptr(), pred.numel(), float(scale), loss.data_ptr(), ws.data_ptr(),
                                    _stream())
    _lib.check(rc, "adalog_rec_loss")
    return loss


def rec_loss_backward(pred, tgt, scale: float, gmul):
    pred, tgt = _f32c(pred, "pred"), _f32c(tgt, "tgt")
    gp = torch.empty_like(pred)
    rc = _lib.load().adalog_rec_loss_backward(pred.data_ptr(), tgt.data_ptr(), pred.numel(), float(scale),
                                             _f32c(gmul, "gmul").data_ptr(), gp.data_ptr(), _stream())
    _lib.check(rc, "adalog_rec_loss_backward")
    return gp


def round_loss_multi(alphas, b, weight: float, want_grads: bool = True, gate=None):
    """weight * sum_t sum(1 - |2h(alpha_t)-1|^b) and its gradients, one launch for all tensors of a block.
    Returns (loss [1], [grad_t]) -- grads None with ``want_grads=False`` (alpha_step_multi computes them itself).  ``gate``: optional
    one-element device tensor multiplied into the VALUE (the gradients stay ungated: the caller's backward applies it)."""
    import ctypes
    alphas = [_f32c(a, "alpha") for a in alphas]
    n = len(alphas)
    grads = [torch.empty_like(a) for a in alphas] if want_grads else None
    PA, NA = ctypes.c_void_p * n, ctypes.c_int64 * n
    ap, ns = PA(*[a.data_ptr() for a in alphas]), NA(*[a.numel() for a in alphas])
    gp = PA(*[g_.data_ptr() for g_ in grads]) if want_grads else None
    lib = _lib.load()
    dev = alphas[0].device
    ws = torch.empty(int(lib.adalog_round_loss_multi_workspace(ns, n)), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    b_dev = _f32c(b, "b") if torch.is_tensor(b) else None
    rc = lib.adalog_round_loss_multi(ap, gp, ns, n, 0.0 if b_dev is not None else float(b), _ptr(b_dev), float(weight),
                                     loss.data_ptr(), ws.data_ptr(), _ptr(None if gate is None else _f32c(gate, "gate")), _stream())
    _lib.check(rc, "adalog_round_loss_multi")
    return loss, grads


# ---------------------------------------------------------------------------------------------- BRECQ training-mode GEMMs
def _mat_layout(t: torch.Tensor):
    """How the kernel reads a logical [..., R, K] operand: (trans, ld, groups, group_stride, Gi, outer_stride) or None.
    trans = 0: K contiguous (rows of ld elements); trans = 1: R contiguous (K-major).  The leading dims form the group index
    g = go * Gi + gi at gi * group_stride + go * outer_stride: they must collapse into one level (Gi = groups) or two."""
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() < 2:
        return None
    R, K = t.shape[-2:]
    sr, sk = t.stride(-2), t.stride(-1)
    if (sk == 1 or K == 1) and (sr >= K or R == 1):
        trans, ld = 0, (sr if R > 1 else max(K, 4))
    elif (sr == 1 or R == 1) and (sk >= R or K == 1):
        trans, ld = 1, (sk if K > 1 else max(R, 4))
    else:
        return None
    G, gs, Gi, go = 1, 0, 1, 0
    lead = [(t.shape[i], t.stride(i)) for i in range(t.dim() - 2) if t.shape[i] != 1]
    if lead:
        levels = []                                  # innermost first: [count, stride] of each run of collapsing dims
        for n_, s_ in reversed(lead):
            if levels and s_ == levels[-1][0] * levels[-1][1]:
                levels[-1][0] *= n_
            else:
                levels.append([n_, s_])
        if len(levels) > 2:
            return None
        Gi, gs = levels[0]
        G = Gi
        if len(levels) == 2:
            G, go = Gi * levels[1][0], levels[1][1]
    if t.data_ptr() % 16:
        return None
    return trans, ld, G, gs, Gi, go


def _group_levels(layouts):
    """A common two-level description (Gi, [(inner, outer) stride per operand]) of operands laid out by _mat_layout, or None."""
    G = layouts[0][2]
    if any(l[2] != G for l in layouts):
        return None
    two = {l[4] for l in layouts if l[4] != G}
    if len(two) > 1:
        return None
    Gi = two.pop() if two else G
    return Gi, [(l[3], l[5] if l[4] != G else Gi * l[3]) for l in layouts]


def gemm_f32x3_ok(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> bool:
    """Whether a @ b^T (a: [..., M, K], b: [..., N, K], same leading shape) runs on adalog_gemm_f32x3[_g2]."""
    la, lb = _mat_layout(a), _mat_layout(b)
    if la is None or lb is None or a.shape[-1] != b.shape[-1]:
        return False
    lays = [la, lb]
    M, N = a.shape[-2], b.shape[-2]
    if out is not None:
        lo = _mat_layout(out)
        if lo is None or lo[0] != 0 or tuple(out.shape[-2:]) != (M, N):
            return False
        lays.append(lo)
    if _group_levels(lays) is None:
        return False
    if a.numel() == 0 or b.numel() == 0:
        return False
    if bias is not None and (N % 16 or bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16
                             or bias.numel() != N):
        return False
    per = 4 * max(M * la[1] if not la[0] else a.shape[-1] * la[1], N * lb[1] if not lb[0] else a.shape[-1] * lb[1])
    return per < (1 << 31)


def gemm_f32x3(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, alpha: float = 1.0,
               allow_split: bool = True, alpha_dev: Optional[torch.Tensor] = None, exact_a: bool = False,
               exact_b: bool = False, out: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """alpha * alpha_dev[0] * a @ b^T (+ bias) for fp32 operands of either memory orientation, at fp32 accuracy on the bf16
    matrix cores (csrc/brecq_gemm.hip).  a: [..., M, K], b: [..., N, K] -> [..., M, N] (contiguous, or written into ``out``: any
    view whose last dim is contiguous and whose leading dims form at most two stride levels).
    exact_a / exact_b: True / 1 = that operand holds integers exactly representable in bf16 (3 products instead of 6 where
    supported); 2 = a two-term split of that operand is enough (hi + mid: 2^-16 relative -- the gradient contractions of a BRECQ
    iteration: 3 products for two such operands, 2 against an exact one)."""
    if not gemm_f32x3_ok(a, b, bias, out):
        raise _lib.AdalogHipError(f"gemm_f32x3: unsupported operand layout {tuple(a.shape)}/{a.stride()} x {tuple(b.shape)}/{b.stride()}")
    la, lb = _mat_layout(a), _mat_layout(b)
    M, K, N, G = a.shape[-2], a.shape[-1], b.shape[-2], la[2]
    if out is None:
        out = torch.empty(a.shape[:-2] + (M, N), dtype=torch.float32, device=a.device)
    lo = _mat_layout(out)
    Gi, ((sa, sao), (sb, sbo), (sc, sco)) = _group_levels([la, lb, lo])
    lib = _lib.load()
    sp, ea, eb = (1 if allow_split else 0), int(exact_a), int(exact_b)
    wsb = int(lib.adalog_gemm_f32x3_workspace_bytes(M, N, K, G, sp, ea, eb, la[0], lb[0]))
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=a.device) if wsb else None
    if addend is not None:
        # ``addend`` (fp32, the shape of the result, contiguous): added to the product -- inside the split product's reduction pass
        # (adalog_gemm_f32x3_add); one level of groups, contiguous result
        addend = _f32c(addend, "addend")
        if Gi != G or addend.numel() != out.numel() or not out.is_contiguous() or addend.data_ptr() % 16:
            raise _lib.AdalogHipError("gemm_f32x3: addend needs a contiguous result of one group level")
        rc = lib.adalog_gemm_f32x3_add(a.data_ptr(), la[1], la[0], b.data_ptr(), lb[1], lb[0], out.data_ptr(), lo[1], M, N, K, G, sa, sb, sc,
                                       _ptr(bias), float(alpha), _ptr(alpha_dev), sp, ea, eb, addend.data_ptr(), _ptr(ws), _stream())
        _lib.check(rc, "adalog_gemm_f32x3_add")
        return out
    rc = lib.adalog_gemm_f32x3_g2(a.data_ptr(), la[1], la[0], b.data_ptr(), lb[1], lb[0], out.data_ptr(), lo[1], M, N, K, G,
                                  sa, sb, sc, Gi, sao, sbo, sco, _ptr(bias), float(alpha), _ptr(alpha_dev), sp, ea, eb, _ptr(ws),
                                  _stream())
    _lib.check(rc, "adalog_gemm_f32x3_g2")
    return out


def gemm_f32x3_planes(a: torch.Tensor, bp: torch.Tensor, K: int, bias: Optional[torch.Tensor] = None, alpha: float = 1.0,
                      allow_split: bool = True, alpha_dev: Optional[torch.Tensor] = None, exact_a: bool = False) -> torch.Tensor:
    """a @ B^T (+ bias) with B handed over pre-split by pack_split3(): bp is its [1, G, N, 3*Kt] bf16 image (hi | mid | lo per
    row).  a: [..., M, K] fp32, K-contiguous -> [..., M, N]."""
    la = _mat_layout(a)
    if la is None or la[0] != 0 or la[4] != la[2] or a.shape[-1] != K or bp.dtype != torch.bfloat16 or not bp.is_contiguous():
        raise _lib.AdalogHipError(f"gemm_f32x3_planes: unsupported operand layout {tuple(a.shape)}/{a.stride()}")
    G, N, Kt = bp.shape[-3], bp.shape[-2], bp.shape[-1] // 3
    M = a.shape[-2]
    if la[2] != G or N % 4 or (bias is not None and N % 16):
        raise _lib.AdalogHipError("gemm_f32x3_planes: group count / N not supported")
    out = torch.empty(a.shape[:-2] + (M, N), dtype=torch.float32, device=a.device)
    lib = _lib.load()
    sp, ea = (1 if allow_split else 0), (1 if exact_a else 0)
    wsb = int(lib.adalog_gemm_f32x3_planes_workspace_bytes(M, N, K, G, sp, ea))
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=a.device) if wsb else None
    rc = lib.adalog_gemm_f32x3_planes(a.data_ptr(), la[1], bp.data_ptr(), Kt, out.data_ptr(), N, M, N, K, G, la[3], M * N,
                                      _ptr(bias), float(alpha), _ptr(alpha_dev), sp, ea, _ptr(ws), _stream())
    _lib.check(rc, "adalog_gemm_f32x3_planes")
    return out


def uniform_int(x: torch.Tensor, scale: torch.Tensor, zero_point: torch.Tensor, n_bits: int) -> torch.Tensor:
    """q - z of the per-tensor asymmetric quantiser as fp32 (exact small integers); (q - z) * scale is uniform_fake_quant(x)."""
    x = _f32c(x, "x")
    y = torch.empty_like(x)
    rc = _lib.load().adalog_uniform_int_f32(x.data_ptr(), y.data_ptr(), x.numel(), _f32c(scale, "scale").data_ptr(),
                                            _f32c(zero_point, "zero_point").data_ptr(), int(n_bits), _stream())
    _lib.check(rc, "adalog_uniform_int_f32")
    return y


def permute_heads(x: torch.Tensor, P: int, H: int, inverse: bool = False) -> torch.Tensor:
    """x [B, N, P*H*D] -> [P, B, H, N, D] in one pass (q, k, v of an attention block: the reshape / permute / unbind of
    reference utils/wrap_net.py:19-33 as contiguous tensors); ``inverse``: [P, B, H, N, D] -> [B, N, P*H*D]."""
    x = _f32c(x, "x")
    if inverse:
        P_, B, H_, N, D = x.shape
        if (P_, H_) != (P, H):
            raise _lib.AdalogHipError("permute_heads: shape does not match P, H")
        out = torch.empty((B, N, P * H * D), dtype=torch.float32, device=x.device)
    else:
        B, N, C = x.shape
        if C % (P * H) or (C // (P * H)) % 4:
            raise _lib.AdalogHipError("permute_heads: the head dimension must be a multiple of 4")
        D = C // (P * H)
        out = torch.empty((P, B, H, N, D), dtype=torch.float32, device=x.device)
    rc = _lib.load().adalog_permute_heads(x.data_ptr(), out.data_ptr(), B, N, int(P), int(H), int(D), int(bool(inverse)), _stream())
    _lib.check(rc, "adalog_permute_heads")
    return out


def _qkv_params(scales, zps, n_bits, H):
    import ctypes
    PA, IA = ctypes.c_void_p * 3, ctypes.c_int * 3
    sc = [_f32c(t.reshape(-1), "scale") for t in scales]
    zp = [_f32c(t.reshape(-1), "zero_point") for t in zps]
    for a_, b_ in zip(sc, zp):
        if a_.numel() not in (1, H) or b_.numel() != a_.numel():
            raise _lib.AdalogHipError("qkv quantisers: one (scale, zero point) per tensor or per head")
    ph = [1 if t.numel() == H and H > 1 else 0 for t in sc]
    return sc, zp, PA(*[t.data_ptr() for t in sc]), PA(*[t.data_ptr() for t in zp]), IA(*ph), IA(*[int(b) for b in n_bits])


def qkv_split_quant(x: torch.Tensor, H: int, scales, zps, n_bits):
    """x [B, N, 3*H*D] -> (q_sim, k_sim, v_sim) [B, H, N, D]: the head split and the three asymmetric uniform fake-quantisations
    (per tensor or per head) in one pass.  D = 32 or 64."""
    x = _f32c(x, "x")
    B, N, C = x.shape
    D = C // (3 * H)
    if C != 3 * H * D or D not in (32, 64):
        raise _lib.AdalogHipError("qkv_split_quant: head dimension 32 or 64")
    ys = [torch.empty((B, H, N, D), dtype=torch.float32, device=x.device) for _ in range(3)]
    sc, zp, psc, pzp, ph, nb = _qkv_params(scales, zps, n_bits, H)
    rc = _lib.load().adalog_qkv_split_quant(x.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), B, N, int(H), int(D), psc,
                                           pzp, ph, nb, _stream())
    _lib.check(rc, "adalog_qkv_split_quant")
    return tuple(ys)


def qkv_merge_quant_backward(gys, x: torch.Tensor, H: int, scales, zps, n_bits, want_gx: bool = True):
    """Gradients of qkv_split_quant: gys = 3 tensors [B, H, N, D] (None = zeros) -> (gx [B, N, 3*H*D] | None, [gscale_p] shaped like
    scales[p])."""
    import ctypes
    x = _f32c(x, "x")
    B, N, C = x.shape
    D = C // (3 * H)
    gs = [None if g_ is None else _f32c(g_, "gy") for g_ in gys]
    lib = _lib.load()
    sc, zp, psc, pzp, ph, nb = _qkv_params(scales, zps, n_bits, H)
    gx = torch.empty_like(x) if want_gx else None
    gsc = [torch.empty_like(t) for t in sc]
    nch = int(lib.adalog_qkv_quant_chunks(B, N, int(D)))
    ws = torch.empty(3 * H * nch, dtype=torch.float32, device=x.device)
    PA = ctypes.c_void_p * 3
    rc = lib.adalog_qkv_merge_quant_backward(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), x.data_ptr(), _ptr(gx), B, N, int(H), int(D), psc, pzp,
                                             ph, nb, PA(*[t.data_ptr() for t in gsc]), ws.data_ptr(), _stream())
    _lib.check(rc, "adalog_qkv_merge_quant_backward")
    return gx, [g_.view_as(s_) for g_, s_ in zip(gsc, scales)]


def scaled_softmax(x: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(x * scale, dim=-1) in one pass (rows of <= 1024 values)."""
    x = _f32c(x, "x")
    y = torch.empty_like(x)
    n = x.shape[-1]
    rc = _lib.load().adalog_scaled_softmax(x.data_ptr(), y.data_ptr(), x.numel() // max(n, 1), n, float(scale), _stream())
    _lib.check(rc, "adalog_scaled_softmax")
    return y


def scaled_softmax_backward(gy: torch.Tensor, y: torch.Tensor, scale: float) -> torch.Tensor:
    """d/dx of softmax(x * scale) given y = its output: scale * y * (gy - sum(gy * y))."""
    gy, y = _f32c(gy, "gy"), _f32c(y, "y")
    gx = torch.empty_like(y)
    n = y.shape[-1]
    rc = _lib.load().adalog_scaled_softmax_backward(gy.data_ptr(), y.data_ptr(), gx.data_ptr(), y.numel() // max(n, 1), n, float(scale),
                                                   _stream())
    _lib.check(rc, "adalog_scaled_softmax_backward")
    return gx


def brecq_prepare(src_in, src_out, idx, dst_in, dst_out, sched_row=None, sched_dev=None):
    """dst_in.copy_(src_in[idx]); dst_out.copy_(src_out[idx]); sched_dev.copy_(sched_row) in one launch (adalog_brecq_prepare): what a
    BRECQ iteration does before its graph is replayed.  False when the tensors do not qualify (the caller takes the three steps)."""
    ts = (src_in, src_out, dst_in, dst_out)
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts):
        return False
    bs = idx.numel()
    row_in, row_out = src_in[0].numel(), src_out[0].numel()
    if (idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous() or row_in % 4 or row_out % 4 or dst_in.numel() != bs * row_in
            or dst_out.numel() != bs * row_out):
        return False
    ns = 0 if sched_row is None else sched_row.numel()
    if ns and not (ns <= 8 and sched_row.is_cuda and sched_row.dtype == torch.float32 and sched_row.is_contiguous()
                   and sched_dev is not None and sched_dev.numel() >= ns and sched_dev.is_contiguous()):
        return False
    rc = _lib.load().adalog_brecq_prepare(src_in.data_ptr(), src_out.data_ptr(), idx.data_ptr(), dst_in.data_ptr(), dst_out.data_ptr(), bs,
                                         row_in, row_out, _ptr(sched_row if ns else None), _ptr(sched_dev if ns else None), ns, _stream())
    _lib.check(rc, "adalog_brecq_prepare")
    return True


def merge_heads(parts, B: int, N: int, H: int, D: int) -> torch.Tensor:
    """The P <= 4 tensors [B, H, N, D] (None = zeros) -> [B, N, P*H*D]: the gradient of permute_heads from its parts' gradients."""
    P = len(parts)
    ps = [None if t is None else _f32c(t, "part") for t in parts]
    dev = next(t for t in ps if t is not None).device
    out = torch.empty((B, N, P * H * D), dtype=torch.float32, device=dev)
    ptr = [_ptr(t) for t in ps] + [None] * (4 - P)
    rc = _lib.load().adalog_merge_heads(ptr[0], ptr[1], ptr[2], ptr[3], out.data_ptr(), B, N, P, int(H), int(D), _stream())
    _lib.check(rc, "adalog_merge_heads")
    return out


def alpha_step_multi(alphas, ws, gws, scales, zps, exp_avg, exp_avg_sq, inners, n_bits, step_dev, lr, beta1: float, beta2: float,
                     eps: float, b, weight: float, gmul, gate=None):
    """AdaRound's alpha of up to 16 layers: gradient (through w_sim from gws[t] = dL/dw_sim or None, plus ``gmul`` times the
    rounding regulariser's) and Adam step in ONE launch.  ``b``: float or device tensor [1]; ``gmul``: device tensor [1] or None;
    ``lr``: float or device tensor [1]; ``step_dev``: device fp32 [1], steps taken so far (advanced here)."""
    import ctypes
    n = len(alphas)
    PA, NA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    for t in list(alphas) + list(ws) + [g_ for g_ in gws if g_ is not None] + list(scales) + list(zps) + list(exp_avg) + list(exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.AdalogHipError("alpha_step_multi: contiguous fp32 device tensors expected")
    for a_, w_, g_ in zip(alphas, ws, gws):
        if w_.numel() != a_.numel() or (g_ is not None and g_.numel() != a_.numel()):
            raise _lib.AdalogHipError("alpha_step_multi: alpha, w and dL/dw_sim must have the same number of elements")
    lr_dev = lr if torch.is_tensor(lr) else None
    b_dev = b if torch.is_tensor(b) else None
    rc = _lib.load().adalog_alpha_step_multi(
        PA(*[t.data_ptr() for t in alphas]), PA(*[t.data_ptr() for t in ws]), PA(*[None if t is None else t.data_ptr() for t in gws]),
        PA(*[t.data_ptr() for t in scales]), PA(*[t.data_ptr() for t in zps]), PA(*[t.data_ptr() for t in exp_avg]),
        PA(*[t.data_ptr() for t in exp_avg_sq]), NA(*[t.numel() for t in alphas]), NA(*[int(i) for i in inners]),
        IA(*[int(i) for i in n_bits]), n, 0.0 if lr_dev is not None else float(lr), _ptr(lr_dev), float(beta1), float(beta2), float(eps),
        step_dev.data_ptr(), 0.0 if b_dev is not None else float(b), _ptr(b_dev), float(weight), _ptr(gmul), _ptr(gate), _stream())
    _lib.check(rc, "adalog_alpha_step_multi")


def adam_multi(params, grads, exp_avg, exp_avg_sq, step_dev, lr, beta1: float, beta2: float, eps: float):
    """One Adam step (torch.optim.Adam defaults) for up to 16 fp32 tensors in one launch; ``lr``: float or device tensor [1];
    ``step_dev``: device fp32 [1] holding the steps taken so far (advanced here)."""
    import ctypes
    n = len(params)
    PA, NA = ctypes.c_void_p * n, ctypes.c_int64 * n
    for t in list(params) + list(grads) + list(exp_avg) + list(exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.AdalogHipError("adam_multi: contiguous fp32 device tensors expected")
    lr_dev = lr if torch.is_tensor(lr) else None
    rc = _lib.load().adalog_adam_multi(PA(*[t.data_ptr() for t in params]), PA(*[t.data_ptr() for t in grads]),
                                      PA(*[t.data_ptr() for t in exp_avg]), PA(*[t.data_ptr() for t in exp_avg_sq]),
                                      NA(*[t.numel() for t in params]), n, 0.0 if lr_dev is not None else float(lr), _ptr(lr_dev),
                                      float(beta1), float(beta2), float(eps), step_dev.data_ptr(), _stream())
    _lib.check(rc, "adalog_adam_multi")
