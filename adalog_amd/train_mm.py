"""Training-mode (BRECQ) contractions of the quantised layers on the hand-written fp32-accurate MFMA kernel.

One BRECQ iteration (reference utils/block_recon.py:116-121) runs, per Linear layer, F.linear(x_sim, w_sim, bias) forward and the
two backward products dL/dx_sim = dL/dy . w_sim and dL/dw_sim = dL/dy^T . x_sim (reference quant_layers/linear.py:46-50 under
autograd).  Here these are csrc/brecq_gemm.hip launches (adalog_gemm_f32x3 / _planes: fp32 operands split in registers into three
bf16 terms, six bf16 MFMA products, fp32 accumulation -- fp32-class accuracy), with the structure of the operands used:

  * a uniformly fake-quantised activation is  x_sim = s_a * x_int  with x_int = q - z a small integer: the kernels take x_int
    (exact in ONE bf16 term: 3 products instead of 6, no split) and apply the trained scale s_a in the epilogue; the
    straight-through gradients of (x, s_a) come from the same fused kernel as before, fed with dL/dx_sim;
  * the soft-rounded weights w_sim (small, re-read by every row tile) are split once per iteration by the packer, in both
    orientations, instead of once per tile in the GEMM;
  * dL/dw_sim reads dL/dy and x_int K-major straight from their row-major tensors (no transposed copies).

ADALOG_BRECQ_MM=0 falls back to the fp32 library GEMMs (torch / rocBLAS) for A/B measurements.
"""
import os
import weakref

import torch
import torch.nn.functional as F

from . import backend

ENABLED = os.environ.get("ADALOG_BRECQ_MM", "1") != "0"
# pre-split w_sim planes (1) or the split in the GEMM's registers (0, default: measured 571 against 546 it/s on a deit_small block --
# the eight packer launches per iteration cost more than the per-tile splits they save)
WEIGHT_PLANES = os.environ.get("ADALOG_BRECQ_WPLANES", "0") != "0"
SPLIT_HEADS = os.environ.get("ADALOG_BRECQ_SPLIT_HEADS", "1") != "0"      # q, k, v by one permute launch each way
HEADS_LAST = os.environ.get("ADALOG_BRECQ_HEADS_LAST", "1") != "0"        # softmax.v writes [B, N, H, D] in place
# forward products read the weights K-major (from a transposed copy made once per iteration): with BOTH operands K-contiguous every
# LDS-DMA request of a 16-element K-step fetches half cache lines -- measured 13-24 % slower than any form with one K-major operand
W_KMAJOR = os.environ.get("ADALOG_BRECQ_WT", "1") != "0"
QKV_FUSED = os.environ.get("ADALOG_BRECQ_QKV_QUANT", "1") != "0"          # q / k / v split + their three quantisers as one pass
FUSED_SOFTMAX = os.environ.get("ADALOG_BRECQ_SOFTMAX", "1") != "0"         # attn * scale + softmax as one pass each way
ADDEND_FUSED = os.environ.get("ADALOG_BRECQ_ADDEND", "1") != "0"           # the residual behind fc2 added inside the product's reduction pass
INT_ACT = os.environ.get("ADALOG_BRECQ_INT_ACT", "1") != "0"            # integer activation operand (0: s_a * x_int as fp32)
# bf16 terms per general operand of the GRADIENT contractions (dL/dx, dL/dw, the attention products' backward): 2 = hi + mid (2^-16
# relative: three MFMA products instead of six, two instead of three against the integer activation); 3 = the forward's three-term
# split (<= rocBLAS fp32 error).  The forward products always take three terms.  north_star asks for 1e-3: the trajectory test
# (tests/test_gpu_layers.py) holds the trained values to the reference's own loop at that bar with either setting.
GRAD_TERMS = 2 if os.environ.get("ADALOG_BRECQ_GRAD_TERMS", "2") == "2" else 0
# ... and of the FORWARD contractions' general operands (the soft-rounded weights; q, k, v and the probabilities): 2 = hi + mid as well
# (outputs 1.5e-5 relative from the fp64 product; ADALOG_BRECQ_FWD_TERMS=3 restores the three-term split).  The trained values after
# 20 iterations of the reference's own loop (tests/golden/brecq_traj.npz) agree to 8e-5 with two-term gradients and forward, 6.5e-6
# with three terms everywhere; 20 000 iterations reach the same reconstruction error (profiles/r06_brecq_convergence.json).
# Only BRECQ's training iterations take these products: the calibrated / reconstructed model's own forward (validate, quant_forward)
# runs on the integer MFMA kernels.
FWD_TERMS = 2 if os.environ.get("ADALOG_BRECQ_FWD_TERMS", "2") == "2" else 0


def _usable(x2, w2, bias):
    if not (ENABLED and x2.is_cuda and x2.dtype == torch.float32 and w2.dtype == torch.float32):
        return False
    be = backend.get()
    if not hasattr(be, "gemm_f32x3"):
        return False
    M, K = x2.shape
    N = w2.shape[0]
    # every form used below: K-contiguous and K-major reads of x2 / gy / w2 need 16-byte aligned rows in both orientations
    return K % 4 == 0 and N % 4 == 0 and M >= 1 and x2.data_ptr() % 16 == 0 and w2.data_ptr() % 16 == 0 \
        and (bias is None or (N % 16 == 0 and bias.is_contiguous() and bias.data_ptr() % 16 == 0))


def _planes(be, w2):
    """w2 [N, K] -> its pre-split image for adalog_gemm_f32x3_planes (rows of hi | mid | lo)."""
    return be.pack_split3(w2.unsqueeze(0), 64)


_KMAJOR_OFFER = {}                                    # data_ptr of a [N, K] tensor -> (its [K, N] image, N, K): one entry per layer


def offer_kmajor(w2, w2_t):
    """The producer of w2 [N, K] (the AdaRound forward) hands over its [K, N] image; the next forward product on w2 takes it."""
    if len(_KMAJOR_OFFER) > 64:
        _KMAJOR_OFFER.clear()
    # (keyed by address: the entry remembers WHICH tensor stood there -- a weak reference and its version -- so that a later tensor
    # of the same shape allocated at the same address cannot pick up a stale image)
    _KMAJOR_OFFER[w2.data_ptr()] = (w2_t, w2.shape[0], w2.shape[1], weakref.ref(w2), w2._version)


def reset_offers():
    """Start of a BRECQ iteration: forget images nobody took (their keys are addresses that may be reused)."""
    _KMAJOR_OFFER.clear()


def _kmajor(w2):
    """w2 [N, K] as the same logical matrix over K-major storage (a [K, N] copy, viewed back)."""
    if not W_KMAJOR or w2.shape[0] % 4:
        return w2
    hit = _KMAJOR_OFFER.pop(w2.data_ptr(), None)       # made by the producer of w_sim when it has one (views share the pointer)
    if hit is not None and hit[1:3] == (w2.shape[0], w2.shape[1]) and w2.is_contiguous():
        src = hit[3]()                                   # the offering tensor must still be alive (nothing else can then sit at its
        if src is not None and src._version == hit[4]:   # address) and unmodified since the offer
            return hit[0].t()
    return w2.t().contiguous().t()


class _LinearFn(torch.autograd.Function):
    """y = x2 @ w2^T + bias with general fp32 operands (x2 [M, K], w2 [N, K])."""

    @staticmethod
    def forward(ctx, x2, w2, bias, addend=None):
        """``addend`` [M, N] (optional): added to the product -- in the reduction pass of a K-split product (the residual stream of a
        transformer block behind fc2: no separate add launch)."""
        be = backend.get()
        ctx.save_for_backward(x2, w2)
        ctx.has_addend = addend is not None
        if WEIGHT_PLANES:
            out = be.gemm_f32x3_planes(x2, _planes(be, w2), x2.shape[1], bias)
            return out if addend is None else out + addend
        if addend is not None:
            return be.gemm_f32x3(x2, _kmajor(w2), bias, exact_a=FWD_TERMS, exact_b=FWD_TERMS, addend=addend)
        return be.gemm_f32x3(x2, _kmajor(w2), bias, exact_a=FWD_TERMS, exact_b=FWD_TERMS)

    @staticmethod
    def backward(ctx, gy):
        x2, w2 = ctx.saved_tensors
        be = backend.get()
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = be.gemm_f32x3_planes(gy, _planes(be, w2.t()), w2.shape[0]) if WEIGHT_PLANES \
                else be.gemm_f32x3(gy, w2.t(), exact_a=GRAD_TERMS, exact_b=GRAD_TERMS)
        if ctx.needs_input_grad[1]:
            gw = be.gemm_f32x3(gy.t(), x2.t(), exact_a=GRAD_TERMS, exact_b=GRAD_TERMS)
        gb = gy.sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, (gy if ctx.has_addend and ctx.needs_input_grad[3] else None)


class _QuantLinearFn(torch.autograd.Function):
    """y = (s_a * x_int) @ w2^T + bias,  x_int = clamp(rne(x / s_a) + z, 0, 2^bits - 1) - z   (per-tensor uniform quantiser
    with the straight-through estimator, reference quantizers/uniform.py:29-35 + _ste.py:5-6, fused with the layer's GEMMs)."""

    @staticmethod
    def forward(ctx, x2, a_scale, a_zp, w2, bias, n_bits):
        be = backend.get()
        xi = be.uniform_int(x2, a_scale, a_zp, n_bits)
        ctx.save_for_backward(x2, xi, a_scale, a_zp, w2)
        ctx.n_bits = n_bits
        if WEIGHT_PLANES:
            return be.gemm_f32x3_planes(xi, _planes(be, w2), x2.shape[1], bias, alpha_dev=a_scale, exact_a=True)
        return be.gemm_f32x3(xi, _kmajor(w2), bias, alpha_dev=a_scale, exact_a=True, exact_b=FWD_TERMS)

    @staticmethod
    def backward(ctx, gy):
        x2, xi, a_scale, a_zp, w2 = ctx.saved_tensors
        be = backend.get()
        gy = gy.contiguous()
        gx = gs = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gxs = be.gemm_f32x3_planes(gy, _planes(be, w2.t()), w2.shape[0]) if WEIGHT_PLANES \
                else be.gemm_f32x3(gy, w2.t(), exact_a=GRAD_TERMS, exact_b=GRAD_TERMS)
            gx, gs, _ = be.uniform_fake_quant_backward(gxs, x2, a_scale, a_zp, ctx.n_bits, False, ctx.needs_input_grad[1], False)
            if not ctx.needs_input_grad[0]:
                gx = None
        if ctx.needs_input_grad[3]:
            gw = be.gemm_f32x3(gy.t(), xi.t(), alpha_dev=a_scale, exact_a=GRAD_TERMS, exact_b=True)      # s_a * dL/dy^T . x_int
        gb = gy.sum(0) if ctx.needs_input_grad[4] else None
        return gx, gs, None, gw, gb, None


def linear(x_sim, w_sim, bias, addend=None):
    """F.linear(x_sim, w_sim, bias) (+ addend, the shape of the result) for a BRECQ iteration."""
    lead = x_sim.shape[:-1]
    x2 = x_sim.reshape(-1, x_sim.shape[-1])
    if not (torch.is_grad_enabled() and _usable(x2, w_sim, bias)):
        out = F.linear(x_sim, w_sim, bias)
        return out if addend is None else addend + out
    if addend is not None and ADDEND_FUSED and addend.shape == lead + (w_sim.shape[0],) and addend.dtype == torch.float32:
        a2 = addend.reshape(-1, w_sim.shape[0]).contiguous()
        if a2.data_ptr() % 16 == 0:
            return _LinearFn.apply(x2.contiguous(), w_sim.contiguous(), bias, a2).view(*lead, w_sim.shape[0])
    out = _LinearFn.apply(x2.contiguous(), w_sim.contiguous(), bias).view(*lead, w_sim.shape[0])
    return out if addend is None else addend + out


def quant_linear(x, a_quantizer, w_sim, bias, pre_gelu=False, addend=None):
    """F.linear(a_quantizer(x), w_sim, bias) (+ addend) for a BRECQ iteration; fuses a per-tensor asymmetric uniform activation
    quantiser.  ``pre_gelu``: the quantiser's input is GELU(x), applied by the quantiser itself (AdaLogQuantizer.forward)."""
    from .quantizers.uniform import UniformQuantizer
    if pre_gelu:
        return linear(a_quantizer(x, pre_gelu=True), w_sim, bias, addend)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    aq = a_quantizer
    fused = (INT_ACT and type(aq) is UniformQuantizer and aq.training_mode and not aq.sym and aq.n_bits < 8
             and aq.scale.numel() == 1 and torch.is_grad_enabled() and _usable(x2, w_sim, bias))
    if not fused:
        return linear(aq(x), w_sim, bias, addend)
    out = _QuantLinearFn.apply(x2.contiguous(), aq.scale.view(1), aq.zero_point.view(1), w_sim.contiguous(), bias, aq.n_bits)
    out = out.view(*lead, w_sim.shape[0])
    return out if addend is None else addend + out


class _SplitHeadsFn(torch.autograd.Function):
    """x [B, N, P*H*D] -> the P tensors [B, H, N, D]: q, k, v of an attention block as contiguous tensors in one pass
    (adalog_permute_heads); the gradient is assembled from the P parts' gradients by one pass (adalog_merge_heads).  Autograd's
    own route through reshape / permute / unbind hands strided views to the quantisers (three copies) and stacks + copies the
    three gradients."""

    @staticmethod
    def forward(ctx, x, P, H):
        B, N, C = x.shape
        ctx.dims = (B, N, H, C // (P * H))
        out = backend.get().permute_heads(x, P, H)
        return tuple(out[i] for i in range(P))

    @staticmethod
    def backward(ctx, *gys):
        B, N, H, D = ctx.dims
        return backend.get().merge_heads([None if g_ is None else g_.contiguous() for g_ in gys], B, N, H, D), None, None


def split_heads(x, P, H):
    """x [B, N, P*H*D] -> the P tensors [B, H, N, D]  (x.reshape(B, N, P, H, D).permute(2, 0, 3, 1, 4).unbind(0), reference
    utils/wrap_net.py:21-22).  In a BRECQ iteration on the GPU: one launch each way instead of views + copies."""
    B, N, C = x.shape
    D = C // (P * H)
    if (ENABLED and SPLIT_HEADS and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()
            and D % 4 == 0 and P <= 4 and hasattr(backend.get(), "merge_heads")):
        return _SplitHeadsFn.apply(x.contiguous(), P, H)
    return x.reshape(B, N, P, H, D).permute(2, 0, 3, 1, 4).unbind(0)   # one backward node (a stack) instead of P zero-filled selects


class _QkvQuantFn(torch.autograd.Function):
    """x [B, N, 3*H*D] -> (q_sim, k_sim, v_sim) [B, H, N, D]: the head split fused with the three straight-through uniform
    quantisers that follow it (q.k^T's A and B quantiser, softmax.v's B quantiser): adalog_qkv_split_quant forward,
    adalog_qkv_merge_quant_backward for the gradient of x and of the three scales."""

    @staticmethod
    def forward(ctx, x, s0, z0, s1, z1, s2, z2, H, bits):
        be = backend.get()
        ctx.save_for_backward(x, s0, z0, s1, z1, s2, z2)
        ctx.H, ctx.bits = H, bits
        return be.qkv_split_quant(x, H, [s0, s1, s2], [z0, z1, z2], bits)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        x, s0, z0, s1, z1, s2, z2 = ctx.saved_tensors
        gx, gs = backend.get().qkv_merge_quant_backward([g0, g1, g2], x, ctx.H, [s0, s1, s2], [z0, z1, z2], ctx.bits,
                                                        want_gx=ctx.needs_input_grad[0])
        need = ctx.needs_input_grad
        return gx, (gs[0] if need[1] else None), None, (gs[1] if need[3] else None), None, (gs[2] if need[5] else None), None, None, None


def qkv_split_quant(x, H, mm1, mm2):
    """The fused route of an attention block inside a BRECQ iteration: q, k, v split from x [B, N, 3*H*D] AND passed through the
    input quantisers of the two attention products (mm1.A_quantizer for q, mm1.B_quantizer for k -- elementwise per head, so it
    commutes with the transpose that follows -- and mm2.B_quantizer for v).  -> (q_sim, k_sim, v_sim), or None when the route does
    not apply (the caller then splits and lets the products quantise their inputs)."""
    from .quantizers.uniform import UniformQuantizer
    if not (ENABLED and QKV_FUSED and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()
            and x.dim() == 3 and hasattr(backend.get(), "qkv_split_quant")):
        return None
    D = x.shape[-1] // (3 * H)
    if x.shape[-1] != 3 * H * D or D not in (32, 64):
        return None
    if getattr(mm1, "mode", None) != "quant_forward" or getattr(mm2, "mode", None) != "quant_forward":
        return None
    qs = (getattr(mm1, "A_quantizer", None), getattr(mm1, "B_quantizer", None), getattr(mm2, "B_quantizer", None))
    for q_ in qs:
        if not (type(q_) is UniformQuantizer and q_.training_mode and q_.inited and not q_.sym and q_.n_bits <= 8
                and q_.scale.numel() in (1, H) and q_.zero_point.numel() == q_.scale.numel() and not q_.zero_point.requires_grad):
            return None
    return _QkvQuantFn.apply(x.contiguous(), qs[0].scale, qs[0].zero_point, qs[1].scale, qs[1].zero_point, qs[2].scale,
                             qs[2].zero_point, H, tuple(q_.n_bits for q_ in qs))


class _ScaledSoftmaxFn(torch.autograd.Function):
    """softmax(x * scale, dim=-1): one pass forward, one backward (adalog_scaled_softmax[_backward]) instead of ATen's multiply,
    softmax, softmax backward and the multiply's gradient."""

    @staticmethod
    def forward(ctx, x, scale):
        y = backend.get().scaled_softmax(x, scale)
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return backend.get().scaled_softmax_backward(gy.contiguous(), y, ctx.scale), None


def scaled_softmax(x, scale):
    """(x * scale).softmax(dim=-1)  (reference utils/wrap_net.py:26-27); fused inside a BRECQ iteration on the GPU."""
    if (ENABLED and FUSED_SOFTMAX and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled()
            and x.shape[-1] <= 1024 and hasattr(backend.get(), "scaled_softmax")):
        return _ScaledSoftmaxFn.apply(x.contiguous(), float(scale))
    return (x * scale).softmax(dim=-1)


class _MatmulFn(torch.autograd.Function):
    """A @ B batched over the leading dims (the attention products q.k^T and softmax.v, reference quant_layers/matmul.py:41-44):
    forward and both backward products on adalog_gemm_f32x3, every operand read in place (K-contiguous or K-major).
    heads_last (4-D operands [B, H, R, K] @ [B, H, K, D]): the result is written as [B, R, H, D] storage and returned as its
    [B, H, R, D] view, so the transpose(1, 2).reshape(B, R, H * D) that follows softmax.v in an attention block (reference
    utils/wrap_net.py:31) is a view, and so is the gradient that comes back through it (two-level group strides in the kernel)."""

    @staticmethod
    def forward(ctx, A, B, heads_last):
        ctx.save_for_backward(A, B)
        be = backend.get()
        Bt = B.transpose(-1, -2)
        if heads_last:
            b_, h_, r_, d_ = A.shape[0], A.shape[1], A.shape[2], B.shape[-1]
            out = torch.empty((b_, r_, h_, d_), dtype=torch.float32, device=A.device).permute(0, 2, 1, 3)
            if be.gemm_f32x3_ok(A, Bt, None, out):
                return be.gemm_f32x3(A, Bt, out=out, exact_a=FWD_TERMS, exact_b=FWD_TERMS)
        return be.gemm_f32x3(A, Bt, exact_a=FWD_TERMS, exact_b=FWD_TERMS)

    @staticmethod
    def backward(ctx, gy):
        A, B = ctx.saved_tensors
        be = backend.get()
        At, gyt = A.transpose(-1, -2), gy.transpose(-1, -2)
        if not (be.gemm_f32x3_ok(gy, B) and be.gemm_f32x3_ok(At, gyt)):
            gy = gy.contiguous()
            gyt = gy.transpose(-1, -2)
        gt = GRAD_TERMS
        gA = be.gemm_f32x3(gy, B, exact_a=gt, exact_b=gt) if ctx.needs_input_grad[0] else None   # gy . B^T
        gB = None
        if ctx.needs_input_grad[1]:
            if not B.is_contiguous() and B.transpose(-1, -2).is_contiguous():
                gB = be.gemm_f32x3(gyt, At, exact_a=gt, exact_b=gt).transpose(-1, -2)   # B is a transposed view (k^T): its gradient in the same storage order
            else:
                gB = be.gemm_f32x3(At, gyt, exact_a=gt, exact_b=gt)                          # A^T . gy
        return gA, gB, None


def matmul(A_sim, B_sim, heads_last=False):
    """A_sim @ B_sim for a BRECQ iteration."""
    ok = (ENABLED and torch.is_grad_enabled() and A_sim.is_cuda and A_sim.dtype == torch.float32 and B_sim.dtype == torch.float32
          and A_sim.dim() >= 3 and A_sim.shape[:-2] == B_sim.shape[:-2] and hasattr(backend.get(), "gemm_f32x3"))
    if ok:
        be = backend.get()

        def fits(a_, b_):
            return be.gemm_f32x3_ok(a_, b_.transpose(-1, -2)) and be.gemm_f32x3_ok(a_.transpose(-1, -2), a_.transpose(-1, -2))
        if not fits(A_sim, B_sim):                          # operands the kernel cannot read in place: contiguous copies
            A_sim, B_sim = A_sim.contiguous(), B_sim.contiguous()
        ok = fits(A_sim, B_sim)
    if not ok:
        return A_sim @ B_sim
    return _MatmulFn.apply(A_sim, B_sim, bool(heads_last and HEADS_LAST and A_sim.dim() == 4))
