"""UniformQuantizer -- same constructor, attributes and state_dict keys as reference quantizers/uniform.py:8-39;
the forward pass is one HIP kernel (adalog_uniform_fake_quant_f32, csrc/fakequant.hip) instead of five ATen passes.

``scale`` / ``zero_point`` are attached by the owning layer as nn.Parameters exactly like the reference does
(linear.py:258-263, matmul.py:129-133, conv.py:222-224), so checkpoints interchange.
"""
import torch
import torch.nn as nn

from .. import backend
from ._ste import round_ste


class _UniformSTE(torch.autograd.Function):
    """Training form (uniform.py:29 with round_ste): HIP forward, straight-through backward.

    Gradients (asymmetric): with q = clamp(rne(x/s) + z, 0, 2L-1) and inside = [0 <= rne(x/s)+z <= 2L-1]
        dy/dx = inside;   dy/ds = (q - z) - (x/s) * inside;   dy/dz = 0 if inside else -s
    """

    @staticmethod
    def forward(ctx, x, scale, zero_point, n_bits, sym):
        be = backend.get()
        # A transposed view of a contiguous tensor (k^T of an attention block) is quantised in its storage order and handed back
        # as the same view: the quantiser is elementwise and its parameters do not vary along the last two dims.
        ctx.swap = (x.dim() >= 2 and not x.is_contiguous() and x.transpose(-1, -2).is_contiguous()
                    and all(t is None or t.numel() == 1 or (t.dim() == x.dim() and t.shape[-1] == 1 and t.shape[-2] == 1)
                            for t in (scale, None if sym else zero_point)))
        if ctx.swap:
            x = x.transpose(-1, -2)
        x = x.contiguous()                               # saved in the layout the backward kernel reads (one copy, not two)
        if sym:
            y = be.uniform_fake_quant(x, scale, None, n_bits, sym=True)
            ctx.save_for_backward(x, scale, None)
        else:
            y = be.uniform_fake_quant(x, scale, zero_point, n_bits, sym=False)
            ctx.save_for_backward(x, scale, zero_point)
        ctx.n_bits, ctx.sym = n_bits, sym
        return y.transpose(-1, -2) if ctx.swap else y

    @staticmethod
    def backward(ctx, gy):
        x, scale, zp = ctx.saved_tensors
        be = backend.get()
        if ctx.swap:
            gy = gy.transpose(-1, -2)
        gx, gs, gz = be.uniform_fake_quant_backward(gy, x, scale, zp, ctx.n_bits, ctx.sym,
                                                    ctx.needs_input_grad[1], ctx.needs_input_grad[2] and zp is not None)
        return (gx.transpose(-1, -2) if ctx.swap else gx), gs, gz, None, None


class UniformQuantizer(nn.Module):
    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False):
        super().__init__()
        self.sym = symmetric
        self.n_bits = n_bits
        self.n_levels = 2 ** (self.n_bits - 1)
        self.channel_wise = channel_wise
        self.drop_prob = 1.0
        self.inited = False
        self.training_mode = False
        self.use_clip_forward = False

    def init_training(self):
        self.training_mode = True

    def end_training(self):
        self.training_mode = False

    def forward(self, x):
        if self.n_bits == 32:
            return x
        assert self.inited
        if self.training_mode and torch.is_grad_enabled():
            return _UniformSTE.apply(x, self.scale, None if self.sym else self.zero_point, self.n_bits, self.sym)
        return backend.get().uniform_fake_quant(x, self.scale.data, None if self.sym else self.zero_point.data,
                                                self.n_bits, sym=self.sym)

    def bins(self, x):
        """Integer bin indices (uint8) of the asymmetric form -- what an integer deployment would store."""
        assert self.inited and not self.sym
        return backend.get().uniform_fake_quant(x, self.scale.data, self.zero_point.data, self.n_bits, sym=False,
                                                want_bins=True, want_y=False)[1]

    def __repr__(self):
        return f'{self.__class__.__name__}(n_bits={self.n_bits}, sym={self.sym}, channel_wise={self.channel_wise})'
