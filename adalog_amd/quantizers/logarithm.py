"""AdaLogQuantizer / ShiftAdaLogQuantizer -- API of reference quantizers/logarithm.py:68-102,127-135.

q (int64 buffer), table1/table2 (fp32 [2L] buffers), scale (Parameter attached by the layer), shift (Parameter) and
bias_reparamed (bool buffer) keep the reference's names and shapes so state_dicts interchange.  forward() is one HIP
kernel (adalog_log_fake_quant_f32, csrc/fakequant.hip).  The plain log2 / log-sqrt2 ablation quantisers of the
reference (logarithm.py:8-65,105-124) are out of scope.
"""
import math

import torch
import torch.nn as nn

from .. import backend


class _AdaLogSTE(torch.autograd.Function):
    """Training form (logarithm.py:88-92): y = 2^(-k*q/37) * s * [k<2L], k = round_ste(-log2(u) * 37/q).

    With STE on the rounding, y ~= u*s inside the representable range, so
        dy/dx = [u in (1e-15,1)] * [0 <= k <= 2L-1 unclamped] ;  dy/ds = y/s - dy/dx * (x+shift)/s
    """

    @staticmethod
    def forward(ctx, x, scale, q, n_bits, shift, sub_shift, pre_gelu=False):
        """``pre_gelu``: the quantiser's input is GELU(x) -- x is fc1's output; the activation function and its derivative run inside
        the two kernels (no GELU pass, no stored GELU output: utils/models.py Mlp.forward inside a BRECQ iteration)."""
        be = backend.get()
        kw = {"pre_gelu": True} if pre_gelu else {}
        y = be.log_fake_quant(x, scale, q, None, None, n_bits, shift=shift, sub_shift=sub_shift, train_form=True, **kw)
        ctx.save_for_backward(x, scale, q, shift, y)
        ctx.n_bits, ctx.sub_shift, ctx.pre_gelu = n_bits, sub_shift, bool(pre_gelu)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, q, shift, y = ctx.saved_tensors
        be = backend.get()
        kw = {"pre_gelu": True} if ctx.pre_gelu else {}
        gx, gs = be.log_fake_quant_backward(gy, x, y, scale, q, ctx.n_bits, shift, ctx.sub_shift, **kw)
        return gx, gs, None, None, None, None, None


class AdaLogQuantizer(nn.Module):
    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False):
        super().__init__()
        self.sym = symmetric
        self.n_bits = n_bits
        self.n_levels = 2 ** (self.n_bits - 1)
        self.inited = False
        self.drop_prob = 1.0
        self.channel_wise = channel_wise
        self.training_mode = False
        self.r = 37.0
        self.register_buffer('q', torch.tensor([int(self.r)]))
        self.register_buffer('table1', torch.zeros((self.n_levels * 2)))
        self.register_buffer('table2', torch.zeros((self.n_levels * 2)))
        self.update_table()

    def init_training(self):
        self.training_mode = True

    def end_training(self):
        self.training_mode = False

    @staticmethod
    def make_tables(q: int, n_levels: int, r: float = 37.0):
        """logarithm.py:77-81: table1[i] = floor(i*q/r), table2[i] = round(2^(-((q*i) mod r)/r) * (4L-2)) / (4L-2),
        evaluated in Python float64 and stored as fp32 (one host->device copy instead of 2L .item() syncs)."""
        t1 = [math.floor(i * q / r) for i in range(2 * n_levels)]
        t2 = [round((2 ** (-((q * i) % r) / r)) * (4 * n_levels - 2)) / (4 * n_levels - 2) for i in range(2 * n_levels)]
        return torch.tensor(t1, dtype=torch.float32), torch.tensor(t2, dtype=torch.float32)

    def update_table(self, q: int = None):
        """Rebuild the LUTs for the current (or given) q.  Passing q avoids the device->host read of self.q."""
        qi = int(self.q.item()) if q is None else int(q)
        t1, t2 = self.make_tables(qi, self.n_levels, self.r)
        self.table1.data.copy_(t1)
        self.table2.data.copy_(t2)

    def _shift_args(self):
        return None, False

    def forward(self, x, pre_gelu=False):
        """``pre_gelu`` (training form on the GPU only; callers ask ``fused_gelu_ok`` first): quantise GELU(x)."""
        if self.n_bits == 32:
            return torch.nn.functional.gelu(x) if pre_gelu else x
        assert self.inited
        shift, sub = self._shift_args()
        if self.training_mode and torch.is_grad_enabled():
            if pre_gelu and not self.fused_gelu_ok(x):
                x, pre_gelu = torch.nn.functional.gelu(x), False
            return _AdaLogSTE.apply(x, self.scale, self.q, self.n_bits, shift, sub, pre_gelu)
        if pre_gelu:
            x = torch.nn.functional.gelu(x)
        return backend.get().log_fake_quant(x, self.scale.data, self.q, self.table1, self.table2, self.n_bits,
                                            shift=None if shift is None else shift.data, sub_shift=sub,
                                            train_form=self.training_mode)

    def fused_gelu_ok(self, x):
        be = backend.get()
        return bool(x.is_cuda and x.dtype == torch.float32 and getattr(be, "QF_EXTRAS", False) and self.n_bits != 32)

    def bins(self, x):
        """Integer bin index k (uint8; 255 marks the masked 'below the last bin' code)."""
        shift, sub = self._shift_args()
        return backend.get().log_fake_quant(x, self.scale.data, self.q, self.table1, self.table2, self.n_bits,
                                            shift=None if shift is None else shift.data, sub_shift=sub,
                                            want_bins=True, want_y=False)[1]

    def __repr__(self):
        return (f'{self.__class__.__name__}(n_bits={self.n_bits}, sym={self.sym}, '
                f'channel_wise={self.channel_wise}, q={self.q.item()})')


class ShiftAdaLogQuantizer(AdaLogQuantizer):
    def __init__(self, n_bits: int = 8, symmetric: bool = False, channel_wise: bool = False):
        super().__init__(n_bits, symmetric, channel_wise)
        self.shift = nn.Parameter(torch.zeros((1)))
        self.register_buffer('bias_reparamed', torch.tensor(False))
        self._reparamed_host = None              # host mirror of the flag: no device read per forward

    def _shift_args(self):
        if self._reparamed_host is None:
            self._reparamed_host = bool(self.bias_reparamed.item())
        return self.shift, (not self._reparamed_host)

    def mark_bias_reparamed(self):
        self.bias_reparamed.data.copy_(torch.tensor(True))
        self._reparamed_host = True

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._reparamed_host = None
