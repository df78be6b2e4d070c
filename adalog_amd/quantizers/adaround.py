"""AdaRoundQuantizer -- API of reference quantizers/adaround.py:7-76 (learned hard-sigmoid rounding, arXiv 2004.10568),
used by BRECQ block reconstruction.  Forward, d/d alpha and the rounding regulariser run as HIP kernels
(csrc/brecq.hip); alpha is the only trained tensor (block_recon.py:97-108 optimises alpha and activation scales).
"""
import torch
from torch import nn

from .. import backend, train_mm
from .uniform import UniformQuantizer


class AlphaCollector:
    """A captured BRECQ iteration on one GPU (utils/block_recon.py): instead of materialising d/d alpha per layer (one launch
    each), the regulariser's gradients, their product with the upstream factor, four accumulations and the optimiser's launch,
    the backward nodes hand their inputs to this collector and ``flush`` issues ONE launch (adalog_alpha_step_multi) that
    computes the gradient of every alpha and takes its Adam step.  Active only while ``COLLECT`` is set (the graph capture)."""

    def __init__(self, quantizers, state_of, step_dev, lr, betas, eps):
        self.q = list(quantizers)                              # the block's AdaRoundQuantizers, in the regulariser's order
        self.ptr = {q_.alpha.data_ptr(): i for i, q_ in enumerate(self.q)}
        self.state_of, self.step_dev, self.lr, self.betas, self.eps = state_of, step_dev, lr, betas, eps
        self.begin()

    def begin(self):
        self.gw = [None] * len(self.q)
        self.w = [None] * len(self.q)
        self.round = None                                      # (b, weight, upstream gradient, gate) of the regulariser

    def take(self, alpha2, w2, gy):
        i = self.ptr.get(alpha2.data_ptr())
        if i is None or not self.q[i].soft_targets:
            return False
        self.gw[i], self.w[i] = gy.contiguous(), w2
        return True

    def flush(self):
        be = backend.get()
        qs = self.q
        ws = [w_ if w_ is not None else q_._w_last for w_, q_ in zip(self.w, qs)]
        sts = [self.state_of(q_.alpha) for q_ in qs]
        b, weight, g, gate = self.round if self.round is not None else (1.0, 0.0, None, None)
        be.alpha_step_multi([q_.alpha.data for q_ in qs], ws, self.gw, [q_.scale.data.view(-1) for q_ in qs],
                            [q_.zero_point.data.view(-1) for q_ in qs], [st['exp_avg'] for st in sts], [st['exp_avg_sq'] for st in sts],
                            [q_.alpha.numel() // q_.scale.numel() for q_ in qs], [q_.n_bits for q_ in qs], self.step_dev, self.lr,
                            self.betas[0], self.betas[1], self.eps, b, weight, g, gate)


COLLECT = None                                                 # the AlphaCollector in force (set around a graph capture only)


class _AdaRoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w2, alpha2, scale, zero_point, n_bits, soft, want_t=False):
        ctx.save_for_backward(w2, alpha2, scale, zero_point)
        ctx.n_bits, ctx.soft = n_bits, soft
        be = backend.get()
        # ``want_t`` is decided by the caller: autograd runs Function.forward with grad mode OFF, so torch.is_grad_enabled()
        # read in here is always False (round 5 read it here and the K-major image was never written)
        if want_t and w2.is_cuda and hasattr(be, "adaround_t"):
            # a BRECQ iteration: the same launch leaves the K-major image the layer's forward product reads (train_mm._kmajor)
            y, y_t = be.adaround_t(w2, alpha2, scale, zero_point, n_bits, soft)
            train_mm.offer_kmajor(y, y_t)
            return y
        return be.adaround(w2, alpha2, scale, zero_point, n_bits, soft)

    @staticmethod
    def backward(ctx, gy):
        w2, alpha2, scale, zero_point = ctx.saved_tensors
        if COLLECT is not None and ctx.soft and COLLECT.take(alpha2, w2, gy):
            return None, None, None, None, None, None, None    # the collector's launch computes and applies d/d alpha
        ga = backend.get().adaround(w2, alpha2, scale, zero_point, ctx.n_bits, ctx.soft, gy=gy.contiguous())
        return None, ga, None, None, None, None, None


class _RoundLossFn(torch.autograd.Function):
    """sum(1 - |2h(alpha)-1|^b), block_recon.py:209-210; value and gradient in one fused kernel each."""

    @staticmethod
    def forward(ctx, alpha, b):
        ctx.save_for_backward(alpha)
        ctx.b = b
        return backend.get().round_loss(alpha, b).view(())

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        ga = torch.empty_like(alpha)
        backend.get().round_loss(alpha, ctx.b, galpha=ga, gscale=1.0, want_loss=False, gmul=g.reshape(1), overwrite=True)
        return ga, None


class AdaRoundQuantizer(nn.Module):
    def __init__(self, uq: UniformQuantizer, weight_tensor: torch.Tensor, round_mode='learned_hard_sigmoid'):
        super().__init__()
        self.n_bits = uq.n_bits
        self.n_levels = uq.n_levels
        self.channel_wise = uq.channel_wise
        self.sym = uq.sym
        self.scale = nn.Parameter(uq.scale)
        self.zero_point = nn.Parameter(uq.zero_point)
        self.round_mode = round_mode
        self.alpha = None
        self.soft_targets = False
        self.inited = True
        self.training_mode = False
        # params for sigmoid function
        self.gamma, self.zeta = -0.1, 1.1
        self.beta = 2 / 3
        self.init_alpha(x=weight_tensor.clone())

    def init_training(self):
        self.training_mode = True

    def end_training(self):
        self.training_mode = False

    def _rows(self, x):
        """View x as [rows, inner] matching the per-row scale ([n_V, rows, 1] or [oc, 1])."""
        rows = self.scale.numel()
        return x.reshape(rows, -1)

    def forward(self, x):
        if self.sym:
            raise NotImplementedError("symmetric AdaRound is not used by the shipped configs")
        if self.round_mode in ('nearest', 'nearest_ste'):
            # after reconstruct_model(): weights already hold the hard values; plain round-to-nearest (adaround.py:39-42)
            y = backend.get().uniform_fake_quant(x, self.scale.data, self.zero_point.data, self.n_bits, sym=False)
            return y
        if self.round_mode != 'learned_hard_sigmoid':
            raise ValueError('Wrong rounding mode')
        shp = x.shape
        self._w_last = self._rows(x).detach()                   # (the collector's launch reads the weights of a layer without gradient)
        # the K-major image of w_sim is written only where a BRECQ iteration will consume it (train_mm._kmajor): under no_grad
        # -- evaluation after BRECQ -- or with the training contractions switched off it would be written for nobody
        want_t = bool(train_mm.W_KMAJOR and train_mm.ENABLED and torch.is_grad_enabled() and self.alpha.requires_grad)
        y = _AdaRoundFn.apply(self._w_last if not x.requires_grad else self._rows(x), self._rows(self.alpha), self.scale.view(-1),
                              self.zero_point.view(-1), self.n_bits, bool(self.soft_targets), want_t)
        return y.view(shp)

    def get_soft_targets(self):
        return torch.clamp(torch.sigmoid(self.alpha) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def round_loss(self, b):
        """sum(1 - |2h-1|^b) over this quantiser's weights (block_recon.py:209-210), fused.  ``b`` may be a one-element
        device tensor (captured BRECQ iterations read the exponent on the device)."""
        return _RoundLossFn.apply(self.alpha, b if torch.is_tensor(b) else float(b))

    def init_alpha(self, x: torch.Tensor):
        """adaround.py:62-69: alpha such that h(alpha) equals the fractional part of w/s."""
        x_floor = torch.floor(x / self.scale)
        if self.round_mode == 'learned_hard_sigmoid':
            rest = (x / self.scale) - x_floor
            alpha = -torch.log((self.zeta - self.gamma) / (rest - self.gamma) - 1)
            self.alpha = nn.Parameter(alpha)
        else:
            raise NotImplementedError

    def get_hard_value(self, x):
        init_shape = x.shape
        return ((torch.floor(x.reshape_as(self.alpha) / self.scale) + (self.alpha >= 0).float()) * self.scale).reshape(
            *init_shape)

    def __repr__(self):
        return (f'{self.__class__.__name__}(n_bits={self.n_bits}, sym={self.sym}, channel_wise={self.channel_wise}, '
                f'round_mode={self.round_mode})')
