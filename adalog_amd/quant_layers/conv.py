"""Quantised nn.Conv2d (patch embedding) -- module API of reference quant_layers/conv.py
(MinMaxQuantConv2d -> PTQSLQuantConv2d -> PTQSLBatchingQuantConv2d -> AsymmetricallyBatchingQuantConv2d).

The ViT/DeiT/Swin patch embedding is a non-overlapping convolution (kernel == stride, no padding), i.e. a plain GEMM
over patches: M = N*gh*gw, K = ic*kh*kw, N = oc.  The weight search (conv.py:226-263) therefore reuses the scoring GEMM
with fp32 MFMA (the input stays unquantised: qconv_a_bit = 8, conv.py:55-58 and configs/*.py:12), scoring all 128
weight candidates of every output channel in one launch.  Overlapping / padded convolutions are rejected loudly.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import backend, search, train_mm
from ..ops import BF16, F32, Strided, pad_k
from ..quantizers.uniform import UniformQuantizer

MAX_PACK_BYTES = int(os.environ.get('ADALOG_MAX_PACK_GIB', '8')) << 30
SPLIT3_INPUT = os.environ.get('ADALOG_CONV_SPLIT3', '1') != '0'   # unquantised input as three exact bf16 terms (else fp32 MFMA)


class MinMaxQuantConv2d(nn.Conv2d):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1, groups: int = 1,
                 bias: bool = True, padding_mode: str = 'zeros', mode='raw', w_bit=8, a_bit=8):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.mode = mode
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=True, channel_wise=False)
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=True, channel_wise=False)
        self.raw_input = None
        self.raw_out = None
        self.tmp_input = None
        self.tmp_out = None
        self.calibrated = False

    def _conv(self, x, w, b):
        kh, kw = self.kernel_size
        if (x.is_cuda and tuple(self.stride) == (kh, kw) and tuple(self.padding) == (0, 0) and tuple(self.dilation) == (1, 1)
                and self.groups == 1 and x.shape[-2] % kh == 0 and x.shape[-1] % kw == 0):
            # non-overlapping patches: the convolution IS a GEMM over [N*gh*gw, ic*kh*kw] (rocBLAS; MIOpen has no tuned
            # kernel for 16x16/16 fp32 and falls back to a naive direct convolution, 3 ms per call)
            n, ic, H, W = x.shape
            gh, gw = H // kh, W // kw
            patches = x.reshape(n, ic, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(n * gh * gw, ic * kh * kw)
            w2 = w.reshape(w.shape[0], -1)
            out = train_mm.linear(patches, w2, b) if w2.requires_grad else F.linear(patches, w2, b)
            return out.reshape(n, gh, gw, -1).permute(0, 3, 1, 2)
        return F.conv2d(x, w, b, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, x):
        if self.mode == 'raw':
            return self._conv(x, self.weight, self.bias)
        if self.mode == "quant_forward":
            return self.quant_forward(x)
        if self.mode == 'debug_only_quant_weight':
            return self.debug_only_quant_weight(x)
        if self.mode == 'debug_only_quant_act':
            return self.debug_only_quant_act(x)
        raise NotImplementedError

    def quant_weight_bias(self):
        return self.w_quantizer(self.weight), self.bias

    def quant_input(self, x):
        if self.a_quantizer.n_bits >= 8:               # conv.py:55-58: 8-bit input is left in fp32
            return x
        return self.a_quantizer(x)

    def quant_forward(self, x):
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        w_sim, bias_sim = self.quant_weight_bias()
        return self._conv(self.quant_input(x), w_sim, bias_sim)

    def debug_only_quant_weight(self, x):
        w_sim, bias_sim = self.quant_weight_bias()
        return self._conv(x, w_sim, bias_sim)

    def debug_only_quant_act(self, x):
        return self._conv(self.quant_input(x), self.weight, self.bias)


class PTQSLQuantConv2d(MinMaxQuantConv2d):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1, groups: int = 1,
                 bias: bool = True, padding_mode: str = 'zeros', mode='raw', w_bit=8, a_bit=8, search_round=1, eq_n=100):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode,
                         mode, w_bit, a_bit)
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=True, channel_wise=True)
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=True, channel_wise=False)
        self.search_round = search_round
        self.eq_n = eq_n
        self.parallel_eq_n = eq_n
        self.w_quantizer.scale = nn.Parameter(torch.zeros((self.out_channels, 1)))
        self.a_quantizer.scale = nn.Parameter(torch.zeros((1, 1, 1, 1)))

    def quant_weight_bias(self):
        oc, ic, kw, kh = self.weight.data.shape
        w_sim = self.w_quantizer(self.weight.view(oc, ic * kw * kh)).view(oc, ic, kw, kh)
        return w_sim, self.bias


class PTQSLBatchingQuantConv2d(PTQSLQuantConv2d):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1, groups: int = 1,
                 bias: bool = True, padding_mode: str = 'zeros', mode='raw', w_bit=8, a_bit=8, calib_batch_size=32,
                 search_round=1, eq_n=100):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode,
                         mode, w_bit, a_bit, search_round, eq_n)
        self.calib_batch_size = calib_batch_size

    def _initialize_calib_parameters(self):
        self.calib_size = self.raw_input.shape[0]
        self.parallel_eq_n = self.eq_n


class AsymmetricallyBatchingQuantConv2d(PTQSLBatchingQuantConv2d):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1, groups: int = 1,
                 bias: bool = True, padding_mode: str = 'zeros', mode='raw', w_bit=8, a_bit=8, calib_batch_size=32,
                 search_round=1, eq_n=100, fpcs=False, steps=4):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode,
                         mode, w_bit, a_bit, calib_batch_size, search_round, eq_n)
        self.fpcs = fpcs
        self.steps = steps
        del self.w_quantizer
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=False, channel_wise=True)
        self.w_quantizer.scale = nn.Parameter(torch.zeros((self.out_channels, 1)))
        self.w_quantizer.zero_point = nn.Parameter(torch.zeros((self.out_channels, 1)))

    # ------------------------------------------------------------------ patch-GEMM views
    def _check_patch_conv(self):
        k, s = tuple(self.kernel_size), tuple(self.stride)
        pad = self.padding if isinstance(self.padding, tuple) else (self.padding, self.padding)
        if k != s or tuple(pad) != (0, 0) or tuple(self.dilation) != (1, 1) or self.groups != 1:
            raise NotImplementedError("only non-overlapping patch-embedding convolutions (kernel == stride, no padding, "
                                      "groups = 1) are on the accelerated calibration path")

    def _patches(self, x):
        """[N, ic, H, W] -> [N*gh*gw, ic*kh*kw] (im2col of a non-overlapping conv is a pure permutation)."""
        N, ic, Hh, Ww = x.shape
        kh, kw = self.kernel_size
        gh, gw = Hh // kh, Ww // kw
        p = x[:, :, :gh * kh, :gw * kw].reshape(N, ic, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5)
        return p.reshape(N * gh * gw, ic * kh * kw), gh, gw

    def _w2(self):
        return self.weight.data.view(self.out_channels, -1)

    def _pack_x(self, patches):
        """The unquantised input of the weight search (conv.py:55-58 with qconv_a_bit = 8) as a GEMM operand: three exact
        bf16 terms per value (the candidate weights, exact small integers, are then repeated three times along K and the
        products run on the bf16 MFMA at 5x the fp32 rate), or a zero-padded fp32 copy (ADALOG_CONV_SPLIT3=0)."""
        be = backend.get()
        if SPLIT3_INPUT:
            return be.pack_split3(patches.unsqueeze(0))
        return be.pack_raw(patches.unsqueeze(0))

    def _score_w(self, xp, ref, M, fmap, scale, zp, defer=False):
        """conv.py:226-255 -> scores [P, oc] = -sum_images mean_{fw,fh} (raw_out - conv(x, fq_p(W)) - b)^2."""
        be = backend.get()
        oc = self.out_channels
        K = self._w2().shape[1]
        P = scale.shape[0]
        split3 = xp.dtype == torch.bfloat16
        dt = BF16 if split3 else F32
        chunk = max(1, min(P, MAX_PACK_BYTES // max(1, oc * pad_k(K, F32) * (6 if split3 else 4))))
        ones = search.const_tensor([1.0], xp.device)
        bias = None if self.bias is None else Strided(self.bias.data, n=1)
        out = []
        for s in range(0, P, chunk):
            e = min(P, s + chunk)
            sc, zc = scale[s:e].contiguous(), zp[s:e].contiguous()
            wp = be.pack_uniform(self._w2().unsqueeze(0), sc, zc, e - s, oc, 1, 0, 1, self.w_quantizer.n_bits, dt,
                                 c_inner=True)
            if split3:
                wp = wp.repeat(1, 1, 1, 3)              # [hi | mid | lo] of x each meet the same integers
            out.append(be.gemm_score(dt, xp, wp, M, oc, e - s, 1, 1, ref, Strided(ones), Strided(sc, c=oc, n=1), bias,
                                     False, True, 1.0 / fmap, ref_div=e - s, order=2, ref_transposed=True,
                                     defer=defer and chunk >= P))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def weight_fpcs(self, fpcs_width=16, steps=4):
        """conv.py:292-311."""
        be = backend.get()
        self._check_patch_conv()
        patches, gh, gw = self._patches(self.raw_input)
        M = patches.shape[0]
        xp = self._pack_x(patches)
        ref = self.raw_out.permute(1, 0, 2, 3).reshape(1, self.out_channels, M).contiguous()      # [1, oc, tokens]
        scale, zp, delta = search.weight_grid(self._w2(), self.w_quantizer.n_bits, self.eq_n, conv=True)
        fn = lambda s, z, t: self._score_w(xp, ref, M, gh * gw, s, z, defer=True)
        wq = self.w_quantizer
        res = search.fpcs(scale, zp, None, delta, fn, steps, fpcs_width, self.eq_n, None,
                          commit_to=search.commit_targets(wq.scale, wq.zero_point, None))
        if res is not None:
            search.commit_param(wq.scale, res[0])               # (no copy when the search's last kernel wrote them in place)
            search.commit_param(wq.zero_point, res[1])

    def hyperparameter_searching(self):
        """conv.py:313-334 for the shipped configuration (qconv_a_bit = 8: the input is not quantised and the loop
        breaks after the first weight FPCS, conv.py:328-331)."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        if not self.fpcs:
            raise NotImplementedError("non-FPCS search is not part of the accelerated path")
        if self.a_quantizer.n_bits < 8:
            raise NotImplementedError("input quantisation of the patch embedding (<8 bit) is dead code in the reference "
                                      "(conv.py:267,329 reference undefined names) and is not implemented")
        self._initialize_calib_parameters()
        scale, zp, _ = search.weight_grid(self._w2(), self.w_quantizer.n_bits, self.eq_n, conv=True)
        self.w_quantizer.scale.data.copy_(scale[-2].view(-1, 1))           # conv.py:319-320
        self.w_quantizer.zero_point.data.copy_(zp[-2].view(-1, 1))
        self.w_quantizer.inited = True
        self.weight_fpcs(steps=self.steps)
        self.calibrated = True
        search.forget_grids()
        del self.raw_input, self.raw_out
        return None
