"""Quantised nn.Linear family -- module API of reference quant_layers/linear.py (class names, constructor signatures,
attributes, state_dict keys, mode strings), with every search and forward pass running as HIP kernels on the MI355X.

Class chain kept for isinstance dispatch by the calibrator / block reconstructor (calibrator.py:40-53):
  MinMaxQuantLinear -> PTQSLQuantLinear -> PTQSLBatchingQuantLinear -> AsymmetricallyBatchingQuantLinear
      -> AsymmetricallyChannelWiseBatchingQuantLinear   (qkv / fc1 / reduction: per-channel search + LayerNorm fold)
      -> PostGeluLogBasedBatchingQuantLinear            (fc2: shifted AdaLog activation, joint (scale, base) search)

What differs from the reference, by design:
  * captured calibration tensors (raw_input / raw_out) stay in HBM for the whole search (no .cpu()/.cuda() round
    trips, no memory-derived candidate chunking: linear.py:111-121 has no counterpart);
  * a scoring call = pack operands once (int8 / bf16) + ONE MFMA GEMM with a fused squared-error epilogue;
  * the FPCS loop is sync-free; winners are committed on the device;
  * images may be sharded over ranks (adalog_amd.parallel): raw_input/raw_out then hold the local shard.
Out of scope: the symmetric PTQSL search (dead code in the reference, linear.py:171) and
PostGeluTwinUniformBatchingQuantLinear (ptq4vit ablation).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import backend, search, train_mm
from ..ops import BF16, BF16_FP8, FP8, I8, Strided, pad_k  # noqa: F401  (dtype codes / epilogue parameter helper; pure metadata)
from ..quantizers.logarithm import ShiftAdaLogQuantizer
from ..quantizers.uniform import UniformQuantizer

GELU_SHIFT = 0.16997124254703522      # -min(gelu(x)), reference linear.py:749
GELU_SHIFT32 = float(torch.tensor(GELU_SHIFT, dtype=torch.float32))     # the Parameter's fp32 value
import os as _os
FUSED_ACT_SEARCH = _os.environ.get("ADALOG_FUSED_ACT", "1") != "0"      # A/B switch: 0 = pack + streaming GEMM (round 1)
# candidates are scored in chunks when a packed operand would exceed this (8 GiB of 288: vit_large's fc2 activation search --
# 51.6 MB per candidate -- stays whole; a chunk of 124 is not a candidate count the streaming kernel takes)
MAX_PACK_BYTES = int(os.environ.get('ADALOG_MAX_PACK_GIB', '8')) << 30
# searches whose launches run on the slab kernel store their <= 4-bit operands as fp8 (exact, search.int_operand_dtype): no
# int -> float conversion per output (K <= 384: 256-column slabs; K <= 768: 128-column slabs).  Weight searches gain 10-20 %,
# activation searches (row scale in the epilogue) 2-4 % -- in round 1 the latter lost 4 %, the compiler spilled there
FP8_WEIGHT_SEARCH_MAX_K = int(os.environ.get('ADALOG_FP8_WEIGHT_MAX_K', '768'))
FIXED_GRID_ONLY = os.environ.get('ADALOG_FIXED_GRID_ONLY', '1') != '0'   # fp8 decided by the fixed operand's quantiser alone (_int_dt)
# self-MSE searches scored from the sorted tensor (csrc/sorted_score.hip); 0 = one pass over the tensor per step (round 1/2)
# uniform activation searches: candidate operand generated in the slab kernel (1) or packed to HBM first (0, rounds 1-2)
GEN_ACT_SEARCH = os.environ.get('ADALOG_GEN_ACT', '1') != '0'
# uniform weight searches on the slab kernel: candidate operand generated in the kernel from the fp32 weight rows (1) or packed (0)
GEN_W_SEARCH = os.environ.get('ADALOG_GEN_W', '1') != '0'
# output-MSE weight searches against a per-tensor uniform activation quantiser: scored from the Gram matrix of the quantised
# activation (csrc/gram.hip: built once per weight_fpcs call, K^2 instead of tokens * K multiply-adds per candidate row) where
# backend.gram_ok says it pays; ADALOG_GRAM_W=0 keeps every such search on the token-form kernels
SORTED_SELF_SEARCH = os.environ.get('ADALOG_SORTED_SELF', '1') != '0'
RUN_DEAD_W_SELF = os.environ.get('ADALOG_DEAD_W_SELF', '0') == '1'
MIXED_W_SEARCH = os.environ.get('ADALOG_MIXED_W', '1') != '0'     # bf16 activations x fp8 weight candidates (wide streaming kernel)


class MinMaxQuantLinear(nn.Linear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8):
        super().__init__(in_features, out_features, bias)
        self.mode = mode
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=True, channel_wise=False)
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=True, channel_wise=False)
        self.raw_input = None
        self.raw_out = None
        self.tmp_input = None
        self.tmp_out = None
        self.calibrated = False

    def forward(self, x):
        if self.mode == 'raw':
            return F.linear(x, self.weight, self.bias)
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
        return self.a_quantizer(x)

    def quant_forward(self, x, pre_gelu=False, addend=None):
        """``pre_gelu``: the layer's input is GELU(x) -- handed to an activation quantiser that applies it in its own kernels
        (AdaLogQuantizer.forward), else applied here.  ``addend``: added to the result (inside the product where that is free)."""
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        if pre_gelu and not hasattr(self.a_quantizer, "fused_gelu_ok"):
            x, pre_gelu = F.gelu(x), False
        w_sim, bias_sim = self.quant_weight_bias()
        if w_sim.requires_grad:                          # a BRECQ iteration: the contractions run on csrc/brecq_gemm.hip
            return train_mm.quant_linear(x, self.a_quantizer, w_sim, bias_sim, pre_gelu=pre_gelu, addend=addend)
        out = F.linear(self.a_quantizer(x, pre_gelu=True) if pre_gelu else self.quant_input(x), w_sim, bias_sim)
        return out if addend is None else addend + out

    def debug_only_quant_weight(self, x):
        w_sim, bias_sim = self.quant_weight_bias()
        if w_sim.requires_grad:
            return train_mm.linear(x, w_sim, bias_sim)
        return F.linear(x, w_sim, bias_sim)

    def debug_only_quant_act(self, x):
        return F.linear(self.quant_input(x), self.weight, self.bias)


class PTQSLQuantLinear(MinMaxQuantLinear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 search_round=1, eq_n=100, n_V=1):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit)
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=True, channel_wise=True)
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=True, channel_wise=False)
        self.search_round = search_round
        self.eq_n = eq_n
        self.parallel_eq_n = eq_n
        self.n_V = n_V
        self.crb_rows = out_features // n_V
        self.w_quantizer.scale = nn.Parameter(torch.zeros((n_V, self.crb_rows, 1)))
        self.a_quantizer.scale = nn.Parameter(torch.zeros((1)))

    def quant_weight_bias(self):
        w_sim = self.w_quantizer(self.weight.view(self.n_V, self.crb_rows, self.in_features)).view(
            self.out_features, self.in_features)
        return w_sim, self.bias


class PTQSLBatchingQuantLinear(PTQSLQuantLinear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 calib_batch_size=32, search_round=1, eq_n=100, n_V=1):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         search_round=search_round, eq_n=eq_n, n_V=n_V)
        self.calib_batch_size = calib_batch_size

    def _initialize_calib_parameters(self):
        """The reference sizes candidate chunks from free GPU memory (linear.py:111-121); with the operands packed to
        int8/bf16 and the error reduced in registers there is nothing to size -- only the shard's image count."""
        self.calib_size = self.raw_input.shape[0]
        self.parallel_eq_n = self.eq_n


class AsymmetricallyBatchingQuantLinear(PTQSLBatchingQuantLinear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 calib_batch_size=32, search_round=1, eq_n=100, n_V=1, fpcs=False, steps=4):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         calib_batch_size=calib_batch_size, search_round=search_round, eq_n=eq_n, n_V=n_V)
        self.fpcs = fpcs
        self.steps = steps
        del self.a_quantizer, self.w_quantizer
        self.w_quantizer = UniformQuantizer(n_bits=w_bit, symmetric=False, channel_wise=True)
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=False, channel_wise=False)
        self.a_quantizer.scale = nn.Parameter(torch.zeros((1)))
        self.a_quantizer.zero_point = nn.Parameter(torch.zeros((1)))
        self.w_quantizer.scale = nn.Parameter(torch.zeros((n_V, self.crb_rows, 1)))
        self.w_quantizer.zero_point = nn.Parameter(torch.zeros((n_V, self.crb_rows, 1)))

    # ------------------------------------------------------------------ min/max initialisation (linear.py:265-294)
    def _initialize_weight_scale(self):
        self.w_quantizer._zp_on_grid = False         # min/max initialisation: the zero point may fall outside [0, 2^bits - 1]
        be = backend.get()
        mn, mx = be.minmax_rows(self.weight.data.view(self.out_features, self.in_features))
        L2 = 2 * self.w_quantizer.n_levels - 1
        scale = ((mx - mn) / L2).view(self.n_V, self.crb_rows, 1)
        self.w_quantizer.scale.data.copy_(scale)
        self.w_quantizer.zero_point.data.copy_(-mn.view(self.n_V, self.crb_rows, 1) / scale)
        self.w_quantizer.inited = True
        self.invalidate_packed_weight()

    def _initialize_activation_scale(self):
        self.a_quantizer._zp_on_grid = False
        from .. import parallel
        be = backend.get()
        x2 = self.raw_input.reshape(-1, self.in_features)
        amn, amx = be.absminmax(x2, per_channel=self.a_quantizer.channel_wise)     # note: of |x| (linear.py:283-287)
        amx, amn = parallel.all_reduce_max(amx), parallel.all_reduce_min(amn)
        scale = (amx - amn) / (2 * self.a_quantizer.n_levels - 1)
        self.a_quantizer.scale.data.copy_(scale.view(self.a_quantizer.scale.shape))
        self.a_quantizer.zero_point.data.copy_((-amn / scale).view(self.a_quantizer.zero_point.shape))
        self.a_quantizer.inited = True

    # ------------------------------------------------------------------ helpers
    def _tokens_per_image(self):
        x = self.raw_input
        return x.numel() // (x.shape[0] * x.shape[-1])

    def _x2(self):
        return self.raw_input.reshape(-1, self.in_features)

    def _ref2(self):
        return self.raw_out.reshape(1, -1, self.out_features)

    def _ref2_t(self):
        """raw_out as [1, O, tokens] (one transposing copy per layer): the weight searches walk the reference along the
        token axis, which the scoring kernel stages through LDS when that axis is contiguous."""
        key = self.raw_out.data_ptr()
        if getattr(self, "_ref_t_key", None) != key:
            self._ref_t = self.raw_out.reshape(-1, self.out_features).t().contiguous().unsqueeze(0)
            self._ref_t_key = key
        return self._ref_t

    def _w2(self):
        return self.weight.data.view(self.out_features, self.in_features)

    def invalidate_packed_weight(self):
        """Drop quant_forward's cached packed weight image (_pack_w_cached).  Called wherever the weight or its quantiser is
        written through ``.data`` / a raw pointer -- writes that do not bump the Parameter's ``_version``, which the cache key
        cannot see: _commit_w, the min/max initialisation, reparam(), BRECQ's hard-rounding commit."""
        self.__dict__.pop("_wp_cache", None)

    def _commit_w(self, scale, zp):
        search.commit_param(self.w_quantizer.scale, scale)              # (no copy when the search's last kernel wrote them in place)
        search.commit_param(self.w_quantizer.zero_point, zp)
        self.w_quantizer.inited = True
        self.w_quantizer._zp_on_grid = True          # zero point taken from an FPCS grid: inside [0, 2^bits - 1]
        self.invalidate_packed_weight()

    def _commit_a(self, scale, zp):
        search.commit_param(self.a_quantizer.scale, scale)
        search.commit_param(self.a_quantizer.zero_point, zp)
        self.a_quantizer.inited = True
        self.a_quantizer._zp_on_grid = True

    def _int_dt(self, rows, prefer_fp8=False, fixed=None):
        """Storage type of the integer operand pair of an output-based search over `rows` candidate rows.  fp8 needs every
        zero point inside [0, 2^bits - 1]: the candidates' always are (FPCS grids), so what decides is the quantiser of the FIXED
        operand (``fixed``: the activation quantiser for a weight search, the weight quantiser for an activation search) -- the
        searched operand's own current parameters never enter the product.  (Asking both made the first round's weight search of
        the default schedule, which skips the dead weight self-search, run on int8: 26-39 % slower launches.)"""
        chunk = self._cand_chunk(rows, pad_k(self.in_features, I8))
        qs = (self.w_quantizer, self.a_quantizer) if (fixed is None or not FIXED_GRID_ONLY) else (fixed,)
        on_grid = all(getattr(q_, "_zp_on_grid", False) for q_ in qs)
        return search.int_operand_dtype(self.w_quantizer.n_bits, self.a_quantizer.n_bits, chunk, on_grid, prefer_fp8)

    def _cand_chunk(self, rows, kp_bytes):
        return max(1, min(self.eq_n, MAX_PACK_BYTES // max(1, rows * kp_bytes)))

    # ------------------------------------------------------------------ scoring calls (one = eq_n candidates)
    def _pack_x_fixed(self):
        """Activation operand quantised with the current a_quantizer -> (dtype, packed [1,1,M,Kp], sa, sa_mul, extra)."""
        be = backend.get()
        aq = self.a_quantizer
        x3 = self._x2().unsqueeze(0)
        # weight searches of short-K layers run on the slab kernel, where fp8 storage saves the epilogue's conversions
        dt = self._int_dt(self.out_features, prefer_fp8=self.in_features <= FP8_WEIGHT_SEARCH_MAX_K, fixed=aq)
        xp = be.pack_uniform(x3, aq.scale.data.view(-1), aq.zero_point.data.view(-1), 1, 0, 1, 0, 0, aq.n_bits, dt)
        return dt, xp, Strided(aq.scale.data.view(-1)), 1.0, None

    def _score_w(self, fixed, scale, zp, defer=False):
        """linear.py:355-384 -> scores [P, O] = -sum_images mean_tokens (raw_out - q_a(x) . fq_p(W)^T - b)^2."""
        be = backend.get()
        dt, xp, sa, sa_mul, shift = fixed
        M = xp.shape[2]
        P = scale.shape[0]
        wq = self.w_quantizer
        out = []
        chunk = self._cand_chunk(self.out_features, pad_k(self.in_features, dt) * (2 if dt == BF16 else 1))
        # bf16 activations (AdaLog values) against <= 4-bit weight candidates: the candidates go out as fp8 (exact) and the wide
        # streaming kernel converts them in registers -- 3/4 of the bytes through the L2 -> LDS path that bounds it
        mixed = (dt == BF16 and MIXED_W_SEARCH and wq.n_bits <= 4 and chunk >= P and xp.shape[-1] % 64 == 0
                 and hasattr(be, "gemm_mixed_ok") and be.gemm_mixed_ok(M, self.out_features, 1, 1, P, self.in_features))
        cdt, gdt, kal = (FP8, BF16_FP8, 64) if mixed else (dt, dt, 128)
        if (GEN_W_SEARCH and shift is None and dt in (I8, FP8) and chunk >= P and hasattr(be, "score_w_gen")
                and be.score_w_gen_ok(dt, M, self.out_features, self.in_features, xp.shape[-1], P)):
            # the candidate operand is generated inside the slab kernel from the fp32 weight rows: nothing is packed
            return be.score_w_gen(dt, xp, self._w2(), scale, zp, wq.n_bits, self._ref2_t(), sa.t, None if self.bias is None else self.bias.data,
                                  1.0 / self._tokens_per_image(), defer=defer)
        for s in range(0, P, chunk):
            e = min(P, s + chunk)
            sc, zc = scale[s:e].contiguous(), zp[s:e].contiguous()
            bias = None if self.bias is None else Strided(self.bias.data, n=1)
            # columns = (out channel, candidate): the e-s candidates of a channel share one reference column
            if shift is None:
                wp = be.pack_uniform(self._w2().unsqueeze(0), sc, zc, e - s, self.out_features, 1, 0, 1, wq.n_bits, cdt,
                                     c_inner=True, k_align=kal)
            else:
                # post-GELU operand is y*s - shift: the -shift term is a per-(candidate, row) constant
                #   -shift * s_w[p,o] * sum_i (q_w - z_w)  folded into the bias (cf. reparam_bias, linear.py:999-1006)
                wp, rowsum = be.pack_uniform(self._w2().unsqueeze(0), sc, zc, e - s, self.out_features, 1, 0, 1,
                                             wq.n_bits, cdt, want_rowsum=True, c_inner=True, k_align=kal)
                fold = be.shift_fold(rowsum.view(e - s, -1), sc, shift, None if self.bias is None else self.bias.data)
                bias = Strided(fold, c=self.out_features, n=1)
            out.append(be.gemm_score(gdt, xp, wp, M, self.out_features, e - s, 1, 1, self._ref2_t(), sa,
                                     Strided(sc, c=self.out_features, n=1), bias, False, True,
                                     1.0 / self._tokens_per_image(), sa_mul=sa_mul, ref_div=e - s, order=2,
                                     ref_transposed=True, defer=defer and chunk >= P))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def _pack_w_fixed(self, dt=I8, want_rowsum=False):
        be = backend.get()
        wq = self.w_quantizer
        return be.pack_uniform(self._w2().unsqueeze(0), wq.scale.data.view(-1), wq.zero_point.data.view(-1), 1, 0, 1, 0, 1,
                               wq.n_bits, dt, want_rowsum=want_rowsum)

    def _pack_w_cached(self, dt=I8, want_rowsum=False):
        """_pack_w_fixed for quant_forward (linear.py:46-51 re-quantises the weight on every call): the packed image is a pure
        function of (weight, scale, zero point), so it is kept until one of them changes (validate() runs thousands of forwards
        on unchanged weights).  The key -- storage address, in-place version and shape of each -- catches re-assigned tensors and
        autograd-visible in-place writes; writes through ``.data`` or a raw pointer do NOT bump ``_version``, so every such
        site calls invalidate_packed_weight()."""
        wq = self.w_quantizer
        key = (dt, want_rowsum, wq.n_bits) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (self.weight, wq.scale, wq.zero_point))
        hit = self.__dict__.get("_wp_cache")
        if hit is None or hit[0] != key:
            hit = (key, self._pack_w_fixed(dt, want_rowsum))
            self.__dict__["_wp_cache"] = hit
        return hit[1]

    def _score_a(self, wp, scale, zp, defer=False):
        """linear.py:394-423 -> scores [P, 1] = -sum_images mean_{tokens,out} (raw_out - fq_p(x) . q_w(W)^T - b)^2.

        Evaluated transposed, out^T = q_w(W) . fq_p(x)^T: GEMM rows = output channels, GEMM columns = (token, candidate)
        with the candidates innermost, so the 128 candidates of a token share one reference column and the weight scale /
        bias become per-row factors of the epilogue."""
        be = backend.get()
        aq = self.a_quantizer
        x3 = self._x2().unsqueeze(0)
        M = x3.shape[1]
        P = scale.shape[0]
        out = []
        dt = getattr(wp, "int_dt", I8)
        norm = 1.0 / (self._tokens_per_image() * self.out_features)
        if GEN_ACT_SEARCH and scale.shape[1] == 1 and hasattr(be, "score_act_gen") and \
                be.score_act_gen_ok(dt, self.out_features, M, self.in_features, wp.shape[-1], P):
            # the candidate operand is generated inside the slab kernel: nothing is packed (gemm_k_slab.inc, GEN form)
            return be.score_act_gen(dt, wp, x3[0], scale, zp, aq.n_bits, self.raw_out.reshape(-1, self.out_features),
                                    self.w_quantizer.scale.data.view(-1), None if self.bias is None else self.bias.data, norm,
                                    defer=defer)
        chunk = self._cand_chunk(M, pad_k(self.in_features, I8))
        for s in range(0, P, chunk):
            e = min(P, s + chunk)
            sc, zc = scale[s:e].contiguous(), zp[s:e].contiguous()
            xp = be.pack_uniform(x3, sc, zc, e - s, 1, 1, 0, 0, aq.n_bits, dt, c_inner=True)
            out.append(be.gemm_score(dt, wp, xp, self.out_features, M, e - s, 1, 1, self._ref2(),
                                     Strided(search.const_tensor([1.0], x3.device)), Strided(sc, c=1), None,
                                     False, False, 1.0 / (self._tokens_per_image() * self.out_features),
                                     ref_div=e - s, order=2, ref_transposed=True,
                                     row_scale=self.w_quantizer.scale.data.view(-1),
                                     row_bias=None if self.bias is None else self.bias.data, defer=defer and chunk >= P))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def _score_w_self(self, scale, zp, tail=None):
        """linear.py:296-318: scores [P, O] = -mean_i (W - fq_p(W))^2 from the sorted weight rows (csrc/sorted_score.hip).
        ``tail``: the FPCS step's ranking / next grid / commit, run by the same launch (search.fpcs)."""
        be = backend.get()
        w2 = self._w2()
        if SORTED_SELF_SEARCH and be.sorted_prefix_ok(w2.shape[0], w2.shape[1], self.w_quantizer.n_bits):
            sp = search.memo_tensor_fn("spw", w2, (), lambda: be.sorted_prefix(w2))
            if tail is not None:
                return be.score_self_sorted(sp, scale, zp, self.w_quantizer.n_bits, 1.0 / w2.shape[1], tail=tail)
            return be.score_self_sorted(sp, scale, zp, self.w_quantizer.n_bits, 1.0 / w2.shape[1])
        return search.honour_tail(be.score_w_self(w2, scale, zp, self.w_quantizer.n_bits), tail)

    def _score_a_self(self, scale, zp, tail=None):
        """linear.py:320-353: scores [P, I | 1] = -sum_images mean_tokens (x - fq_p(x))^2.  The captured activation is sorted
        once per search (per channel: one segment per channel; per tensor: one segment), every step then costs 2^bits
        bisections per candidate instead of a pass over the tensor."""
        be = backend.get()
        cw = self.a_quantizer.channel_wise
        T = self._tokens_per_image()
        norm = 1.0 / T if cw else 1.0 / (T * self.in_features)
        x2 = self._x2()
        S, n = (x2.shape[1], x2.shape[0]) if cw else (1, x2.numel())
        if SORTED_SELF_SEARCH and be.sorted_prefix_ok(S, n, self.a_quantizer.n_bits):
            sp = search.memo_tensor_fn("spa", self.raw_input, (cw,),
                                       lambda: be.sorted_prefix(x2.t().contiguous() if cw else x2.reshape(1, -1)))
            if tail is not None:
                return be.score_self_sorted(sp, scale, zp, self.a_quantizer.n_bits, norm, tail=tail)
            return be.score_self_sorted(sp, scale, zp, self.a_quantizer.n_bits, norm)
        return search.honour_tail(be.score_a_self(x2, scale, zp, cw, self.a_quantizer.n_bits, norm), tail)

    # ------------------------------------------------------------------ FPCS (linear.py:483-523)
    def weight_fpcs(self, fpcs_width=16, steps=6, search_strategy="output"):
        if search_strategy != "self" and search.round_is_redundant(self, "w", self.a_quantizer):
            return                                 # the activation quantiser is what this search saw last round: same winner
        scale, zp, delta = search.weight_grid(self._w2(), self.w_quantizer.n_bits, self.eq_n)
        if search_strategy == "self":
            fn = lambda s, z, t, tail=None: self._score_w_self(s, z, tail=tail)
            fn.fused_tail = True
        else:
            score = self._w_scorer()
            if getattr(score, "fused_tail", False):
                fn = lambda s, z, t, tail=None: score(s, z, defer=True, tail=tail)
                fn.fused_tail = True
            else:
                fn = lambda s, z, t: score(s, z, defer=True)
            fn.global_scores = getattr(score, "global_scores", False)
        wq = self.w_quantizer
        res = search.fpcs(scale, zp, None, delta, fn, steps, fpcs_width, self.eq_n, None,
                          commit_to=search.commit_targets(wq.scale, wq.zero_point, None))
        if res is not None:
            self._commit_w(res[0], res[1])

    def _w_scorer(self):
        """The scoring function (scale, zp, defer=False) -> scores [P, O] | PendingScores of one output-MSE weight_fpcs call, with
        whatever is fixed for the call (the activation quantiser: linear.py:483-503) prepared once: the Gram-form state where it
        pays, else the packed activation image of the token-form kernels."""
        gram = self._gram_state()
        if gram is not None:
            norm = 1.0 / self._tokens_per_image()
            if getattr(backend.get(), "FpcsTail", None) is not None:
                fn = lambda s, z, defer=False, tail=None: gram.score_w(self._w2(), s, z, self.w_quantizer.n_bits, norm, tail=tail)
                fn.fused_tail = True                     # score_w runs the FPCS tail itself (same call)
            else:
                fn = lambda s, z, defer=False: gram.score_w(self._w2(), s, z, self.w_quantizer.n_bits, norm)
            # under image sharding the state holds the all-reduced G, c, S0: the scores are final on every rank (search.fpcs)
            fn.global_scores = bool(getattr(gram, "global_scores", False))
            return fn
        fixed = self._pack_x_fixed()
        return lambda s, z, defer=False: self._score_w(fixed, s, z, defer=defer)

    def _a_scorer(self):
        """The scoring function (scale, zp, defer=False) -> scores [P, 1] | PendingScores of one output-MSE activation_fpcs call, with
        whatever is fixed for the call (the weight quantiser: linear.py:505-523) prepared once: the Gram-form state (csrc/gram_act.hip:
        the candidates' own Gram matrices X_p^T X_p against H = Wq^T Wq, K^2 / 2 instead of O K multiply-adds per token) where
        backend.gram_act_ok takes the search, else the packed weight image of the token-form kernels."""
        be = backend.get()
        aq, wq = self.a_quantizer, self.w_quantizer
        x2 = self._x2()
        # (the weight image enters the build's int8 product as q - z: its zero point must lie on the grid, like _gram_state's)
        if (hasattr(be, "gram_act_ok") and type(aq) is UniformQuantizer and not aq.channel_wise and aq.scale.numel() == 1
                and getattr(wq, "_zp_on_grid", False)
                and be.gram_act_ok(x2.shape[0], self.out_features, self.in_features, aq.n_bits, wq.n_bits, self.eq_n)):
            prep = search.memo_tensor_fn("gact", self.raw_input, (), lambda: be.GramActPrepared(x2))
            st = be.GramActState(prep, self.raw_out.reshape(-1, self.out_features), None if self.bias is None else self.bias.data,
                                 self._w2(), wq.scale.data.view(-1), wq.zero_point.data.view(-1), wq.n_bits, aq.n_bits, self.eq_n)
            norm = 1.0 / (self._tokens_per_image() * self.out_features)
            if getattr(be, "FpcsTail", None) is not None:
                fn = lambda s, z, defer=False, tail=None: st.score(s, z, norm, tail=tail)
                fn.fused_tail = True                     # the finish kernel's last block ranks and writes the next grid
                return fn
            return lambda s, z, defer=False: st.score(s, z, norm)
        dt = self._int_dt(self.raw_input.numel() // self.in_features, prefer_fp8=self.in_features <= FP8_WEIGHT_SEARCH_MAX_K,
                          fixed=wq)
        wp = self._pack_w_fixed(dt)
        wp.int_dt = dt
        return lambda s, z, defer=False: self._score_a(wp, s, z, defer=defer)

    def _gram_state(self):
        """The Gram-form state of an output-MSE weight search (linear.py:355-392 with the activation quantiser fixed for the whole
        weight_fpcs call), or None where the token-form kernels serve the search: a non-uniform or per-channel activation
        quantiser, a shape adalog_gram_ok declines (K not a multiple of 32, too few tokens per K for the form to pay)."""
        be = backend.get()
        aq = self.a_quantizer
        if not hasattr(be, "gram_ok") or type(aq) is not UniformQuantizer or aq.channel_wise or aq.scale.numel() != 1:
            return None
        # gram.hip stores (q - z) as int8 and bounds G by T (2^b - 1)^2: that needs the zero point inside [0, 2^bits - 1], which
        # only a committed FPCS grid point guarantees (a min/max initialisation still in force, or a loaded quantiser, may not)
        if not getattr(aq, "_zp_on_grid", False):
            return None
        x2 = self._x2()
        # (image-sharded ranks: the state is built from ALL ranks' tokens -- ops.GramState all-reduces G, c, S0 once -- so the form is
        # priced on the global token count, not on this rank's share)
        from .. import parallel
        if not be.gram_ok(x2.shape[0] * parallel.world_size(), self.out_features, self.in_features, aq.n_bits,
                          self.w_quantizer.n_bits, self.eq_n):
            return None
        return be.GramState(x2, aq.scale.data, aq.zero_point.data, aq.n_bits, self._ref2_t(),
                            None if self.bias is None else self.bias.data)

    def activation_fpcs(self, fpcs_width=16, steps=6, search_strategy="output"):
        aq = self.a_quantizer
        if search_strategy != "self" and search.round_is_redundant(self, "a", self.w_quantizer):
            return
        scale, zp, delta = search.activation_grid(self.raw_input, aq.n_bits, self.eq_n, aq.channel_wise)
        if search_strategy == "self":
            fn = lambda s, z, t, tail=None: self._score_a_self(s, z, tail=tail)
            fn.fused_tail = True
        else:
            score = self._a_scorer()
            if getattr(score, "fused_tail", False):
                fn = lambda s, z, t, tail=None: score(s, z, defer=True, tail=tail)
                fn.fused_tail = True
            else:
                fn = lambda s, z, t: score(s, z, defer=True)
        res = search.fpcs(scale, zp, None, delta, fn, steps, fpcs_width, self.eq_n, 1e-4,
                          commit_to=search.commit_targets(aq.scale, aq.zero_point, None))
        if res is not None:
            self._commit_a(res[0], res[1])

    def hyperparameter_searching(self):
        """linear.py:525-545 with fpcs=True (the only mode the shipped configs use, configs/*.py:20)."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        if not self.fpcs:
            raise NotImplementedError("non-FPCS single-pass search is not part of the accelerated path (configs use fpcs=True)")
        self._initialize_calib_parameters()
        search.begin_rounds(self)
        # linear.py:536: the weights' self-MSE search.  Its result is overwritten by the first round's output-MSE weight search
        # (which reads the activation quantiser and the percentile grid, never the current weight parameters) before anything
        # reads it, so with search_round >= 1 it is dead work and is not run (ADALOG_DEAD_W_SELF=1 runs it, as the reference does).
        if self.search_round < 1 or RUN_DEAD_W_SELF:
            self.weight_fpcs(steps=self.steps, search_strategy="self")
        self.activation_fpcs(steps=self.steps, search_strategy="self")
        for _ in range(self.search_round):
            self.weight_fpcs(steps=self.steps, search_strategy="output")
            self.activation_fpcs(steps=self.steps, search_strategy="output")
        self.calibrated = True
        search.begin_rounds(self)
        search.forget_grids()
        del self.raw_input, self.raw_out
        self._ref_t = self._ref_t_key = None
        return None

    # ------------------------------------------------------------------ quantised forward (linear.py:46-51), fused
    def quant_forward(self, x, addend=None):
        """``addend`` (optional, the shape of the output): added to the result -- in the GEMM's epilogue on the fused routes (the
        residual stream of a transformer block: x + proj(...), utils/models.py)."""
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        if torch.is_grad_enabled() and (self.w_quantizer.training_mode or self.a_quantizer.training_mode
                                        or not isinstance(self.w_quantizer, UniformQuantizer)):
            out = super().quant_forward(x)                     # BRECQ: differentiable fake-quant + GEMM
            return out if addend is None else addend + out
        be = backend.get()
        aq = self.a_quantizer
        if not isinstance(self.w_quantizer, UniformQuantizer) or aq.n_bits > 7 or self.w_quantizer.n_bits > 7:
            out = super().quant_forward(x)
            return out if addend is None else addend + out
        lead = x.shape[:-1]
        x3 = x.reshape(1, -1, self.in_features)
        wp = self._pack_w_cached()
        sa_, sb_ = Strided(aq.scale.data.view(-1)), Strided(self.w_quantizer.scale.data.view(-1), n=1)
        bias_ = None if self.bias is None else Strided(self.bias.data, n=1)
        fused_add = addend is not None and getattr(be, "QF_EXTRAS", False) and addend.shape == lead + (self.out_features,)
        add_ = addend.reshape(1, -1, self.out_features) if fused_add else None
        if aq.scale.numel() == 1 and hasattr(be, "gemm_out_gen") and be.gemm_out_gen_ok(x3, wp.shape[-1], aq.n_bits):
            # ONE launch for the layer: the activation is quantised in the GEMM's loader (k_gemm_cand<GENA>), the weight image is cached
            out = be.gemm_out_gen(x3, aq.scale.data, aq.zero_point.data, aq.n_bits, wp, self.out_features, 1, sa_, sb_, bias_,
                                  **({"addend": add_} if fused_add else {}))
        else:
            xp = be.pack_uniform(x3, aq.scale.data.view(-1), aq.zero_point.data.view(-1), 1, 0, 1, 0, 0, aq.n_bits, I8)
            out = be.gemm_out(I8, xp, wp, x3.shape[1], self.out_features, 1, 1, sa_, sb_, bias_,
                              **({"addend": add_} if fused_add else {}))
        out = out.view(*lead, self.out_features)
        return out if (addend is None or fused_add) else addend + out


class AsymmetricallyChannelWiseBatchingQuantLinear(AsymmetricallyBatchingQuantLinear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 calib_batch_size=None, search_round=1, eq_n=100, n_V=1, fpcs=False, steps=4):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         calib_batch_size=calib_batch_size, search_round=search_round, eq_n=eq_n, n_V=n_V,
                         fpcs=fpcs, steps=steps)
        del self.a_quantizer
        self.a_quantizer = UniformQuantizer(n_bits=a_bit, symmetric=False, channel_wise=True)
        self.a_quantizer.scale = nn.Parameter(torch.zeros((in_features)))
        self.a_quantizer.zero_point = nn.Parameter(torch.zeros((in_features)))
        self._prev_layer = None

    def __setattr__(self, name, value):
        if name == "prev_layer":                     # keep the LayerNorm out of _modules (linear.py:571-575)
            self.__dict__['_prev_layer'] = value
        else:
            super().__setattr__(name, value)

    @property
    def prev_layer(self):
        return self._prev_layer

    def quant_forward(self, x, addend=None):
        if self.a_quantizer.channel_wise:            # only between search and reparam; plain composition
            out = MinMaxQuantLinear.quant_forward(self, x)
            return out if addend is None else addend + out
        return super().quant_forward(x, addend)

    def hyperparameter_searching(self):
        """linear.py:585-594: per-channel activation FPCS against the activation's own MSE."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        assert self.a_quantizer.channel_wise and self.w_quantizer.channel_wise
        if not self.fpcs:
            raise NotImplementedError("non-FPCS search is not part of the accelerated path")
        self._initialize_calib_parameters()
        self.activation_fpcs(steps=self.steps, search_strategy="self")
        self.calibrated = True

    def reparam_step1(self):
        """linear.py:596-612: fold the per-channel (scale, zero point) into the preceding LayerNorm and this weight."""
        self.calibrated = False
        aq = self.a_quantizer
        channel_min = -aq.zero_point * aq.scale
        target_channel_scale = torch.mean(aq.scale).view(1)
        target_channel_zero_point = torch.mean(aq.zero_point).round().view(1)
        target_channel_min = -target_channel_zero_point * target_channel_scale
        r = (aq.scale / target_channel_scale)
        b = channel_min / r - target_channel_min
        self.prev_layer.weight.data = self.prev_layer.weight.data / r
        self.prev_layer.bias.data = self.prev_layer.bias.data / r.view(-1) - b
        self.weight.data = self.weight.data * r.view(1, -1)
        extra = torch.mm(self.weight.data, b.reshape(-1, 1)).reshape(-1)
        if self.bias is not None:
            self.bias.data = self.bias.data + extra
        else:
            self.bias = nn.Parameter(torch.zeros(self.out_features, device=self.weight.device))
            self.bias.data = extra
        return r, b, target_channel_scale, target_channel_zero_point

    def reparam(self):
        """linear.py:614-621."""
        with torch.no_grad():
            r, b, t_scale, t_zp = self.reparam_step1()
            self.invalidate_packed_weight()
            self.raw_input = self.raw_input / r - b
            del self.a_quantizer.scale, self.a_quantizer.zero_point
            self.a_quantizer.channel_wise = False
            self.a_quantizer.scale = nn.Parameter(t_scale.detach().clone())
            self.a_quantizer.zero_point = nn.Parameter(t_zp.detach().clone())
            AsymmetricallyBatchingQuantLinear.hyperparameter_searching(self)


class PostGeluLogBasedBatchingQuantLinear(AsymmetricallyBatchingQuantLinear):
    """fc2: activation = ShiftAdaLog (log base 2^(-q/37), q searched), bias re-parameterised after the search."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, mode="raw", w_bit=8, a_bit=8,
                 calib_batch_size=None, search_round=1, eq_n=100, n_V=1, quantizer='adalog', fpcs=False, steps=4):
        super().__init__(in_features, out_features, bias=bias, mode=mode, w_bit=w_bit, a_bit=a_bit,
                         calib_batch_size=calib_batch_size, search_round=search_round, eq_n=eq_n, n_V=n_V,
                         fpcs=fpcs, steps=steps)
        if quantizer != 'adalog':
            raise NotImplementedError("only the 'adalog' post-GELU quantiser is accelerated (log2/logsqrt2 are ablations)")
        del self.a_quantizer
        self.a_quantizer = ShiftAdaLogQuantizer(n_bits=a_bit, symmetric=False, channel_wise=False)
        self.a_quantizer.scale = nn.Parameter(torch.zeros((1)))
        self.a_quantizer.shift.data.copy_(torch.tensor(GELU_SHIFT))
        # search-time mantissa table (linear.py:750-752), fp32 ops as in the reference
        self.table = torch.tensor([2 ** (-j / self.a_quantizer.r) for j in range(120)])
        self.table_scale = 1. / (4 * self.a_quantizer.n_levels - 2)
        self.table = torch.round(self.table / self.table_scale) * self.table_scale
        self._q_host = 37

    def _mant37(self, device):
        """Integer numerators of the table (15..30 for 4 bit), the bf16-exact MFMA operand."""
        num = torch.round(self.table[:37] / self.table_scale)
        return search.const_tensor(num.tolist(), device)

    def _ts32(self):
        return float(torch.tensor(self.table_scale, dtype=torch.float32))

    # ---- percentile candidates (linear.py:763-814)
    def calculate_percentile_activation_candidates(self, l=0.9, r=1.0):
        return search.memo_tensor_fn("pp", self.raw_input, (l, r, self.eq_n),
                                     lambda: self._percentile_activation_candidates(l, r))

    def _percentile_activation_candidates(self, l=0.9, r=1.0):
        from .. import parallel
        be = backend.get()
        qs = torch.tensor([l, r]).tolist()
        if parallel.is_dist():
            # one global segment spread over the ranks: distributed radix select (the rank of the percentile comes from
            # the GLOBAL count of positive entries, known after the first all-reduced histogram)
            sel = be.ShardedSelect(self.raw_input.reshape(1, -1), 1, 2, 0, 1, 0, qfrac=qs)
            for p in range(4):
                sel.hist_pass(p)
                parallel.all_reduce_sum(sel.hist)
                sel.pick(p)
            pp = sel.values()                                                                   # [2, 1]
        else:
            pp = be.positive_percentile_rows(self.raw_input.reshape(1, -1), qs)                 # [2, 1]
        cand = (pp.view(1, 2) + self.a_quantizer.shift.data.view(1, 1))
        frac = search.const_tensor([i / (self.eq_n - 1) for i in range(self.eq_n)], cand.device).view(1, -1)
        scales = cand[:, 0:1] + (cand[:, 1:] - cand[:, 0:1]) * frac
        return cand, scales

    # ---- fixed operands
    def _pack_x_fixed(self):
        be = backend.get()
        aq = self.a_quantizer
        dev = self.weight.device
        x3 = self._x2().unsqueeze(0)
        qv = search.const_tensor([float(self._q_host)], dev)
        xp = be.pack_adalog(x3, aq.scale.data.view(-1), qv, 1, 0, 1, 0, aq.n_bits, self._mant37(dev),
                            shift=aq.shift.data, clamp_u=True)
        return BF16, xp, Strided(aq.scale.data.view(-1)), self._ts32(), aq.shift.data

    def _shift32(self):
        """The quantiser's shift as the fp32 host value the fused kernels take (one host read per search: a loaded
        checkpoint or a future searched shift need not equal GELU_SHIFT32, and the packed path reads aq.shift)."""
        if getattr(self, "_shift_host", None) is None:
            self._shift_host = float(self.a_quantizer.shift.data.reshape(-1)[0])
        return self._shift_host

    def _log2_x(self):
        """log2(raw_input + shift), once per captured tensor (input of the fused activation search)."""
        key = (self.raw_input.data_ptr(), self.raw_input._version)
        if getattr(self, "_lx_key", None) != key:
            self._lx = backend.get().log2_shift(self._x2(), self._shift32())
            self._lx_key = key
        return self._lx

    def _score_scale_logbase(self, wp, bias_fold, scale, qv, defer=False):
        """linear.py:816-848 / 856-890 / 898-931 -> scores [P, 1] for per-candidate (scale_p, q_p); transposed like
        _score_a (rows = output channels, columns = (token, candidate))."""
        be = backend.get()
        aq = self.a_quantizer
        dev = self.weight.device
        x3 = self._x2().unsqueeze(0)
        M = x3.shape[1]
        P = scale.shape[0]
        norm = 1.0 / (self._tokens_per_image() * self.out_features)
        if FUSED_ACT_SEARCH and hasattr(be, "score_act_fused") and \
                be.score_act_fused_ok(self.out_features, M, self.in_features, wp.shape[-1], P, aq.n_bits):
            # quantise-in-the-loader kernel: the [M*P, K] candidate operand is never written (gemm_fused.hip)
            return be.score_act_fused(wp, x3[0], self._log2_x(), self.raw_out.reshape(-1, self.out_features),
                                      self.w_quantizer.scale.data.view(-1), bias_fold, scale.reshape(-1), qv.reshape(-1),
                                      aq.n_bits, self._mant37(dev), self._shift32(), True, self._ts32(), norm)
        out = []
        chunk = self._cand_chunk(M, pad_k(self.in_features, BF16) * 2)
        for s in range(0, P, chunk):
            e = min(P, s + chunk)
            sc, qc = scale[s:e].contiguous(), qv[s:e].contiguous()
            xp = be.pack_adalog(x3, sc, qc, e - s, 1, 1, 0, aq.n_bits, self._mant37(dev), shift=aq.shift.data, clamp_u=True,
                                c_inner=True)
            out.append(be.gemm_score(BF16, wp, xp, self.out_features, M, e - s, 1, 1, self._ref2(),
                                     Strided(search.const_tensor([1.0], dev)), Strided(sc, c=1), None, False, False,
                                     1.0 / (self._tokens_per_image() * self.out_features), sa_mul=self._ts32(),
                                     ref_div=e - s, order=2, ref_transposed=True,
                                     row_scale=self.w_quantizer.scale.data.view(-1), row_bias=bias_fold,
                                     defer=defer and chunk >= P))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def activation_fpcs(self, ud_candidates, base_num=8, scale_num=16, fpcs_width=32, steps=6):
        """linear.py:941-967: 128 log bases -> top-8 bases x 16 scales -> width-32 FPCS with 4 neighbours."""
        be = backend.get()
        aq = self.a_quantizer
        if search.round_is_redundant(self, "a", self.w_quantizer, aq):     # (starts from its own current scale: part of the input)
            return
        dev = self.weight.device
        wp, rowsum = self._pack_w_fixed(BF16, want_rowsum=True)
        bias_fold = be.shift_fold(rowsum.view(1, -1), self.w_quantizer.scale.data.view(1, -1), aq.shift.data,
                                  None if self.bias is None else self.bias.data).view(-1)
        q_all = search.const_tensor([float(i) for i in range(10, 11 + self.eq_n)], dev)[:self.eq_n].view(-1, 1)
        cur_scale = aq.scale.data.view(1, 1).expand(self.eq_n, 1).contiguous()
        s0 = self._score_scale_logbase(wp, bias_fold, cur_scale, q_all)
        q_idx = search.argbest(s0, base_num)                                       # [8, 1]
        top_q = be.fpcs_next(q_all, None, None, q_idx, base_num, 1, search.const_tensor([0.5], dev),
                             torch.zeros(1, device=dev), None)[0]                  # gather: [8, 1]
        frac = search.const_tensor([i / (scale_num - 1) for i in range(scale_num)], dev).view(-1, 1)
        ud = ud_candidates.view(-1)
        scales16 = ud[0:1].view(1, 1) + (ud[1:2] - ud[0:1]).view(1, 1) * frac       # [16, 1]
        delta = (scales16[1:2] - scales16[0:1]).view(1).contiguous()
        scale = scales16.repeat(base_num, 1).contiguous()                           # a.repeat(1, base_num) layout
        qv = top_q.repeat_interleave(scale_num, dim=0).contiguous()
        fn = lambda s, z, t: self._score_scale_logbase(wp, bias_fold, s, t, defer=True)
        res = search.fpcs(scale, None, qv, delta, fn, steps, fpcs_width, self.eq_n, None)
        if res is not None:
            aq.scale.data.copy_(res[0].view(aq.scale.shape))
            aq.q.data.copy_(res[2].view(aq.q.shape).to(aq.q.dtype))
            self._q_host = int(aq.q.item())                     # one host read per activation_fpcs (LUT rebuild)
            aq.update_table(self._q_host)

    def hyperparameter_searching(self):
        """linear.py:969-997 with fpcs=True."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        if not self.fpcs:
            raise NotImplementedError("non-FPCS search is not part of the accelerated path")
        self._initialize_calib_parameters()
        self._shift_host = None
        search.begin_rounds(self)
        try:
            self.weight_fpcs(steps=self.steps, search_strategy="self")
            ud_candidates, input_scale_candidates = self.calculate_percentile_activation_candidates()
            self.a_quantizer.scale.data.copy_(input_scale_candidates[:, -2])
            self.a_quantizer.inited = True
            self._q_host = int(self.a_quantizer.q.item())
            for _ in range(self.search_round):
                self.activation_fpcs(ud_candidates=ud_candidates, steps=self.steps)
                self.weight_fpcs(steps=self.steps, search_strategy="output")
        finally:
            # the caches are keyed by storage address: they must not outlive this search, whether it ends or raises
            search.forget_grids()
            search.begin_rounds(self)
            self._ref_t = self._ref_t_key = None
            self._lx = self._lx_key = None
        self.calibrated = True
        del self.raw_input, self.raw_out

    def reparam_bias(self):
        """linear.py:999-1006: bias += (-shift * 1^T) . q_w(W)^T, then the quantiser stops subtracting the shift."""
        aq = self.a_quantizer
        if aq._shift_args()[1] is False:
            return
        be = backend.get()
        _, rowsum = self._pack_w_fixed(BF16, want_rowsum=True)
        fold = be.shift_fold(rowsum.view(1, -1), self.w_quantizer.scale.data.view(1, -1), aq.shift.data,
                             None if self.bias is None else self.bias.data).view(-1)
        if self.bias is None:
            self.bias = nn.Parameter(torch.zeros(self.out_features, device=self.weight.device))
        self.bias.data.copy_(fold)
        aq.mark_bias_reparamed()

    def fused_ok(self):
        """quant_forward takes the fused route (packer + bf16 MFMA product): the callers that hand over fc1's output with
        ``pre_gelu`` (utils/models.py: Mlp) ask first."""
        return (self.calibrated and isinstance(self.w_quantizer, UniformQuantizer) and not self.a_quantizer.training_mode
                and not (torch.is_grad_enabled() and self.w_quantizer.training_mode) and getattr(backend.get(), "QF_EXTRAS", False))

    def quant_forward(self, x, addend=None, pre_gelu=False):
        """``pre_gelu``: x is fc1's output and the layer's input is GELU(x) -- applied in the packer's loader (one pass less over the
        widest activation of the block); ``addend``: added to the result in the GEMM's epilogue (the residual stream)."""
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        aq = self.a_quantizer
        if (torch.is_grad_enabled() and (self.w_quantizer.training_mode or aq.training_mode)) \
                or not isinstance(self.w_quantizer, UniformQuantizer) or aq.training_mode:
            # (a BRECQ iteration: GELU inside the quantiser's kernels, the residual inside the product's reduction pass)
            return MinMaxQuantLinear.quant_forward(self, x, pre_gelu=pre_gelu, addend=addend)
        be = backend.get()
        dev = x.device
        lead = x.shape[:-1]
        x3 = x.reshape(1, -1, self.in_features)
        if self._q_host is None:
            self._q_host = int(aq.q.item())
        qv = search.const_tensor([float(self._q_host)], dev)
        if pre_gelu and not (getattr(be, "QF_EXTRAS", False) and x3.stride(-1) == 1):
            x3, pre_gelu = torch.nn.functional.gelu(x3), False
        xp = be.pack_adalog(x3, aq.scale.data.view(-1), qv, 1, 0, 1, 0, aq.n_bits, self._mant37(dev),
                            shift=aq.shift.data, clamp_u=True, **({"pre_gelu": True} if pre_gelu else {}))
        _, sub = aq._shift_args()
        if sub:
            wp, rowsum = self._pack_w_cached(BF16, want_rowsum=True)
            bias = be.shift_fold(rowsum.view(1, -1), self.w_quantizer.scale.data.view(1, -1), aq.shift.data,
                                 None if self.bias is None else self.bias.data).view(-1)
        else:
            wp = self._pack_w_cached(BF16)
            bias = None if self.bias is None else self.bias.data
        fused_add = addend is not None and getattr(be, "QF_EXTRAS", False) and addend.shape == lead + (self.out_features,)
        out = be.gemm_out(BF16, xp, wp, x3.shape[1], self.out_features, 1, 1, Strided(aq.scale.data.view(-1)),
                          Strided(self.w_quantizer.scale.data.view(-1), n=1),
                          None if bias is None else Strided(bias, n=1), sa_mul=self._ts32(),
                          **({"addend": addend.reshape(1, -1, self.out_features)} if fused_add else {}))
        out = out.view(*lead, self.out_features)
        return out if (addend is None or fused_add) else addend + out

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._q_host = None
