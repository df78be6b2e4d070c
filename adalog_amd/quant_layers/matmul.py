"""Quantised A @ B for the attention products -- module API of reference quant_layers/matmul.py
(MinMaxQuantMatMul -> PTQSLQuantMatMul -> PTQSLBatchingQuantMatMul -> AsymmetricallyBatchingQuantMatMul
 -> PostSoftmaxAsymmetricallyBatchingQuantMatMul), searches and forward on HIP kernels.

Shapes (matmul.py:49-56):  q@k^T: A [N,H,S,C], B [N,H,C,S];  softmax@v: A [N,H,S,S], B [N,H,S,C].
Every (image, head) pair is one GEMM group g = n*H + h; per-head parameters are addressed by g % H inside the
kernels, so a scoring call for 128 candidates is one batched MFMA launch over [P, N*H] groups.
B enters the GEMM K-contiguous, i.e. as B^T -- taken as a strided *view*, packed by the operand kernel (no copy).
"""
import os

import torch
import torch.nn as nn

from .. import backend, search, train_mm
from ..ops import BF16, BF16_FP8, FP8, I8, Strided, pad_k  # noqa: F401
from ..quantizers.logarithm import AdaLogQuantizer
from ..quantizers.uniform import UniformQuantizer

MAX_PACK_BYTES = int(os.environ.get('ADALOG_MAX_PACK_GIB', '8')) << 30
# uniform attention candidates generated inside the scoring kernel instead of packed (adalog_gemm_score_gen): 'all' = wherever the
# window kernel (swin) or the wave-private q.k^T kernel (197-token ViTs) serves the search, 'win' = windows only, '0' = never.
# Same-box A/Bs: the kernels get slower (VALU: swin_base 148 -> 176 ms, deit_small 51 -> 69 ms per calibration) but the 0.3-1.1 GB
# operand is neither written nor read: swin_base 2 978 -> 2 863 ms, deit_small 1 052 -> 1 033 ms per calibration
GEN_MM = os.environ.get('ADALOG_GEN_MM', 'all')
# the log-base search of softmax.v with its 128 AdaLog quantisations generated inside the kernel (adalog_gemm_score_avq); 0 = packed
GEN_AVQ = os.environ.get('ADALOG_GEN_AVQ', '1') != '0'
MIXED_B_SEARCH = os.environ.get('ADALOG_MIXED_B', '1') != '0'     # softmax.v weight search: fp8 candidates against the bf16 probabilities


class MinMaxQuantMatMul(nn.Module):
    def __init__(self, A_bit=8, B_bit=8, mode="raw"):
        super().__init__()
        self.mode = mode
        self.A_quantizer = UniformQuantizer(n_bits=A_bit, symmetric=True, channel_wise=False)
        self.B_quantizer = UniformQuantizer(n_bits=B_bit, symmetric=True, channel_wise=False)
        self.raw_input = None
        self.raw_out = None
        self.tmp_input = None
        self.tmp_out = None
        self.calibrated = False

    def forward(self, A, B, a_pre=False, b_pre=False):
        """a_pre / b_pre (quant_forward only): that operand already went through this module's input quantiser (an attention
        block quantises q, k, v in the pass that splits them, train_mm.qkv_split_quant)."""
        if self.mode == 'raw':
            return A @ B
        if self.mode == "quant_forward":
            return self.quant_forward(A, B, a_pre, b_pre)
        raise NotImplementedError

    def quant_input_A(self, x):
        return self.A_quantizer(x)

    def quant_input_B(self, x):
        return self.B_quantizer(x)

    def quant_forward(self, A, B, a_pre=False, b_pre=False):
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        a_sim = A if a_pre else self.quant_input_A(A)
        b_sim = B if b_pre else self.quant_input_B(B)
        if a_sim.requires_grad or b_sim.requires_grad:       # a BRECQ iteration: the contractions run on csrc/brecq_gemm.hip
            return train_mm.matmul(a_sim, b_sim, heads_last=getattr(self, 'out_heads_last', False))
        return a_sim @ b_sim


class PTQSLQuantMatMul(MinMaxQuantMatMul):
    def __init__(self, A_bit=8, B_bit=8, mode="raw", search_round=1, eq_n=100, head_channel_wise=True, num_heads=12):
        super().__init__(A_bit, B_bit, mode)
        self.A_quantizer = UniformQuantizer(n_bits=A_bit, symmetric=True, channel_wise=head_channel_wise)
        self.B_quantizer = UniformQuantizer(n_bits=B_bit, symmetric=True, channel_wise=head_channel_wise)
        self.search_round = search_round
        self.eq_n = eq_n
        self.head_channel_wise = head_channel_wise
        self.num_heads = num_heads
        target_shape = [1, self.num_heads, 1, 1] if self.head_channel_wise else [1, 1, 1, 1]
        self.A_quantizer.scale = nn.Parameter(torch.zeros(*target_shape))
        self.B_quantizer.scale = nn.Parameter(torch.zeros(*target_shape))


class PTQSLBatchingQuantMatMul(PTQSLQuantMatMul):
    def __init__(self, A_bit=8, B_bit=8, mode="raw", calib_batch_size=32, search_round=1, eq_n=100,
                 head_channel_wise=True, num_heads=12):
        super().__init__(A_bit, B_bit, mode, search_round, eq_n, head_channel_wise, num_heads)
        self.calib_batch_size = calib_batch_size

    def _initialize_calib_parameters(self):
        self.calib_size = self.raw_input[0].shape[0]
        self.parallel_eq_n = self.eq_n            # no memory-derived chunking (matmul.py:95-106 has no counterpart)


class AsymmetricallyBatchingQuantMatMul(PTQSLBatchingQuantMatMul):
    def __init__(self, A_bit=8, B_bit=8, mode="raw", calib_batch_size=32, search_round=1, eq_n=128,
                 head_channel_wise=True, num_heads=12, fpcs=False, steps=4):
        super().__init__(A_bit, B_bit, mode, calib_batch_size, search_round, eq_n, head_channel_wise, num_heads)
        self.fpcs = fpcs
        self.steps = steps
        del self.A_quantizer, self.B_quantizer
        self.A_quantizer = UniformQuantizer(n_bits=A_bit, symmetric=False, channel_wise=head_channel_wise)
        self.B_quantizer = UniformQuantizer(n_bits=B_bit, symmetric=False, channel_wise=head_channel_wise)
        target_shape = [1, self.num_heads, 1, 1] if self.head_channel_wise else [1, 1, 1, 1]
        self.A_quantizer.scale = nn.Parameter(torch.zeros(*target_shape))
        self.B_quantizer.scale = nn.Parameter(torch.zeros(*target_shape))
        self.A_quantizer.zero_point = nn.Parameter(torch.zeros(*target_shape))
        self.B_quantizer.zero_point = nn.Parameter(torch.zeros(*target_shape))

    # ------------------------------------------------------------------ views
    def _heads(self):
        return self.num_heads if self.head_channel_wise else 1

    def _norm(self, A, S, Sp):
        """mean over (S, S') per head (matmul.py:154-155) or over (H, S, S') per tensor (:156-157), summed over images."""
        return 1.0 / (S * Sp) if self.head_channel_wise else 1.0 / (A.shape[1] * S * Sp)

    @staticmethod
    def _a3(A):
        """[N,H,S,K] -> [G, S, K] view (rows = output rows, K contiguous)."""
        return A.reshape(-1, A.shape[-2], A.shape[-1])

    def _bt3_packable(self, B):
        """B^T view for the search packs.  softmax@v hands over v [N,H,S,C], whose transpose is not K-contiguous: make ONE
        K-contiguous copy per layer so the 100+ candidate packs read coalesced rows."""
        bt = self._bt3(B)
        if bt.stride(-1) == 1:
            return bt
        key = B.data_ptr()
        if getattr(self, "_bt_key", None) != key:
            self._bt_c = bt.contiguous()
            self._bt_key = key
        return self._bt_c

    @staticmethod
    def _bt3(B):
        """[N,H,K,S'] -> B^T as a [G, S', K] view: K-contiguous when B itself was a transposed view (q@k^T), otherwise
        row-contiguous (softmax@v) -- the operand kernel coalesces along whichever stride is 1.  Copies only if the
        layout cannot be expressed as a 3-D strided view."""
        Bt = B.transpose(-2, -1)
        return Bt.reshape(-1, Bt.shape[-2], Bt.shape[-1])

    def _dims(self):
        A, B = self.raw_input
        S, K, Sp = A.shape[-2], A.shape[-1], B.shape[-1]
        G = A.numel() // (S * K)
        return G, S, K, Sp

    def _ref3(self):
        G, S, K, Sp = self._dims()
        return self.raw_out.reshape(G, S, Sp)

    def _ref3_t(self):
        """raw_out transposed per group, [G, S', S] (one copy per layer) for the B-operand searches, whose GEMM rows run
        along S: the kernel stages the reference through LDS when its row axis is contiguous."""
        key = self.raw_out.data_ptr()
        if getattr(self, "_ref_t_key", None) != key:
            self._ref_t = self._ref3().transpose(1, 2).contiguous()
            self._ref_t_key = key
        return self._ref_t

    def _cand_chunk(self, bytes_per_cand):
        return max(1, min(self.eq_n, MAX_PACK_BYTES // max(1, bytes_per_cand)))

    def _q_params(self, quantizer):
        return quantizer.scale.data.view(-1), quantizer.zero_point.data.view(-1)

    # ------------------------------------------------------------------ scoring calls
    def _kalign(self, dt=None):
        """Row padding (bytes) of the operands packed for the searches: 64 when every scoring call takes all candidates in one
        launch of the streaming kernel (whose K-step is 64 bytes: q.k^T with head_dim 64 then packs and reads half the
        bytes), else the general 128.  32 for int8 / fp8 operands of K <= 32 that the window kernel takes (swin's
        49 x 49 x 32 windows: these launches stream 0.8 GB of candidate operand, half of it padding at 64)."""
        G, S, K, Sp = self._dims()
        if self.eq_n not in (64, 128, 256):
            return 128
        if dt in (I8, FP8) and K <= 32 and self._cand_chunk(G * max(S, Sp) * 32) >= self.eq_n:
            be, H = backend.get(), self._heads()
            if be.gemm_win_ok(dt, Sp, S, G, H, self.eq_n, K) and be.gemm_win_ok(dt, S, Sp, G, H, self.eq_n, K):
                return 32
        esz = 1 if dt in (I8, FP8) else 2
        whole = self._cand_chunk(G * max(S, Sp) * pad_k(K, BF16 if esz == 2 else I8, 64) * esz) >= self.eq_n
        return 64 if whole else 128

    def _pack_fixed(self, which, dt=I8):
        be = backend.get()
        H = self._heads()
        A, B = self.raw_input
        al = self._kalign(dt)
        if which == "A":
            s, z = self._q_params(self.A_quantizer)
            return be.pack_uniform(self._a3(A), s, z, 1, 0, H, 1 if H > 1 else 0, 0, self.A_quantizer.n_bits, dt, k_align=al)
        s, z = self._q_params(self.B_quantizer)
        return be.pack_uniform(self._bt3_packable(B), s, z, 1, 0, H, 1 if H > 1 else 0, 0, self.B_quantizer.n_bits, dt,
                               k_align=al)

    def _score(self, which, fixed, scale, zp, dt=I8, fixed_sa=None, sa_mul=1.0, defer=False):
        """matmul.py:135-163 (which='A') / :173-201 (which='B') -> scores [P, H].

        The candidates go into the GEMM's COLUMN axis (packed candidates-innermost): the 128 candidates of one output
        row/column share one reference element per accumulator row, and tiles are full along that axis.  For the
        A-operand search the product is evaluated transposed (D^T = B . A_p^T), reading raw_out transposed in place.
        """
        be = backend.get()
        H = self._heads()
        G, S, K, Sp = self._dims()
        A, B = self.raw_input
        P = scale.shape[0]
        src = self._a3(A) if which == "A" else self._bt3_packable(B)
        bits = self.A_quantizer.n_bits if which == "A" else self.B_quantizer.n_bits
        rows = S if which == "A" else Sp
        mixed = dt == BF16_FP8                     # fixed operand bf16 [.., 256], candidates fp8 [.., 256] (ops.gemm_mixed_ok)
        cdt = FP8 if mixed else dt                 # what the candidates are packed as
        esz = 2 if dt == BF16 else 1
        al = (64 if K <= 64 else 256) if mixed else self._kalign(dt)      # fp8 rows of the two mixed shape families
        chunk = self._cand_chunk(G * rows * pad_k(K, cdt, al) * esz)
        if mixed and chunk < P:
            raise RuntimeError("the mixed softmax.v search scores all candidates in one launch")
        pg = 1 if H > 1 else 0
        mrows, ncols = (Sp, S) if which == "A" else (S, Sp)
        if (GEN_MM != '0' and dt in (I8, FP8) and chunk >= P and K % 16 == 0 and K <= 64 and hasattr(be, "gemm_score_gen")
                and (dt != FP8 or bits <= 4) and src.dtype == torch.float32
                and be.gemm_score_gen_ok(dt, mrows, ncols, G, H, P, K, fixed.shape[-1])
                and (GEN_MM == 'all' or be.gemm_win_ok(dt, mrows, ncols, G, H, P, K))):
            sc, zc = scale.contiguous(), zp.contiguous()
            sb = Strided(sc, c=H, g=pg)
            if which == "A":
                sa, ref = Strided(self.B_quantizer.scale.data.view(-1), g=pg), self._ref3()
            else:
                sa = fixed_sa if fixed_sa is not None else Strided(self.A_quantizer.scale.data.view(-1), g=pg)
                ref = self._ref3_t()
            pend = be.gemm_score_gen(dt, fixed, src, zc, bits, mrows, ncols, P, G, H, ref, sa, sb, self.head_channel_wise,
                                     self._norm(A, S, Sp), sa_mul=sa_mul)
            return pend if defer else pend.finish()
        out = []
        for s0 in range(0, P, chunk):
            e = min(P, s0 + chunk)
            sc, zc = scale[s0:e].contiguous(), zp[s0:e].contiguous()
            cand = be.pack_uniform(src, sc, zc, e - s0, H, H, pg, 0, bits, cdt, c_inner=True, k_align=al)
            sb = Strided(sc, c=H, g=pg)
            if which == "A":
                sa = Strided(self.B_quantizer.scale.data.view(-1), g=pg)
                out.append(be.gemm_score(dt, fixed, cand, Sp, S, e - s0, G, H, self._ref3(), sa, sb, None,
                                         self.head_channel_wise, False, self._norm(A, S, Sp), sa_mul=sa_mul,
                                         ref_div=e - s0, order=2, ref_transposed=True, defer=defer and chunk >= P))
            else:
                sa = fixed_sa if fixed_sa is not None else Strided(self.A_quantizer.scale.data.view(-1), g=pg)
                out.append(be.gemm_score(dt, fixed, cand, S, Sp, e - s0, G, H, self._ref3_t(), sa, sb, None,
                                         self.head_channel_wise, False, self._norm(A, S, Sp), sa_mul=sa_mul,
                                         ref_div=e - s0, order=2, ref_transposed=True, defer=defer and chunk >= P))
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def _commit(self, quantizer, scale, zp):
        search.commit_param(quantizer.scale, scale)             # (no copy when the search's last kernel wrote them in place)
        search.commit_param(quantizer.zero_point, zp)

    def _fpcs(self, which, fpcs_width=16, steps=6, fixed=None, dt=I8, fixed_sa=None, sa_mul=1.0, checked=False):
        """matmul.py:243-262."""
        if not checked and search.round_is_redundant(self, which, self.B_quantizer if which == "A" else self.A_quantizer):
            return                                 # the other operand's quantiser is what this search saw last round
        x = self.raw_input[0] if which == "A" else self.raw_input[1]
        scale, zp, delta = search.matmul_grid(x, self.B_quantizer.n_bits, self.eq_n, self.head_channel_wise)
        if fixed is None:
            fixed = self._pack_fixed("B" if which == "A" else "A", dt)
        fn = lambda s, z, t: self._score(which, fixed, s, z, dt, fixed_sa, sa_mul, defer=True)
        q = self.A_quantizer if which == "A" else self.B_quantizer
        res = search.fpcs(scale, zp, None, delta, fn, steps, fpcs_width, self.eq_n, None,
                          commit_to=search.commit_targets(q.scale, q.zero_point, None))
        if res is not None:
            self._commit(q, res[0], res[1])

    def _init_from_grid(self, which):
        """matmul.py:266-271: parameters start at candidate [-2] of the percentile grid."""
        x = self.raw_input[0] if which == "A" else self.raw_input[1]
        scale, zp, _ = search.matmul_grid(x, self.B_quantizer.n_bits, self.eq_n, self.head_channel_wise)
        q = self.A_quantizer if which == "A" else self.B_quantizer
        self._commit(q, scale[-2], zp[-2])
        q.inited = True

    def hyperparameter_searching(self):
        """matmul.py:264-283 with fpcs=True."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        if not self.fpcs:
            raise NotImplementedError("non-FPCS search is not part of the accelerated path")
        self._initialize_calib_parameters()
        search.begin_rounds(self)
        self._init_from_grid("A")
        self._init_from_grid("B")
        G, S, K, Sp = self._dims()
        # both zero points come from the percentile grid (_init_from_grid, then FPCS commits): fp8 storage when <= 4 bit
        dt = search.int_operand_dtype(self.A_quantizer.n_bits, self.B_quantizer.n_bits,
                                      self._cand_chunk(G * max(S, Sp) * pad_k(K, I8)), prefer_fp8=K <= 64)
        for _ in range(self.search_round):
            self._fpcs("A", steps=self.steps, dt=dt)
            self._fpcs("B", steps=self.steps, dt=dt)
        self.calibrated = True
        search.begin_rounds(self)
        search.forget_grids()
        del self.raw_input, self.raw_out
        self._ref_t = self._ref_t_key = None
        self._bt_c = self._bt_key = None
        return None

    # ------------------------------------------------------------------ fused quantised forward (matmul.py:43-45)
    def _training(self):
        return torch.is_grad_enabled() and (self.A_quantizer.training_mode or self.B_quantizer.training_mode)

    def quant_forward(self, A, B, a_pre=False, b_pre=False):
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        if a_pre or b_pre or self._training() or self.A_quantizer.n_bits > 7 or self.B_quantizer.n_bits > 7:
            return super().quant_forward(A, B, a_pre, b_pre)
        be = backend.get()
        H = self._heads()
        pg = 1 if H > 1 else 0
        lead = A.shape[:-2]
        S, Sp = A.shape[-2], B.shape[-1]
        sA, zA = self._q_params(self.A_quantizer)
        sB, zB = self._q_params(self.B_quantizer)
        a3, bt3 = self._a3(A), self._bt3(B)
        ap = be.pack_uniform(a3, sA, zA, 1, 0, H, pg, 0, self.A_quantizer.n_bits, I8)
        bp = be.pack_uniform(bt3, sB, zB, 1, 0, H, pg, 0, self.B_quantizer.n_bits, I8)
        out = be.gemm_out(I8, ap, bp, S, Sp, a3.shape[0], H, Strided(sA, g=pg), Strided(sB, g=pg), None)
        return out.view(*lead, S, Sp)


class PostSoftmaxAsymmetricallyBatchingQuantMatMul(AsymmetricallyBatchingQuantMatMul):
    """softmax @ v: A is quantised with AdaLog (scale fixed to 1, log base 2^(-q/37) searched), B uniformly."""

    def __init__(self, A_bit=8, B_bit=8, mode="raw", calib_batch_size=32, search_round=1, eq_n=100,
                 head_channel_wise=True, num_heads=12, fpcs=False, steps=4, quantizer='adalog'):
        super().__init__(A_bit, B_bit, mode, calib_batch_size, search_round, eq_n, head_channel_wise, num_heads,
                         fpcs, steps)
        if quantizer != 'adalog':
            raise NotImplementedError(f"quantizer {quantizer} not implemented on the accelerated path "
                                      "(log2/logsqrt2 are ablation baselines)")
        del self.A_quantizer
        self.A_quantizer = AdaLogQuantizer(n_bits=A_bit, symmetric=False, channel_wise=False)
        self.table = torch.tensor([2 ** (-j / self.A_quantizer.r) for j in range(120)])
        self.table_scale = 1. / (4 * self.A_quantizer.n_levels - 2)
        self.table = torch.round(self.table / self.table_scale) * self.table_scale
        self.A_quantizer.scale = nn.Parameter(torch.ones([1, 1, 1, 1]))
        self.A_quantizer.inited = True
        self._q_host = 37

    def _mant37(self, device):
        return search.const_tensor(torch.round(self.table[:37] / self.table_scale).tolist(), device)

    def _ts32(self):
        return float(torch.tensor(self.table_scale, dtype=torch.float32))

    def _pack_A_adalog(self, A3, qv, scale, C, clamp_u, c_inner=False, k_align=128):
        be = backend.get()
        return be.pack_adalog(A3, scale, qv, C, 1 if C > 1 else 0, 1, 0, self.A_quantizer.n_bits,
                              self._mant37(A3.device), shift=None, clamp_u=clamp_u, c_inner=c_inner, k_align=k_align)

    def _score_A_log_base(self):
        """matmul.py:321-351 -> (q_all [P], scores [P, 1]): per-tensor output MSE of the 128 log bases q = 10..137."""
        be = backend.get()
        H = self._heads()
        pg = 1 if H > 1 else 0
        G, S, K, Sp = self._dims()
        A, B = self.raw_input
        dev = A.device
        bp = self._pack_fixed("B", BF16)
        q_all = search.const_tensor([float(i) for i in range(10, 11 + self.eq_n)], dev)[:self.eq_n]
        P = self.eq_n
        ones = search.const_tensor([1.0] * P, dev)
        if (GEN_AVQ and hasattr(be, "gemm_score_avq") and A.dtype == torch.float32
                and be.gemm_score_avq_ok(Sp, S, G, H, P, K, bp.shape[-1], self.A_quantizer.n_bits)):
            # the 128 quantisations of the probabilities are generated inside the kernel (adalog_gemm_score_avq): nothing is packed
            key = (P, self.A_quantizer.n_bits, str(dev))
            if getattr(self, "_avq_lut_key", None) != key:
                self._avq_lut, self._avq_lut_key = be.adalog_value_lut(q_all, self.A_quantizer.n_bits, self._mant37(dev)), key
            return q_all, be.gemm_score_avq(bp, self._a3(A), q_all, self._avq_lut, self.A_quantizer.n_bits, Sp, S, P, G, H,
                                            self._ref3(), Strided(self.B_quantizer.scale.data.view(-1), g=pg), Strided(ones, c=1),
                                            1.0 / (A.shape[1] * S * Sp), sa_mul=self._ts32())
        chunk = self._cand_chunk(G * S * pad_k(K, BF16, self._kalign()) * 2)
        out = []
        for s0 in range(0, P, chunk):
            e = min(P, s0 + chunk)
            # transposed product: rows = head-dim columns of v, GEMM columns = (attention row, candidate base)
            ap = self._pack_A_adalog(self._a3(A), q_all[s0:e].contiguous(), ones[s0:e].contiguous(), e - s0, False,
                                     c_inner=True, k_align=self._kalign())
            out.append(be.gemm_score(BF16, bp, ap, Sp, S, e - s0, G, H, self._ref3(),
                                     Strided(self.B_quantizer.scale.data.view(-1), g=pg), Strided(ones[s0:e].contiguous(), c=1),
                                     None, False, False, 1.0 / (A.shape[1] * S * Sp), sa_mul=self._ts32(),
                                     ref_div=e - s0, order=2, ref_transposed=True))
        return q_all, (out[0] if len(out) == 1 else torch.cat(out, 0))

    def _mixed_B_search(self):
        """bf16 rows x fp8 candidate columns for the B (v) search: needs exact fp8 candidates (<= 4 bit), one of the kernels' shape
        families (ops.gemm_mixed_ok: 197-token groups, rows of 256 elements; windows of <= 64 keys, rows of 64) and all candidates in
        one launch."""
        if not MIXED_B_SEARCH or self.B_quantizer.n_bits > 4 or self.eq_n not in (64, 128, 256):
            return False
        be = backend.get()
        G, S, K, Sp = self._dims()
        if not hasattr(be, "gemm_mixed_ok") or self._cand_chunk(G * Sp * (64 if K <= 64 else 256)) < self.eq_n:
            return False
        return bool(be.gemm_mixed_ok(S, Sp, G, self._heads(), self.eq_n, K))

    def _search_best_A_log_base(self):
        """matmul.py:321-358: score the 128 bases, commit the best."""
        be = backend.get()
        aq = self.A_quantizer
        if search.round_is_redundant(self, "Aq", self.B_quantizer):
            return
        q_all, scores = self._score_A_log_base()
        idx = search.argbest(scores, 1)
        best_q = be.fpcs_next(q_all.view(-1, 1), None, None, idx, 1, 0, None, None, None)[0]
        aq.q.data.copy_(best_q.view(aq.q.shape).to(aq.q.dtype))
        self._q_host = int(aq.q.item())                           # one host read per round (LUT rebuild)
        aq.update_table(self._q_host)

    def hyperparameter_searching(self):
        """matmul.py:360-378."""
        search.forget_grids()                     # percentile grids are memoised per search call only
        if not self.fpcs:
            raise NotImplementedError("non-FPCS search is not part of the accelerated path")
        self._initialize_calib_parameters()
        search.begin_rounds(self)
        self._init_from_grid("B")
        A = self.raw_input[0]
        dev = A.device
        for _ in range(self.search_round):
            self._search_best_A_log_base()
            if search.round_is_redundant(self, "B", self.A_quantizer):
                continue                           # same log base as last round: the B search would repeat itself
            # B search against q_A(A): eval-form AdaLog of A (clamped, scale 1) is the fixed bf16 operand.  When the shape allows
            # (197 keys, <= 4-bit v) the 128 x candidates are packed as fp8 and converted to bf16 inside the kernel: the operand
            # a launch streams is 256 instead of 448 bytes per column
            mixed = self._mixed_B_search()
            qv = search.const_tensor([float(self._q_host)], dev)
            K = A.shape[-1]
            ap = self._pack_A_adalog(self._a3(A), qv, self.A_quantizer.scale.data.view(-1), 1, True,
                                     k_align=(128 if K <= 64 else 512) if mixed else self._kalign())
            self._fpcs("B", steps=self.steps, fixed=ap, dt=BF16_FP8 if mixed else BF16,
                       fixed_sa=Strided(self.A_quantizer.scale.data.view(-1)), sa_mul=self._ts32(), checked=True)
        self.calibrated = True
        search.begin_rounds(self)
        search.forget_grids()
        del self.raw_input, self.raw_out
        self._ref_t = self._ref_t_key = None
        self._bt_c = self._bt_key = None
        return None

    def quant_forward(self, A, B, a_pre=False, b_pre=False):
        assert self.calibrated, f"Module should be calibrated before run quant_forward for {self}"
        if a_pre or b_pre or self._training() or self.A_quantizer.training_mode:
            return MinMaxQuantMatMul.quant_forward(self, A, B, a_pre, b_pre)
        be = backend.get()
        H = self._heads()
        pg = 1 if H > 1 else 0
        lead = A.shape[:-2]
        S, Sp = A.shape[-2], B.shape[-1]
        if self._q_host is None:
            self._q_host = int(self.A_quantizer.q.item())
        qv = search.const_tensor([float(self._q_host)], A.device)
        a3, bt3 = self._a3(A), self._bt3(B)
        ap = self._pack_A_adalog(a3, qv, self.A_quantizer.scale.data.view(-1), 1, True)
        sB, zB = self._q_params(self.B_quantizer)
        bp = be.pack_uniform(bt3, sB, zB, 1, 0, H, pg, 0, self.B_quantizer.n_bits, BF16)
        out = be.gemm_out(BF16, ap, bp, S, Sp, a3.shape[0], H, Strided(self.A_quantizer.scale.data.view(-1)),
                          Strided(sB, g=pg), None, sa_mul=self._ts32())
        return out.view(*lead, S, Sp)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._q_host = None
