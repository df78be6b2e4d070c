#!/usr/bin/env python3
"""Timing of one post-GELU activation-search scoring call (deit_small fc2: 32 x 197 tokens, 1536 -> 384, 128 candidates):
the fused quantise-in-loader kernel against pack_adalog + streaming GEMM.   python tools/bench_fused.py [I O T N bits]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_kernels import _postgelu_layer  # noqa: E402
from adalog_amd import backend  # noqa: E402
from adalog_amd.ops import BF16  # noqa: E402
from adalog_amd.quant_layers import linear as LM  # noqa: E402

I, Oc, T, N, bits = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (1536, 384, 197, 32, 4)))
be = backend.get()
lay, scs, qs = _postgelu_layer(I, Oc, T, N, bits, 7, 0.0)
aq = lay.a_quantizer
with torch.no_grad():
    wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
    fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data, lay.bias.data).view(-1)
    for fused in (False, True):
        LM.FUSED_ACT_SEARCH = fused
        for _ in range(3):
            s = lay._score_scale_logbase(wp, fold, scs, qs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            s = lay._score_scale_logbase(wp, fold, scs, qs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fl = 2.0 * Oc * N * T * 128 * I
        print(f"fused={fused}: {dt * 1e3:.3f} ms per call  {fl / dt / 1e12:.0f} TFLOP/s  score[0]={float(s[0]):.6e}")
