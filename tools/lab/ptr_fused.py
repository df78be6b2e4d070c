#!/usr/bin/env python3
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_kernels import _postgelu_layer
from adalog_amd import backend
from adalog_amd.ops import BF16
be = backend.get()
lay, scs, qs = _postgelu_layer(1536, 384, 197, 4, 4, 7, 0.0)
aq = lay.a_quantizer
with torch.no_grad():
    wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
    fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data, lay.bias.data).view(-1)
    x2 = lay._x2(); lx = lay._log2_x(); ref = lay.raw_out.reshape(-1, 384)
    print("ref", hex(ref.data_ptr()), "size", hex(ref.numel() * 4), "W", hex(wp.data_ptr()), "x", hex(x2.data_ptr()), "L", hex(lx.data_ptr()),
          "rs", hex(lay.w_quantizer.scale.data.data_ptr()), "fold", hex(fold.data_ptr()), flush=True)
    s = lay._score_scale_logbase(wp, fold, scs, qs)
    torch.cuda.synchronize()
    print("ok", float(s[0]))
