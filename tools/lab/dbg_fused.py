#!/usr/bin/env python3
"""Debug build only (FUSED_DEBUG): calls adalog_score_act_fused with a visible workspace and prints the scalars the asm kernel dumped."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_kernels import _postgelu_layer
from adalog_amd import backend, _lib
from adalog_amd.ops import BF16
be = backend.get()
lay, scs, qs = _postgelu_layer(1536, 384, 197, 4, 4, 7, 0.0)
aq = lay.a_quantizer
with torch.no_grad():
    wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
    fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data, lay.bias.data).view(-1)
    x2 = lay._x2(); lx = lay._log2_x(); ref = lay.raw_out.reshape(-1, 384)
    lib = _lib.load()
    nb = lib.adalog_score_act_fused_workspace_bytes(T, Kp)
    ws = torch.zeros(nb // 8, dtype=torch.float64, device="cuda")
    scores = torch.empty(128, device="cuda")
    rs = lay.w_quantizer.scale.data.view(-1).contiguous()
    rc = lib.adalog_score_act_fused(wp.data_ptr(), 384, wp.shape[-1], x2.data_ptr(), lx.data_ptr(), x2.shape[0], 1536, ref.data_ptr(),
                                    rs.data_ptr(), fold.data_ptr(), scs.data_ptr(), qs.data_ptr(), 128, 4, lay._mant37(x2.device).data_ptr(),
                                    0.16997124254703522, 1, lay._ts32(), 1.0, ws.data_ptr(), nb, scores.data_ptr(), None)
    torch.cuda.synchronize()
    d = ws.view(torch.int32)[:64].cpu().tolist()
    names = ["pRef_lo", "pRef_hi", "t20", "t21", "M", "T", "t2(tok0)", "t3(m0)", "c_tile", "n_rt", "w", "oFin", "c_pair", "c_rt", "pair0", "rt0"]
    for wv in range(4):
        print("wave", wv, " ".join(f"{n}={d[16 * wv + i] & 0xffffffff:x}" for i, n in enumerate(names)))
    print("ref ptr   ", hex(ref.data_ptr()), "W ptr", hex(wp.data_ptr()))
