#!/usr/bin/env python3
"""Phase timers of the hand-scheduled fused kernel (build the loop with FUSED_PROF=1 first):
   FUSED_PROF=1 FUSED_NRB=12 FUSED_FNS=4 python tools/gen_fused_asm.py && make -C adalog_amd/csrc
   python tools/lab/prof_fused.py        -> average shader cycles per K-step: barrier wait, unit 0, unit 1, between steps
(the kernel then returns cycle counts instead of scores: regenerate without FUSED_PROF afterwards)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_kernels import _postgelu_layer
from adalog_amd import backend, _lib
from adalog_amd.ops import BF16
be = backend.get()
I, O, T, N = 1536, 384, 197, 32
lay, scs, qs = _postgelu_layer(I, O, T, N, 4, 7, 0.0)
aq = lay.a_quantizer
with torch.no_grad():
    wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
    fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data, lay.bias.data).view(-1)
    x2 = lay._x2(); lx = lay._log2_x(); ref = lay.raw_out.reshape(-1, O).contiguous()
    lib = _lib.load()
    Tt, Kp = x2.shape[0], wp.shape[-1]
    nb = lib.adalog_score_act_fused_workspace_bytes(Tt, Kp)
    ws = torch.zeros((nb + 7) // 8, dtype=torch.float64, device="cuda")
    scores = torch.empty(128, device="cuda")
    rs = lay.w_quantizer.scale.data.view(-1).contiguous()
    for _ in range(3):
        rc = lib.adalog_score_act_fused(wp.data_ptr(), O, Kp, x2.data_ptr(), lx.data_ptr(), Tt, I, ref.data_ptr(),
                                        rs.data_ptr(), fold.data_ptr(), scs.data_ptr(), qs.data_ptr(), 128, 4, lay._mant37(x2.device).data_ptr(),
                                        float(aq.shift.item()), 1, lay._ts32(), 1.0, ws.data_ptr(), nb, scores.data_ptr(), None)
    torch.cuda.synchronize()
    nwg = torch.cuda.get_device_properties(0).multi_processor_count
    acc = ws[:nwg * 128].view(nwg, 128).cpu()
    nk = Kp * 2 // 64
    ntile = (Tt + 1) // 2
    tot = {"barrier": 0.0, "unit0": 0.0, "unit1": 0.0, "between": 0.0}
    steps = 0
    for b in range(nwg):
        st = len(range(b, ntile, nwg)) * nk
        steps += st
        tot["barrier"] += (acc[b, 0] + acc[b, 64]).item() / 4
        tot["unit1"] += (acc[b, 16] + acc[b, 64 + 16]).item() / 4
        tot["unit0"] += (acc[b, 32] + acc[b, 64 + 32]).item() / 4
        tot["between"] += (acc[b, 48] + acc[b, 64 + 48]).item() / 4
    print("average cycles per K-step and wave:", {k: round(v_ / steps, 1) for k, v_ in tot.items()}, "sum", round(sum(tot.values()) / steps, 1))
