#!/usr/bin/env python3
"""Lab probe: time the uniform weight-candidate pack (per-row parameters) at the deit_small fc2 / qkv shapes per storage type."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adalog_amd import backend  # noqa: E402
ops = backend.get()
dev = "cuda"
torch.manual_seed(0)
P = 128
for name, O, I, dt, esz in (("fc2 bf16", 384, 1536, ops.BF16, 2), ("qkv fp8", 1152, 384, ops.FP8, 1), ("fc1 fp8", 1536, 384, ops.FP8, 1), ("fc2 i8", 384, 1536, ops.I8, 1)):
    W = (torch.randn(1, O, I, device=dev) * 0.05)
    sc = torch.rand(P, O, device=dev) * 0.01 + 0.005
    zc = torch.randint(4, 12, (P, O), device=dev).float()
    for ci in (True,):
        f = lambda: ops.pack_uniform(W, sc, zc, P, O, 1, 0, 1, 4, dt, c_inner=ci)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = out.numel() * esz
        print(f"{name:10s} c_inner={ci}: {ms*1e3:7.1f} us  {nbytes/1e6:6.1f} MB written  {nbytes/ms/1e9:5.2f} TB/s")
