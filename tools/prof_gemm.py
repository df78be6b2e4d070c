#!/usr/bin/env python3
"""Launch a handful of scoring-GEMM calls at deit_small shapes (for rocprofv3 --pmc / --kernel-trace runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adalog_amd import backend  # noqa: E402

ops = backend.get()
dev = "cuda"
torch.manual_seed(0)
N, T, P = 32, 197, 128
M = N * T
S = ops.Strided
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
if which == "qk":                   # attention q.k^T search: 192 (image, head) groups, fp8 storage, group kernel
    G, H, D = N * 6, 6, 64
    q = torch.randn(G, T, D, device=dev)
    k = torch.randn(G, T, D, device=dev)
    sq, zq = torch.full((H,), 0.3, device=dev), torch.full((H,), 8.0, device=dev)
    cs = torch.rand(P, H, device=dev) * 0.2 + 0.2
    cz = torch.randint(4, 12, (P, H), device=dev).float()
    qp = ops.pack_uniform(q, sq, zq, 1, 0, H, 1, 0, 4, ops.FP8, k_align=64)
    kc = ops.pack_uniform(k, cs, cz, P, H, H, 1, 0, 4, ops.FP8, c_inner=True, k_align=64)
    ref3 = torch.randn(G, T, T, device=dev)
    for _ in range(3):
        ops.gemm_score(ops.FP8, qp, kc, T, T, P, G, H, ref3, S(sq, g=1), S(cs, c=H, g=1), None, True, False, 1.0,
                       ref_div=P, order=2, ref_transposed=True)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
I, O = {"qkv": (384, 1152), "fc1": (384, 1536), "fc2": (1536, 384), "proj": (384, 384)}[which]
DT = ops.FP8 if os.environ.get("PROF_DT", "i8") == "fp8" else ops.I8      # operand storage (fp8 = what K <= 768 searches use)
x = torch.randn(1, M, I, device=dev)
W = torch.randn(1, O, I, device=dev) * 0.05
ref = torch.randn(1, M, O, device=dev)
bias = torch.zeros(O, device=dev)
xs, xz = torch.tensor([0.3], device=dev), torch.tensor([8.0], device=dev)
ws, wz = torch.full((O,), 0.01, device=dev), torch.full((O,), 8.0, device=dev)
csw = torch.rand(P, O, device=dev) * 0.01 + 0.005; czw = torch.randint(4, 12, (P, O), device=dev).float()
csa = torch.rand(P, 1, device=dev) * 0.2 + 0.2; cza = torch.randint(4, 12, (P, 1), device=dev).float()
xp = ops.pack_uniform(x, xs, xz, 1, 0, 1, 0, 0, 4, DT)
wp = ops.pack_uniform(W, csw, czw, P, O, 1, 0, 1, 4, DT, c_inner=True)
wfix = ops.pack_uniform(W, ws, wz, 1, 0, 1, 0, 1, 4, DT)
xP = ops.pack_uniform(x, csa, cza, P, 1, 1, 0, 0, 4, DT, c_inner=True)
ref_t = ref.transpose(1, 2).contiguous()
one = torch.ones(1, device=dev)
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
for _ in range(3):
    if mode in ("both", "w"):       # weight search: rows = tokens, columns = (out channel, candidate)
        ops.gemm_score(DT, xp, wp, M, O, P, 1, 1, ref_t, S(xs), S(csw, c=O, n=1), S(bias, n=1), False, True, 1.0 / T,
                       ref_div=P, order=2, ref_transposed=True)
    if mode == "g":                 # activation search, candidate operand generated inside the slab kernel (round 3 default)
        ops.score_act_gen(DT, wfix, x[0], csa, cza, 4, ref[0], ws, bias, 1.0 / (T * O))
    if mode in ("both", "a"):       # activation search (transposed): rows = out channels, columns = (token, candidate)
        ops.gemm_score(DT, wfix, xP, O, M, P, 1, 1, ref, S(one), S(csa, c=1), None, False, False, 1.0 / (T * O),
                       ref_div=P, order=2, ref_transposed=True, row_scale=ws, row_bias=bias)
torch.cuda.synchronize()
print("done")
