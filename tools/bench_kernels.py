#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the deit_small / 32-image layer shapes (run on the GPU box).

Times each kernel with events on the launch stream, prints achieved algorithmic TFLOP/s (scoring GEMMs) or GB/s
(packing / elementwise kernels).  Usage: python tools/bench_kernels.py [--reps 5] [--only qkv|proj|fc1|fc2|matmul|elem]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adalog_amd import backend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="")
args = ap.parse_args()
ops = backend.get()
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, reps=args.reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


N, T = 32, 197
M = N * T
P = 128
S = ops.Strided
for name, I, O in (("qkv", 384, 1152), ("proj", 384, 384), ("fc1", 384, 1536), ("fc2", 1536, 384)):
    if args.only and args.only not in name:
        continue
    x = torch.randn(1, M, I, device=dev)
    W = torch.randn(1, O, I, device=dev) * 0.05
    ref = torch.randn(1, M, O, device=dev)
    ref_t = ref.transpose(1, 2).contiguous()
    bias = torch.zeros(O, device=dev)
    xs, xz = torch.tensor([0.3], device=dev), torch.tensor([8.0], device=dev)
    ws, wz = torch.full((O,), 0.01, device=dev), torch.full((O,), 8.0, device=dev)
    csw = torch.rand(P, O, device=dev) * 0.01 + 0.005; czw = torch.randint(4, 12, (P, O), device=dev).float()
    csa = torch.rand(P, 1, device=dev) * 0.2 + 0.2; cza = torch.randint(4, 12, (P, 1), device=dev).float()
    fl = 2.0 * M * I * O * P
    xp = ops.pack_uniform(x, xs, xz, 1, 0, 1, 0, 0, 4, ops.I8)
    t_pw = timeit(lambda: ops.pack_uniform(W, csw, czw, P, O, 1, 0, 1, 4, ops.I8, c_inner=True))
    wp = ops.pack_uniform(W, csw, czw, P, O, 1, 0, 1, 4, ops.I8, c_inner=True)
    t_gw = timeit(lambda: ops.gemm_score(ops.I8, xp, wp, M, O, P, 1, 1, ref_t, S(xs), S(csw, c=O, n=1), S(bias, n=1), False, True,
                                         1.0 / T, ref_div=P, order=2, ref_transposed=True))
    wfix = ops.pack_uniform(W, ws, wz, 1, 0, 1, 0, 1, 4, ops.I8)
    one = torch.ones(1, device=dev)
    t_pa = timeit(lambda: ops.pack_uniform(x, csa, cza, P, 1, 1, 0, 0, 4, ops.I8, c_inner=True))
    xP = ops.pack_uniform(x, csa, cza, P, 1, 1, 0, 0, 4, ops.I8, c_inner=True)
    t_ga = timeit(lambda: ops.gemm_score(ops.I8, wfix, xP, O, M, P, 1, 1, ref, S(one), S(csa, c=1), None, False, False,
                                         1.0 / (T * O), ref_div=P, order=2, ref_transposed=True, row_scale=ws, row_bias=bias))
    print(f"{name:5s} i8  W-search: pack {t_pw*1e3:7.0f} us  gemm {t_gw*1e3:7.0f} us = {fl/t_gw/1e9:7.1f} TOPS | "
          f"A-search: pack {t_pa*1e3:7.0f} us ({P*M*I/t_pa/1e6:6.1f} GB/s out) gemm {t_ga*1e3:7.0f} us = {fl/t_ga/1e9:7.1f} TOPS", flush=True)
    if name == "fc2":
        mant = torch.arange(30, 30 - 37, -1, device=dev).float().clamp(min=15)
        qv = torch.randint(20, 60, (P,), device=dev).float()
        cs = torch.rand(P, device=dev) * 2 + 2
        sh = torch.tensor([0.17], device=dev)
        xg = torch.nn.functional.gelu(x * 2)
        t_pl = timeit(lambda: ops.pack_adalog(xg, cs, qv, P, 1, 1, 0, 4, mant, sh, True, c_inner=True))
        xL = ops.pack_adalog(xg, cs, qv, P, 1, 1, 0, 4, mant, sh, True, c_inner=True)
        wb = ops.pack_uniform(W, ws, wz, 1, 0, 1, 0, 1, 4, ops.BF16)
        t_gl = timeit(lambda: ops.gemm_score(ops.BF16, wb, xL, O, M, P, 1, 1, ref, S(one), S(cs, c=1), None, False, False,
                                             1.0 / (T * O), sa_mul=1 / 30.0, ref_div=P, order=2, ref_transposed=True,
                                             row_scale=ws, row_bias=bias))
        print(f"{name:5s} bf16 A-search (AdaLog): pack {t_pl*1e3:7.0f} us ({2*P*M*I/t_pl/1e6:6.1f} GB/s out)  gemm {t_gl*1e3:7.0f} us "
              f"= {fl/t_gl/1e9:7.1f} TFLOPS", flush=True)
if not args.only or "matmul" in args.only:
    H, C = 6, 64
    G = N * H
    A = torch.randn(N, H, T, C, device=dev); Bk = torch.randn(N, H, T, C, device=dev)
    ref = torch.randn(G, T, T, device=dev)
    cs = torch.rand(P, H, device=dev) * 0.2 + 0.2; cz = torch.randint(4, 12, (P, H), device=dev).float()
    fs, fz = torch.full((H,), 0.3, device=dev), torch.full((H,), 8.0, device=dev)
    a3 = A.reshape(G, T, C); bt3 = Bk.reshape(G, T, C)
    bfix = ops.pack_uniform(bt3, fs, fz, 1, 0, H, 1, 0, 4, ops.I8)
    t_p = timeit(lambda: ops.pack_uniform(a3, cs, cz, P, H, H, 1, 0, 4, ops.I8, c_inner=True))
    aP = ops.pack_uniform(a3, cs, cz, P, H, H, 1, 0, 4, ops.I8, c_inner=True)
    t_g = timeit(lambda: ops.gemm_score(ops.I8, bfix, aP, T, T, P, G, H, ref, S(fs, g=1), S(cs, c=H, g=1), None, True, False,
                                        1.0 / (T * T), ref_div=P, order=2, ref_transposed=True))
    fl = 2.0 * G * T * T * C * P
    print(f"qk^T  i8  A-search: pack {t_p*1e3:7.0f} us  gemm {t_g*1e3:7.0f} us = {fl/t_g/1e9:7.1f} TOPS", flush=True)
    Asm = torch.softmax(torch.randn(N, H, T, T, device=dev), -1); V = torch.randn(N, H, T, C, device=dev)
    ref2 = torch.randn(G, T, C, device=dev)
    mant = torch.arange(30, 30 - 37, -1, device=dev).float().clamp(min=15)
    qv = torch.arange(10, 10 + P, device=dev).float(); ones = torch.ones(P, device=dev)
    vt3 = V.reshape(G, T, C).transpose(1, 2)
    vfix = ops.pack_uniform(vt3, fs, fz, 1, 0, H, 1, 0, 4, ops.BF16)
    t_p = timeit(lambda: ops.pack_adalog(Asm.reshape(G, T, T), ones, qv, P, 1, 1, 0, 4, mant, None, False, c_inner=True))
    aL = ops.pack_adalog(Asm.reshape(G, T, T), ones, qv, P, 1, 1, 0, 4, mant, None, False, c_inner=True)
    t_g = timeit(lambda: ops.gemm_score(ops.BF16, vfix, aL, C, T, P, G, H, ref2, S(fs, g=1), S(ones, c=1), None, False, False,
                                        1.0 / (H * T * C), sa_mul=1 / 30.0, ref_div=P, order=2, ref_transposed=True))
    fl = 2.0 * G * T * T * C * P
    print(f"sm@v  bf16 log-base: pack {t_p*1e3:7.0f} us  gemm {t_g*1e3:7.0f} us = {fl/t_g/1e9:7.1f} TFLOPS", flush=True)
    t_p = timeit(lambda: ops.pack_uniform(vt3, cs, cz, P, H, H, 1, 0, 4, ops.BF16))
    print(f"sm@v  bf16 B-cand pack (transposed source): {t_p*1e3:7.0f} us", flush=True)
if not args.only or "elem" in args.only:
    x = torch.randn(32, 197, 1536, device=dev)
    n = x.numel()
    s1, z1 = torch.tensor([0.1], device=dev), torch.tensor([8.0], device=dev)
    t = timeit(lambda: ops.uniform_fake_quant(x, s1, z1, 4))
    print(f"uniform fq   {n/1e6:.1f} M elems: {t*1e3:6.0f} us = {8*n/t/1e6:7.1f} GB/s (8 B/elem)")
    q = torch.tensor([37], device=dev)
    t1 = torch.arange(16, device=dev).float(); t2 = torch.ones(16, device=dev)
    xg = torch.nn.functional.gelu(x)
    t = timeit(lambda: ops.log_fake_quant(xg, torch.tensor([3.0], device=dev), q, t1, t2, 4, shift=torch.tensor([0.17], device=dev), sub_shift=True))
    print(f"adalog fq    {n/1e6:.1f} M elems: {t*1e3:6.0f} us = {8*n/t/1e6:7.1f} GB/s (8 B/elem)")
    x2 = x.view(-1, 1536)
    cs = torch.rand(P, 1, device=dev) * 0.2 + 0.1; cz = torch.randint(4, 12, (P, 1), device=dev).float()
    t = timeit(lambda: ops.score_a_self(x2, cs, cz, False, 4, 1.0))
    print(f"a self-MSE   {n/1e6:.1f} M elems x128 cands: {t*1e3:6.0f} us = {4*n/t/1e6:7.1f} GB/s (4 B/elem once)")
    qs = [0.9, 1.0, 0.1, 0.0]
    t = timeit(lambda: ops.quantile_rows(x.view(1, -1), qs, 1))
    print(f"quantile     {n/1e6:.1f} M elems: {t*1e3:6.0f} us = {16*n/t/1e6:7.1f} GB/s (4 passes x 4 B)")
