"""adalog_gemm_f32x3 (csrc/brecq_gemm.hip) against fp64 and against rocBLAS fp32 on the products of a BRECQ iteration.

    python tools/lab/bq_gemm_bench.py [--model deit_small|vit_base] [--images 32] [--iters 50] [--terms 2|3] [--int-act]

--terms: bf16 terms per general operand (2 = what a BRECQ iteration issues since round 6: 3 products; 3 = six products, fp32-class error).
--int-act: the activation operand holds small integers (the uniformly fake-quantised input of qkv / proj / fc1: exact in one term),
as in the forward and dL/dw products of those layers.
"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adalog_amd import ops


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="deit_small")
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--terms", type=int, default=3, choices=(2, 3))
    ap.add_argument("--int-act", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bq_gemm_bench.json"))
    args = ap.parse_args()
    D, H = {"deit_tiny": (192, 3), "deit_small": (384, 6), "vit_base": (768, 12)}[args.model]
    T = 197
    M = args.images * T
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    layers = [("qkv", D, 3 * D), ("proj", D, D), ("fc1", D, 4 * D), ("fc2", 4 * D, D)]
    rows = []
    tot_new = tot_ref = 0.0
    ex_gen = 2 if args.terms == 2 else 0
    for name, I, O in layers:
        int_act = args.int_act and name != "fc2"           # fc2's input is AdaLog-quantised: not integers
        x = torch.randn(M, I, device=dev, generator=g)
        if int_act:
            x = torch.round(x * 4).clamp(-8, 7)
        ex_x = 1 if int_act else ex_gen
        w = torch.randn(O, I, device=dev, generator=g) * 0.05
        b = torch.randn(O, device=dev, generator=g)
        gy = torch.randn(M, O, device=dev, generator=g)
        cases = [
            ("fwd", lambda: ops.gemm_f32x3(x, w, b, exact_a=ex_x, exact_b=ex_gen), lambda: torch.addmm(b, x, w.t()), lambda: x.double() @ w.double().t() + b.double(), M, O, I),
            ("dx", lambda: ops.gemm_f32x3(gy, w.t(), exact_a=ex_gen, exact_b=ex_gen), lambda: gy @ w, lambda: gy.double() @ w.double(), M, I, O),
            ("dw", lambda: ops.gemm_f32x3(gy.t(), x.t(), exact_a=ex_gen, exact_b=ex_x), lambda: gy.t() @ x, lambda: gy.double().t() @ x.double(), O, I, M),
        ]
        for cname, fn, ref, exact, m_, n_, k_ in cases:
            out = fn()
            ex = exact()
            err = ((out.double() - ex).abs().max() / ex.abs().max()).item()
            err_ref = ((ref().double() - ex).abs().max() / ex.abs().max()).item()
            t_new, t_ref = timeit(fn, args.iters), timeit(ref, args.iters)
            from adalog_amd import _lib
            kern = _lib.load().adalog_last_kernel().decode()
            flops = 2.0 * m_ * n_ * k_
            one_exact = int_act and cname != "dx"                       # forward and dL/dw read the integer activation
            nprod = (3 if one_exact else 6) if args.terms == 3 else (2 if one_exact else 3)
            rows.append(dict(layer=name, product=cname, M=m_, N=n_, K=k_, us=round(t_new, 1), rocblas_us=round(t_ref, 1),
                             tflops_fp32_equiv=round(flops / t_new / 1e6, 1), bf16_tflops=round(nprod * flops / t_new / 1e6, 1), products=nprod,
                             rel_err=err, rocblas_rel_err=err_ref, kernel=kern))
            tot_new += t_new
            tot_ref += t_ref
            print(f"{name:5s} {cname:3s} {m_:5d}x{n_:5d}x{k_:5d}  {t_new:7.1f} us ({nprod} products, {nprod * flops / t_new / 1e6:7.1f} bf16 TF/s)  rocBLAS {t_ref:7.1f} us"
                  f"   err {err:.2e} (rocBLAS {err_ref:.2e})  {kern}", flush=True)
    print(f"sum of the 12 products: {tot_new:.1f} us   rocBLAS: {tot_ref:.1f} us")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(model=args.model, images=args.images, terms=args.terms, int_act=args.int_act, sum_us=tot_new, rocblas_sum_us=tot_ref, rows=rows), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
