"""adalog_gemm_f32x3 on one product shape in the four operand orientations (N = K-contiguous rows, T = K-major): what the
LDS-DMA request pattern of an orientation costs (an N-form stage holds 64 bytes of each row: half cache lines)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adalog_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, N, K in ((6304, 1536, 384), (6304, 1152, 384), (6304, 384, 1536), (6304, 384, 384)):
    a_n = torch.randn(M, K, device=dev, generator=g)
    b_n = torch.randn(N, K, device=dev, generator=g)
    a_t = a_n.t().contiguous().t()          # same values, K-major storage
    b_t = b_n.t().contiguous().t()
    ref = a_n.double() @ b_n.double().t()
    out = []
    for la, a in (("N", a_n), ("T", a_t)):
        for lb, b in (("N", b_n), ("T", b_t)):
            us = timed(lambda: ops.gemm_f32x3(a, b))
            err = ((ops.gemm_f32x3(a, b).double() - ref).abs().max() / ref.abs().max()).item()
            out.append(f"{la}{lb} {us:6.1f} us (err {err:.1e})")
    print(f"{M}x{N}x{K}: " + "   ".join(out))
