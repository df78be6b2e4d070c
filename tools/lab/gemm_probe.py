"""Times the plain GEMM shapes of one BRECQ iteration (deit_small block) under the BLAS back ends torch offers."""
import torch, time
dev = "cuda"
shapes = [("qkv fwd", 6304, 1152, 384), ("proj fwd", 6304, 384, 384), ("fc1 fwd", 6304, 1536, 384), ("fc2 fwd", 6304, 384, 1536),
          ("qkv dW", 1152, 384, 6304), ("fc1 dW", 1536, 384, 6304), ("fc2 dW", 384, 1536, 6304), ("qkv dX", 6304, 384, 1152)]

def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

for lib in ("default", "hipblaslt"):
    if lib != "default":
        try: torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e: print("cannot select", lib, e); continue
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
        us = bench(lambda: a @ b.t())
        ab, bb = a.bfloat16(), b.bfloat16()
        us16 = bench(lambda: ab @ bb.t())
        try:
            us16f = bench(lambda: torch.mm(ab, bb.t(), out_dtype=torch.float32))
        except Exception as e:
            us16f = float("nan")
        print(f"{lib:9s} {name:8s} M{M} N{N} K{K}: fp32 {us:7.1f} us ({2*M*N*K/us/1e6:6.1f} TF)  bf16 {us16:6.1f} us  bf16->f32 {us16f:6.1f} us", flush=True)
torch.backends.cuda.matmul.allow_tf32 = True
for name, M, N, K in shapes[:3]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    print("tf32 flag", name, bench(lambda: a @ b.t()))
