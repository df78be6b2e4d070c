"""Accuracy and kernel time of fp32 / 'allow_tf32' / bf16-split GEMMs on the BRECQ shapes (run under rocprofv3 for kernel times)."""
import torch
dev = "cuda"
shapes = [("qkv fwd", 6304, 1152, 384), ("fc2 fwd", 6304, 384, 1536), ("fc1 dW", 1536, 384, 6304)]
def relerr(y, ref): return ((y.double() - ref).abs().max() / ref.abs().max()).item(), ((y.double() - ref).norm() / ref.norm()).item()
for name, M, N, K in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
    ref = a.double() @ b.double().t()
    torch.backends.cuda.matmul.allow_tf32 = False
    for _ in range(3): y32 = a @ b.t()
    torch.backends.cuda.matmul.allow_tf32 = True
    for _ in range(3): ytf = a @ b.t()
    torch.backends.cuda.matmul.allow_tf32 = False
    ah = a.bfloat16(); al = (a - ah.float()).bfloat16(); bh = b.bfloat16(); bl = (b - bh.float()).bfloat16()
    for _ in range(3):
        y3 = torch.mm(ah, bh.t(), out_dtype=torch.float32) + torch.mm(ah, bl.t(), out_dtype=torch.float32) + torch.mm(al, bh.t(), out_dtype=torch.float32)
    # K-concatenated single call
    A3 = torch.cat([ah, ah, al], 1); B3 = torch.cat([bh, bl, bh], 1)
    for _ in range(3): y3c = torch.mm(A3, B3.t(), out_dtype=torch.float32)
    y1 = torch.mm(ah, bh.t(), out_dtype=torch.float32)
    print(name, "fp32", relerr(y32, ref), "tf32flag", relerr(ytf, ref), "bf16x3", relerr(y3, ref), "bf16x3cat", relerr(y3c, ref), "bf16", relerr(y1, ref), flush=True)
torch.cuda.synchronize()
