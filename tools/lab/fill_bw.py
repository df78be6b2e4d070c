import torch, time
x = torch.empty(2_480_000_000 // 2, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(lambda: x.fill_(1.0)); print(f"fill 2.48 GB: {t*1e3:.0f} us = {2.48e9/t/1e9:.0f} GB/s... {2.48/t*1e3:.2f} TB/s")
t = timeit(lambda: y.copy_(x)); print(f"copy 2.48 GB: {t*1e3:.0f} us = {2*2.48/t*1e3:.2f} TB/s (r+w)")
z = torch.empty(620_000_000, dtype=torch.float32, device="cuda")
t = timeit(lambda: torch.neg(x, out=y)); print(f"neg bf16 2.48 GB: {t*1e3:.0f} us = {2*2.48/t*1e3:.2f} TB/s (r+w)")
