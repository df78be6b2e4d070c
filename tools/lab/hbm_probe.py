import sys, torch, json
sys.path.insert(0, "/root/repo")
import bench
from adalog_amd import backend
ops = backend.get()
for r in bench.hbm_kernels(ops, torch.device("cuda")): print(r)
x = torch.randn(32*197*1536, device="cuda"); y = torch.empty_like(x)
import time
torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): y.copy_(x)
b.record(); torch.cuda.synchronize(); print("torch copy GB/s", 2*x.numel()*4*20/a.elapsed_time(b)/1e6)
