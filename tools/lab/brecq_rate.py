import os, sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
print(json.dumps(bench.brecq_rate("deit_small", 4, dev, iters=int(os.environ.get("ITERS", "1500")))))
