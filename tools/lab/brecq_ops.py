"""Lists the ATen ops / kernels of BRECQ iterations on one deit_small block (torch.profiler), to see what to fuse."""
import copy, importlib.util, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adalog_amd.utils.block_recon import BlockReconstructor
from adalog_amd.utils.calibrator import QuantCalibrator
from adalog_amd.utils.models import create_model
from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
spec = importlib.util.spec_from_file_location("cfg", os.path.join(ROOT, "configs", "4bit.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
cfg = mod.Config()
dev = torch.device("cuda")
torch.manual_seed(5)
base = create_model(os.environ.get("MODEL", "deit_small"), depth=1).eval()
full = copy.deepcopy(base).to(dev).eval()
model = wrap_modules_in_net(base, cfg, reparam=True).to(dev)
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
model = wrap_reparamed_modules_in_net(model)
for m in model.modules():
    if hasattr(m, "reparam_bias"):
        m.reparam_bias()
opt = torch.randn(64, 3, 224, 224, generator=torch.Generator().manual_seed(6)).to(dev)
rec = BlockReconstructor(model, full, [(opt[i:i + 32], None) for i in range(0, 64, 32)])
name = "blocks.0"
block, fblock = rec.blocks[name], rec.full_blocks[name]
rec.init_block_raw_data(block, fblock, name, dev)
from torch.profiler import profile, ProfilerActivity
N = int(os.environ.get("ITERS", "40"))
SHAPES = os.environ.get("SHAPES", "0") != "0"
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=SHAPES) as prof:
    rec.reconstruct_single_block(name, block, dev, quant_act=True, iters=N)
    torch.cuda.synchronize()
rows = prof.key_averages()
if SHAPES:          # the small ATen ops by input shape: which copies / adds / fills an iteration still holds
    for r in sorted(prof.key_averages(group_by_input_shape=True), key=lambda r: -getattr(r, "device_time_total", 0)):
        if r.key in ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::cat", "aten::clone", "aten::sum",
                     "aten::contiguous", "aten::zero_", "aten::zeros", "aten::index_select", "aten::mul_", "aten::div", "aten::stack"):
            print(f"{r.count / N:6.2f}/it dev {r.device_time_total / N:7.1f} us/it  {r.key:18s} {str(r.input_shapes)[:110]}")
out = []
for r in sorted(rows, key=lambda r: -getattr(r, "device_time_total", 0)):
    dt = getattr(r, "device_time_total", 0)
    out.append(f"{r.count / N:7.2f}/it  dev {dt / N:8.1f} us/it  cpu {r.self_cpu_time_total / N:8.1f} us/it  {r.key[:100]}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "brecq_ops.txt"), "w").write("\n".join(out))
print("\n".join(out[:70]))
