"""Lab: the kernels of one quant_forward pass (deit_small W4A4, 32 images): run under rocprofv3 --kernel-trace --stats."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import load_cfg
from adalog_amd.utils.calibrator import QuantCalibrator
from adalog_amd.utils.models import create_model
from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net

model_name = sys.argv[1] if len(sys.argv) > 1 else "deit_small"
dev = torch.device("cuda")
cfg = load_cfg(4)
cfg.search_round, cfg.steps = 1, 2                       # a quick calibration: only the forward is of interest here
torch.manual_seed(5)
model = wrap_modules_in_net(create_model(model_name).eval(), cfg, reparam=True).to(dev)
x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
QuantCalibrator(model, [(x, None)], capture="block").batching_quant_calib()
model = wrap_reparamed_modules_in_net(model).to(dev).eval()
for m in model.modules():
    if hasattr(m, "reparam_bias"):
        m.reparam_bias()
    if hasattr(m, "mode"):
        m.mode = "quant_forward"
with torch.no_grad():
    model(x)
    torch.cuda.synchronize()
    torch.tril(torch.ones(64, 64, device=dev))          # marker kernel (triu_tril): the passes after it are the measured ones
    for _ in range(int(os.environ.get("QF_REPS", "20"))):
        model(x)
    torch.cuda.synchronize()
