"""Lab aid: aten::copy_ / aten::contiguous / aten::clone calls by input shape during one block's calibration (reference schedule)."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import load_cfg
from adalog_amd.utils.calibrator import QuantCalibrator
from adalog_amd.utils.models import create_model
from adalog_amd.utils.wrap_net import wrap_modules_in_net
from adalog_amd import search as _search
from adalog_amd.quant_layers import linear as _linear
_search.SKIP_CONVERGED, _linear.RUN_DEAD_W_SELF = False, True
dev = torch.device("cuda")
cfg = load_cfg(4)
torch.manual_seed(5)
base = create_model("deit_small", depth=1).eval()
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
import copy
for rep in range(2):
    model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
    if rep == 0:
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib(); torch.cuda.synchronize(); continue
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib(); torch.cuda.synchronize()
    cnt = collections.Counter()
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy") and e.input_shapes:
            n = 1
            for d in (e.input_shapes[0] or []): n *= d
            if n >= 200_000:
                st = [s for s in (e.stack or []) if "adalog_amd" in s][:3]
                cnt[(e.name, str(e.input_shapes[:2]), " <- ".join(x.split("adalog_amd/")[-1][:60] for x in st))] += 1
    for k, v in cnt.most_common(25):
        print(v, k)
