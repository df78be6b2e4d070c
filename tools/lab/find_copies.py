"""Lab aid (not product): which host-side ops launch device-to-device copies / fills during one block's calibration.
Usage on the GPU box: python tools/lab/find_copies.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import load_cfg                                                        # noqa: E402
from adalog_amd.utils.calibrator import QuantCalibrator                           # noqa: E402
from adalog_amd.utils.models import create_model                                  # noqa: E402
from adalog_amd.utils.wrap_net import wrap_modules_in_net                         # noqa: E402

dev = torch.device("cuda")
cfg = load_cfg(4)
torch.manual_seed(5)
base = create_model("deit_small", depth=1).eval()
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
for rep in range(2):
    model = wrap_modules_in_net(__import__("copy").deepcopy(base), cfg, reparam=True).to(dev)
    if rep == 0:
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
        torch.cuda.synchronize()
        continue
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                with_stack=True) as prof:
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::to",
                       "aten::_to_copy"):
            stack = [s for s in (ev.stack or []) if ".py" in s and "torch/" not in s]
            cnt[(ev.name, " <- ".join(st.split("/")[-1] for st in stack[:3]) if stack else "?")] += 1
    for (name, where), n in cnt.most_common(40):
        print(f"{n:6d}  {name:18s} {where}")
