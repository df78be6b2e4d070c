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
base = create_model("deit_small", depth=int(os.environ.get("DEPTH", "1"))).eval()
from adalog_amd import search as _search
from adalog_amd.quant_layers import linear as _linear
if os.environ.get("REF_SCHEDULE", "1") == "1":
    _search.SKIP_CONVERGED, _linear.RUN_DEAD_W_SELF = False, True
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
for rep in range(2):
    model = wrap_modules_in_net(__import__("copy").deepcopy(base), cfg, reparam=True).to(dev)
    if rep == 0:
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
        torch.cuda.synchronize()
        continue
    import traceback
    from torch.overrides import TorchFunctionMode

    WATCH = {"copy_", "to", "clone", "contiguous", "float", "cat", "stack", "zeros", "zeros_like", "full", "tensor", "repeat",
             "index_select", "fill_", "zero_", "expand_as", "double", "int", "long", "item", "tolist", "cpu",
             "add", "sub", "mul", "div", "neg", "mean", "round", "pow", "__getitem__", "repeat_interleave", "equal", "all", "__eq__",
             "eq", "sum", "empty", "empty_like", "ones", "arange", "linspace", "where", "abs", "clamp"}
    cnt = collections.Counter()

    class Spy(TorchFunctionMode):
        def __torch_function__(self, func, types, args=(), kwargs=None):
            name = getattr(func, "__name__", str(func))
            big = None
            if os.environ.get("BIG_COPIES") and name in ("contiguous", "clone", "copy_", "reshape", "to", "flatten", "view_as", "t", "float"):
                for a in args[:2]:
                    if torch.is_tensor(a) and a.is_cuda and a.numel() >= 200_000 and not a.is_contiguous():
                        big = (tuple(a.shape), tuple(a.stride()))
                        break
            if big is not None and name in ("contiguous", "clone", "reshape", "copy_", "flatten"):
                fr = [f for f in traceback.extract_stack()[:-1] if "adalog_amd" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1]) if fr else "?"
                cnt[("BIG " + name + " " + str(big), where)] += 1
            if name in WATCH:
                fr = [f for f in traceback.extract_stack()[:-1] if "adalog_amd" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1]) if fr else "?"
                cnt[(name, where)] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
        torch.cuda.synchronize()
    for (name, where), n in cnt.most_common(150):
        print(f"{n:6d}  {name:14s} {where}")
