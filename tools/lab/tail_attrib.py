"""Lab aid (not product): attribute the NON-scoring launches of one calibration (copies, fills, ATen elementwise kernels, the
package's own small kernels) to the host call site that issued them.  torch.profiler with Python stacks over ONE warm calibration on
the reference schedule; prints launches and GPU microseconds per (kernel, aten op, first adalog_amd frame).
Usage on the GPU box: python tools/lab/tail_attrib.py [model] [depth] > gpurun_out/tail_attrib.txt"""
import collections
import copy
import os
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import load_cfg                                                        # noqa: E402
from adalog_amd import search as _search                                          # noqa: E402
from adalog_amd.quant_layers import linear as _linear                             # noqa: E402
from adalog_amd.utils.calibrator import QuantCalibrator                           # noqa: E402
from adalog_amd.utils.models import create_model                                  # noqa: E402
from adalog_amd.utils.wrap_net import wrap_modules_in_net                         # noqa: E402

model_name = sys.argv[1] if len(sys.argv) > 1 else "deit_small"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda")
cfg = load_cfg(4)
torch.manual_seed(5)
base = wrap_modules_in_net(create_model(model_name, depth=depth).eval(), cfg, reparam=True).to(dev)
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
_search.SKIP_CONVERGED, _linear.RUN_DEAD_W_SELF = False, True                    # the reference schedule (bench.py's `value`)

QuantCalibrator(copy.deepcopy(base), [(imgs, None)], capture="block").batching_quant_calib()     # warm
torch.cuda.synchronize()
model = copy.deepcopy(base)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
    torch.cuda.synchronize()

SCORING = ("k_act_fused", "k_gemm_", "k_ga_quad", "k_ga_rect", "k_gram_score")
agg = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    frames = [f for f in (e.stack or []) if "adalog_amd" in f or "bench.py" in f]
    where = " <- ".join(re.sub(r".*/adalog_amd/", "", f).split(":")[0] for f in frames[:2]) if frames else "?"
    for k in ks:
        nm = re.sub(r"\(anonymous namespace\)::", "", k.name)
        nm = re.sub(r"^void ", "", nm)
        short = nm.split("(")[0][:70]
        if any(s in short for s in SCORING):
            tot["scoring"][0] += 1
            tot["scoring"][1] += k.duration
            continue
        tot["tail"][0] += 1
        tot["tail"][1] += k.duration
        key = (short, e.name, where)
        agg[key][0] += 1
        agg[key][1] += k.duration

print(f"# {model_name} depth={depth}: scoring {tot['scoring'][0]} launches {tot['scoring'][1] / 1e3:.1f} ms; "
      f"tail {tot['tail'][0]} launches {tot['tail'][1] / 1e3:.1f} ms")
print(f"{'launches':>8s} {'ms':>8s}  kernel | aten op | call site")
for (short, op, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:140]:
    print(f"{n:8d} {us / 1e3:8.2f}  {short} | {op} | {where}")
print("# by kernel")
byk = collections.defaultdict(lambda: [0, 0.0])
for (short, op, where), (n, us) in agg.items():
    byk[short][0] += n
    byk[short][1] += us
for short, (n, us) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n:8d} {us / 1e3:8.2f}  {short}")
