"""Lab aid: how often does an output-MSE search of round r see exactly the inputs it saw in round r-1 (the other operand's
quantiser unchanged bit for bit), per search kind -- and, for the per-head attention searches, per head."""
import collections, copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import load_cfg
from adalog_amd import search
from adalog_amd.utils.calibrator import QuantCalibrator
from adalog_amd.utils.models import create_model
from adalog_amd.utils.wrap_net import wrap_modules_in_net

dev = torch.device("cuda")
cfg = load_cfg(int(os.environ.get("BITS", "4")))
torch.manual_seed(5)
base = create_model(os.environ.get("MODEL", "deit_small"), depth=int(os.environ.get("DEPTH", "3"))).eval()
imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
stats = collections.Counter()
orig = search.round_is_redundant
search.SKIP_CONVERGED = False
search.COLLECT_ROUND_STATS = True


def spy(module, tag, *qs):
    seen = module.__dict__.setdefault("_round_inputs", {})
    prev = seen.get(tag)
    r = orig(module, tag, *qs)
    cur = seen.get(tag)
    kind = f"{type(module).__name__}.{tag}"
    if prev is not None:
        stats[kind + " checked"] += 1
        same = prev.shape == cur.shape and bool(torch.equal(prev, cur))
        stats[kind + " unchanged"] += int(same)
        sc = getattr(qs[0], "scale", None)
        if not same and torch.is_tensor(sc) and sc.numel() > 1 and sc.numel() <= 64 and prev.numel() == 2 * sc.numel():
            H = sc.numel()                                  # per-head (scale, zero point): heads whose pair is unchanged
            eq = (prev.view(2, H) == cur.view(2, H)).all(0)
            stats[kind + " heads unchanged (of changed modules)"] += int(eq.sum())
            stats[kind + " heads total (of changed modules)"] += H
    return r


search.round_is_redundant = spy
QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
torch.cuda.synchronize()
for k in sorted(stats):
    print(f"{stats[k]:5d}  {k}")
