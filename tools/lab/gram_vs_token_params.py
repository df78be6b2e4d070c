"""Lab: calibrate one model twice -- Gram forms on (default) and off (token-form kernels) -- and compare every committed quantiser
parameter.  python tools/lab/gram_vs_token_params.py [model] [bits]"""
import copy
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(tag):
    from bench import load_cfg
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net
    model_name, bits = sys.argv[2], int(sys.argv[3])
    dev = torch.device("cuda")
    cfg = load_cfg(bits)
    torch.manual_seed(5)
    base = create_model(model_name).eval()
    imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
    model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
    QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if "quantizer" in k}
    torch.save(sd, f"/tmp/params_{tag}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("gram", "token"):
        run(sys.argv[1])
        sys.exit(0)
    model = sys.argv[1] if len(sys.argv) > 1 else "deit_small"
    bits = sys.argv[2] if len(sys.argv) > 2 else "4"
    for tag, env in (("gram", {}), ("token", {"ADALOG_GRAM_W": "0", "ADALOG_GRAM_A": "0"})):
        subprocess.check_call([sys.executable, __file__, tag, model, bits], env={**os.environ, **env})
    a, b = torch.load("/tmp/params_gram.pt"), torch.load("/tmp/params_token.pt")
    nd, worst = 0, 0.0
    for k in a:
        if not torch.equal(a[k], b[k]):
            nd += 1
            rel = ((a[k].float() - b[k].float()).abs() / b[k].float().abs().clamp_min(1e-12)).max().item()
            frac = (a[k] != b[k]).float().mean().item()
            worst = max(worst, rel)
            print(f"differs: {k}  max rel {rel:.3e}  entries {frac:.4f}")
    print(f"{model} W{bits}A{bits}: {len(a)} quantiser tensors, {nd} differ, worst relative difference {worst:.3e}")
