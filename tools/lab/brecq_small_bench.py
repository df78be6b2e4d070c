"""Times the elementwise / reduction kernels of a BRECQ iteration one by one at the shapes of a deit_small (or vit_base)
block (32 images): the STE backward passes, the losses, AdaRound, Adam.  For each: microseconds per launch and the HBM
rate its algorithmic bytes imply.  `--no-ticket` variants (where the op has one) show what the last-block reduction costs."""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from adalog_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="deit_small")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--out", default=None)
a = ap.parse_args()
C, H = {"deit_small": (384, 6), "vit_base": (768, 12), "deit_tiny": (192, 3)}[a.model]
T, S = 32 * 197, 197
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


rows = []


def row(name, us, nbytes):
    rows.append({"op": name, "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1)})
    print(f"{name:44s} {us:8.2f} us   {nbytes / us / 1e3:8.1f} GB/s")


sc, zp = torch.tensor([0.05], device=dev), torch.tensor([7.0], device=dev)
for label, n in (("uniform_bwd x[T,C]", T * C), ("uniform_bwd x[T,4C]", T * 4 * C)):
    x, gy = rn(n), rn(n)
    row(label + " +gscale", timed(lambda: ops.uniform_fake_quant_backward(gy, x, sc, zp, 4, False, True, False)), 12 * n)
    row(label + " no ticket", timed(lambda: ops.uniform_fake_quant_backward(gy, x, sc, zp, 4, False, False, False)), 12 * n)
q = torch.tensor([37], device=dev, dtype=torch.int64)
for label, n in (("adalog_bwd probs[32,H,S,S]", 32 * H * S * S), ("adalog_bwd gelu[T,4C]", T * 4 * C)):
    x = torch.rand(n, device=dev, generator=g)
    gy = rn(n)
    y = x.clone()
    s1 = torch.tensor([1.0], device=dev)
    row(label, timed(lambda: ops.log_fake_quant_backward(gy, x, y, s1, q, 4, None, False)), 12 * n)
for label, n in (("rec_loss [T,C]", T * C),):
    p_, t_ = rn(n), rn(n)
    row(label, timed(lambda: ops.rec_loss(p_, t_, 1.0)), 8 * n)
    gm = torch.ones(1, device=dev)
    row("rec_loss_backward [T,C]", timed(lambda: ops.rec_loss_backward(p_, t_, 1.0, gm)), 12 * n)
alphas = [rn(3 * C, C), rn(C, C), rn(4 * C, C), rn(C, 4 * C)]
nal = sum(t.numel() for t in alphas)
b = torch.tensor([10.0], device=dev)
row("round_loss_multi (4 layers)", timed(lambda: ops.round_loss_multi(alphas, b, 0.01)), 8 * nal)
for t in alphas[:1] + alphas[2:3]:
    w = rn(*t.shape)
    s_ = torch.full((t.shape[0],), 0.02, device=dev)
    z_ = torch.full((t.shape[0],), 8.0, device=dev)
    row(f"adaround fwd {tuple(t.shape)}", timed(lambda: ops.adaround(w, t, s_, z_, 4, True)), 12 * t.numel())
    gyw = rn(*t.shape)
    row(f"adaround bwd {tuple(t.shape)}", timed(lambda: ops.adaround(w, t, s_, z_, 4, True, gyw)), 16 * t.numel())
x = rn(T * C)
row("uniform_int [T,C]", timed(lambda: ops.uniform_int(x, sc, zp, 4)), 8 * T * C)
xs = rn(T * C)
row("uniform_fake_quant fwd [T,C]", timed(lambda: ops.uniform_fake_quant(xs, sc, zp, 4)), 8 * T * C)
if a.out:
    json.dump({"model": a.model, "rows": rows}, open(a.out, "w"), indent=1)
