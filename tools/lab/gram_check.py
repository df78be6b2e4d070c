"""Lab: Gram-form weight-search scores (ops.GramState) against the token-form slab kernel and the CPU oracle, plus timings."""
import os, sys, time
os.environ.setdefault("ADALOG_GRAM_W", "2")      # every supported shape, profitable or not
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adalog_amd import ops
from adalog_amd.ops import FP8, I8
DEV = "cuda"


def case(T, O, K, bits, P=128, tokens=197, check_oracle=False, seed=0):
    gen = torch.Generator().manual_seed(seed + T + O + K + bits)
    x = torch.randn(T, K, generator=gen)
    W = torch.randn(O, K, generator=gen) * 0.05
    b = torch.randn(O, generator=gen) * 0.1
    qmax = 2 ** bits - 1
    w_lo, w_hi = W.min(1).values, W.max(1).values
    sc = ((w_hi - w_lo) / qmax).view(1, O) * torch.linspace(0.6, 1.2, P).view(P, 1)
    zp = torch.round(-w_lo.view(1, O) / sc).clamp(0, qmax)
    W[:, 5] = sc[17] * 2.5
    W[:, 9] = sc[min(90, P - 1)] * -1.5
    a_s = torch.tensor([x.abs().max().item() * 2 / qmax])
    a_z = torch.tensor([float(2 ** (bits - 1))])
    ref = torch.nn.functional.linear(x, W, b)
    d = lambda t: t.to(DEV).contiguous()
    xd, Wd, scd, zpd, asd, azd, bd = d(x), d(W), d(sc), d(zp), d(a_s), d(a_z), d(b)
    ref_t = d(ref.t())
    norm = 1.0 / tokens
    assert ops.gram_ok(T, O, K, bits, bits, P) or os.environ.get("ADALOG_GRAM_W") == "2", "gram_ok declined"
    torch.cuda.synchronize()
    t0 = time.time()
    gs = ops.GramState(xd, asd, azd, bits, ref_t, bd)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    got = gs.score_w(Wd, scd, zpd, bits, norm)
    torch.cuda.synchronize()
    res = {"shape": (T, O, K, bits, P)}
    dt = FP8 if bits <= 4 else I8
    xp = ops.pack_uniform(xd.unsqueeze(0), asd, azd, 1, 0, 1, 0, 0, bits, dt)
    if ops.score_w_gen_ok(dt, T, O, K, xp.shape[-1], P):
        want = ops.score_w_gen(dt, xp, Wd, scd, zpd, bits, ref_t, asd, bd, norm)
        res["vs_slab"] = float(((got - want).abs().max() / want.abs().max()).item())
        res["vs_slab_elem"] = float(((got - want).abs() / want.abs()).max().item())
    if check_oracle:
        from oracle import adalog_oracle as Orc
        xq = (torch.clamp((x / a_s).round() + a_z, 0, qmax) - a_z) * a_s
        imgs = T // tokens
        o = Orc.score_w(xq.view(imgs, tokens, K), W.view(1, O, K), b, ref.view(imgs, tokens, O), sc.view(P, 1, O, 1), zp.view(P, 1, O, 1), bits)
        o = o.view(P, O)
        res["vs_oracle"] = float(((got.cpu() - o).abs().max() / o.abs().max()).item())
    # timings
    def timeit(f, n=20):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res["build_us"] = timeit(lambda: ops.GramState(xd, asd, azd, bits, ref_t, bd), 10)
    res["score_us"] = timeit(lambda: gs.score_w(Wd, scd, zpd, bits, norm))
    if "vs_slab" in res:
        res["slab_us"] = timeit(lambda: ops.score_w_gen(dt, xp, Wd, scd, zpd, bits, ref_t, asd, bd, norm, defer=True))
    if os.environ.get("GRAM_TIMELINE"):
        import ctypes
        from adalog_amd import _lib
        lib = _lib.load()
        tl = torch.zeros(1024 * 2 * 8, dtype=torch.int64, device=DEV)
        lib.adalog_gram_set_timeline.argtypes = [ctypes.c_void_p]
        lib.adalog_gram_set_timeline(ctypes.c_void_p(tl.data_ptr()))
        gs.score_w(Wd, scd, zpd, bits, norm)
        torch.cuda.synchronize()
        lib.adalog_gram_set_timeline(None)
        t = tl.view(-1, 2, 8).cpu()
        used = t[:, 0, 0] != 0
        t = t[used]
        d = lambda a, b_: (t[:, 0, b_] - t[:, 0, a]).float()
        res["tl_pass0_cycles(100MHz ticks?)"] = {"gen": d(0, 1).mean().item(), "wc": d(1, 2).mean().item(), "main": d(2, 3).mean().item(),
                                 "tail": d(3, 4).mean().item(), "n_wg": int(used.sum())}
        u2 = t[:, 1, 0] != 0
        if u2.any():
            t2 = t[u2]
            res["tl_pass1"] = {"gen": (t2[:, 1, 1] - t2[:, 1, 0]).float().mean().item(), "main": (t2[:, 1, 3] - t2[:, 1, 2]).float().mean().item(),
                               "n": int(u2.sum())}
        span = (t[:, :, 4].max() - t[:, 0, 0].min()).item()
        res["tl_span"] = span
    print(res, flush=True)
    return res


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "prof":
    case(6304, 1152, 384, 4)
    case(6304, 384, 384, 4)
    case(6304, 2304, 768, 4)
    case(100352, 384, 128, 4, tokens=3136)
elif __name__ == "__main__":
    case(197 * 4, 96, 96, 4, check_oracle=True)
    case(197 * 8, 384, 384, 4, check_oracle=True)
    case(197 * 8, 384, 384, 6, check_oracle=True)
    case(6304, 1152, 384, 4)
    case(6304, 384, 384, 4)
    case(6304, 1536, 384, 4)
    case(6304, 1152, 384, 3)
    case(6304, 1152, 384, 6)
    case(6304, 576, 192, 4)
    case(6304, 2304, 768, 4)
    case(100352, 384, 128, 4, tokens=3136)
    case(25088, 768, 256, 4, tokens=784)
    case(6272, 1536, 512, 4, tokens=196)
