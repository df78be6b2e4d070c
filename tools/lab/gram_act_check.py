"""Lab: Gram-form activation-search scores (ops.GramActState) against an fp64 evaluation and the token-form slab kernel; timings."""
import os, sys, time
os.environ.setdefault("ADALOG_GRAM_A", "2")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from adalog_amd import ops
from adalog_amd.ops import FP8, I8
DEV = "cuda"


def case(T, O, K, bits, P=128, tokens=197, seed=0):
    gen = torch.Generator().manual_seed(seed + T + O + K + bits)
    x = torch.randn(T, K, generator=gen)
    x[:, : max(1, K // 8)] *= 3.0
    W = torch.randn(O, K, generator=gen) * 0.05
    b = torch.randn(O, generator=gen) * 0.1
    qmax = 2 ** bits - 1
    w_lo, w_hi = W.min(1).values, W.max(1).values
    sw = (w_hi - w_lo) / qmax
    zw = torch.round(-w_lo / sw).clamp(0, qmax)
    wq = (torch.clamp(torch.round(W / sw[:, None]) + zw[:, None], 0, qmax) - zw[:, None])
    Wq = wq * sw[:, None]
    ref = torch.nn.functional.linear(x, W, b)
    amax = x.abs().max().item()
    sc = (2 * amax / qmax) * torch.linspace(0.5, 1.1, P)
    zp = torch.round(torch.linspace(qmax / 2 - 3, qmax / 2 + 3, P)).clamp(0, qmax)
    x[5, 3] = (sc[17] * 2.5).item()
    d = lambda t: t.to(DEV).contiguous()
    xd, Wd, scd, zpd, bd, swd, zwd, refd = d(x), d(W), d(sc), d(zp), d(b), d(sw), d(zw), d(ref)
    norm = 1.0 / (tokens * O)
    assert ops.gram_act_ok(T, O, K, bits, bits, P)
    torch.cuda.synchronize()
    prep = ops.GramActPrepared(xd)
    st = ops.GramActState(prep, refd, bd, Wd, swd, zwd, bits, bits, P)
    got = st.score(scd.view(P, 1), zpd.view(P, 1), norm)
    torch.cuda.synchronize()
    res = {"shape": (T, O, K, bits, P)}
    # fp64 truth on a subset of candidates
    sub = list(range(0, P, max(1, P // 8)))
    xd64, Wq64, r64 = xd.double(), d(Wq).double(), (refd - bd).double()
    tr = []
    for p_ in sub:
        s_, z_ = sc[p_].item(), zp[p_].item()
        xq = (torch.clamp(torch.round(xd / s_) + z_, 0, qmax) - z_).double()
        out = (xq @ Wq64.t()) * float(torch.tensor(s_, dtype=torch.float32))
        tr.append(-norm * ((r64 - out) ** 2).sum().item())
    tr = torch.tensor(tr)
    g_ = got.view(-1)[sub].double().cpu()
    res["vs_fp64"] = float(((g_ - tr).abs() / tr.abs()).max())
    dt = FP8 if bits <= 4 else I8
    wp = ops.pack_uniform(Wd.unsqueeze(0), swd, zwd, 1, 0, 1, 0, 1, bits, dt)
    if ops.score_act_gen_ok(dt, O, T, K, wp.shape[-1], P):
        want = ops.score_act_gen(dt, wp, xd, scd.view(P, 1), zpd.view(P, 1), bits, refd, swd, bd, norm)
        res["vs_slab"] = float(((got.view(-1) - want.view(-1)).abs() / want.view(-1).abs()).max())

    def timeit(f, n=10):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    res["prepare_us"] = timeit(lambda: ops.GramActPrepared(xd), 5)
    res["build_us"] = timeit(lambda: ops.GramActState(prep, refd, bd, Wd, swd, zwd, bits, bits, P), 5)
    res["score_us"] = timeit(lambda: st.score(scd.view(P, 1), zpd.view(P, 1), norm))
    if "vs_slab" in res:
        res["slab_us"] = timeit(lambda: ops.score_act_gen(dt, wp, xd, scd.view(P, 1), zpd.view(P, 1), bits, refd, swd, bd, norm, defer=True))
    if os.environ.get("GA_TIMELINE"):
        import ctypes
        from adalog_amd import _lib
        lib = _lib.load()
        tl = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=DEV)
        lib.adalog_gram_act_set_timeline.argtypes = [ctypes.c_void_p]
        lib.adalog_gram_act_set_timeline(ctypes.c_void_p(tl.data_ptr()))
        st.score(scd.view(P, 1), zpd.view(P, 1), norm)
        torch.cuda.synchronize()
        lib.adalog_gram_act_set_timeline(None)
        t = tl.view(-1, 4, 8).cpu().float()
        t = t[t[:, 0, 4] > 0]
        res["tl_per_chunk"] = {"mfma": [round((t[:, w, 0] / t[:, w, 4]).mean().item()) for w in range(4)],
                               "gen": [round((t[:, w, 1] / t[:, w, 4]).mean().item()) for w in range(4)],
                               "barrier": [round((t[:, w, 2] / t[:, w, 4]).mean().item()) for w in range(4)],
                               "total": [round((t[:, w, 3] / t[:, w, 4]).mean().item()) for w in range(4)], "chunks": t[0, 0, 4].item()}
    print(res, flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "split":
    case(6272, 1536, 512, 4, tokens=196)       # swin_base stage 2: qkv, fc1, proj
    case(6272, 2048, 512, 4, tokens=196)
    case(6272, 512, 512, 4, tokens=196)
    case(6304, 2304, 768, 4)                   # vit_base: qkv, fc1, proj
    case(6304, 3072, 768, 4)
    case(6304, 768, 768, 4)
    case(6304, 2304, 768, 6)
    case(6304, 1152, 384, 4)
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "swin":
    for T, K, tok in ((100352, 128, 3136), (25088, 256, 784)):          # swin_base stages 0 / 1: qkv, proj, fc1
        for O in (3 * K, K, 4 * K):
            case(T, O, K, 4, tokens=tok)
    for O in (288, 96, 384):                                             # swin_small / tiny stage 0 (K = 96)
        case(100352, O, 96, 4, tokens=3136)
elif __name__ == "__main__" and len(sys.argv) > 1:
    case(6304, 1152, 384, 4)
    case(25088, 768, 256, 4, tokens=784)
elif __name__ == "__main__":
    case(197 * 4, 96, 96, 4)
    case(197 * 8, 384, 128, 4)
    case(6304, 1152, 384, 4)
    case(6304, 384, 384, 4)
    case(6304, 1536, 384, 4)
    case(6304, 1152, 384, 3)
    case(6304, 1152, 384, 6)
    case(6304, 576, 192, 4)
    case(100352, 384, 128, 4, tokens=3136)
    case(25088, 768, 256, 4, tokens=784)
