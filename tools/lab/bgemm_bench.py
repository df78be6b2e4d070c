"""EXPERIMENT (round 3, not part of the product): a hand-written fp32-MFMA GEMM (tools/lab/brecq_gemm_lab.hip) timed against
torch (rocBLAS) on the twelve Linear products of one BRECQ iteration of a deit_small block, both checked against fp64.
Outcome (profiles/r03_notes.md section 5): 0.93-1.06 ms against rocBLAS's 0.80 ms -- the product keeps F.linear.
ADALOG_BGEMM_TM / ADALOG_BGEMM_SPLITS force the plan (one setting per process); -DBG_LAB_NO_LOADS / -DBG_LAB_NO_MFMA builds time
the two halves of the main loop (LAB_DEFS)."""
import ctypes, json, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "adalog_amd", "csrc")
defs = os.environ.get("LAB_DEFS", "")
so = os.path.join(HERE, "libbgemm_lab%s.so" % defs.replace("-D", "_").replace(" ", ""))
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", CSRC]
                          + defs.split() + [os.path.join(HERE, "brecq_gemm_lab.hip"), os.path.join(CSRC, "errors.hip"), "-o", so])
lib = ctypes.CDLL(so)
i32, i64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
lib.adalog_brecq_gemm_workspace_bytes.restype = i64
lib.adalog_brecq_gemm_workspace_bytes.argtypes = [i32] * 4
lib.adalog_brecq_gemm.argtypes = [vp, i64, i32, vp, i64, i32, vp, vp, i64, i32, i32, i32, i32, i64, i64, i64, vp, i64, vp]


class ops:   # the call the product would make
    @staticmethod
    def brecq_gemm(a, b, a_kmajor=False, b_kmajor=False, bias=None):
        M, K = (a.shape[1], a.shape[0]) if a_kmajor else a.shape
        N = b.shape[1] if b_kmajor else b.shape[0]
        out = torch.empty(M, N, device=a.device)
        wsb = lib.adalog_brecq_gemm_workspace_bytes(M, N, K, 1)
        ws = torch.empty(max(wsb // 4, 1), device=a.device)
        rc = lib.adalog_brecq_gemm(a.data_ptr(), a.stride(0), int(a_kmajor), b.data_ptr(), b.stride(0), int(b_kmajor),
                                   None if bias is None else bias.data_ptr(), out.data_ptr(), N, M, N, K, 1, 0, 0, M * N,
                                   ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out

dev = torch.device("cuda")
T = int(os.environ.get("TOKENS", "6304"))
D = int(os.environ.get("DIM", "384"))
H, S, C = D // 64, 197, 64
NB = T // S


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
rows = []
for name, (I, O) in {"qkv": (D, 3 * D), "proj": (D, D), "fc1": (D, 4 * D), "fc2": (4 * D, D)}.items():
    x = torch.randn(T, I, device=dev, generator=g)
    w = torch.randn(O, I, device=dev, generator=g) * 0.05
    b = torch.randn(O, device=dev, generator=g)
    dy = torch.randn(T, O, device=dev, generator=g)
    cases = [
        ("fwd", lambda: ops.brecq_gemm(x, w, False, False, b), lambda: torch.nn.functional.linear(x, w, b), 2.0 * T * I * O,
         lambda: (x.double() @ w.double().t() + b.double())),
        ("dx", lambda: ops.brecq_gemm(dy, w, False, True), lambda: dy @ w, 2.0 * T * I * O, lambda: dy.double() @ w.double()),
        ("dw", lambda: ops.brecq_gemm(dy, x, True, True), lambda: dy.t() @ x, 2.0 * T * I * O, lambda: dy.double().t() @ x.double()),
    ]
    for kind, mine, ref, fl, exact in cases:
        e = exact()
        em = ((mine().double() - e).abs().max() / e.abs().max()).item()
        er = ((ref().double() - e).abs().max() / e.abs().max()).item()
        tm, tr = timeit(mine), timeit(ref)
        rows.append(dict(op=f"{name}.{kind}", us_mfma=round(tm, 1), us_rocblas=round(tr, 1), tflops_mfma=round(fl / tm / 1e6, 1),
                         tflops_rocblas=round(fl / tr / 1e6, 1), err_mfma=em, err_rocblas=er))
        print(rows[-1], flush=True)
# attention products, batched over (image, head)
q = torch.randn(NB * H, S, C, device=dev, generator=g)
k = torch.randn(NB * H, S, C, device=dev, generator=g)
v = torch.randn(NB * H, S, C, device=dev, generator=g)
pm = torch.softmax(torch.randn(NB * H, S, S, device=dev, generator=g), -1)
ds = torch.randn(NB * H, S, S, device=dev, generator=g)
do = torch.randn(NB * H, S, C, device=dev, generator=g)
if os.environ.get("BATCHED", "1") == "1" and S % 4 != 0:
    # S = 197 rows are not 16-byte aligned in the [S, S] operands: those products need padded leading dimensions
    Sp = (S + 3) // 4 * 4
    pmp = torch.zeros(NB * H, S, Sp, device=dev); pmp[..., :S] = pm
    dsp = torch.zeros(NB * H, S, Sp, device=dev); dsp[..., :S] = ds
    fl = 2.0 * NB * H * S * S * C
    for nm, mine, ref in [
        ("qk.fwd", None, lambda: q @ k.transpose(1, 2)),
        ("av.fwd", None, lambda: pm @ v),
    ]:
        tr = timeit(ref)
        rows.append(dict(op=nm, us_rocblas=round(tr, 1), tflops_rocblas=round(fl / tr / 1e6, 1)))
        print(rows[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = os.environ.get("TAG", "default")
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"bgemm_bench_{tag}.json"), "w"), indent=1)
tot_m = sum(r.get("us_mfma", 0) for r in rows if "us_mfma" in r)
tot_r = sum(r["us_rocblas"] for r in rows if "us_mfma" in r)
print(f"linear products: mfma {tot_m:.0f} us, rocBLAS {tot_r:.0f} us")
