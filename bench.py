#!/usr/bin/env python3
"""Benchmark of the AdaLog calibration hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One *step* = one complete calibration (QuantCalibrator.batching_quant_calib: capture + every module's FPCS search) of
BASELINE.json configs[1]: deit_small, W4A4 (configs/4bit.py), 32 synthetic 224x224 calibration images PER GPU
(weak scaling: the global calibration set is 32*N images sharded by rank, scores all-reduced over RCCL).
Inputs (images, random-init weights) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line:
  value      = calibration images per second, whole job  (= 32*N*K / wall, wall = max over ranks)
  roofline   = the dominant scoring kernel (largest summed launch time; the library reports which kernel each launch used):
               algorithmic flops of its launches / their summed duration, measured live with events on the launch stream
               during the timed steps; config.scoring_kernels lists every kernel the same way
  hbm_kernels = the HBM-class kernels of the path (fake-quant, self-MSE scoring, quantile, packing) against 8 TB/s
  brecq      = BRECQ iterations per second on one block (bounded sample)
  cpu_baseline = the CPU oracle (a port of the reference's algorithm, oracle/) timed on the host cores on a bounded
               sample of the same workload and scaled to images/s with the work model of BASELINE.md section 2
"""
import argparse
import copy
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TOPS = {0: 5000.0, 1: 2500.0, 2: 157.3, 3: 5000.0}      # dense MFMA peaks, TFLOP/s (MI355X_MICROARCH.md): i8 = fp8 = 2x bf16
DT_NAME = {0: "i8", 1: "bf16", 2: "f32", 3: "fp8"}


def load_cfg(bits):
    spec = importlib.util.spec_from_file_location(f"cfg{bits}", os.path.join(ROOT, "configs", f"{bits}bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Config()


class GemmProfiler:
    """Collects the event pairs adalog_amd.ops records around every scoring launch (on the launch stream) together with
    the true (un-padded) K of the operands and the name of the kernel the library picked (adalog_last_kernel), so that
    ALGORITHMIC flops = 2*M*N*K*C*G are divided by kernel time, per kernel."""

    def __init__(self, ops):
        self.ops = ops

    def start(self):
        self.ops.GEMM_EVENTS = []

    def stop(self):
        ev, self.ops.GEMM_EVENTS = self.ops.GEMM_EVENTS, None
        torch.cuda.synchronize()
        by, shapes, kern = {}, {}, {}
        for dt, M, N, K, C, G, s, e, name in ev:
            ms = s.elapsed_time(e)
            for d, key in ((by, dt), (shapes, (dt, M, N, K, C, G, name)), (kern, (name, dt))):
                b = d.setdefault(key, [0.0, 0.0, 0])
                b[0] += 2.0 * M * N * K * C * G
                b[1] += ms
                b[2] += 1
        self.shapes, self.kernels = shapes, kern
        return by

    def top_shapes(self, steps, n=14):
        """Per-shape totals (rows x columns x K, candidates, groups), largest time first."""
        rows = sorted(self.shapes.items(), key=lambda kv: -kv[1][1])[:n]
        return [{"kernel": k[6], "dtype": DT_NAME[k[0]], "M": k[1], "N": k[2], "K": k[3], "cands": k[4], "groups": k[5],
                 "launches_per_step": v[2] / steps, "ms_per_step": round(v[1] / steps, 2),
                 "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1)} for k, v in rows]

    def by_kernel(self, steps):
        rows = sorted(self.kernels.items(), key=lambda kv: -kv[1][1])
        return [{"kernel": k[0], "dtype": DT_NAME[k[1]], "launches_per_step": v[2] / steps, "ms_per_step": round(v[1] / steps, 2),
                 "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1), "frac_of_peak": round(v[0] / (v[1] * 1e-3) / 1e12 / PEAK_TOPS[k[1]], 3)}
                for k, v in rows]


def pmc_traffic(kernel, workload_is_default):
    """HBM bytes per launch of `kernel`, REPLAYED from the committed PMC passes of this same command
    (tools/pmc_bench.sh -> profiles/r02_pmc_bench_traffic.json: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, averaged
    over every dispatch of one calibration step).  None for any other workload or when the summary is absent."""
    if not workload_is_default:
        return None
    path = os.path.join(ROOT, "profiles", "r02_pmc_bench_traffic.json")
    try:
        with open(path) as f:
            rows = json.load(f)
    except OSError:
        return None
    base = kernel.split("<")[0]
    tot, n = 0.0, 0
    for name, v in rows.items():
        if name.startswith(base) and v.get("hbm_read_bytes_per_launch") is not None:
            tot += (v["hbm_read_bytes_per_launch"] + (v.get("hbm_write_bytes_per_launch") or 0.0)) * v["launches"]
            n += v["launches"]
    return tot / n if n else None


def hbm_kernels(ops, dev):
    """Class-E (HBM-bound) kernels of the path at the deit_small / 32-image layer shapes: algorithmic bytes
    over the median event time, against 8 TB/s."""
    def timeit(fn, reps=5, inner=10):
        """median over `reps` of the time of `inner` back-to-back launches / inner (steady-state rate: one launch between two
        events is mostly launch latency at these sizes: 77 MB is 10 us at 8 TB/s)"""
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(inner):
                fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / inner)
        return sorted(ts)[len(ts) // 2]
    g = torch.Generator().manual_seed(3)
    M = 32 * 197
    out = []
    x = torch.randn(M, 1536, generator=g).to(dev)                          # fc2 input sized tensor (38.7 MB)
    xg = torch.nn.functional.gelu(x)
    sc1, zp1 = torch.tensor([0.21], device=dev), torch.tensor([7.0], device=dev)
    P = 128
    cs = (torch.rand(P, 1, generator=g) * 0.2 + 0.1).to(dev)
    cz = torch.randint(4, 12, (P, 1), generator=g).float().to(dev)
    W = (torch.randn(1536, 384, generator=g) * 0.05).to(dev)
    csw = (torch.rand(P, 1536, generator=g) * 0.01 + 0.005).to(dev)
    czw = torch.randint(4, 12, (P, 1536), generator=g).float().to(dev)
    q = torch.tensor([37], dtype=torch.int64, device=dev)
    from adalog_amd.quantizers.logarithm import AdaLogQuantizer
    t1, t2 = AdaLogQuantizer.make_tables(37, 8)                             # 4 bit: n_levels = 8, 16-entry tables
    t1, t2 = t1.to(dev), t2.to(dev)
    shift = torch.tensor([0.17], device=dev)
    x384 = x[:, :384].contiguous().unsqueeze(0)
    nb = x.numel() * 4
    rows = [
        ("k_uniform_rows (K1 uniform fake-quant, fp32 in/out)", 2 * nb, lambda: ops.uniform_fake_quant(x, sc1, zp1, 4)),
        ("k_adalog (K2/K3 shifted AdaLog fake-quant)", 2 * nb,
         lambda: ops.log_fake_quant(xg, sc1, q, t1, t2, 4, shift=shift, sub_shift=True)),
        ("k_score_a_self (K10, 128 candidates, x read once)", nb, lambda: ops.score_a_self(x, cs, cz, False, 4, 1.0)),
        ("k_score_w_self (K9, 128 candidates)", W.numel() * 4, lambda: ops.score_w_self(W, csw, czw, 4)),
        ("k_sel_hist/pick (K5 quantile, 4 radix passes)", 4 * nb, lambda: ops.quantile_rows(x.view(1, -1), [0.9, 1.0, 0.1, 0.0], 1)),
        ("k_log2_shift (input of the fused search, once per layer)", 2 * nb, lambda: ops.log2_shift(xg, 0.17)),
        ("k_pack_uniform_i8_fast (128 candidates -> int8 operand)", x384.numel() * 4 + x384.numel() * P,
         lambda: ops.pack_uniform(x384, cs, cz, P, 1, 1, 0, 0, 4, ops.I8, c_inner=True)),
    ]
    for name, nbytes, fn in rows:
        try:
            ms = timeit(fn)
            out.append({"kernel": name, "algorithmic_bytes": int(nbytes), "ms": round(ms, 4), "achieved_GBps": round(nbytes / ms / 1e6, 1),
                        "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000.0, 3)})
        except Exception as ex:                                   # a micro-benchmark must never take the headline down
            out.append({"kernel": name, "error": repr(ex)[:200]})
    return out


def brecq_rate(model_name, bits, dev, iters=2000):
    """BRECQ (utils/block_recon.py:84-137) iterations per second on blocks.0 of the benchmarked model: a bounded sample
    (the reference runs 20 000 iterations per block) with the reference's schedule in proportion (first 20 % without the
    rounding regulariser); wall time for one block and the whole model follow by scaling."""
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    cfg = load_cfg(bits)
    torch.manual_seed(5)
    base = create_model(model_name).eval()
    full = copy.deepcopy(base).to(dev).eval()
    model = wrap_modules_in_net(base, cfg, reparam=True).to(dev)
    imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
    QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    for m in model.modules():
        if hasattr(m, "reparam_bias"):
            m.reparam_bias()
    opt = torch.randn(64, 3, 224, 224, generator=torch.Generator().manual_seed(6)).to(dev)
    rec = BlockReconstructor(model, full, [(opt[:32], None), (opt[32:], None)])
    name = "blocks.0"
    block, fblock = rec.blocks[name], rec.full_blocks[name]
    rec.init_block_raw_data(block, fblock, name, dev)
    rec.reconstruct_single_block(name, block, dev, quant_act=True, iters=100)         # warm-up (releases the block data)
    rec.init_block_raw_data(block, fblock, name, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec.reconstruct_single_block(name, block, dev, quant_act=True, iters=iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nblk = len(rec.blocks)
    return {"iters_per_s": round(iters / dt, 1), "ms_per_iter": round(dt / iters * 1e3, 3), "block": name, "batch": 32,
            "sample_iters": iters, "blocks_in_model": nblk,
            "extrapolated_s_per_block_20000_iters": round(20000 * dt / iters, 1),
            "extrapolated_s_whole_model": round(20000 * dt / iters * nblk, 1)}


def _cpu_sample(threads, n_cand, min_seconds=0.0):
    """Seconds the oracle needs for one weight-scoring + one activation-scoring call of ``n_cand`` candidates each at
    deit_small attn.proj size (32 images x 197 tokens, 384 -> 384, W4A4) on ``threads`` host threads."""
    from oracle import adalog_oracle as O
    torch.set_num_threads(threads)
    O.PCHUNK = 16
    g = torch.Generator().manual_seed(5)
    N, T, I, Oc, bits = 32, 197, 384, 384, 4
    x = torch.randn(N, T, I, generator=g)
    W = torch.randn(Oc, I, generator=g) * 0.05
    b = torch.zeros(Oc)
    ro = torch.nn.functional.linear(x, W, b)
    w3 = W.view(1, Oc, I)
    scw, zpw = O.weight_candidates(w3, bits)
    sca, zpa = O.activation_candidates(x, bits, False)
    xq = O.uniform_fake_quant(x, sca[:, 60], zpa[:, 60].float(), bits)[0]
    wq = O.uniform_fake_quant(w3, scw[60], zpw[60].float(), bits)[0].view(Oc, I)
    t0 = time.perf_counter()
    reps = 0
    while True:
        ref_w = O.score_w(xq, w3, b, ro, scw[:n_cand], zpw[:n_cand], bits, 32)
        ref_a = O.score_a(x, wq, b, ro, sca[:, :n_cand], zpa[:, :n_cand], bits, 32)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    return dt, reps, (x, W, b, ro, scw, zpw, sca, zpa, ref_w, ref_a)


def cpu_baseline(nproc, with_hip=True):
    """The CPU oracle (a port of the reference's algorithm, validated against the reference's golden traces) on a bounded
    sample, timed at 8 / 32 / nproc host threads -- the best thread count is the baseline and is stated (the reference's
    elementwise chains over [N,T,I,P] temporaries are memory-bound: more threads than memory channels only adds
    contention).  Scaled to images/s with the candidate-GEMM work of the whole model (1354 TFLOP
    for deit_small at 32 images).  BASELINE.md section 3.1 measured the reference's own code at 24-46 GFLOP/s on 8 vCPUs."""
    N, T, I, Oc, bits = 32, 197, 384, 384, 4
    tried = {}
    for th in sorted({min(8, nproc), min(32, nproc), nproc}):
        dt, reps, _ = _cpu_sample(th, 16)                       # probe: 16 candidates per call
        tried[th] = reps * 2 * 2.0 * N * T * I * Oc * 16 / dt / 1e9
    best = max(tried, key=tried.get)
    dt, reps, (x, W, b, ro, scw, zpw, sca, zpa, ref_w, ref_a) = _cpu_sample(best, 128, 12.0)
    flops = reps * 2 * 2.0 * N * T * I * Oc * 128
    rate = flops / dt                                   # candidate-GEMM flop/s of the CPU path
    total = 1354e12                                     # deit_small, 32 images
    out = {"value": 32.0 / (total / rate), "unit": "images/s", "cores": best, "kind": "port",
           "sample": f"{reps} x (oracle score_w + score_a, 128 candidates each), deit_small attn.proj 32x197x384->384 W4A4 on {best} "
                     f"threads (best of {sorted(tried)}; host has {nproc}): {dt:.1f} s = {rate / 1e9:.1f} GFLOP/s "
                     f"candidate-GEMM rate; scaled by 1354 TFLOP per 32-image calibration",
           "sample_seconds": dt, "gflops_by_threads": {str(k): round(v, 1) for k, v in tried.items()},
           "reference_code_gflops_8vcpu": [24.1, 45.9]}
    if not with_hip:
        return out
    # the oracle as the CHECKER at full layer size: the same two scoring calls through the product path (HIP kernels)
    from adalog_amd import quant_layers as Q
    dev = torch.device("cuda")
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", bits, bits, calib_batch_size=32, search_round=1, eq_n=128,
                                              n_V=1, fpcs=True, steps=6).to(dev)
    lay.weight.data.copy_(W); lay.bias.data.copy_(b)
    lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
    lay.a_quantizer.scale.data.copy_(sca[:, 60].view(-1)); lay.a_quantizer.zero_point.data.copy_(zpa[:, 60].float().view(-1))
    lay.w_quantizer.scale.data.copy_(scw[60]); lay.w_quantizer.zero_point.data.copy_(zpw[60].float())
    got_w = lay._score_w(lay._pack_x_fixed(), scw.reshape(128, -1).to(dev), zpw.reshape(128, -1).float().to(dev)).cpu()
    got_a = lay._score_a(lay._pack_w_fixed(), sca.t().contiguous().to(dev), zpa.t().contiguous().float().to(dev)).cpu()
    rw, ra = ref_w.reshape(128, -1), ref_a.reshape(-1, 128).t()
    err_w = float(((got_w - rw).abs() / rw.abs()).max())
    err_a = float(((got_a - ra).abs() / ra.abs()).max())
    same_top = bool(torch.equal(torch.topk(got_a[:, 0], 16).indices.sort().values, torch.topk(ra[:, 0], 16).indices.sort().values))
    out["parity_vs_hip"] = {"max_rel_err_weight_scores": err_w, "max_rel_err_activation_scores": err_a,
                            "same_top16_activation_candidates": same_top,
                            "note": "same inputs and candidates through adalog_amd (HIP) at full layer size; bar 1e-3"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="deit_small")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--images-per-gpu", type=int, default=32)
    ap.add_argument("--images-total", type=int, default=None,
                    help="fixed calibration-set size sharded over the GPUs (strong scaling; BASELINE config 4 = 1024 over 8); "
                         "default: --images-per-gpu x N (weak scaling)")
    ap.add_argument("--depth", type=int, default=None, help="truncate the block count (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched as `python bench.py --gpus N`: start one rank per GPU under torch.distributed.run as a CHILD process
        # (nothing in this process has touched the GPU yet) and leave with its exit code
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("ADALOG_DIST_BACKEND", "nccl") != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        # RCCL ("nccl") is the product path.  ADALOG_DIST_BACKEND=gloo lets several ranks share ONE GPU: a functional check of
        # the sharded code path on a single-GPU box (collectives then stage through the host: not a performance mode)
        be_name = os.environ.get("ADALOG_DIST_BACKEND", "nccl")
        if be_name == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=be_name)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from adalog_amd import backend, parallel
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net
    ops = backend.get()

    cfg = load_cfg(args.bits)
    if args.images_total is not None:
        assert args.images_total % world == 0, "--images-total must divide over the GPUs"
        args.images_per_gpu = args.images_total // world
    cfg.calib_size = args.images_per_gpu * world
    torch.manual_seed(5)                                           # reference default seed (test_quant.py:77)
    base = create_model(args.model, depth=args.depth).eval()
    base = wrap_modules_in_net(base, cfg, reparam=True).to(dev)
    images = torch.randn(cfg.calib_size, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    lo, hi = rank * args.images_per_gpu, (rank + 1) * args.images_per_gpu
    local = images[lo:hi].to(dev)
    loader = [(local[i:i + cfg.calib_batch_size], None) for i in range(0, local.shape[0], cfg.calib_batch_size)]

    prof = GemmProfiler(ops)

    cals = []

    def one_step(model):
        cal = QuantCalibrator(model, loader, capture="block")
        cal.batching_quant_calib()
        cals.append(cal)

    models = [copy.deepcopy(base) for _ in range(args.warmup + args.steps)]
    for i in range(args.warmup):
        one_step(models[i])
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    prof.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(models[args.warmup + i])
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    by = prof.stop()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = cfg.calib_size * args.steps / wall
        # the dominant KERNEL (largest summed launch time over the timed steps) carries the roofline object
        (dom_name, dom), (fl, ms, n) = max(prof.kernels.items(), key=lambda kv: kv[1][1]) if prof.kernels else (("", 0), (0.0, 1.0, 1))
        achieved = fl / (ms * 1e-3) / 1e12
        gemm_ms_total = sum(v[1] for v in by.values())
        timed = cals[args.warmup:]
        fpcs = [sum(c.fpcs_seconds().values()) for c in timed]
        capt = [c.capture_device_seconds() for c in timed]
        default_workload = (args.model == "deit_small" and args.bits == 4 and args.images_per_gpu == 32 and world == 1
                            and args.depth is None)
        result = {
            "metric": "calib_images_per_sec", "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.images_total is not None else "weak",
            "vs_baseline": None, "dtype": DT_NAME[dom], "data": "synthetic",
            "config": {"workload": f"{args.model} W{cfg.w_bit}A{cfg.a_bit} --calibrate, {args.images_per_gpu} calib images "
                                   f"per GPU ({cfg.calib_size} total), eq_n=128, 3 rounds, FPCS 6 steps",
                       "calib_wall_s_per_step": wall / args.steps,
                       "scoring_gemm_ms_per_step": gemm_ms_total / args.steps,
                       "scoring_gemm_by_dtype": {DT_NAME[d]: {"launches_per_step": v[2] / args.steps,
                                                              "ms_per_step": v[1] / args.steps,
                                                              "tflops": v[0] / (v[1] * 1e-3) / 1e12} for d, v in by.items()},
                       "scoring_kernels": prof.by_kernel(args.steps),
                       "scoring_gemm_top_shapes": prof.top_shapes(args.steps),
                       "fpcs_seconds_per_step": sum(fpcs) / max(len(fpcs), 1),
                       "capture_seconds_per_step": sum(capt) / max(len(capt), 1),
                       "fpcs_note": "device events on the search stream around every module's hyperparameter_searching (+ reparam), "
                                    "summed over the modules (rank 0); capture = the FP forward passes that record the activations",
                       "depth_override": args.depth},
            "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": achieved,
                         "peak": PEAK_TOPS[dom], "unit": "TFLOP/s", "frac": achieved / PEAK_TOPS[dom],
                         "traffic": pmc_traffic(dom_name, default_workload),
                         "traffic_note": "replayed from profiles/r02_pmc_bench_traffic.json (rocprofv3 --pmc passes of this same "
                                         "command); null for any other workload",
                         "launches": n, "avg_launch_ms": ms / max(n, 1)},
        }
        if not args.no_cpu_baseline and world == 1:
            result["hbm_kernels"] = hbm_kernels(ops, dev)
            try:
                result["brecq"] = brecq_rate(args.model, args.bits, dev)
            except Exception as ex:
                result["brecq"] = {"error": repr(ex)[:300]}
            result["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
