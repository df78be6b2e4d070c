#!/usr/bin/env python3
"""BRECQ / AdaRound block reconstruction throughput (iterations per second) on one transformer block.

    python tools/bench_brecq.py [--model deit_small] [--bits 4] [--iters 300] [--images 128]

Calibrates the model on synthetic images, then times `reconstruct_single_block` on blocks.0 (batch 32, quant_act as in
test_quant.py --optimize).  Prints wall time per iteration and the share of it the GPU was busy."""
import argparse
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="deit_small")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--block", default="blocks.0")
    ap.add_argument("--kernel-stats", default=None,
                    help="write a BRECQ-ONLY per-kernel table (name, calls, total / average microseconds) of the reconstruct_single_block "
                         "call to this CSV: the kernel activity records of torch.profiler (roctracer) around that call alone -- the "
                         "calibration that precedes it is outside the window")
    args = ap.parse_args()
    import copy
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    spec = importlib.util.spec_from_file_location("cfg", os.path.join(ROOT, "configs", f"{args.bits}bit.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cfg = mod.Config()
    dev = torch.device("cuda")
    torch.manual_seed(5)
    base = create_model(args.model).eval()
    full = copy.deepcopy(base).to(dev).eval()
    model = wrap_modules_in_net(base, cfg, reparam=True).to(dev)
    imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
    QuantCalibrator(model, [(imgs, None)], capture="block").batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    for m in model.modules():
        if hasattr(m, "reparam_bias"):
            m.reparam_bias()
    opt_imgs = torch.randn(args.images, 3, 224, 224, generator=torch.Generator().manual_seed(6)).to(dev)
    loader = [(opt_imgs[i:i + 32], None) for i in range(0, args.images, 32)]
    rec = BlockReconstructor(model, full, loader)
    name = args.block
    block, fblock = rec.blocks[name], rec.full_blocks[name]
    rec.init_block_raw_data(block, fblock, name, dev)
    # steady state: the clock is read (after a device synchronisation) when iteration iters / 4 and the last one have finished
    marks = {}

    def hook(it, loss_func):
        if it in (args.iters // 4, args.iters):
            torch.cuda.synchronize()
            marks[it] = time.perf_counter()
    rec.iter_hook = hook
    torch.cuda.synchronize()
    prof = None
    if args.kernel_stats:
        from torch.profiler import ProfilerActivity, profile
        prof = profile(activities=[ProfilerActivity.CUDA])
        prof.__enter__()
    t0 = time.perf_counter()
    rec.reconstruct_single_block(name, block, dev, quant_act=True, iters=args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        prof.__exit__(None, None, None)
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
        tot = sum(e.device_time_total for e in rows) or 1.0
        with open(args.kernel_stats, "w") as f:
            f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage,CallsPerIteration\n")
            for e in rows:
                if e.device_time_total <= 0:
                    continue
                f.write('"%s",%d,%.1f,%.2f,%.2f,%.2f\n' % (e.key.replace('"', "'"), e.count, e.device_time_total,
                                                         e.device_time_total / max(e.count, 1), 100.0 * e.device_time_total / tot,
                                                         e.count / args.iters))
        print(f"kernel stats of the reconstruct_single_block call ({args.iters} iterations, profiler attached): {args.kernel_stats}")
    print(f"{args.model} W{args.bits}A{args.bits} {name}: {args.iters} iterations in {dt:.2f} s = {args.iters / dt:.1f} it/s "
          f"({dt / args.iters * 1e3:.2f} ms per iteration)")
    if len(marks) == 2:
        n_ = args.iters - args.iters // 4
        ds = marks[args.iters] - marks[args.iters // 4]
        print(f"steady state (last {n_} iterations): {n_ / ds:.1f} it/s ({ds / n_ * 1e3:.3f} ms per iteration)")


if __name__ == "__main__":
    main()
