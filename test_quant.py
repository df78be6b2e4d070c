#!/usr/bin/env python3
"""Entry point with the flags and flow of the reference's test_quant.py (test_quant.py:45-81,136-241):

    python test_quant.py --model deit_small --config ./configs/4bit.py --calibrate [--optimize]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 test_quant.py --model swin_base \
        --config ./configs/3bit.py --calibrate --calib-size 1024          # images sharded over the 8 GPUs

parse args -> import Config from the file path -> CLI overrides -> seed -> build model -> wrap -> calibrate (FPCS) ->
un-wrap channel-wise layers -> reparam_bias -> save checkpoint -> validate -> [--optimize: BRECQ] -> save.

Differences forced by the environment (no network, no timm / torchvision, no ImageNet on the build/GPU boxes):
  * models come from adalog_amd.utils.models (timm-compatible names); `./checkpoints/vit_raw/<timm name>.bin` is loaded
    when present (test_quant.py:181-182), otherwise seeded random-init weights are used;
  * `--dataset synthetic` (default) draws calibration/validation images from torch.randn with the run's seed and
    reports *fidelity to the FP model* (top-1 agreement, logit SQNR) instead of ImageNet accuracy;
  * `--dataset <root>` walks an ImageNet folder tree (train/, val/) with adalog_amd.utils.datasets (PIL + a table of the
    timm data configs: no torchvision) and reports Prec@1 / Prec@5 through adalog_amd.utils.test_utils.validate.
The output directory is created when main() runs, not at import time (the reference does it on import, test_quant.py:21-29).
"""
import argparse
import copy
import importlib
import logging
import os
import sys
import time
from datetime import datetime

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from adalog_amd import parallel  # noqa: E402
from adalog_amd.utils.calibrator import QuantCalibrator  # noqa: E402
from adalog_amd.utils.models import MODEL_ZOO, create_model  # noqa: E402
from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net  # noqa: E402


def get_args_parser():
    parser = argparse.ArgumentParser(add_help=False)
    parser.add_argument("--model", default="deit_small", choices=list(MODEL_ZOO), help="model")
    parser.add_argument('--config', type=str, default="./configs/4bit.py", help="File path to import Config class from")
    parser.add_argument('--dataset', default="synthetic", help="'synthetic' or the root of an ImageNet folder tree (train/ and val/)")
    parser.add_argument("--calib-size", default=argparse.SUPPRESS, type=int, help="size of calibration set")
    parser.add_argument("--calib-batch-size", default=argparse.SUPPRESS, type=int, help="batchsize of calibration set")
    parser.add_argument("--val-batch-size", default=200, type=int, help="batchsize of validation set")
    parser.add_argument("--val-size", default=256, type=int, help="synthetic validation images")
    parser.add_argument("--num-workers", default=8, type=int, help="number of data loading workers (default: 8)")
    parser.add_argument("--device", default="cuda", type=str, help="device")
    calibrate_mode_group = parser.add_mutually_exclusive_group()
    calibrate_mode_group.add_argument('--calibrate', action='store_true', help="Calibrate the model")
    calibrate_mode_group.add_argument('--load-calibrate-checkpoint', type=str, default=None,
                                      help="Path to the calibrated checkpoint.")
    parser.add_argument('--test-calibrate-checkpoint', action='store_true', help='validate the calibrated checkpoint.')
    optimize_mode_group = parser.add_mutually_exclusive_group()
    optimize_mode_group.add_argument('--optimize', action='store_true', help="Optimize the model")
    optimize_mode_group.add_argument('--load-optimize-checkpoint', type=str, default=None,
                                     help="Path to the optimized checkpoint.")
    parser.add_argument('--test-optimize-checkpoint', action='store_true', help='validate the optimized checkpoint.')
    parser.add_argument("--print-freq", default=10, type=int, help="print frequency")
    parser.add_argument("--seed", default=5, type=int, help="seed")
    parser.add_argument('--w_bit', type=int, default=argparse.SUPPRESS, help='bit-precision of weights')
    parser.add_argument('--a_bit', type=int, default=argparse.SUPPRESS, help='bit-precision of activation')
    parser.add_argument('--s_bit', type=int, default=argparse.SUPPRESS, help='bit-precision of post softmax activation')
    parser.add_argument('--optim-iters', type=int, default=20000, help='BRECQ iterations per block (reference: 20000)')
    parser.add_argument('--output-dir', type=str, default=None, help='default ./checkpoints/quant_result/<timestamp>')
    return parser


def seed_all(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_cur_time():
    return datetime.now().strftime("%Y-%m-%d %H:%M:%S")


def save_model(model, args, cfg, root_path, mode='calibrate'):
    """test_quant.py:95-106: same file naming, plain state_dict."""
    assert mode in ['calibrate', 'optimize']
    if mode == 'calibrate':
        auto_name = '{}_w{}_a{}_s{}_calibsize_{}.pth'.format(args.model, cfg.w_bit, cfg.a_bit, cfg.s_bit, cfg.calib_size)
    else:
        auto_name = '{}_w{}_a{}_s{}_optimsize_{}.pth'.format(args.model, cfg.w_bit, cfg.a_bit, cfg.s_bit, cfg.optim_size)
    save_path = os.path.join(root_path, auto_name)
    if parallel.rank() == 0:
        logging.info(f"Saving checkpoint to {save_path}")
        torch.save(model.state_dict(), save_path)
    return save_path


def load_model(model, ckpt_path, device):
    """test_quant.py:109-127."""
    for name, module in model.named_modules():
        if hasattr(module, 'mode'):
            module.calibrated = True
            module.mode = 'quant_forward'
        if isinstance(module, nn.Linear) and 'reduction' in name:
            module.bias = nn.Parameter(torch.zeros(module.out_features))
        for attr in ['a_quantizer', 'w_quantizer', 'A_quantizer', 'B_quantizer']:
            if hasattr(module, attr):
                getattr(module, attr).inited = True
    ckpt = torch.load(ckpt_path, map_location="cpu")
    result = model.load_state_dict(ckpt, strict=False)
    logging.info(str(result))
    model.to(device)
    model.eval()
    return model


def finish_training(model):
    for name, module in model.named_modules():
        if hasattr(module, 'mode') and hasattr(module, 'reparam_bias'):
            module.reparam_bias()


def synthetic_images(n, seed, img_size=224):
    return torch.randn(n, 3, img_size, img_size, generator=torch.Generator().manual_seed(seed))


def make_loader(images, batch_size, device, shard=True):
    if shard:
        lo, hi = parallel.shard_slice(images.shape[0])
        images = images[lo:hi]
    images = images.to(device)
    return [(images[i:i + batch_size], None) for i in range(0, images.shape[0], batch_size)]


def imagenet_batches(loader, device, shard=True):
    """materialises a DataLoader as the list of (images on the device, labels) batches the calibrator / BRECQ walk,
    keeping this rank's shard of the images when several ranks calibrate together"""
    xs, ys = zip(*[(x, y) for x, y in loader])
    x, y = torch.cat(xs), torch.cat(ys)
    bs = xs[0].shape[0]
    if shard:
        lo, hi = parallel.shard_slice(x.shape[0])
        x, y = x[lo:hi], y[lo:hi]
    x = x.to(device)
    return [(x[i:i + bs], y[i:i + bs].to(device)) for i in range(0, x.shape[0], bs)]


def validate_dataset(loader, model, full_model, device):
    """top-1 / top-5 of the quantised model on labelled validation data (reference test_utils.validate) next to the FP model's"""
    from adalog_amd.utils.test_utils import validate
    crit = nn.CrossEntropyLoss().to(device)
    _, q1, q5 = validate(loader, model, crit, device=device)
    _, f1, f5 = validate(loader, full_model, crit, device=device)
    logging.info(f" * quantised Prec@1 {q1:.3f} Prec@5 {q5:.3f}   (FP model Prec@1 {f1:.3f} Prec@5 {f5:.3f})")
    return q1, q5


@torch.no_grad()
def validate_fidelity(loader, model, full_model):
    """Synthetic-data stand-in for test_utils.validate: agreement of the quantised model with the FP model."""
    from adalog_amd.utils.graph_forward import GraphedForward
    agree, total, num, den = 0, 0, 0.0, 0.0
    model = GraphedForward(model)                                # (launch-bound at these batch sizes: captured once per shape)
    for x, _ in loader:
        q, f = model(x), full_model(x)
        agree += (q.argmax(-1) == f.argmax(-1)).sum().item()
        total += x.shape[0]
        num += (f ** 2).sum().item()
        den += ((q - f) ** 2).sum().item()
    sqnr = 10 * np.log10(num / max(den, 1e-30))
    logging.info(f" * top-1 agreement with FP model {100.0 * agree / max(total, 1):.2f}%   logit SQNR {sqnr:.2f} dB")
    return agree / max(total, 1), sqnr


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        # RCCL ("nccl") is the product path.  ADALOG_DIST_BACKEND=gloo lets several ranks share ONE GPU (a functional check
        # of the sharded path on a single-GPU box: the collectives then stage through the host)
        be_name = os.environ.get("ADALOG_DIST_BACKEND", "nccl")
        if args.device.startswith("cuda") and be_name == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=parallel.dist_timeout())
        else:
            if args.device.startswith("cuda"):
                torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend="gloo", timeout=parallel.dist_timeout())
    root_path = args.output_dir or './checkpoints/quant_result/{}'.format(datetime.now().strftime("%Y%m%d_%H%M"))
    if parallel.rank() == 0:
        os.makedirs(root_path, exist_ok=True)
    parallel.barrier()
    logging.basicConfig(level=logging.INFO if parallel.rank() == 0 else logging.WARNING, format='%(message)s',
                        handlers=[logging.StreamHandler()] +
                                 ([logging.FileHandler('{}/output.log'.format(root_path))] if parallel.rank() == 0 else []))
    logging.info("{} - start the process.".format(get_cur_time()))
    logging.info(str(args))
    dir_path = os.path.dirname(os.path.abspath(args.config))
    if dir_path not in sys.path:
        sys.path.append(dir_path)
    module_name = os.path.splitext(os.path.basename(args.config))[0]
    Config = getattr(importlib.import_module(module_name), 'Config')
    logging.info("Successfully imported Config class!")
    cfg = Config()
    for k in ('calib_size', 'calib_batch_size', 'w_bit', 'a_bit', 's_bit'):
        if hasattr(args, k):
            setattr(cfg, k, getattr(args, k))
    for name, value in vars(cfg).items():
        logging.info(f"{name}: {value}")

    if args.device.startswith('cuda'):
        device = torch.device('cuda', (local_rank % max(1, torch.cuda.device_count())) if world > 1
                              else (int(args.device.split(':')[1]) if ':' in args.device else 0))
        torch.cuda.set_device(device)
    else:
        device = torch.device(args.device)
    seed_all(args.seed)

    logging.info('Building model ...')
    model = create_model(args.model)
    raw_ckpt = './checkpoints/vit_raw/{}.bin'.format(MODEL_ZOO[args.model])
    if os.path.exists(raw_ckpt):
        logging.info(f"loading FP weights from {raw_ckpt}")
        logging.info(str(model.load_state_dict(torch.load(raw_ckpt, map_location="cpu"), strict=False)))
    else:
        logging.info("no FP checkpoint found: seeded random-init weights")
    full_model = copy.deepcopy(model).to(device).eval()
    model.to(device).eval()

    img_size = 384 if args.model.endswith("384") else 224
    loader_gen = None
    if args.dataset != "synthetic":
        # reference test_quant.py:167-171: ViTImageNetLoaderGenerator(root, val_batch_size, num_workers, kwargs={"model": model})
        from adalog_amd.utils.datasets import ViTImageNetLoaderGenerator
        loader_gen = ViTImageNetLoaderGenerator(args.dataset, args.val_batch_size, args.num_workers, kwargs={"model": args.model})
        logging.info("known deviation from the reference's preprocessing: timm is not available, so the model's data config comes "
                     "from a table of the timm 0.9.2 defaults and the calibration-side transform omits timm's colour jitter; "
                     "Prec@1 is not bit-comparable with the reference's (adalog_amd/utils/datasets.py)")
        val_loader = loader_gen.val_loader()
        validate_any = lambda m: validate_dataset(val_loader, m, full_model, device)
    else:
        val_loader = make_loader(synthetic_images(args.val_size, args.seed + 1, img_size), args.val_batch_size, device, shard=False)
        validate_any = lambda m: validate_fidelity(val_loader, m, full_model)

    reparam = args.load_calibrate_checkpoint is None and args.load_optimize_checkpoint is None
    logging.info('Wraping quantiztion modules (reparam: {}) ...'.format(reparam))
    model = wrap_modules_in_net(model, cfg, reparam=reparam)
    model.to(device).eval()

    if not args.load_optimize_checkpoint:
        if args.load_calibrate_checkpoint:
            logging.info(f"Restoring checkpoint from '{args.load_calibrate_checkpoint}'")
            model = load_model(model, args.load_calibrate_checkpoint, device)
            if args.test_calibrate_checkpoint:
                validate_any(model)
        else:
            logging.info("{} - start calibration".format(get_cur_time()))
            if loader_gen is not None:
                calib_loader = imagenet_batches(loader_gen.calib_loader(num=cfg.calib_size, batch_size=cfg.calib_batch_size,
                                                                        seed=args.seed), device)
            else:
                calib_loader = make_loader(synthetic_images(cfg.calib_size, args.seed, img_size), cfg.calib_batch_size, device)
            t0 = time.perf_counter()
            calibrator = QuantCalibrator(model, calib_loader)
            calibrator.batching_quant_calib()
            if device.type == "cuda":
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            model = wrap_reparamed_modules_in_net(model)
            model.to(device)
            logging.info("{} - calibration finished: {:.2f} s, {:.2f} calib-images/s on {} GPU(s)".format(
                get_cur_time(), dt, cfg.calib_size / dt, world))
            if not args.optimize:
                finish_training(model)
            save_model(model, args, cfg, root_path, mode='calibrate')
            logging.info('Validating after calibration ...')
            validate_any(model)

    if args.optimize:
        from adalog_amd.utils.block_recon import BlockReconstructor
        logging.info('Building calibrator ...')
        if loader_gen is not None:
            calib_loader = imagenet_batches(loader_gen.calib_loader(num=cfg.optim_size, batch_size=cfg.optim_batch_size,
                                                                    seed=args.seed), device)
        else:
            calib_loader = make_loader(synthetic_images(cfg.optim_size, args.seed, img_size), cfg.optim_batch_size, device)
        logging.info("{} - start block reconstruction".format(get_cur_time()))
        block_reconstructor = BlockReconstructor(model, full_model, calib_loader)
        block_reconstructor.reconstruct_model(quant_act=cfg.train_act, keep_gpu=cfg.keep_gpu, iters=args.optim_iters)
        finish_training(model)
        logging.info("{} - block reconstruction finished.".format(get_cur_time()))
        save_model(model, args, cfg, root_path, mode='optimize')
    if args.load_optimize_checkpoint:
        model = load_model(model, args.load_optimize_checkpoint, device)
    if args.optimize or args.test_optimize_checkpoint:
        logging.info('Validating after block reconstruction ...')
        validate_any(model)
    logging.info("{} - finished the process.".format(get_cur_time()))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(parents=[get_args_parser()])
    main(parser.parse_args())
