"""Top-1 / top-5 validation with the interface of the reference's ``utils/test_utils.py:10-54`` (``validate`` returns
``(loss, top1, top5)`` averaged over the loader; ``accuracy``; ``AverageMeter``).  The forward passes run on whatever the
model's layers dispatch to -- calibrated ``quant_forward`` layers use the HIP kernels."""
import logging
import time

import torch


class AverageMeter:
    """running value / sum / count / mean"""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


def accuracy(output, target, topk=(1,)):
    """percentage of samples whose label is among the k largest logits, for every k in ``topk``"""
    kmax = max(topk)
    pred = output.topk(kmax, dim=1, largest=True, sorted=True).indices          # [B, kmax]
    hit = pred.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum() * (100.0 / target.numel()) for k in topk]


@torch.no_grad()
def validate(val_loader, model, criterion, print_freq=10, device='cuda:0', graph=True):
    """reference test_utils.py:10-54.  ``graph``: replay the forward as a captured HIP graph per batch shape
    (utils/graph_forward.py; the forward is launch-bound at validation batch sizes) -- same kernels, same results."""
    from .graph_forward import GraphedForward
    losses, top1, top5, batch_time = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    model.eval()
    model = GraphedForward(model, enabled=graph and str(device).startswith('cuda'))
    t_start = t_last = time.time()
    n_batches = len(val_loader)
    for i, (data, target) in enumerate(val_loader):
        data, target = data.to(device), target.to(device)
        output = model(data)
        loss = criterion(output, target)
        p1, p5 = accuracy(output, target, topk=(1, min(5, output.shape[1])))
        n = data.size(0)
        losses.update(float(loss), n)
        top1.update(float(p1), n)
        top5.update(float(p5), n)
        now = time.time()
        batch_time.update(now - t_last)
        t_last = now
        if i % print_freq == 0:
            logging.info(f"Test: [{i}/{n_batches}]\tTime {batch_time.val:.3f} ({batch_time.avg:.3f})\tLoss {losses.val:.4f} "
                         f"({losses.avg:.4f})\tPrec@1 {top1.val:.3f} ({top1.avg:.3f})\tPrec@5 {top5.val:.3f} ({top5.avg:.3f})")
    logging.info(f" * Prec@1 {top1.avg:.3f} Prec@5 {top5.avg:.3f} Loss {losses.avg:.3f} Time {time.time() - t_start:.3f}")
    return losses.avg, top1.avg, top5.avg
