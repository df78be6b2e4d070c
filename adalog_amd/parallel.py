"""Image-sharded data parallelism for calibration (one process per GPU, torch.distributed; backend "nccl" = RCCL).

The reference is single-device (test_quant.py:156-160).  Calibration images are independent: every search score is a
sum over images, so each rank keeps its contiguous slice of every module's captured activations and the
only data-path collective is an all-reduce(SUM) of the small [P, cols] score tensor per scoring call (<= 1.5 MB,
latency-bound on xGMI).  RCCL delivers bit-identical sums to every rank, so the deterministic top-k that follows picks
the same survivors everywhere.  Global order statistics (percentile candidates) use a distributed radix select: the
per-rank histograms are all-reduced (search._sharded_quantiles), the activations stay where they are.
"""
import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def shard_slice(n_images: int):
    """Contiguous image slice [lo, hi) owned by this rank; n_images must divide evenly (weak scaling keeps it so)."""
    ws = world_size()
    if n_images % ws != 0:
        raise ValueError(f"calibration set of {n_images} images does not divide over {ws} ranks")
    per = n_images // ws
    return rank() * per, (rank() + 1) * per


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_reduce_min(t: torch.Tensor) -> torch.Tensor:
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


def gather_images(x: torch.Tensor) -> torch.Tensor:
    """All ranks' shards concatenated along the image axis (dim 0), in rank order."""
    if not is_dist():
        return x
    x = x.contiguous()
    out = torch.empty((x.shape[0] * world_size(),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out


def barrier():
    if is_dist():
        dist.barrier()
