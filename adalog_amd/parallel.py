"""Image-sharded data parallelism for calibration (one process per GPU, torch.distributed; backend "nccl" = RCCL).

The reference is single-device (test_quant.py:156-160).  Calibration images are independent: every search score is a
sum over images, so each rank keeps its contiguous slice of every module's captured activations and the
only data-path collective is an all-reduce(SUM) of the small [P, cols] score tensor per scoring call (<= 1.5 MB,
latency-bound on xGMI).  RCCL delivers bit-identical sums to every rank, so the deterministic top-k that follows picks
the same survivors everywhere.  Global order statistics (percentile candidates) use a distributed radix select: the
per-rank histograms are all-reduced (search._sharded_quantiles), the activations stay where they are.
"""
import threading

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


# ------------------------------------------------------------------------------------------------ lanes
# ~4 000 score all-reduces per ViT calibration (one per scoring call: <= 1.5 MB, latency-bound on xGMI), each followed by a
# top-k that needs its result: issued on the search stream they serialise with the GEMMs.  Modules are calibrated
# independently of each other (every capture is of the FP model, reference utils/calibrator.py:34-67), so the calibrator runs
# TWO modules' searches side by side -- each in its own host thread, on its own HIP stream and its own communicator (a "lane");
# while one lane waits for its all-reduce the other lane's GEMMs keep the GPU busy.  Within a lane the sequence of collectives
# is the same on every rank (modules are dealt to the lanes round-robin in named_modules() order), and RCCL gives every rank
# bit-identical sums, so the results are those of the sequential schedule.
_tls = threading.local()
_LANE_GROUPS = []
STATS = {"collectives": 0, "bytes": 0}
_EVENTS = []                               # (start, end) device events around every collective (HIP streams only)
_stats_lock = threading.Lock()


def lane_groups(n: int):
    """n process groups over all ranks, created once, in the same order on every rank (dist.new_group is collective)."""
    while len(_LANE_GROUPS) < n:
        _LANE_GROUPS.append(dist.new_group(ranks=list(range(dist.get_world_size()))))
    return _LANE_GROUPS[:n]


def set_lane(group):
    """Collectives of the calling THREAD use `group` from now on (None: the default group)."""
    _tls.group = group


def _group():
    return getattr(_tls, "group", None)


def _all_reduce(t: torch.Tensor, op) -> torch.Tensor:
    if not is_dist():
        return t
    ev = None
    if t.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_reduce(t, op=op, group=_group())
    if ev is not None:
        ev[1].record()
    with _stats_lock:
        STATS["collectives"] += 1
        STATS["bytes"] += t.numel() * t.element_size()
        if ev is not None:
            _EVENTS.append(ev)
    return t


def reset_stats():
    with _stats_lock:
        STATS["collectives"] = 0
        STATS["bytes"] = 0
        _EVENTS.clear()


def collective_stats():
    """-> {collectives, bytes, device_ms}: count / payload / summed stream time of the collectives since reset_stats() (this
    rank).  Synchronises the device."""
    with _stats_lock:
        evs = list(_EVENTS)
        out = dict(STATS)
    if evs:
        torch.cuda.synchronize()
        out["device_ms"] = sum(a.elapsed_time(b) for a, b in evs)
    else:
        out["device_ms"] = None
    return out


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    return _all_reduce(t, dist.ReduceOp.SUM)


def all_reduce_max(t: torch.Tensor) -> torch.Tensor:
    return _all_reduce(t, dist.ReduceOp.MAX)


def all_reduce_min(t: torch.Tensor) -> torch.Tensor:
    return _all_reduce(t, dist.ReduceOp.MIN)


def all_reduce_mean_bucket(tensors):
    """Mean over the ranks of every tensor in `tensors`, in place, with ONE collective: the tensors are flattened into one
    bucket, all-reduced and copied back (BRECQ: the gradients of alpha and of the activation scales, a few MB per iteration -- one
    ring all-reduce over xGMI instead of one latency-bound call per parameter)."""
    if not is_dist() or not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    _all_reduce(flat, dist.ReduceOp.SUM)
    flat.div_(world_size())
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


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
