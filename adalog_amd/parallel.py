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

_tls = threading.local()


def _real_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def is_dist() -> bool:
    """Several ranks AND the calling thread is not inside solo() (rank-local work such as a block owner's BRECQ training)."""
    return _real_dist() and not getattr(_tls, "solo", False)


class solo:
    """Context: the calling thread behaves as a single process (world_size() == 1, rank() == 0, collectives are no-ops).
    Used where the work is partitioned over the ranks with no exchange at all: block-parallel BRECQ (each rank trains the
    blocks it owns exactly as one process would)."""

    def __enter__(self):
        self._prev = getattr(_tls, "solo", False)
        _tls.solo = True
        return self

    def __exit__(self, *exc):
        _tls.solo = self._prev


def real_world_size() -> int:
    return dist.get_world_size() if _real_dist() else 1


def real_rank() -> int:
    return dist.get_rank() if _real_dist() else 0


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
_LANE_GROUPS = []
STATS = {"collectives": 0, "bytes": 0}
_EVENTS = []                               # (start, end) device events around every collective (HIP streams only)
_stats_lock = threading.Lock()


def lane_groups(n: int):
    """n process groups over all ranks, created once, in the same order on every rank (dist.new_group is collective)."""
    while len(_LANE_GROUPS) < n:
        _LANE_GROUPS.append(dist.new_group(ranks=list(range(dist.get_world_size()))))
    return _LANE_GROUPS[:n]


class Sequencer:
    """One global issue order for the collectives of all lanes: call c of lane L goes out after call c of every lane before L
    and call c - 1 of every lane after L -- a pure function of (call index, lane), hence the same on every rank.  RCCL / NCCL
    require collectives on DIFFERENT communicators of one device to be issued in the same relative order everywhere; per-lane
    order alone (what the two host threads give) leaves lane A against lane B to thread timing.  A lane that has finished its
    modules retires (the number of its collectives is the same on every rank, so it retires at the same point everywhere)."""

    def __init__(self, n):
        self.n, self.count, self.done = n, [0] * n, [False] * n
        self.cv = threading.Condition()
        self.order = []                                    # (lane, call index) in issue order (tests read it)

    def _my_turn(self, lane):
        c = self.count[lane]
        for o in range(self.n):
            if o == lane or self.done[o]:
                continue
            if self.count[o] < (c + 1 if o < lane else c):
                return False
        return True

    def issue(self, lane, fn):
        with self.cv:
            self.cv.wait_for(lambda: self._my_turn(lane))
            try:
                return fn()                                # enqueued while holding the turn: host order == the global order
            finally:
                self.order.append((lane, self.count[lane]))
                self.count[lane] += 1
                self.cv.notify_all()

    def finish(self, lane):
        with self.cv:
            self.done[lane] = True
            self.cv.notify_all()


def set_lane(group, sequencer=None, lane=0):
    """Collectives of the calling THREAD use `group` from now on (None: the default group), issued through `sequencer`."""
    _tls.group = group
    _tls.seq = sequencer if group is not None else None
    _tls.lane = lane


def _group():
    return getattr(_tls, "group", None)


# what this rank WOULD put on the fabric with several ranks: every all-reduce site is counted even in a one-rank run (the payloads --
# [P, cols] score tensors, min / max scalars, radix histograms -- do not depend on the world size), so that a one-GPU bench line
# already states the collective count and bytes of the N-GPU job (bench.py: collectives.planned_per_step)
PLANNED = {"collectives": 0, "bytes": 0}


def _all_reduce(t: torch.Tensor, op) -> torch.Tensor:
    if not getattr(_tls, "solo", False):
        with _stats_lock:
            PLANNED["collectives"] += 1
            PLANNED["bytes"] += t.numel() * t.element_size()
    if not is_dist():
        return t
    ev = None
    if t.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    seq = getattr(_tls, "seq", None)
    if seq is not None:
        seq.issue(_tls.lane, lambda: dist.all_reduce(t, op=op, group=_group()))
    else:
        dist.all_reduce(t, op=op, group=_group())
    if ev is not None:
        ev[1].record()
    with _stats_lock:
        STATS["collectives"] += 1
        STATS["bytes"] += t.numel() * t.element_size()
        if ev is not None:
            _EVENTS.append(ev)
    return t


def note_planned(nbytes: int, count: int = 1):
    """an all-reduce SITE that this (one-rank) run passed without issuing a collective -- e.g. an FPCS step whose scoring kernel ranked
    its own scores, where an N-rank run all-reduces the [P, cols] scores first: keeps PLANNED what an N-GPU job would issue"""
    if not getattr(_tls, "solo", False):
        with _stats_lock:
            PLANNED["collectives"] += int(count)
            PLANNED["bytes"] += int(nbytes)


def reset_stats():
    with _stats_lock:
        STATS["collectives"] = 0
        STATS["bytes"] = 0
        PLANNED["collectives"] = 0
        PLANNED["bytes"] = 0
        _EVENTS.clear()


def collective_stats():
    """-> {collectives, bytes, device_ms}: count / payload / summed stream time of the collectives since reset_stats() (this
    rank).  Synchronises the device."""
    with _stats_lock:
        evs = list(_EVENTS)
        out = dict(STATS)
        out["planned_collectives"], out["planned_bytes"] = PLANNED["collectives"], PLANNED["bytes"]
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


def broadcast(t: torch.Tensor, src: int) -> torch.Tensor:
    if is_dist():
        dist.broadcast(t, src=src)
    return t


def dist_timeout():
    """The timeout test_quant.py / bench.py hand to init_process_group: block-parallel BRECQ lets a rank wait in a collective for as
    long as another rank trains a block (minutes for Swin stage 0 at 20 000 iterations); the library default of 10 minutes would
    abort such a job.  ADALOG_DIST_TIMEOUT_MIN (default 120)."""
    import datetime
    import os
    return datetime.timedelta(minutes=int(os.environ.get("ADALOG_DIST_TIMEOUT_MIN", "120")))


def barrier():
    if is_dist():
        dist.barrier()
