"""QuantCalibrator -- the outer calibration loop of reference utils/calibrator.py:9-67, device-resident.

Same contract: for every module that has ``calibrated == False`` (visited in ``named_modules()`` order) capture its
input(s) and output over the whole calibration set with all modules in 'raw' mode, attach them as ``raw_input`` /
``raw_out``, call ``hyperparameter_searching()`` and, for layers with a ``prev_layer``, ``reparam()``; finally switch
every module to 'quant_forward'.

What is different, by design:
  * captures stay on the GPU (the reference moves every activation to the host and back, calibrator.py:17-28 and
    linear.py:134,145,...); 288 GB of HBM holds a whole block's captures for 1024 images;
  * ``capture='block'`` (default) records all un-calibrated modules that share a transformer block in ONE forward
    pass instead of one full-model pass per module (74 passes for ViT, 149 for Swin).  This is legal because every
    module stays 'raw' until the end (calibrator.py:65-67) and the LayerNorm fold of reparam() is function preserving
    (linear.py:604-611), so later captures are unchanged up to fp32 rounding.  ``capture='module'`` reproduces the
    reference's pass structure exactly;
  * the forward pass stops right after the last hooked module of the group has run;
  * with torch.distributed initialised the loader is expected to yield this rank's image shard; searches all-reduce
    their scores (adalog_amd.parallel).
"""
import os
import re
import time

import torch

from ..quant_layers import MinMaxQuantConv2d, MinMaxQuantLinear, MinMaxQuantMatMul


class _StopForward(Exception):
    pass


class QuantCalibrator:
    def __init__(self, model, calib_loader, capture: str = "block", verbose: bool = False):
        assert capture in ("block", "module")
        self.model = model
        self.calib_loader = calib_loader
        self.capture = capture
        self.verbose = verbose
        self.timings = {}                    # module name -> HOST seconds enqueueing hyperparameter_searching (+ reparam)
        self.capture_seconds = 0.0
        self._events = {}                    # module name -> (start, end) device events around its search (+ reparam)
        self._capture_events = []
        # Finished blocks answer from a cache.  The reference re-runs the whole network for every module it calibrates
        # (calibrator.py:44-47); with one pass per block that is still (depth + 1) / 2 full forwards per calibration.  A block
        # whose modules are all calibrated (and re-parameterised) no longer changes, nor does anything upstream of it, so its
        # output for calibration batch i is a constant: it is recorded the first time the block runs after being finished (during
        # the NEXT block's capture pass) and returned from then on instead of being recomputed -- the tensors downstream are
        # bit for bit the ones a full forward produces.  ADALOG_CAPTURE_CACHE=0: recompute.
        # the cache replays a block's outputs by batch POSITION: it needs a loader that yields the same batches on every pass
        # (an in-memory list, which is what test_quant.py / bench.py hand over) -- a DataLoader with a random transform does not
        self._cache_blocks = (capture == "block" and os.environ.get("ADALOG_CAPTURE_CACHE", "1") != "0"
                              and isinstance(calib_loader, (list, tuple)))
        self._finished = []                  # finished blocks not cached yet
        self._patched = []                   # blocks whose forward returns the cached outputs
        self._bi = 0                         # index of the calibration batch in flight

    # hooks keep the reference's names (calibrator.py:14-28); tensors stay on the device
    def single_input_forward_hook(self, module, inp, outp):
        if module.tmp_input is None:
            module.tmp_input = []
        module.tmp_input.append(inp[0].detach())

    def double_input_forward_hook(self, module, inp, outp):
        if module.tmp_input is None:
            module.tmp_input = [[], []]
        module.tmp_input[0].append(inp[0].detach())
        module.tmp_input[1].append(inp[1].detach())

    def outp_forward_hook(self, module, inp, outp):
        if module.tmp_out is None:
            module.tmp_out = []
        module.tmp_out.append(outp.detach())

    @staticmethod
    def _group_key(name: str) -> str:
        m = re.match(r"^(.*?blocks\.\d+)\.", name)
        return m.group(1) if m else name

    def _pending(self):
        return [(n, m) for n, m in self.model.named_modules() if hasattr(m, 'calibrated') and not m.calibrated]

    def _capture(self, group):
        """One pass over the calibration set recording inputs/outputs of every module in ``group``."""
        t0 = time.perf_counter()
        device = next(self.model.parameters()).device
        cev = self._event_pair(device)
        hooks = []
        last = group[-1][1]
        for _, module in group:
            hooks.append(module.register_forward_hook(self.outp_forward_hook))
            if isinstance(module, (MinMaxQuantLinear, MinMaxQuantConv2d)):
                hooks.append(module.register_forward_hook(self.single_input_forward_hook))
            if isinstance(module, MinMaxQuantMatMul):
                hooks.append(module.register_forward_hook(self.double_input_forward_hook))

        def stop(module, inp, outp):
            raise _StopForward()
        hooks.append(last.register_forward_hook(stop))
        recording = []
        for block in self._finished:
            outs = []
            recording.append((block, outs))
            hooks.append(block.register_forward_hook(lambda m, i, o, _outs=outs: _outs.append(o.detach() if torch.is_tensor(o) else o)))
        n_batches = 0
        with torch.no_grad():
            for bi, (inp, _) in enumerate(self.calib_loader):
                self._bi = bi
                n_batches += 1
                try:
                    self.model(inp.to(device, non_blocking=True))
                except _StopForward:
                    pass
        for h in hooks:
            h.remove()
        for block, outs in recording:
            if len(outs) == n_batches:                       # (the pass ran through it for every batch)
                # Only the most recently patched block's cache is ever consumed (a patched block ignores its input): the
                # blocks patched before it keep ONE batch of theirs, sliced to the incoming batch size, so that the glue code
                # between blocks still sees tensors of the right shape -- depth x calib_size x tokens x dim of cache (7 GB for
                # vit_base at 1024 images) becomes one block's worth.
                for old in self._patched:
                    first = getattr(old, "_adalog_cache_first", None)
                    if first is not None and torch.is_tensor(first):
                        old.forward = lambda *a, _o=first, **k: _o[:a[0].shape[0]] if torch.is_tensor(a[0]) else _o
                block.forward = lambda *a, _outs=outs, **k: _outs[self._bi]
                block._adalog_cache_first = outs[0] if outs and torch.is_tensor(outs[0]) else None
                self._patched.append(block)
                self._finished.remove(block)
        for _, module in group:
            module.raw_out = torch.cat(module.tmp_out, dim=0)
            if isinstance(module, (MinMaxQuantLinear, MinMaxQuantConv2d)):
                module.raw_input = torch.cat(module.tmp_input, dim=0)
            if isinstance(module, MinMaxQuantMatMul):
                module.raw_input = [torch.cat(t, dim=0) for t in module.tmp_input]
            module.tmp_input = module.tmp_out = None
        self.capture_seconds += time.perf_counter() - t0
        if cev is not None:
            cev[1].record()
            self._capture_events.append(cev)

    @staticmethod
    def _event_pair(device):
        """(start, end) events on the current stream, start already recorded; None off-GPU"""
        if torch.device(device).type != "cuda":
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def fpcs_seconds(self):
        """DEVICE seconds per module: events on the search stream around hyperparameter_searching (+ reparam), i.e. the
        reference's `FPCS wall-clock = sum over modules` without the host running ahead.  Synchronises."""
        if not self._events:
            return {}
        torch.cuda.synchronize()
        return {n: a.elapsed_time(b) * 1e-3 for n, (a, b) in self._events.items()}

    def capture_device_seconds(self):
        if not self._capture_events:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._capture_events) * 1e-3

    def _search_one(self, name, module, device):
        t0 = time.perf_counter()
        ev = self._event_pair(device)
        with torch.no_grad():
            module.hyperparameter_searching()
            if hasattr(module, 'prev_layer') and module.prev_layer is not None:
                module.reparam()
        if ev is not None:
            ev[1].record()
            self._events[name] = ev
        self.timings[name] = time.perf_counter() - t0      # host-side enqueue time; the stream runs behind
        if self.verbose:
            print(f"calibrated {name}")

    def _lanes(self, device):
        """Two (communicator, stream) lanes when the images are sharded over several ranks (adalog_amd.parallel): one module's
        score all-reduce then hides under the other module's scoring GEMMs.  ADALOG_INTERLEAVE=0: the sequential schedule."""
        from .. import parallel
        on_gpu = torch.device(device).type == "cuda"
        if not parallel.is_dist():
            # one process: ADALOG_LANES=n (>= 2) runs n modules' searches side by side on n streams (no communicators)
            n = int(os.environ.get("ADALOG_LANES", "1"))
            return [(None, torch.cuda.Stream(device=device)) for _ in range(n)] if n >= 2 and on_gpu else None
        # Two communicators driven from two host threads are opt-in (ADALOG_INTERLEAVE=1): the schedule has run under gloo and
        # with two ranks sharing one GPU only, never on a multi-GPU RCCL node, and concurrent collectives on different
        # communicators must reach the device in the same relative order on every rank (parallel.Sequencer enforces one global
        # host-side issue order, a pure function of (lane, call index)) -- until that is measured on hardware the default is
        # the sequential schedule: one communicator, one stream, one order.
        if os.environ.get("ADALOG_INTERLEAVE", "0") != "1":
            return None
        groups = parallel.lane_groups(2)
        return [(g, torch.cuda.Stream(device=device) if on_gpu else None) for g in groups]

    def _search_interleaved(self, group, device, lanes):
        """The group's modules dealt round-robin to the lanes (same deal on every rank: each lane's collectives then follow the
        same sequence everywhere), one host thread per lane.  Modules are independent (their captures are taken, all of the FP
        model: reference utils/calibrator.py:34-67; reparam() touches only the module's own weight and its LayerNorm)."""
        import threading
        from .. import parallel
        on_gpu = torch.device(device).type == "cuda"
        ready = torch.cuda.Event() if on_gpu else None
        if on_gpu:
            ready.record()                                     # the captures were written on the calling stream
        errors, done = [], []
        seq = parallel.Sequencer(len(lanes)) if parallel.is_dist() else None

        def work(li):
            grp, stream = lanes[li]
            parallel.set_lane(grp, seq, li)
            try:
                if on_gpu:
                    torch.cuda.set_device(device)
                    with torch.cuda.stream(stream):
                        stream.wait_event(ready)
                        for name, module in group[li::len(lanes)]:
                            for t in self._captured_tensors(module):
                                t.record_stream(stream)        # allocated on the capture stream, read (and freed) on this one
                            self._search_one(name, module, device)
                        ev = torch.cuda.Event()
                        ev.record()
                        done.append(ev)
                else:
                    for name, module in group[li::len(lanes)]:
                        self._search_one(name, module, device)
            except BaseException as ex:                        # re-raised on the calling thread
                errors.append(ex)
            finally:
                if seq is not None:
                    seq.finish(li)
                parallel.set_lane(None)

        threads = [threading.Thread(target=work, args=(li,), name=f"adalog-lane{li}") for li in range(len(lanes))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        for ev in done:
            torch.cuda.current_stream().wait_event(ev)

    @staticmethod
    def _captured_tensors(module):
        out = []
        for v in (getattr(module, "raw_input", None), getattr(module, "raw_out", None)):
            if isinstance(v, (list, tuple)):
                out += [t for t in v if torch.is_tensor(t) and t.is_cuda]
            elif torch.is_tensor(v) and v.is_cuda:
                out.append(v)
        return out

    def batching_quant_calib(self):
        pending = self._pending()
        groups = []
        for name, module in pending:
            key = self._group_key(name) if self.capture == "block" else name
            if groups and groups[-1][0] == key:
                groups[-1][1].append((name, module))
            else:
                groups.append((key, [(name, module)]))
        device = next(self.model.parameters()).device
        lanes = self._lanes(device)
        try:
            for key, group in groups:
                self._capture(group)
                if lanes is None or len(group) < 2:
                    for name, module in group:
                        self._search_one(name, module, device)
                else:
                    self._search_interleaved(group, device, lanes)
                if self._cache_blocks and re.search(r"blocks\.\d+$", key):
                    self._finished.append(self.model.get_submodule(key))
        finally:
            for block in self._patched:                      # back to computing (the instance attribute shadowed the method)
                del block.forward
                block.__dict__.pop("_adalog_cache_first", None)
            self._patched, self._finished = [], []
        for _, module in self.model.named_modules():
            if hasattr(module, 'mode'):
                module.mode = "quant_forward"
