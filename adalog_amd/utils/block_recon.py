"""BRECQ / AdaRound block reconstruction -- API and flow of reference utils/block_recon.py:17-238.

For every block (PatchEmbed, each transformer Block / SwinTransformerBlock, PatchMerging, and the classifier ``head``):
capture the block's input/output from the FP twin model, wrap the weight quantisers in AdaRoundQuantizers, and train the
rounding variables ``alpha`` (Adam, lr 1e-3) and -- with ``quant_act`` -- the activation scales (Adam lr 4e-5, cosine
decay) for ``iters`` steps of batch 32 against  rec_loss = sum_1(|delta|^2).mean()/10 (KL for the head)  plus, after
the 20 % warm-up, the rounding regulariser 0.01 * sum(1 - |2h-1|^b), b: 20 -> 2.  Finally commit hard rounding.

MI355X specifics: the straight-through fake-quant forward/backward passes, the AdaRound weight quantiser and the
regulariser are fused HIP kernels (csrc/brecq.hip); block inputs/outputs stay in HBM; with torch.distributed the
mini-batch is split over the ranks and the gradients of alpha / activation scales are all-reduced every iteration
(RCCL over xGMI; identical seeds keep randperm in lock-step).
"""
import logging
import math
import os

import torch
import torch.nn.functional as F

from .. import backend, parallel, train_mm
from ..quant_layers import MinMaxQuantConv2d, MinMaxQuantLinear, MinMaxQuantMatMul
from ..quantizers.adaround import AdaRoundQuantizer
from ..quantizers import adaround as adaround_mod
from .calibrator import QuantCalibrator
from . import models as M

# gather of the mini-batch (inputs, outputs) and the iteration's schedule row in ONE launch before a graph replay (adalog_brecq_prepare)
PREPARE_FUSED = os.environ.get("ADALOG_BRECQ_PREPARE", "1") != "0"


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam with its default options (what reference utils/block_recon.py:108-109 constructs), stepping ALL tensors
    of the optimiser in one launch of adalog_adam_multi (csrc/brecq.hip).  torch's fused multi-tensor kernel hands a
    65 536-element chunk to a block: ~40 blocks for the 2.4 M trained values of a deit_small block, 52 us per iteration.
    State (exp_avg, exp_avg_sq, a device step counter) lives on the device, so a captured HIP graph replays it; ``lr`` may be a
    device tensor (the cosine schedule of the activation scales writes it from the host)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._steps = {}               # (group, ids of the chunk's tensors) -> device step counter of that chunk
        self._hstep = {}               # id(param) -> steps taken so far (host mirror: decides which tensors may share a counter)

    def step_counter(self, params, group_index=0):
        """the device counter of the chunk made of exactly these tensors (None before their first step)"""
        return self._steps.get((group_index, tuple(id(p) for p in params)))

    @torch.no_grad()
    def step(self, closure=None):
        be = backend.get()
        for gi, group in enumerate(self.param_groups):
            # torch.optim.Adam skips a tensor without a gradient and counts steps per tensor: tensors are chunked with others of the
            # SAME step count only, and a chunk's device counter is keyed by its exact members -- a tensor that sits an iteration out
            # can neither shift others into a counter with a different history nor inherit one (in BRECQ every trained tensor has a
            # gradient in every iteration: one bucket, the same chunks and counters each time, nothing allocated after the first step)
            buckets = {}
            for p in group['params']:
                if p.grad is not None:
                    buckets.setdefault(self._hstep.get(id(p), 0), []).append(p)
            for hs, ps in buckets.items():
                for c0 in range(0, len(ps), 16):
                    chunk = ps[c0:c0 + 16]
                    for p in chunk:
                        st = self.state[p]
                        if not st:
                            st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
                    key = (gi, tuple(id(p) for p in chunk))
                    if key not in self._steps:
                        self._steps[key] = torch.full((1,), float(hs), dtype=torch.float32, device=chunk[0].device)
                    grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                    be.adam_multi([p.data for p in chunk], grads, [self.state[p]['exp_avg'] for p in chunk],
                                  [self.state[p]['exp_avg_sq'] for p in chunk], self._steps[key], group['lr'],
                                  group['betas'][0], group['betas'][1], group['eps'])
                    for p in chunk:
                        self._hstep[id(p)] = hs + 1

    def note_external_steps(self, params, n=1):
        """steps of these tensors taken outside step() on their shared device counter (the one-launch alpha update)"""
        for p in params:
            self._hstep[id(p)] = self._hstep.get(id(p), 0) + n


class BlockReconstructor(QuantCalibrator):
    def __init__(self, model, full_model, calib_loader):
        super().__init__(model, calib_loader)
        self.full_model = full_model
        self.blocks = {}
        self.full_blocks = {}
        types_of_block = (M.PatchEmbed, M.Block, M.SwinTransformerBlock, M.PatchMerging)
        for name, module in self.model.named_modules():
            if isinstance(module, types_of_block) or name.split('.')[-1] == 'head':
                self.blocks[name] = module
                BlockReconstructor._prepare_module_data_init(module)
        for name, module in self.full_model.named_modules():
            if isinstance(module, types_of_block) or name.split('.')[-1] == 'head':
                self.full_blocks[name] = module
                BlockReconstructor._prepare_module_data_init(module)

    @staticmethod
    def _prepare_module_data_init(module):
        module.raw_input = module.tmp_input = None
        module.raw_out = module.tmp_out = None

    @staticmethod
    def _alpha_collector(block, loss_func, w_optimizer, w_params):
        """The one-launch alpha update of a captured iteration (quantizers/adaround.py AlphaCollector), or None when it does not
        apply: HipAdam over exactly the block's soft-target AdaRound alphas (<= 16, fp32, contiguous), ADALOG_BRECQ_ALPHA_STEP != 0."""
        be = backend.get()
        if os.environ.get("ADALOG_BRECQ_ALPHA_STEP", "1") == "0" or not isinstance(w_optimizer, HipAdam) \
                or not hasattr(be, "alpha_step_multi") or loss_func.round_loss != 'relaxation':
            return None
        qs = [m.w_quantizer for _, m in block.named_modules()
              if hasattr(m, 'w_quantizer') and isinstance(m.w_quantizer, AdaRoundQuantizer)]
        if not (1 <= len(qs) <= 16) or len(w_optimizer.param_groups) != 1:
            return None
        if {id(q_.alpha) for q_ in qs} != {id(p_) for p_ in w_params} or len(w_params) != len(qs):
            return None
        if not all(q_.soft_targets and q_.round_mode == 'learned_hard_sigmoid' and not q_.sym and q_.alpha.is_cuda
                   and q_.alpha.dtype == torch.float32 and q_.alpha.is_contiguous() for q_ in qs):
            return None
        group = w_optimizer.param_groups[0]
        counter = w_optimizer.step_counter(group['params'])  # all alphas in one chunk, in the optimiser's order (<= 16 tensors)
        if counter is None or any(not w_optimizer.state[q_.alpha] for q_ in qs):
            return None                                      # the eager warm-up iterations create the Adam state
        return adaround_mod.AlphaCollector(qs, lambda p_: w_optimizer.state[p_], counter, group['lr'],
                                           group['betas'], group['eps'])

    def set_block_mode(self, block, mode='raw'):
        for _, module in block.named_modules():
            if hasattr(module, 'mode'):
                module.mode = mode

    def wrap_quantizers_in_net(self, block, name):
        for _, module in block.named_modules():
            if hasattr(module, 'w_quantizer'):
                if isinstance(module, MinMaxQuantLinear):
                    module.w_quantizer = AdaRoundQuantizer(
                        uq=module.w_quantizer,
                        weight_tensor=module.weight.view(module.n_V, module.crb_rows, module.in_features),
                        round_mode='learned_hard_sigmoid')
                elif isinstance(module, MinMaxQuantConv2d):
                    module.w_quantizer = AdaRoundQuantizer(
                        uq=module.w_quantizer, weight_tensor=module.weight.view(module.weight.shape[0], -1),
                        round_mode='learned_hard_sigmoid')
                module.w_quantizer.soft_targets = True

    def init_block_raw_data(self, block, full_block, name, device, keep_gpu=True):
        self.init_block_raw_inp_outp(block, full_block, name, device)
        if not keep_gpu:
            block.raw_input, block.raw_out = block.raw_input.cpu(), block.raw_out.cpu()

    def init_block_raw_inp_outp(self, block, full_block, name, device):
        """block_recon.py:67-82: inputs AND targets come from the FP model, so blocks are independent."""
        hooks = [full_block.register_forward_hook(self.outp_forward_hook),
                 full_block.register_forward_hook(self.single_input_forward_hook)]
        with torch.no_grad():
            for inp, _ in self.calib_loader:
                self.full_model(inp.to(device))
        block.raw_out = torch.cat(full_block.tmp_out, dim=0)
        block.raw_input = torch.cat(full_block.tmp_input, dim=0)
        full_block.tmp_input, full_block.tmp_out = None, None
        for hook in hooks:
            hook.remove()

    def reconstruct_single_block(self, name, block, device, batch_size: int = 32, iters: int = 20000,
                                 weight: float = 0.01, b_range: tuple = (20, 2), warmup: float = 0.2, lr: float = 4e-5,
                                 p: float = 2.0, quant_act=False):
        self.wrap_quantizers_in_net(block, name)
        self.set_block_mode(block, 'quant_forward')
        for _, module in block.named_modules():
            if hasattr(module, 'training_mode'):
                module.init_training()
        w_params, a_params = [], []
        for _, module in block.named_modules():
            if hasattr(module, 'mode'):
                if isinstance(module, (MinMaxQuantLinear, MinMaxQuantConv2d)):
                    w_params += [module.w_quantizer.alpha]
                    if quant_act:
                        a_params += [module.a_quantizer.scale]
                    else:
                        module.mode = 'debug_only_quant_weight'
                elif isinstance(module, MinMaxQuantMatMul):
                    if quant_act:
                        a_params += [module.A_quantizer.scale, module.B_quantizer.scale]
                    else:
                        module.mode = 'raw'
        ws = parallel.world_size()
        local_bs = max(1, batch_size // ws)
        n_local = block.raw_input.size(0)
        # One iteration is ~85 kernel launches (1.7 ms of GPU time on MI355X for a deit_small block, 58 % of it the fp32
        # GEMMs), issued eagerly in 2.0 ms.  The iteration is captured once in a HIP graph over static batch buffers and
        # replayed: forward, reconstruction loss, rounding regulariser (its exponent b and on/off weight live in device
        # scalars), backward and -- single process -- both Adam steps (capturable).  20 000 iterations of the block:
        # 496 it/s eager, 568 it/s replayed, same reached error.  With several ranks the gradient all-reduce and the
        # optimiser steps stay eager and the graph is opt-in (ADALOG_BRECQ_GRAPH=1); ADALOG_BRECQ_GRAPH=0 forces eager.
        want_graph = os.environ.get("ADALOG_BRECQ_GRAPH", "1" if ws == 1 else "0") == "1"
        use_graph = (torch.device(device).type == 'cuda' and want_graph and n_local >= local_bs and iters > 8)
        full_graph = use_graph and ws == 1
        if use_graph:
            from .. import _lib                          # the ticket counters are allocated (with a device sync) before any capture
            _lib.check(_lib.load().adalog_brecq_init(), "adalog_brecq_init")
        okw = dict(capturable=True) if full_graph else {}
        if torch.device(device).type == 'cuda':
            okw['fused'] = True                          # one kernel per optimiser step instead of six foreach passes
        # the optimiser steps: one hand-written launch per optimiser (HipAdam) on the HIP device; ADALOG_BRECQ_ADAM=torch (and the
        # CPU tier by default) keeps torch.optim.Adam
        hip_adam = os.environ.get("ADALOG_BRECQ_ADAM", "hip" if torch.device(device).type == 'cuda' else "torch") == "hip"
        a_lr = torch.tensor(lr, dtype=torch.float32, device=device) if full_graph else lr
        # Captured iterations with HipAdam: what changes from one iteration to the next besides the mini-batch -- the
        # regulariser's exponent b, its 0 / 1 warm-up gate and the cosine learning rate of the activation scales -- lives in ONE
        # device triple the captured kernels read, refreshed by one 12-byte copy per iteration from a table uploaded 256 rows at
        # a time (two fills and the scheduler's tensor arithmetic, ~7 launches, before).  The rows are computed on the host in
        # python floats by the reference's own objects (LinearTempDecay, torch's CosineAnnealingLR on a float learning rate).
        table_mode = full_graph and hip_adam
        sched_dev = torch.zeros(3, dtype=torch.float32, device=device) if table_mode else None
        if table_mode:
            a_lr = sched_dev[2:3]
            a_lr.fill_(lr)
        if hip_adam:
            w_optimizer = HipAdam(w_params)
            a_optimizer = HipAdam(a_params, lr=a_lr) if len(a_params) != 0 else None
        else:
            w_optimizer = torch.optim.Adam(w_params, **okw)
            a_optimizer = torch.optim.Adam(a_params, lr=a_lr, **okw) if len(a_params) != 0 else None
        a_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(a_optimizer, T_max=iters, eta_min=0.) \
            if len(a_params) != 0 and not table_mode else None
        twin_opt = twin = None
        if table_mode:
            twin_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
            twin_opt.step()
            twin = torch.optim.lr_scheduler.CosineAnnealingLR(twin_opt, T_max=iters, eta_min=0.)
        loss_func = LossFunction(block, round_loss='relaxation', weight=weight, max_count=iters,
                                 rec_loss='mse' if 'head' not in name else 'kl_div', b_range=b_range, decay_start=0,
                                 warmup=warmup, p=p)
        gen = torch.Generator().manual_seed(1234 + parallel.rank())
        # test hooks (tests/golden/brecq_traj.npz replays the reference's own mini-batch sequence and compares the trained
        # values after given iterations): index_source(it, n, bs) -> LongTensor of bs indices; iter_hook(it) after iteration it
        index_source = getattr(self, 'index_source', None)
        iter_hook = getattr(self, 'iter_hook', None)
        # Only alpha and the activation scales are optimised (block_recon.py:97-108).  The reference leaves every other
        # parameter of the block with requires_grad=True and autograd fills their .grad each iteration (bias sums,
        # LayerNorm gamma/beta, ~40 accumulations) although nothing reads them; freezing them for the duration of the
        # reconstruction changes no trained value and drops those kernels.
        trained = {id(prm) for prm in w_params + a_params}
        frozen = [prm for prm in block.parameters() if prm.requires_grad and id(prm) not in trained]
        for prm in frozen:
            prm.requires_grad_(False)
        # ADALOG_BRECQ_FAST_MM=1: the block's plain GEMMs (linear / bmm, forward and both backward products) use hipBLASLt's
        # "fast fp32" mode (bf16 three-term split products, relative error 4e-6 against 4e-7).  Measured on a deit_small
        # block it is worth 4 % of the iteration (some shapes get slower kernels), so the default stays plain fp32.
        fast_mm = torch.device(device).type == 'cuda' and os.environ.get("ADALOG_BRECQ_FAST_MM", "0") == "1"
        prev_tf32 = torch.backends.cuda.matmul.allow_tf32
        if fast_mm:
            torch.backends.cuda.matmul.allow_tf32 = True
        graph, static_inp, static_out, static_rec, static_rnd = None, None, None, None, None
        b_dev = rw_dev = None
        captured_now, collector = False, None
        params = w_params + a_params

        def optim_steps():
            w_optimizer.step()
            if a_optimizer is not None:
                a_optimizer.step()

        def eager_step(cur_inp, cur_out):
            train_mm.reset_offers()
            for prm in params:
                prm.grad = None
            out_quant = block(cur_inp)
            err = loss_func(out_quant, cur_out)
            err.backward()
            if ws > 1:                                   # data-parallel: mean of the per-rank batch-mean gradients
                parallel.all_reduce_mean_bucket([prm.grad for prm in params if prm.grad is not None])
            optim_steps()
            if a_scheduler is not None:
                a_scheduler.step()

        # The mini-batch indices of the next IDX_AHEAD iterations are drawn together (the same randperm sequence) and uploaded
        # in one non-blocking copy from pinned memory: a pageable host-to-device copy per iteration blocks the host until the
        # previous iteration's kernels have drained, i.e. the GPU idles through the host's ~0.2 ms of per-iteration work.
        IDX_AHEAD = 256
        idx_dev = block.raw_input.device
        idx_block, idx_base = None, 0

        def next_indices(it):
            nonlocal idx_block, idx_base
            if idx_block is None or it >= idx_base + idx_block.shape[0]:
                cnt = min(IDX_AHEAD, iters - it)
                rows = [(index_source(it + k, n_local, local_bs) if index_source is not None
                         else torch.randperm(n_local, generator=gen)[:local_bs]) for k in range(cnt)]
                host = torch.stack(rows)
                if idx_dev.type == 'cuda':
                    host = host.pin_memory()
                idx_block, idx_base = host.to(idx_dev, non_blocking=True), it
            return idx_block[it - idx_base]

        sched_block, sched_base = None, 0

        def next_schedule(it, copy=True):
            """Row `it` of the schedule table -> sched_dev (b of iteration it, its gate, the learning rate it steps with); returned (and
            not copied) with ``copy=False``: the replay path hands it to adalog_brecq_prepare together with the mini-batch gather."""
            nonlocal sched_block, sched_base
            if sched_block is None or it >= sched_base + sched_block.shape[0]:
                cnt = min(IDX_AHEAD, iters - it)
                rows = []
                for k in range(cnt):
                    count = it + k + 1                       # LossFunction.advance(): the counter of that iteration
                    active = not (count < loss_func.loss_start or loss_func.round_loss == 'none')
                    rows.append([float(loss_func.temp_decay(count)) if active else 0.0, 1.0 if active else 0.0,
                                 float(twin_opt.param_groups[0]['lr'])])
                    twin.step()                              # (block_recon.py:124-125: the scheduler steps after the optimiser)
                host = torch.tensor(rows, dtype=torch.float32).pin_memory()
                sched_block, sched_base = host.to(device, non_blocking=True), it
            if copy:
                sched_dev.copy_(sched_block[it - sched_base])
            return sched_block[it - sched_base]

        def fixed_rec_loss():
            """the block's reconstruction loss as it stands (soft rounding targets, training-form quantisers: the forward values of an
            iteration) on a FIXED set of its optimisation images -- what the iterations are meant to lower (block_recon.py:189-198)"""
            n_eval = min(n_local, 2 * local_bs)
            train_mm.reset_offers()
            with torch.enable_grad():                        # the iteration's own forward path (contractions on csrc/brecq_gemm.hip)
                pred = block(block.raw_input[:n_eval].to(device)).detach()
            with torch.no_grad():
                return float(loss_func.rec_term(pred, block.raw_out[:n_eval].to(device)))

        report = os.environ.get("ADALOG_BRECQ_REPORT", "1") != "0" and n_local > 0
        rec_before = fixed_rec_loss() if report else None
        try:
            for it in range(iters):
                idx = next_indices(it)
                replaying = use_graph and graph is not None and it >= 3
                sched_row = None
                if table_mode:
                    sched_row = next_schedule(it, copy=not replaying)
                if not use_graph or it < 3:                  # eager (and the warm-up iterations before the capture)
                    eager_step(block.raw_input[idx].to(device), block.raw_out[idx].to(device))
                    if iter_hook is not None:
                        iter_hook(it + 1, loss_func)
                    continue
                if graph is None:
                    static_inp, static_out = block.raw_input[idx].to(device).clone(), block.raw_out[idx].to(device).clone()
                    if table_mode:
                        b_dev, rw_dev = sched_dev[0:1], sched_dev[1:2]
                    else:
                        b_dev = torch.zeros(1, dtype=torch.float32, device=device)
                        rw_dev = torch.zeros(1, dtype=torch.float32, device=device)
                else:
                    be_ = backend.get() if device.type == 'cuda' else None
                    one = (PREPARE_FUSED and be_ is not None and hasattr(be_, "brecq_prepare")
                           and block.raw_input.device == static_inp.device
                           and be_.brecq_prepare(block.raw_input, block.raw_out, idx, static_inp, static_out, sched_row, sched_dev))
                    if not one:                                  # (gather + gather + schedule copy as separate launches)
                        if sched_row is not None:
                            sched_dev.copy_(sched_row)
                        if block.raw_input.device == static_inp.device:
                            torch.index_select(block.raw_input, 0, idx, out=static_inp)
                            torch.index_select(block.raw_out, 0, idx, out=static_out)
                        else:                                    # keep_gpu=False: block data lives on the host
                            static_inp.copy_(block.raw_input[idx])
                            static_out.copy_(block.raw_out[idx])
                active = loss_func.advance()                 # iteration counter, b of this iteration
                if not table_mode:
                    b_dev.fill_(float(loss_func.b))
                    rw_dev.fill_(1.0 if active else 0.0)
                if graph is None:
                    for prm in params:
                        prm.grad = None
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    train_mm.reset_offers()
                    collector = self._alpha_collector(block, loss_func, w_optimizer, w_params) if full_graph else None
                    with torch.cuda.graph(graph):
                        adaround_mod.COLLECT = collector
                        try:
                            static_rec = loss_func.rec_term(block(static_inp), static_out)
                            static_rnd = loss_func.round_sum(b_dev, gate=rw_dev)
                            (static_rec + static_rnd).backward()
                        finally:
                            adaround_mod.COLLECT = None
                        if full_graph:
                            if collector is not None:
                                collector.flush()            # d/d alpha of every layer and its Adam step: one launch
                            optim_steps()                    # (w_optimizer finds no gradient on the collected alphas)
                    captured_now = True
                graph.replay()                               # grads are overwritten, not accumulated (none existed at capture)
                if full_graph:
                    # the replay stepped the optimisers' DEVICE counters; keep HipAdam's host mirror (which decides chunk
                    # membership and seeds new counters) in step: every trained tensor, except that in the capture iteration
                    # step() itself already counted the tensors it saw a gradient on (all but the collected alphas)
                    for opt, prms in ((w_optimizer, w_params), (a_optimizer, a_params)):
                        if isinstance(opt, HipAdam):
                            if not captured_now:
                                opt.note_external_steps(prms)
                            elif collector is not None and opt is w_optimizer:
                                opt.note_external_steps(prms)
                    captured_now = False
                if not full_graph:
                    parallel.all_reduce_mean_bucket([prm.grad for prm in params if prm.grad is not None])
                    optim_steps()
                if a_scheduler is not None:
                    a_scheduler.step()
                loss_func.cur = (static_rec.detach(), static_rnd.detach())
                loss_func.log(static_rec, static_rnd)
                if iter_hook is not None:
                    iter_hook(it + 1, loss_func)
        finally:                                         # (also when an iteration raises: leave the block as it was found)
            graph = None
            torch.backends.cuda.matmul.allow_tf32 = prev_tf32
            for prm in frozen:
                prm.requires_grad_(True)
        if report:
            rec_after = fixed_rec_loss()
            self.__dict__.setdefault('rec_report', {})[name] = (rec_before, rec_after)
            logging.info('{}: reconstruction loss on its first {} images {:.6g} -> {:.6g} after {} iterations'.format(
                name, min(n_local, 2 * local_bs), rec_before, rec_after, iters))
        for _, module in block.named_modules():
            if hasattr(module, 'w_quantizer'):
                module.w_quantizer.soft_targets = False
            if hasattr(module, 'mode'):
                module.mode = 'raw'
            if hasattr(module, 'training_mode'):
                module.end_training()
        del block.raw_input, block.raw_out
        return loss_func

    def _trained_tensors(self, block, quant_act):
        """What a block's reconstruction trains (block_recon.py:97-108), in named_modules() order."""
        out = []
        for _, module in block.named_modules():
            if hasattr(module, 'mode'):
                if isinstance(module, (MinMaxQuantLinear, MinMaxQuantConv2d)):
                    out.append(module.w_quantizer.alpha)
                    if quant_act:
                        out.append(module.a_quantizer.scale)
                elif isinstance(module, MinMaxQuantMatMul) and quant_act:
                    out += [module.A_quantizer.scale, module.B_quantizer.scale]
        return out

    def _reconstruct_blocks_parallel(self, device, quant_act, keep_gpu, iters):
        """Block-parallel BRECQ over the ranks.  A block's inputs AND targets come from the FP model (block_recon.py:62-82), so
        the blocks are independent: they are dealt to the ranks round-robin, every rank trains the blocks it owns exactly as a
        single process would (full mini-batch, full HIP-graph replay, no collective inside the 20 000 iterations), and the
        trained tensors are broadcast from their owners at the end.  The only exchanges are, per block, one all-gather of its
        captured inputs / targets (the optimisation images are sharded over the ranks) before the training and one broadcast of
        alpha / activation scales after it.  The batch-split mode (one gradient all-reduce per iteration: at 32 / 8 = 4 images per
        rank and ~1.8 ms per iteration it is latency-bound and slower on 8 GPUs than on one) stays behind ADALOG_BRECQ_DP=batch."""
        ws, rk = parallel.world_size(), parallel.rank()
        names = list(self.blocks.keys())
        local = {}
        for name in names:                                   # captures first (rank-local shards; collectives, same order on every rank)
            block, full_block = self.blocks[name], self.full_blocks[name]
            self.init_block_raw_inp_outp(block, full_block, name, device)
            local[name] = (block.raw_input, block.raw_out) if keep_gpu else (block.raw_input.cpu(), block.raw_out.cpu())
            del block.raw_input, block.raw_out
        # deal the blocks by estimated cost (an iteration is dominated by the Linear products: tokens x weights), heaviest first
        # to the least loaded rank -- the same pure function of shapes on every rank.  Round-robin dealing left ranks with the
        # cheap Swin stage-3 blocks waiting in the result broadcast while others still trained stage-0 blocks.
        cost = {n: self._block_cost(self.blocks[n], local[n][0]) for n in names}
        owner, load = {}, [0.0] * ws
        for n in sorted(names, key=lambda n_: (-cost[n_], names.index(n_))):
            r = min(range(ws), key=lambda i: (load[i], i))
            owner[n] = r
            load[r] += cost[n]
        self.block_owner = dict(owner)
        rounds = max(sum(1 for n in names if owner[n] == r) for r in range(ws))
        queue = {r: [n for n in names if owner[n] == r] for r in range(ws)}
        for name in names:                                   # every rank: AdaRound quantisers exist everywhere (to receive alpha)
            if owner[name] != rk:
                self.wrap_quantizers_in_net(self.blocks[name], name)
        for rd in range(rounds):
            # one block per rank and round: its data is gathered right before its owner trains it (all blocks' gathered inputs and
            # targets resident at once was 8x the footprint), every rank joins every gather in the same order
            mine = None
            for r in range(ws):
                if rd >= len(queue[r]):
                    continue
                name = queue[r][rd]
                xin, xout = (t.to(device) for t in local.pop(name))
                gin, gout = parallel.gather_images(xin), parallel.gather_images(xout)
                del xin, xout
                if r == rk:
                    mine = (name, gin, gout)
                del gin, gout
            if mine is not None:
                name, gin, gout = mine
                block = self.blocks[name]
                logging.info('rank {}: reconstructing {} ...'.format(rk, name))
                block.raw_input, block.raw_out = gin, gout
                del mine, gin, gout
                with parallel.solo():
                    self.reconstruct_single_block(name, block, device, quant_act=quant_act, iters=iters)
            # the ranks meet here once per round: a wait is at most one block's training (the process group's timeout -- 
            # ADALOG_DIST_TIMEOUT_MIN, default 120 -- is what bounds it, not the library's 10 minutes)
            parallel.barrier()
        for name in names:                                   # results from their owners
            block = self.blocks[name]
            for t in self._trained_tensors(block, quant_act):
                parallel.broadcast(t.data, src=owner[name])
            if owner[name] != rk:                            # the state reconstruct_single_block leaves behind
                for _, module in block.named_modules():
                    if hasattr(module, 'w_quantizer'):
                        module.w_quantizer.soft_targets = False
                    if hasattr(module, 'mode'):
                        module.mode = 'raw'

    @staticmethod
    def _block_cost(block, shard_input):
        """tokens of the block's input (per image) x quantised weights it multiplies them with: what an iteration's contractions
        scale with; the same number on every rank (shapes only)."""
        # from the PER-IMAGE shape (shape[1:]): a shard's image count differs between ranks and an empty shard (shape[0] == 0) must
        # not make its rank price the block differently -- the owner map has to be the same pure function everywhere
        assert shard_input.dim() >= 2, "block input: [images, ..., features]"
        tokens = max(1, int(math.prod(shard_input.shape[1:-1])) if shard_input.dim() > 2 else 1)
        weights = sum(m.weight.numel() for m in block.modules() if isinstance(m, (MinMaxQuantLinear, MinMaxQuantConv2d)))
        if isinstance(block, (MinMaxQuantLinear, MinMaxQuantConv2d)):
            weights = max(weights, block.weight.numel())
        return float(tokens) * float(max(1, weights))

    def reconstruct_model(self, quant_act: bool = False, keep_gpu: bool = True, iters: int = 20000):
        device = next(self.model.parameters()).device
        for _, module in self.model.named_modules():
            if hasattr(module, 'mode'):
                module.mode = 'raw'
        self.dp_mode = 'single'
        if parallel.world_size() > 1 and os.environ.get("ADALOG_BRECQ_DP", "block") != "batch":
            self.dp_mode = 'block'
            self._reconstruct_blocks_parallel(device, quant_act, keep_gpu, iters)
        else:
            if parallel.world_size() > 1:
                self.dp_mode = 'batch'
            for name in self.blocks.keys():
                block, full_block = self.blocks[name], self.full_blocks[name]
                logging.info('reconstructing {} ...'.format(name))
                self.init_block_raw_data(block, full_block, name, device, keep_gpu=keep_gpu)
                self.reconstruct_single_block(name, block, device, quant_act=quant_act, iters=iters)
        for _, module in self.model.named_modules():
            if hasattr(module, 'mode'):
                module.mode = 'quant_forward'
            if hasattr(module, 'w_quantizer') and isinstance(module.w_quantizer, AdaRoundQuantizer):
                module.weight.data.copy_(module.w_quantizer.get_hard_value(module.weight.data))
                del module.w_quantizer.alpha
                module.w_quantizer.round_mode = "nearest"


class _RecLossFn(torch.autograd.Function):
    """scale * sum (pred - tgt)^2 in one kernel each way (lp_loss with p = 2 is six ATen passes forward, six back)."""

    @staticmethod
    def forward(ctx, pred, tgt, scale):
        ctx.save_for_backward(pred, tgt)
        ctx.scale = scale
        return backend.get().rec_loss(pred, tgt, scale).view(())

    @staticmethod
    def backward(ctx, g):
        pred, tgt = ctx.saved_tensors
        return backend.get().rec_loss_backward(pred, tgt, ctx.scale, g.reshape(1)), None, None


class _RoundLossAllFn(torch.autograd.Function):
    """weight * sum over the block's AdaRound quantisers of sum(1 - |2h-1|^b): value and gradients in ONE launch
    (csrc/brecq.hip k_round_loss_multi); the reference builds ~10 autograd nodes per quantiser."""

    @staticmethod
    def forward(ctx, b, weight, gate, *alphas):
        """gate: None or a one-element device tensor (the 0 / 1 switch of the warm-up): value and gradients are multiplied by it
        inside the kernels instead of by a multiply and a sum around this node (four launches per iteration)."""
        ctx.collect = adaround_mod.COLLECT is not None
        kw = {"want_grads": False} if ctx.collect else {}
        if gate is not None:
            kw["gate"] = gate
        loss, grads = backend.get().round_loss_multi(alphas, b, weight, **kw)
        ctx.gate = gate
        if ctx.collect:
            ctx.b, ctx.weight, ctx.n = b, weight, len(alphas)
        else:
            ctx.save_for_backward(*grads)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        if ctx.collect:                                      # the collector's launch computes these gradients itself
            adaround_mod.COLLECT.round = (ctx.b, ctx.weight, g.reshape(1).contiguous(), ctx.gate)
            return (None, None, None) + (None,) * ctx.n
        if ctx.gate is not None:
            g = g * ctx.gate.reshape(())
        return (None, None, None) + tuple(torch._foreach_mul(list(ctx.saved_tensors), g))


class LossFunction:
    """block_recon.py:160-218."""

    def __init__(self, block, round_loss: str = 'relaxation', weight: float = 1., rec_loss: str = 'mse',
                 max_count: int = 2000, b_range: tuple = (10, 2), decay_start: float = 0.0, warmup: float = 0.0,
                 p: float = 2.):
        self.block = block
        self.round_loss = round_loss
        self.weight = weight
        self.rec_loss = rec_loss
        self.loss_start = max_count * warmup
        self.p = p
        self.temp_decay = LinearTempDecay(max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0
        self.b = 0
        self.last = (0.0, 0.0, 0.0)
        self.cur = (0.0, 0.0)

    @staticmethod
    def lp_loss(pred, tgt, p=2.0, reduction='none', scale=1.0):
        """``scale`` multiplies the result (rec_term passes its /10 here so that the fused kernel absorbs it)."""
        if reduction == 'none':
            if p == 2.0 and pred.dim() >= 2 and pred.shape == tgt.shape and pred.is_contiguous() and tgt.is_contiguous() \
                    and pred.dtype == torch.float32 and pred.requires_grad and not tgt.requires_grad:   # a BRECQ iteration
                # .sum(1).mean() = sum over everything / (numel / size(1))
                return _RecLossFn.apply(pred, tgt, scale * float(pred.size(1)) / float(pred.numel()))
            return (pred - tgt).abs().pow(p).sum(1).mean() * scale
        return (pred - tgt).abs().pow(p).mean() * scale

    def rec_term(self, pred, tgt):
        if self.rec_loss == 'mse':
            return self.lp_loss(pred, tgt, p=self.p, scale=0.1)
        if self.rec_loss == 'kl_div':
            return F.kl_div(F.log_softmax(pred, dim=-1), F.softmax(tgt, dim=-1).detach(), reduction="batchmean")
        raise ValueError('Not supported reconstruction loss function: {}'.format(self.rec_loss))

    def advance(self):
        """Advances the iteration counter and sets ``b``; returns whether the rounding regulariser is active."""
        self.count += 1
        self.b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round_loss == 'none':
            self.b = 0
            return False
        if self.round_loss != 'relaxation':
            raise NotImplementedError
        return True

    def round_sum(self, b, gate=None):
        """weight * sum over the block's AdaRound quantisers of sum(1 - |2h-1|^b); b: float or one-element device tensor;
        gate: optional one-element device tensor multiplied in (captured iterations: 0 during the warm-up, then 1)."""
        alphas = [module.w_quantizer.alpha for _, module in self.block.named_modules()
                  if hasattr(module, 'w_quantizer') and isinstance(module.w_quantizer, AdaRoundQuantizer)]
        if 1 <= len(alphas) <= 16 and all(a.dtype == torch.float32 for a in alphas):
            return _RoundLossAllFn.apply(b, float(self.weight), gate, *alphas)
        round_loss = 0
        for _, module in self.block.named_modules():
            if hasattr(module, 'w_quantizer') and isinstance(module.w_quantizer, AdaRoundQuantizer):
                round_loss = round_loss + self.weight * module.w_quantizer.round_loss(b)
        return round_loss if gate is None else (round_loss * gate).sum()

    def round_term(self):
        """Advances the iteration counter; returns the rounding regulariser of this iteration (0 during the warm-up)."""
        return self.round_sum(self.b) if self.advance() else 0

    def log(self, rec_loss, round_loss):
        if self.count == 1 or self.count % 500 == 0:
            rec = float(rec_loss.detach())
            rnd = float(round_loss.detach()) if torch.is_tensor(round_loss) else float(round_loss)
            self.last = (rec + rnd, rec, rnd)
            logging.info('Total loss:\t{:.3f} (rec:{:.3f}, round:{:.3f})\tb={:.2f}\tcount={}'.format(
                self.last[0], self.last[1], self.last[2], self.b, self.count))

    def __call__(self, pred, tgt):
        rec_loss = self.rec_term(pred, tgt)
        round_loss = self.round_term()
        # this iteration's two terms (detached device scalars, read only by tests / logs: a reference to the live loss would
        # keep the iteration's autograd graph alive and break the HIP-graph capture of the next one)
        self.cur = (rec_loss.detach(), round_loss.detach() if torch.is_tensor(round_loss) else round_loss)
        self.log(rec_loss, round_loss)
        return rec_loss + round_loss


class LinearTempDecay:
    """block_recon.py:221-238."""

    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: int = 10, end_b: int = 2):
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b = start_b
        self.end_b = end_b

    def __call__(self, t):
        if t < self.start_decay:
            return self.start_b
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, (1 - rel_t))
