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

import torch
import torch.nn.functional as F

from .. import parallel
from ..quant_layers import MinMaxQuantConv2d, MinMaxQuantLinear, MinMaxQuantMatMul
from ..quantizers.adaround import AdaRoundQuantizer
from .calibrator import QuantCalibrator
from . import models as M


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
        w_optimizer = torch.optim.Adam(w_params)
        a_optimizer = torch.optim.Adam(a_params, lr=lr) if len(a_params) != 0 else None
        a_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(a_optimizer, T_max=iters, eta_min=0.) \
            if len(a_params) != 0 else None
        loss_func = LossFunction(block, round_loss='relaxation', weight=weight, max_count=iters,
                                 rec_loss='mse' if 'head' not in name else 'kl_div', b_range=b_range, decay_start=0,
                                 warmup=warmup, p=p)
        ws = parallel.world_size()
        local_bs = max(1, batch_size // ws)
        n_local = block.raw_input.size(0)
        gen = torch.Generator().manual_seed(1234 + parallel.rank())
        for it in range(iters):
            idx = torch.randperm(n_local, generator=gen)[:local_bs].to(block.raw_input.device)
            cur_inp = block.raw_input[idx].to(device)
            cur_out = block.raw_out[idx].to(device)
            w_optimizer.zero_grad()
            if a_optimizer is not None:
                a_optimizer.zero_grad()
            out_quant = block(cur_inp)
            err = loss_func(out_quant, cur_out)
            err.backward()
            if ws > 1:                                   # data-parallel: mean of the per-rank batch-mean gradients
                for prm in w_params + a_params:
                    if prm.grad is not None:
                        parallel.all_reduce_sum(prm.grad)
                        prm.grad.div_(ws)
            w_optimizer.step()
            if a_optimizer is not None:
                a_optimizer.step()
                a_scheduler.step()
        for _, module in block.named_modules():
            if hasattr(module, 'w_quantizer'):
                module.w_quantizer.soft_targets = False
            if hasattr(module, 'mode'):
                module.mode = 'raw'
            if hasattr(module, 'training_mode'):
                module.end_training()
        del block.raw_input, block.raw_out
        return loss_func

    def reconstruct_model(self, quant_act: bool = False, keep_gpu: bool = True, iters: int = 20000):
        device = next(self.model.parameters()).device
        for _, module in self.model.named_modules():
            if hasattr(module, 'mode'):
                module.mode = 'raw'
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
        self.last = (0.0, 0.0, 0.0)

    @staticmethod
    def lp_loss(pred, tgt, p=2.0, reduction='none'):
        if reduction == 'none':
            return (pred - tgt).abs().pow(p).sum(1).mean()
        return (pred - tgt).abs().pow(p).mean()

    def __call__(self, pred, tgt):
        self.count += 1
        if self.rec_loss == 'mse':
            rec_loss = self.lp_loss(pred, tgt, p=self.p) / 10
        elif self.rec_loss == 'kl_div':
            rec_loss = F.kl_div(F.log_softmax(pred, dim=-1), F.softmax(tgt, dim=-1).detach(), reduction="batchmean")
        else:
            raise ValueError('Not supported reconstruction loss function: {}'.format(self.rec_loss))
        b = self.temp_decay(self.count)
        if self.count < self.loss_start or self.round_loss == 'none':
            b = round_loss = 0
        elif self.round_loss == 'relaxation':
            round_loss = 0
            for _, module in self.block.named_modules():
                if hasattr(module, 'w_quantizer') and isinstance(module.w_quantizer, AdaRoundQuantizer):
                    round_loss = round_loss + self.weight * module.w_quantizer.round_loss(b)
        else:
            raise NotImplementedError
        total_loss = rec_loss + round_loss
        if self.count == 1 or self.count % 500 == 0:
            self.last = (float(total_loss.detach()), float(rec_loss.detach()),
                         float(round_loss.detach()) if torch.is_tensor(round_loss) else float(round_loss))
            logging.info('Total loss:\t{:.3f} (rec:{:.3f}, round:{:.3f})\tb={:.2f}\tcount={}'.format(
                self.last[0], self.last[1], self.last[2], b, self.count))
        return total_loss


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
