"""Model surgery: swap nn.Conv2d / nn.Linear / MatMul for their quantised counterparts by the reference's naming rules
(reference utils/wrap_net.py:55-172) and, after calibration, swap the channel-wise layers for plain ones (:175-210).

Rules: 'qkv' -> n_V = 3; 'qkv' / 'fc1' / 'reduction' -> channel-wise search + LayerNorm fold when the
activation bit-width equals the weight bit-width and ``reparam`` is set, with prev_layer = norm1 / norm2 / the merging
block's norm; 'fc2' -> post-GELU AdaLog; 'head' -> cfg.qhead_a_bit; 'matmul2' -> post-softmax AdaLog.
The attention modules of adalog_amd.utils.models already expose matmul1/matmul2, so no forward patching is needed.
"""
from torch import nn

from ..quant_layers.conv import AsymmetricallyBatchingQuantConv2d
from ..quant_layers.linear import (AsymmetricallyBatchingQuantLinear, AsymmetricallyChannelWiseBatchingQuantLinear,
                                   PostGeluLogBasedBatchingQuantLinear)
from ..quant_layers.matmul import AsymmetricallyBatchingQuantMatMul, PostSoftmaxAsymmetricallyBatchingQuantMatMul
from .models import MatMul


def _parent_and_leaf(module_dict, name):
    idx = name.rfind('.')
    father_name = name[:idx] if idx != -1 else ''
    if father_name not in module_dict:
        raise RuntimeError(f"father module {father_name} not found")
    return father_name, module_dict[father_name], name[idx + 1:]


def wrap_modules_in_net(model, cfg, reparam=False):
    module_dict = {}
    for name, module in list(model.named_modules()):
        module_dict[name] = module
        if name == '':
            continue
        father_name, father, leaf = _parent_and_leaf(module_dict, name)
        dev = next(module.parameters(), None)
        dev = dev.device if dev is not None else next(model.parameters()).device
        new_module = None
        if isinstance(module, nn.Conv2d):
            new_module = AsymmetricallyBatchingQuantConv2d(
                in_channels=module.in_channels, out_channels=module.out_channels, kernel_size=module.kernel_size,
                stride=module.stride, mode='raw', w_bit=cfg.w_bit, a_bit=cfg.qconv_a_bit,
                calib_batch_size=cfg.calib_batch_size, search_round=cfg.search_round, eq_n=cfg.eq_n, fpcs=cfg.fpcs,
                steps=cfg.steps)
            new_module.weight.data.copy_(module.weight.data)
            new_module.bias.data.copy_(module.bias.data)
        elif isinstance(module, MatMul):
            kw = dict(B_bit=cfg.a_bit, mode='raw', calib_batch_size=cfg.calib_batch_size, search_round=cfg.search_round,
                      eq_n=cfg.eq_n, head_channel_wise=cfg.matmul_head_channel_wise, num_heads=father.num_heads,
                      fpcs=cfg.fpcs, steps=cfg.steps)
            if 'matmul2' in name:
                new_module = PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=cfg.s_bit, **kw,
                                                                         quantizer=cfg.post_softmax_quantizer)
                new_module.out_heads_last = True       # layout hint for BRECQ iterations (train_mm.matmul): the result is merged
                                                       # over heads right after (models.py: transpose(1, 2).reshape)
            else:
                new_module = AsymmetricallyBatchingQuantMatMul(A_bit=cfg.a_bit, **kw)
        elif isinstance(module, nn.Linear):
            cur_a_bit = cfg.qhead_a_bit if 'head' in name else cfg.a_bit
            kw = dict(in_features=module.in_features, out_features=module.out_features, bias=module.bias is not None,
                      mode='raw', w_bit=cfg.w_bit, a_bit=cur_a_bit, calib_batch_size=cfg.calib_batch_size,
                      search_round=cfg.search_round, eq_n=cfg.eq_n, n_V=3 if 'qkv' in name else 1, fpcs=cfg.fpcs,
                      steps=cfg.steps)
            if cur_a_bit == cfg.w_bit and reparam and ('qkv' in name or 'reduction' in name or 'fc1' in name):
                new_module = AsymmetricallyChannelWiseBatchingQuantLinear(**kw)
                idxx = father_name.rfind('.')
                grandfather = module_dict.get(father_name[:idxx] if idxx != -1 else '')
                if 'qkv' in name:
                    new_module.prev_layer = grandfather.norm1
                if 'fc1' in name:
                    new_module.prev_layer = grandfather.norm2
                if 'reduction' in name:
                    new_module.prev_layer = father.norm
            elif 'fc2' in name and cfg.post_gelu_quantizer in ('adalog', 'log2', 'logsqrt2', 'ptq4vit'):
                new_module = PostGeluLogBasedBatchingQuantLinear(**kw, quantizer=cfg.post_gelu_quantizer)
            else:
                new_module = AsymmetricallyBatchingQuantLinear(**kw)
            new_module.weight.data.copy_(module.weight.data)
            if module.bias is not None:
                new_module.bias.data.copy_(module.bias.data)
        if new_module is not None:
            new_module.to(dev)
            setattr(father, leaf, new_module)
            module_dict[name] = new_module
    return model


def wrap_reparamed_modules_in_net(model):
    """After calibration the channel-wise layers carry per-tensor parameters: re-instantiate them as plain layers so the
    checkpoint has the reference's final key/shape layout (wrap_net.py:175-210)."""
    module_dict = {}
    for name, module in list(model.named_modules()):
        module_dict[name] = module
        if name == '' or not isinstance(module, AsymmetricallyChannelWiseBatchingQuantLinear):
            continue
        _, father, leaf = _parent_and_leaf(module_dict, name)
        new_module = AsymmetricallyBatchingQuantLinear(
            in_features=module.in_features, out_features=module.out_features, bias=module.bias is not None,
            mode=module.mode, w_bit=module.w_quantizer.n_bits, a_bit=module.a_quantizer.n_bits,
            calib_batch_size=module.calib_batch_size, search_round=module.search_round, eq_n=module.eq_n, n_V=module.n_V,
            fpcs=module.fpcs, steps=module.steps)
        new_module.load_state_dict(module.state_dict())
        new_module.calibrated = True
        new_module.a_quantizer.inited = True
        new_module.w_quantizer.inited = True
        new_module.to(module.weight.device)
        setattr(father, leaf, new_module)
    return model
