"""Vision Transformer model definitions with timm-0.9.2-compatible module names.

The reference builds its models with ``timm.create_model`` (test_quant.py:181-184); timm is not available on the build
or the GPU box, so the architectures are defined here with exactly the attribute / state_dict names the reference's
wrapper and calibrator rely on (SURVEY 8b "naming rules"):
  ViT / DeiT : patch_embed.proj, cls_token, pos_embed, blocks.N.{norm1, attn.{qkv, proj}, norm2, mlp.{fc1, fc2}}, norm, head
  Swin       : patch_embed.{proj, norm}, layers.S.{downsample.{norm, reduction}, blocks.N.{norm1, attn.{qkv, proj,
               relative_position_bias_table}, norm2, mlp.{fc1, fc2}}}, norm, head.fc
so a timm checkpoint (``*.bin`` state_dict) loads with ``load_state_dict`` and quantised checkpoints keep the
reference's key layout.  Attention exposes the two matrix products as ``matmul1`` / ``matmul2`` sub-modules right
after ``proj`` -- the position the reference's ``setattr`` gives them (wrap_net.py:58-59), which fixes the calibration
order qkv, proj, matmul1, matmul2, fc1, fc2.

Weights: seeded trunc_normal(std=0.02) init (there is no network for pretrained weights); LayerNorm gamma=1, beta=0.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import backend as _backend
from .. import search as _search
from .. import train_mm

# quant_forward of a ViT / DeiT block on the fused route (Attention._fused_quant_forward, Mlp.forward): 0 = module by module
QF_FUSED = os.environ.get("ADALOG_QF_FUSED", "1") != "0"


def _plain_quant_forward(m):
    """m is one of the package's quantised modules, calibrated, in quant_forward mode, and nobody listens on its forward (the
    calibrator's capture hooks need the module-by-module route)."""
    return (getattr(m, "mode", None) == "quant_forward" and getattr(m, "calibrated", False)
            and not m._forward_hooks and not m._forward_pre_hooks)


class MatMul(nn.Module):
    """A @ B as a module, the hook point that quantised matmuls replace (wrap_net.py:14-16)."""

    def forward(self, A, B):
        return A @ B


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.drop1 = nn.Identity()
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features or in_features)
        self.drop2 = nn.Identity()

    def forward(self, x, residual=None):
        """``residual`` (utils/models.py: Block): returned added to the result -- inside fc2's launch on the fused route."""
        fc2 = self.fc2
        if (QF_FUSED and not torch.is_grad_enabled() and x.is_cuda and _plain_quant_forward(fc2) and hasattr(fc2, "fused_ok")
                and fc2.fused_ok() and isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none"
                and all(isinstance(m, nn.Identity) for m in (self.drop1, self.norm, self.drop2))):
            # quant_forward: GELU runs in the loader of fc2's operand packer, the residual is added in fc2's epilogue
            return fc2.quant_forward(self.fc1(x), addend=residual, pre_gelu=True)
        if (QF_FUSED and train_mm.ENABLED and torch.is_grad_enabled() and x.is_cuda and _plain_quant_forward(fc2)
                and getattr(getattr(fc2, "a_quantizer", None), "training_mode", False) and hasattr(fc2.a_quantizer, "fused_gelu_ok")
                and isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none"
                and all(isinstance(m, nn.Identity) for m in (self.drop1, self.norm, self.drop2))):
            # a BRECQ iteration: GELU and its derivative run inside the kernels of fc2's input quantiser (no GELU pass either way)
            return fc2.quant_forward(self.fc1(x), addend=residual, pre_gelu=True)
        out = self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))
        return out if residual is None else residual + out


class Attention(nn.Module):
    """timm.models.vision_transformer.Attention with the forward of wrap_net.py:19-32."""

    def __init__(self, dim, num_heads=8, qkv_bias=True):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()
        self.matmul1 = MatMul()
        self.matmul2 = MatMul()

    def _fused_quant_forward_ok(self, x):
        """quant_forward of the whole block on the fused route (below): every quantised module of the block is in plain
        quant_forward mode with the input quantisers the packers implement, head dimension 64, <= 256 tokens."""
        from ..quant_layers.matmul import AsymmetricallyBatchingQuantMatMul, PostSoftmaxAsymmetricallyBatchingQuantMatMul
        from ..quantizers.uniform import UniformQuantizer
        m1, m2 = self.matmul1, self.matmul2
        if not (QF_FUSED and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and self.head_dim == 64
                and x.shape[1] <= 256 and isinstance(self.q_norm, nn.Identity) and isinstance(self.k_norm, nn.Identity)
                and isinstance(self.attn_drop, nn.Identity) and isinstance(self.proj_drop, nn.Identity)):
            return False
        if not (type(m1) is AsymmetricallyBatchingQuantMatMul and type(m2) is PostSoftmaxAsymmetricallyBatchingQuantMatMul
                and all(_plain_quant_forward(m) for m in (self.qkv, self.proj, m1, m2))):
            return False
        qs = (m1.A_quantizer, m1.B_quantizer, m2.B_quantizer)
        return (all(isinstance(q, UniformQuantizer) and 2 <= q.n_bits <= 7 and not q.training_mode for q in qs)
                and not m2.A_quantizer.training_mode and m1._heads() == m2._heads() and m2.A_quantizer.scale.numel() == 1
                and getattr(_backend.get(), "QF_EXTRAS", False))

    def _fused_quant_forward(self, x, residual):
        """The attention block in quant_forward mode (reference utils/wrap_net.py:19-32 with every product in quant_forward,
        quant_layers/matmul.py:43-45) as five launches after the qkv projection: split + three input quantisers + operand packs;
        q . k^T (int8 MFMA); scale + softmax + AdaLog quantiser + pack; softmax . v (bf16 MFMA) written as [B, N, H, D]; the
        projection with the residual stream added in its epilogue."""
        from ..ops import BF16, I8, Strided
        be = _backend.get()
        B, N, C = x.shape
        H = self.num_heads
        m1, m2 = self.matmul1, self.matmul2
        qkv = self.qkv(x)
        hm = m1._heads()
        pg = 1 if hm > 1 else 0
        sA, zA = m1._q_params(m1.A_quantizer)
        sB, zB = m1._q_params(m1.B_quantizer)
        sV, zV = m2._q_params(m2.B_quantizer)
        qp, kp, vp = be.attn_split_pack(qkv, H, (sA, zA, m1.A_quantizer.n_bits), (sB, zB, m1.B_quantizer.n_bits),
                                        (sV, zV, m2.B_quantizer.n_bits), hm > 1)
        scores = be.gemm_out(I8, qp, kp, N, N, B * H, hm, Strided(sA, g=pg), Strided(sB, g=pg), None)
        if m2._q_host is None:
            m2._q_host = int(m2.A_quantizer.q.item())
        qv = _search.const_tensor([float(m2._q_host)], x.device)
        a_scale = m2.A_quantizer.scale.data.view(-1)
        ap = be.softmax_adalog_pack(scores, self.scale, a_scale, qv, m2.A_quantizer.n_bits, m2._mant37(x.device))
        out = be.gemm_out(BF16, ap, vp, N, self.head_dim, B * H, hm, Strided(a_scale), Strided(sV, g=pg), None, sa_mul=m2._ts32(),
                          heads_last=H)
        return self.proj.quant_forward(out.view(B, N, C), addend=residual)

    def forward(self, x, residual=None):
        """``residual`` (Block): returned added to the result -- inside the projection's launch on the fused route."""
        if self._fused_quant_forward_ok(x):
            return self._fused_quant_forward(x, residual)
        out = self._forward(x)
        return out if residual is None else residual + out

    def _forward(self, x):
        B, N, C = x.shape
        x = self.qkv(x)
        fused = None
        if isinstance(self.q_norm, nn.Identity) and isinstance(self.k_norm, nn.Identity):
            fused = train_mm.qkv_split_quant(x, self.num_heads, self.matmul1, self.matmul2)   # a BRECQ iteration on the GPU
        if fused is not None:
            q, k, v = fused                                    # already through the products' input quantisers
            attn = train_mm.scaled_softmax(self.matmul1(q, k.transpose(-2, -1), a_pre=True, b_pre=True), self.scale)
            x = self.matmul2(self.attn_drop(attn), v, b_pre=True)
        else:
            q, k, v = train_mm.split_heads(x, 3, self.num_heads)
            q, k = self.q_norm(q), self.k_norm(k)
            attn = train_mm.scaled_softmax(self.matmul1(q, k.transpose(-2, -1)), self.scale)
            attn = self.attn_drop(attn)
            x = self.matmul2(attn, v)
        x = x.transpose(1, 2).reshape(B, N, C)
        x = self.proj(x)
        return self.proj_drop(x)


class Block(nn.Module):
    """timm.models.vision_transformer.Block (pre-norm, no layer-scale, no drop-path at eval)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        if all(isinstance(m, nn.Identity) for m in (self.ls1, self.ls2, self.drop_path1, self.drop_path2)):
            x = self.attn(self.norm1(x), residual=x)       # x + attn(...): the sum is taken by the branch (in its last launch when fused)
            return self.mlp(self.norm2(x), residual=x)
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        x = x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))
        return x


class PatchEmbed(nn.Module):
    """timm.layers.patch_embed.PatchEmbed: non-overlapping conv, then (optionally) flatten to tokens."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 output_nhwc=False):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.output_nhwc = output_nhwc
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)          # NCHW -> NLC
        elif self.output_nhwc:
            x = x.permute(0, 2, 3, 1)                 # NCHW -> NHWC
        return self.norm(x)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.pos_drop = nn.Identity()
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes)
        self._init()

    def _init(self):
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed
        x = self.norm_pre(self.patch_drop(self.pos_drop(x)))
        x = self.blocks(x)
        return self.norm(x)

    def forward_head(self, x):
        x = x[:, 0]                                    # class token
        return self.head(self.head_drop(self.fc_norm(x)))

    def forward(self, x):
        return self.forward_head(self.forward_features(x))


# ------------------------------------------------------------------------------------------------ Swin (timm 0.9.2, NHWC blocks)
def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size[0], window_size[0], W // window_size[1], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size[0], window_size[1], C)


def window_reverse(windows, window_size, H, W):
    C = windows.shape[-1]
    x = windows.view(-1, H // window_size[0], W // window_size[1], window_size[0], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)


def get_relative_position_index(win_h, win_w):
    coords = torch.stack(torch.meshgrid([torch.arange(win_h), torch.arange(win_w)], indexing="ij"))
    coords_flatten = torch.flatten(coords, 1)
    rel = coords_flatten[:, :, None] - coords_flatten[:, None, :]
    rel = rel.permute(1, 2, 0).contiguous()
    rel[:, :, 0] += win_h - 1
    rel[:, :, 1] += win_w - 1
    rel[:, :, 0] *= 2 * win_w - 1
    return rel.sum(-1)


class WindowAttention(nn.Module):
    """timm.models.swin_transformer.WindowAttention with the forward of wrap_net.py:35-52."""

    def __init__(self, dim, num_heads, window_size=(7, 7), qkv_bias=True):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        win_h, win_w = window_size
        self.window_area = win_h * win_w
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * win_h - 1) * (2 * win_w - 1), num_heads))
        self.register_buffer("relative_position_index", get_relative_position_index(win_h, win_w), persistent=False)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Identity()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Identity()
        self.softmax = nn.Softmax(dim=-1)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.matmul1 = MatMul()
        self.matmul2 = MatMul()

    def _get_rel_pos_bias(self):
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(
            self.window_area, self.window_area, -1)
        return bias.permute(2, 0, 1).contiguous().unsqueeze(0)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        x = self.qkv(x)
        q, k, v = train_mm.split_heads(x, 3, self.num_heads)
        q = q * self.scale
        attn = self.matmul1(q, k.transpose(-2, -1))
        attn = attn + self._get_rel_pos_bias()
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(-1, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = attn.softmax(dim=-1)
        attn = self.attn_drop(attn)
        x = self.matmul2(attn, v).transpose(1, 2).reshape(B_, N, C)
        x = self.proj(x)
        return self.proj_drop(x)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.dim = dim
        self.input_resolution = input_resolution
        ws = (window_size, window_size)
        ss = (shift_size, shift_size)
        if min(input_resolution) <= window_size:           # window covers the map: no shift (timm _calc_window_shift)
            ws = tuple(input_resolution)
            ss = (0, 0)
        self.window_size, self.shift_size = ws, ss
        self.window_area = ws[0] * ws[1]
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, num_heads=num_heads, window_size=ws, qkv_bias=qkv_bias)
        self.drop_path1 = nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.drop_path2 = nn.Identity()
        if any(self.shift_size):
            H, W = input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
                for w in (slice(0, -ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mask_windows = window_partition(img_mask, ws).view(-1, self.window_area)
            attn_mask = mask_windows.unsqueeze(1) - mask_windows.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
        else:
            attn_mask = None
        self.register_buffer("attn_mask", attn_mask, persistent=False)

    def _attn(self, x):
        B, H, W, C = x.shape
        has_shift = any(self.shift_size)
        shifted = torch.roll(x, shifts=(-self.shift_size[0], -self.shift_size[1]), dims=(1, 2)) if has_shift else x
        xw = window_partition(shifted, self.window_size).view(-1, self.window_area, C)
        aw = self.attn(xw, mask=self.attn_mask)
        aw = aw.view(-1, self.window_size[0], self.window_size[1], C)
        shifted = window_reverse(aw, self.window_size, H, W)
        return torch.roll(shifted, shifts=self.shift_size, dims=(1, 2)) if has_shift else shifted

    def forward(self, x):
        B, H, W, C = x.shape
        x = x + self.drop_path1(self._attn(self.norm1(x)))
        x = x.reshape(B, -1, C)
        x = x + self.drop_path2(self.mlp(self.norm2(x)))
        return x.reshape(B, H, W, C)


class PatchMerging(nn.Module):
    def __init__(self, dim, out_dim=None):
        super().__init__()
        self.dim = dim
        self.out_dim = out_dim or 2 * dim
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, self.out_dim, bias=False)

    def forward(self, x):
        B, H, W, C = x.shape
        x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
        return self.reduction(self.norm(x))


class SwinTransformerStage(nn.Module):
    def __init__(self, dim, out_dim, input_resolution, depth, downsample, num_heads, window_size, mlp_ratio, qkv_bias):
        super().__init__()
        self.output_resolution = tuple(i // 2 for i in input_resolution) if downsample else input_resolution
        self.downsample = PatchMerging(dim, out_dim) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[
            SwinTransformerBlock(out_dim, self.output_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias) for i in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ClassifierHead(nn.Module):
    """timm ClassifierHead for NHWC feature maps: global average pool -> fc."""

    def __init__(self, in_features, num_classes):
        super().__init__()
        self.global_pool = nn.Identity()
        self.drop = nn.Identity()
        self.fc = nn.Linear(in_features, num_classes)
        self.flatten = nn.Identity()

    def forward(self, x):
        return self.fc(self.drop(x.mean(dim=(1, 2))))


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.num_classes = num_classes
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, norm_layer=nn.LayerNorm, flatten=False,
                                      output_nhwc=True)
        grid = self.patch_embed.grid_size
        dims = [int(embed_dim * 2 ** i) for i in range(len(depths))]
        layers, in_dim, scale = [], dims[0], 1
        for i in range(len(depths)):
            if i > 0:
                scale *= 2
            layers.append(SwinTransformerStage(in_dim, dims[i], (grid[0] // scale * (2 if i > 0 else 1),
                                                                  grid[1] // scale * (2 if i > 0 else 1)),
                                               depths[i], i > 0, num_heads[i], window_size, mlp_ratio, qkv_bias))
            in_dim = dims[i]
        self.layers = nn.Sequential(*layers)
        self.norm = nn.LayerNorm(dims[-1])
        self.head = ClassifierHead(dims[-1], num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = self.layers(x)
        return self.head(self.norm(x))


# ------------------------------------------------------------------------------------------------ zoo (test_quant.py:162-176)
MODEL_ZOO = {
    'vit_tiny': 'vit_tiny_patch16_224', 'vit_small': 'vit_small_patch16_224', 'vit_base': 'vit_base_patch16_224',
    'vit_large': 'vit_large_patch16_224',
    'deit_tiny': 'deit_tiny_patch16_224', 'deit_small': 'deit_small_patch16_224', 'deit_base': 'deit_base_patch16_224',
    'swin_tiny': 'swin_tiny_patch4_window7_224', 'swin_small': 'swin_small_patch4_window7_224',
    'swin_base': 'swin_base_patch4_window7_224', 'swin_base_384': 'swin_base_patch4_window12_384',
}

_VIT = {'tiny': (192, 12, 3), 'small': (384, 12, 6), 'base': (768, 12, 12), 'large': (1024, 24, 16)}
_SWIN = {'swin_tiny': (96, (2, 2, 6, 2), (3, 6, 12, 24), 7, 224), 'swin_small': (96, (2, 2, 18, 2), (3, 6, 12, 24), 7, 224),
         'swin_base': (128, (2, 2, 18, 2), (4, 8, 16, 32), 7, 224), 'swin_base_384': (128, (2, 2, 18, 2), (4, 8, 16, 32), 12, 384)}


def create_model(name: str, num_classes: int = 1000, depth: int = None, img_size: int = None):
    """Counterpart of timm.create_model for the reference's model zoo.  ``depth`` truncates the block count (tests)."""
    if name in MODEL_ZOO.values():
        name = {v: k for k, v in MODEL_ZOO.items()}[name]
    if name.startswith(('vit_', 'deit_')):
        dim, d, heads = _VIT[name.split('_')[1]]
        return VisionTransformer(img_size=img_size or 224, embed_dim=dim, depth=depth or d, num_heads=heads,
                                 num_classes=num_classes)
    if name in _SWIN:
        e, depths, heads, ws, size = _SWIN[name]
        if depth is not None:
            depths = tuple(min(x, depth) for x in depths)
        return SwinTransformer(img_size=img_size or size, embed_dim=e, depths=depths, num_heads=heads, window_size=ws,
                               num_classes=num_classes)
    raise ValueError(f"unknown model {name}")
