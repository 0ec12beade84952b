"""ViT encoder / decoder of the VQGAN as parameter holders + operator-level forwards.

State_dict names, shapes, construction order and initialisation follow the reference
(stage1/layers.py:40-152) so checkpoints and seeded random inits are interchangeable.  The forwards
below compose the HIP operators one call at a time (the "operator plug-in" view of the path);
``VQModel.encode/decode`` use the fused C++ engine instead, and tests pin the two against each other.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, packing
from ..modules.attention import ATTENTION_MODES
from ..modules.mlp import SwiGLUFFNFused


def compute_dtype_of(module):
    return getattr(module, "_pm_dtype", torch.float32)


def _xavier_like_reference(root):
    """reference stage1/layers.py:94-104: xavier-uniform Linear weights, zero biases, unit LayerNorm."""
    def init(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
    root.apply(init)


def _norm(x2d, ln, out_dtype):
    if not x2d.is_cuda:                     # CPU branch: torch's own LayerNorm, fp32
        return F.layer_norm(x2d, (x2d.shape[-1],), ln.weight, ln.bias, ln.eps)
    return ops.layernorm(x2d, ln.weight.detach(), ln.bias.detach(), ln.eps, out_dtype)


class Layer(nn.Module):
    ATTENTION_MODES = ATTENTION_MODES

    def __init__(self, dim, dim_head, mlp_dim, num_head=8, dropout=0.0):
        super().__init__()
        attn_cls = self.ATTENTION_MODES["hip"]
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = attn_cls(query_dim=dim, heads=num_head, dim_head=dim_head, dropout=dropout)
        self.norm2 = nn.LayerNorm(dim)
        self.ffnet = SwiGLUFFNFused(in_features=dim, hidden_features=mlp_dim)

    def forward(self, x):
        """x: fp32 residual stream [B,N,dim]; x = attn1(norm1 x)+x ; x = ffnet(norm2 x)+x (layers.py:54-58)."""
        B, N, D = x.shape
        T = compute_dtype_of(self)
        x = x.contiguous()
        y = _norm(x.reshape(B * N, D), self.norm1, T).reshape(B, N, D)
        x = self.attn1.run(y, None, residual=x)
        y = _norm(x.reshape(B * N, D), self.norm2, T).reshape(B, N, D)
        return self.ffnet.run(y, residual=x)


class Transformer(nn.Module):
    def __init__(self, dim, depth, num_head, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.Sequential(*[Layer(dim, dim_head, mlp_dim, num_head, dropout) for _ in range(depth)])

    def forward(self, x):
        return self.layers(x)


class Encoder(nn.Module):
    def __init__(self, image_size, patch_size, dim, depth, num_head, mlp_dim, in_channels=3, out_channels=3,
                 dim_head=64, dropout=0.):
        super().__init__()
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        self.image_size, self.patch_size = image_size, patch_size
        # index 0 keeps the reference key "to_patch_embedding.0.weight"; index 1 is the layout change the
        # reference does with einops, which the patchify kernel already produces
        self.to_patch_embedding = nn.Sequential(
            nn.Conv2d(in_channels, dim, kernel_size=patch_size, stride=patch_size, bias=False), nn.Identity())
        num_patches = (image_size // patch_size) ** 2
        self.position_embedding = nn.Parameter(torch.randn(1, num_patches, dim) * dim ** -0.5)
        self.norm_pre = nn.LayerNorm(dim)
        self.transformer = Transformer(dim, depth, num_head, dim_head, mlp_dim, dropout)
        _xavier_like_reference(self)

    def forward(self, img):
        T = compute_dtype_of(self)
        conv = self.to_patch_embedding[0]
        B = img.shape[0]
        dim = conv.out_channels
        if not img.is_cuda:
            # CPU branch (reference stage1/layers.py:106-112): conv, 'b c h w -> b (h w) c', + pos, norm_pre, layers
            x = conv(img.float()).flatten(2).transpose(1, 2) + self.position_embedding
            return self.transformer(self.norm_pre(x))
        a = ops.patchify(img.contiguous().float(), self.patch_size, T)
        w = packing.cast(conv.weight.reshape(dim, -1), T)
        pos = self.position_embedding.detach()[0].contiguous()
        x0 = ops.gemm(a, w, residual=pos, res_rows=pos.shape[0], out_dtype=torch.float32)
        x = _norm(x0, self.norm_pre, torch.float32)
        return self.transformer(x.reshape(B, -1, dim))


class Decoder(nn.Module):
    def __init__(self, image_size, patch_size, dim, depth, num_head, mlp_dim, in_channels=3, out_channels=3,
                 dim_head=64, dropout=0.):
        super().__init__()
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        self.image_size, self.patch_size, self.out_channels = image_size, patch_size, out_channels
        num_patches = (image_size // patch_size) ** 2
        self.position_embedding = nn.Parameter(torch.randn(1, num_patches, dim) * dim ** -0.5)
        self.transformer = Transformer(dim, depth, num_head, dim_head, mlp_dim, dropout)
        self.norm = nn.LayerNorm(dim)
        self.proj = nn.Linear(dim, out_channels * patch_size * patch_size, bias=True)
        _xavier_like_reference(self)

    def forward(self, x):
        """x fp32 [B,N,dim] (post_quant output) -> un-clamped image [B,C,H,W] (layers.py:145-152)."""
        T = compute_dtype_of(self)
        B, N, D = x.shape
        if not x.is_cuda:
            # CPU branch (reference stage1/layers.py:145-152): + pos, layers, norm, proj, 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)'
            y = self.proj(self.norm(self.transformer(x.float() + self.position_embedding)))
            g, P, C = self.image_size // self.patch_size, self.patch_size, self.out_channels
            return y.reshape(B, g, g, P, P, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, g * P, g * P)
        pos = self.position_embedding.detach()[0].contiguous()
        x = ops.add_rows(x.contiguous().float().reshape(B * N, D), pos).reshape(B, N, D)
        x = self.transformer(x)
        y = _norm(x.reshape(B * N, D), self.norm, T)
        px = ops.gemm(y, packing.cast(self.proj.weight, T), bias=self.proj.bias.detach().float(), out_dtype=torch.float32)
        return ops.unpatchify_clamp(px, B, self.out_channels, self.image_size, self.patch_size, -float("inf"), float("inf"))
