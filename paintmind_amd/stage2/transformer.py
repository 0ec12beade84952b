"""Stage-2 bidirectional conditional transformer holder (reference stage2/transformer.py:28-93)."""
import torch
from torch import nn

from .. import ops, packing
from ..modules.attention import ATTENTION_MODES, _rows
from ..modules.mlp import SwiGLUFFNFused
from ..stage1.layers import _norm, _xavier_like_reference, compute_dtype_of


class Layer(nn.Module):
    ATTENTION_MODES = ATTENTION_MODES

    def __init__(self, dim, dim_head, mlp_dim, num_head=8, dropout=0.0, dim_context=None):
        super().__init__()
        attn_cls = self.ATTENTION_MODES["hip"]
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = attn_cls(query_dim=dim, heads=num_head, dim_head=dim_head, dropout=dropout)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = attn_cls(query_dim=dim, context_dim=dim_context, heads=num_head, dim_head=dim_head, dropout=dropout)
        self.norm3 = nn.LayerNorm(dim)
        self.ffnet = SwiGLUFFNFused(in_features=dim, hidden_features=mlp_dim)

    def forward(self, x, context=None):
        """self-attn, cross-attn (a 2nd self-attn when context is None), SwiGLU (transformer.py:44-49)."""
        B, N, D = x.shape
        T = compute_dtype_of(self)
        x = x.contiguous()
        x = self.attn1.run(_norm(x.reshape(B * N, D), self.norm1, T).reshape(B, N, D), None, residual=x)
        x = self.attn2.run(_norm(x.reshape(B * N, D), self.norm2, T).reshape(B, N, D), context, residual=x)
        return self.ffnet.run(_norm(x.reshape(B * N, D), self.norm3, T).reshape(B, N, D), residual=x)


class CondTransformer(nn.Module):
    def __init__(self, in_dim, dim, len_seq, dim_head, mlp_dim, num_head=8, depth=6, dropout=0.1, context_dim=None,
                 num_classes=8192):
        super().__init__()
        self.token_proj = nn.Linear(in_dim, dim)
        self.position_embedding = nn.Parameter(torch.randn(1, len_seq, dim) * dim ** -0.5)
        self.context_proj = nn.Linear(context_dim, dim, bias=False) if context_dim != dim else nn.Identity()
        self.layers = nn.Sequential()
        for i in range(depth):
            self.layers.add_module("layer" + str(i), Layer(dim, dim_head, mlp_dim, num_head, dropout, dim))
        self.norm = nn.LayerNorm(dim)
        self.to_logits = nn.Linear(dim, num_classes)
        _xavier_like_reference(self)

    def forward(self, x, context=None):
        """operator-level composition of transformer.py:80-93 (Pipeline uses the native engine instead)."""
        T = compute_dtype_of(self)
        B, N, E = x.shape
        dim = self.token_proj.out_features
        if not x.is_cuda:
            # parameters on the CPU (reference stage2/transformer.py:80-93 in plain torch): the context is projected once
            # per forward, a missing context makes attn2 a second self-attention
            h = self.token_proj(x.float()) + self.position_embedding
            if context is not None:
                context = self.context_proj(context.float())
            for layer in self.layers:
                h = layer(h, context)
            return self.to_logits(self.norm(h))
        a = ops.convert_pad(x.contiguous().float().reshape(B * N, E), 64, T)
        pos = self.position_embedding.detach()[0].contiguous()
        h = ops.gemm(a, packing.pad_cols(self.token_proj.weight, 64, T), bias=self.token_proj.bias.detach().float(),
                     residual=pos, res_rows=N, out_dtype=torch.float32).reshape(B, N, dim)
        if context is not None:
            L = context.shape[1]
            c = _rows(context.contiguous().float().reshape(B * L, -1), T)
            if isinstance(self.context_proj, nn.Linear):
                c = ops.gemm(c, packing.pad_cols(self.context_proj.weight, c.shape[1], T))
            context = c.reshape(B, L, dim)
        for layer in self.layers:
            h = layer(h, context)
        y = _norm(h.reshape(B * N, dim), self.norm, T)
        logits = ops.gemm(y, packing.cast(self.to_logits.weight, T), bias=self.to_logits.bias.detach().float(),
                          out_dtype=torch.float32)
        return logits.reshape(B, N, -1)
