"""Attention operator for the reference's plug-point.

The reference selects its attention class through ``Layer.ATTENTION_MODES`` (reference
stage1/layers.py:41-48, stage2/transformer.py:29-36) with two interchangeable entries,
``CrossAttention`` and ``MemoryEfficientCrossAttention`` (modules/attention.py:25-108).  This
class is the third implementation of the same contract: same constructor signature, same
``forward(x, context=None)``, same state_dict keys (to_q/to_k/to_v/to_out.0), with the arithmetic
done by libpaintmind_hip.so: fused q|k|v projection with head-split epilogue, flash-style
softmax(QK^T)V that never materialises the score matrix, and an out-projection with fused bias
(+ residual when called from a Layer).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, packing

XFORMERS_IS_AVAILBLE = False   # name kept for source compatibility (reference attention.py:7-12)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        if dim_head % 16 or not 16 <= dim_head <= 128 or (dim_head * heads) % 64:
            raise ValueError(f"dim_head={dim_head}, heads={heads}: the HIP operators serve dim_head = 16, 32, ..., 128 with "
                             f"heads*dim_head a multiple of 64 (64 runs the tuned MFMA kernels, the rest a plain path)")
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.query_dim = query_dim
        self.context_dim = context_dim
        # creation order == the reference's, so a seeded construction draws identical weights
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self._pack = None

    # -- packed weights, rebuilt whenever a parameter changes ---------------------------------------
    def packed(self, dtype):
        stamp = (packing.params_fingerprint(self), dtype)
        if self._pack is None or self._pack[0] != stamp:
            qd, cd = round_up64(self.query_dim), round_up64(self.context_dim)
            pk = {
                "wq": packing.pad_cols(self.to_q.weight, qd, dtype),
                "wkv": torch.cat([packing.pad_cols(self.to_k.weight, cd, dtype),
                                  packing.pad_cols(self.to_v.weight, cd, dtype)], dim=0),
                "wo": packing.cast(self.to_out[0].weight, dtype),
                "bo": packing.cast(self.to_out[0].bias, torch.float32),
            }
            if qd == cd:
                pk["wqkv"] = torch.cat([pk["wq"], pk["wkv"]], dim=0)
            self._pack = (stamp, pk)
        return self._pack[1]

    def forward(self, x, context=None):
        return self.run(x, context, residual=None)

    def run(self, x, context=None, residual=None):
        """x [B,N,query_dim]; returns to_out(attn) (+ residual) as fp32/bf16 like x."""
        if self.training and self.to_out[1].p > 0:
            raise RuntimeError("paintmind_amd attention is inference-only (dropout>0 in training mode)")
        if not x.is_cuda:
            return self._run_cpu(x, context, residual)
        B, N, D = x.shape
        dtype = x.dtype
        pk = self.packed(dtype)
        fast = dtype == torch.bfloat16
        q_scale = self.scale * (ops.LOG2E if fast else 1.0)
        a = _rows(x.reshape(B * N, D), dtype)
        if context is None:
            if "wqkv" not in pk:
                raise ValueError("self-attention needs context_dim == query_dim")
            q, k, vt = ops.gemm_heads(a, pk["wqkv"], self.heads, N, [ops.PART_Q, ops.PART_K, ops.PART_V], q_scale, self.dim_head)
            n_kv = N
        else:
            L = context.shape[1]
            c = _rows(context.reshape(B * L, context.shape[2]), dtype)
            (q,) = ops.gemm_heads(a, pk["wq"], self.heads, N, [ops.PART_Q], q_scale, self.dim_head)
            k, vt = ops.gemm_heads(c, pk["wkv"], self.heads, L, [ops.PART_K, ops.PART_V], 1.0, self.dim_head)
            n_kv = L
        o = ops.attention(q, k, vt, n_kv, use_exp2=fast)
        res = residual.reshape(B * N, D) if residual is not None else None
        out_dtype = torch.float32 if residual is not None else dtype
        out = ops.gemm(o, pk["wo"], bias=pk["bo"], residual=res, out_dtype=out_dtype)
        return out.reshape(B, N, D)


    def _run_cpu(self, x, context, residual):
        """Parameters on the CPU: the same operator in plain torch (reference modules/attention.py:43-59; BASELINE config 1
        runs the model there).  q is scaled BEFORE the product (:52), no mask, softmax over the keys, head split 'b n (h d)'."""
        B, N, _ = x.shape
        c = x if context is None else context
        h, d = self.heads, self.dim_head
        q = (F.linear(x, self.to_q.weight) * self.scale).reshape(B, N, h, d).permute(0, 2, 1, 3)
        k = F.linear(c, self.to_k.weight).reshape(B, c.shape[1], h, d).permute(0, 2, 1, 3)
        v = F.linear(c, self.to_v.weight).reshape(B, c.shape[1], h, d).permute(0, 2, 1, 3)
        p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, h * d)
        out = F.linear(o, self.to_out[0].weight, self.to_out[0].bias)
        return out if residual is None else out + residual


def round_up64(v):
    return ops.round_up(v, 64)


def _rows(x2d, dtype):
    """contiguous [M,K] in `dtype` with K padded to a multiple of 64 (device-side pad-convert kernel)."""
    k = x2d.shape[1]
    if x2d.dtype == dtype and k % 64 == 0:
        return x2d.contiguous()
    if x2d.dtype != torch.float32:
        raise TypeError("only float32 inputs can be pad-converted")
    return ops.convert_pad(x2d.contiguous(), round_up64(k), dtype)


# both reference names resolve to the HIP operator
MemoryEfficientCrossAttention = CrossAttention
ATTENTION_MODES = {"vanilla": CrossAttention, "xformer": CrossAttention, "hip": CrossAttention}
