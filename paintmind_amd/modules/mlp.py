"""SwiGLU feed-forward for the reference's plug-point (reference modules/mlp.py:13-59).

Same constructor signature, same parameter names (w12, w3) and the same hidden-width rule; the
arithmetic is one GEMM whose epilogue applies silu(x1)*x2 (so the 2*hidden-wide intermediate never
reaches HBM) followed by the w3 GEMM with fused bias (+ residual when called from a Layer).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, packing
from .attention import _rows


class SwiGLUFFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)
        self._pack = None

    def packed(self, dtype):
        stamp = (packing.params_fingerprint(self), dtype)
        if self._pack is None or self._pack[0] != stamp:
            kin = ops.round_up(self.w12.in_features, 64)
            w12p, b12p, hp = packing.pack_w12(self.w12, dtype)
            if kin != self.w12.in_features:
                w12p = packing.pad_cols(w12p, kin, dtype)
            b3 = self.w3.bias if self.w3.bias is not None else torch.zeros(self.w3.out_features, device=w12p.device)
            pk = {"w12p": w12p, "b12p": b12p, "hp": hp, "w3p": packing.pack_w3(self.w3, hp, dtype),
                  "b3": packing.cast(b3, torch.float32)}
            self._pack = (stamp, pk)
        return self._pack[1]

    def forward(self, x):
        return self.run(x, residual=None)

    def run(self, x, residual=None):
        if not x.is_cuda:
            # parameters on the CPU: plain torch (reference modules/mlp.py:27-31: the FIRST half of w12's output is gated)
            x1, x2 = F.linear(x, self.w12.weight, self.w12.bias).chunk(2, dim=-1)
            out = F.linear(F.silu(x1) * x2, self.w3.weight, self.w3.bias)
            return out if residual is None else out + residual
        shape = x.shape
        dtype = x.dtype
        pk = self.packed(dtype)
        a = _rows(x.reshape(-1, shape[-1]), dtype)
        hidden = ops.gemm_swiglu(a, pk["w12p"], pk["b12p"])
        res = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        out_dtype = torch.float32 if residual is not None else dtype
        out = ops.gemm(hidden, pk["w3p"], bias=pk["b3"], residual=res, out_dtype=out_dtype)
        return out.reshape(shape[:-1] + (out.shape[-1],))


SwiGLU = SwiGLUFFN
XFORMERS_AVAILABLE = False


class SwiGLUFFNFused(SwiGLU):
    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        hidden_features = ops.swiglu_hidden(hidden_features)     # (int(h*2/3)+7)//8*8, mlp.py:53
        super().__init__(in_features=in_features, hidden_features=hidden_features, out_features=out_features, bias=bias)
