"""L2-normalised nearest-neighbour quantiser holder (reference stage1/quantize.py:8-44)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops, packing


class VectorQuantizer(nn.Module):
    def __init__(self, n_e, e_dim, beta=0.25):
        super().__init__()
        self.n_e, self.e_dim, self.beta = n_e, e_dim, beta
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.normal_()
        self._prep = None

    def prepared(self):
        """(normalised codebook, its squared row norms): hoisted out of the per-call path."""
        w = self.embedding.weight
        stamp = (w.data_ptr(), w._version)
        if self._prep is None or self._prep[0] != stamp:
            self._prep = (stamp, ops.vq_prepare(w.detach().float().contiguous()))
        return self._prep[1]

    def forward(self, z):
        """z fp32 [..., e_dim] -> (z + (z_q - z), loss, indices) exactly as quantize.py:18-38 returns them."""
        if not z.is_cuda:
            return self._forward_cpu(z)
        en, sq = self.prepared()
        lead = z.shape[:-1]
        z_out, idx, loss = ops.vq_quantize(z.contiguous().float().reshape(-1, self.e_dim), en, sq, self.beta)
        return z_out.reshape(z.shape), loss.reshape(()), idx.reshape(lead)

    def _forward_cpu(self, z):
        """Parameters on the CPU: reference stage1/quantize.py:18-38 in plain torch -- both sides l2-normalised (eps 1e-12),
        d = |z|^2 + |e|^2 - 2 z.e, first minimum, loss = beta * mean((zq - z)^2) + mean((zq - z)^2), straight-through value."""
        z = F.normalize(z.float(), p=2, dim=-1)
        zf = z.reshape(-1, self.e_dim)
        e = F.normalize(self.embedding.weight, p=2, dim=-1)
        d = zf.pow(2).sum(1, keepdim=True) + e.pow(2).sum(1) - 2 * zf @ e.t()
        idx = torch.argmin(d, dim=1).reshape(z.shape[:-1])
        zq = F.normalize(self.embedding(idx), p=2, dim=-1)
        loss = self.beta * torch.mean((zq.detach() - z) ** 2) + torch.mean((zq - z.detach()) ** 2)
        return z + (zq - z).detach(), loss, idx

    def decode_from_indice(self, indices):
        if not indices.is_cuda:
            return F.normalize(self.embedding(indices), p=2, dim=-1)
        en, _ = self.prepared()
        rows = ops.embed_rows(en, indices.contiguous().reshape(-1), self.e_dim, torch.float32)
        return rows.reshape(indices.shape + (self.e_dim,))
