"""L2-normalised nearest-neighbour quantiser holder (reference stage1/quantize.py:8-44)."""
import torch
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
        en, sq = self.prepared()
        lead = z.shape[:-1]
        z_out, idx, loss = ops.vq_quantize(z.contiguous().float().reshape(-1, self.e_dim), en, sq, self.beta)
        return z_out.reshape(z.shape), loss.reshape(()), idx.reshape(lead)

    def decode_from_indice(self, indices):
        en, _ = self.prepared()
        rows = ops.embed_rows(en, indices.contiguous().reshape(-1), self.e_dim, torch.float32)
        return rows.reshape(indices.shape + (self.e_dim,))
