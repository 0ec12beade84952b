"""Host-side weight packing: reference state_dict layout -> the layouts the HIP kernels consume.

All packing is done once per (weights version, compute dtype) with ordinary torch tensor ops on the
device (cat / pad / cast): this is data movement, not the product's arithmetic.
"""
import torch

from .ops import round_up


def cast(t, dtype):
    return t.detach().to(dtype).contiguous()


def pad_cols(w, kpad, dtype):
    """[N,K] -> [N,kpad] zero-padded, cast."""
    w = w.detach()
    n, k = w.shape
    if k == kpad:
        return w.to(dtype).contiguous()
    out = torch.zeros(n, kpad, device=w.device, dtype=dtype)
    out[:, :k] = w.to(dtype)
    return out


def pack_qkv(to_q, to_k, to_v, dtype):
    """rows q | k | v (reference modules/attention.py:34-36), [3*inner, D]."""
    return torch.cat([to_q.weight.detach(), to_k.weight.detach(), to_v.weight.detach()], dim=0).to(dtype).contiguous()


def pack_w12(w12, dtype):
    """SwiGLU w12 (modules/mlp.py:24,28-29): interleave x1 / x2 rows in groups of 16, hidden padded to 64.

    Returns (w12p [2*Hp, D] dtype, b12p [2*Hp] fp32, Hp).  Padded rows are zero, so the padded hidden
    columns evaluate to silu(0)*0 = 0 exactly.
    """
    w = w12.weight.detach()
    two_h, d = w.shape
    h = two_h // 2
    hp = round_up(h, 64)
    b = w12.bias.detach() if w12.bias is not None else torch.zeros(two_h, device=w.device, dtype=w.dtype)

    def halves(t, tail):
        x1 = torch.zeros((hp,) + tail, device=w.device, dtype=torch.float32)
        x2 = torch.zeros((hp,) + tail, device=w.device, dtype=torch.float32)
        x1[:h] = t[:h].float()
        x2[:h] = t[h:].float()
        x1 = x1.reshape((hp // 16, 1, 16) + tail)
        x2 = x2.reshape((hp // 16, 1, 16) + tail)
        return torch.cat([x1, x2], dim=1).reshape((2 * hp,) + tail)

    return halves(w, (d,)).to(dtype).contiguous(), halves(b, ()).contiguous(), hp


def pack_w3(w3, hp, dtype):
    """w3 [D, H] -> [D, Hp] zero-padded in K (modules/mlp.py:25,31)."""
    return pad_cols(w3.weight, hp, dtype)


def ln_fold(w, gamma, beta, dtype=torch.bfloat16):
    """LayerNorm folded into the Linear that consumes it (include/pmhip.h, pmhip_lnfold):
    y . W^T = rstd * (x . Wg^T) - rstd * mean * c + d  with  Wg = gamma (.) W (rounded to `dtype`), c[n] = sum_k Wg[n,k]
    (of the ROUNDED values, so the algebra matches what the matrix core multiplies), d[n] = sum_k beta[k] W[n,k].
    w [N,K] in the packed row order the kernel consumes; returns (Wg [N,K] dtype, c [N] f32, d [N] f32)."""
    w32 = w.detach().float()
    wg = (w32 * gamma.detach().float()[None, :]).to(dtype).contiguous()
    c = wg.float().sum(1).contiguous()
    d = (w32 * beta.detach().float()[None, :]).sum(1).contiguous()
    return wg, c, d


def params_fingerprint(module):
    """Cheap identity+version stamp of every parameter: changes on load_state_dict / .to() / in-place edits."""
    return tuple((p.data_ptr(), p._version, p.dtype) for p in module.parameters())
