"""CPU ORACLE, second form -- test infrastructure, NOT product code.

The same restatement as paintmind_oracle.py, written with torch CPU tensor ops (fp32).  The
reference's CPU path *is* torch/ATen, so this port is what bench.py times for `cpu_baseline`
(kind "port"): same kernels (MKL/oneDNN), same thread count, no Python-module overhead.  It is pinned
by tests/test_oracle_vs_golden.py against the same golden vectors as the numpy oracle.
Parameters: dict name -> torch.float32 tensor keyed by the reference's state_dict names.
"""
import math

import torch
import torch.nn.functional as F


def attention(x, context, p, pre, heads):
    """modules/attention.py:43-59"""
    B, N, _ = x.shape
    c = x if context is None else context
    q, k, v = x @ p[pre + "to_q.weight"].T, c @ p[pre + "to_k.weight"].T, c @ p[pre + "to_v.weight"].T
    dh = q.shape[-1] // heads
    sp = lambda t: t.reshape(B, t.shape[1], heads, dh).permute(0, 2, 1, 3)
    q, k, v = sp(q) * dh ** -0.5, sp(k), sp(v)
    out = ((q @ k.transpose(-2, -1)).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, N, heads * dh)
    return out @ p[pre + "to_out.0.weight"].T + p[pre + "to_out.0.bias"]


def ffn(x, p, pre):
    """modules/mlp.py:27-31"""
    x1, x2 = (x @ p[pre + "w12.weight"].T + p[pre + "w12.bias"]).chunk(2, dim=-1)
    return (F.silu(x1) * x2) @ p[pre + "w3.weight"].T + p[pre + "w3.bias"]


def ln(x, p, pre):
    return F.layer_norm(x, x.shape[-1:], p[pre + "weight"], p[pre + "bias"], 1e-5)


def layer(x, context, p, pre, heads, stage2):
    """stage1/layers.py:54-58 ; stage2/transformer.py:44-49"""
    x = attention(ln(x, p, pre + "norm1."), None, p, pre + "attn1.", heads) + x
    if stage2:
        x = attention(ln(x, p, pre + "norm2."), context, p, pre + "attn2.", heads) + x
        return ffn(ln(x, p, pre + "norm3."), p, pre + "ffnet.") + x
    return ffn(ln(x, p, pre + "norm2."), p, pre + "ffnet.") + x


def vqgan_encode(img, p, cfg):
    """stage1/layers.py:106-112 ; stage1/vqmodel.py:21-25 ; stage1/quantize.py:18-38"""
    e = cfg["enc"]
    x = F.conv2d(img, p["encoder.to_patch_embedding.0.weight"], stride=e["patch_size"]).flatten(2).transpose(1, 2)
    x = ln(x + p["encoder.position_embedding"], p, "encoder.norm_pre.")
    for i in range(e["depth"]):
        x = layer(x, None, p, f"encoder.transformer.layers.{i}.", e["num_head"], False)
    z = F.normalize(x @ p["prev_quant.weight"].T + p["prev_quant.bias"], dim=-1)
    en = F.normalize(p["quantize.embedding.weight"], dim=-1)
    flat = z.reshape(-1, z.shape[-1])
    d = (flat ** 2).sum(1, keepdim=True) + (en ** 2).sum(1) - 2 * flat @ en.T
    idx = d.argmin(1).reshape(z.shape[:-1])
    zq = F.normalize(p["quantize.embedding.weight"][idx], dim=-1)
    loss = cfg["beta"] * ((zq - z) ** 2).mean() + ((zq - z) ** 2).mean()
    return z + (zq - z), loss, idx


def vqgan_decode(z, p, cfg):
    """stage1/vqmodel.py:27-30 ; stage1/layers.py:145-152"""
    d = cfg["dec"]
    x = z @ p["post_quant.weight"].T + p["post_quant.bias"] + p["decoder.position_embedding"]
    for i in range(d["depth"]):
        x = layer(x, None, p, f"decoder.transformer.layers.{i}.", d["num_head"], False)
    x = ln(x, p, "decoder.norm.") @ p["decoder.proj.weight"].T + p["decoder.proj.bias"]
    B, g, P, C = x.shape[0], d["image_size"] // d["patch_size"], d["patch_size"], d["out_channels"]
    x = x.reshape(B, g, g, P, P, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, g * P, g * P)
    return x.clamp(-1.0, 1.0)


def cond_transformer(tokens, context, p, s2cfg, pre="transformer."):
    """stage2/transformer.py:80-93"""
    x = tokens @ p[pre + "token_proj.weight"].T + p[pre + "token_proj.bias"] + p[pre + "position_embedding"]
    if context is not None and (pre + "context_proj.weight") in p:
        context = context @ p[pre + "context_proj.weight"].T
    for i in range(s2cfg["depth"]):
        x = layer(x, context, p, f"{pre}layers.layer{i}.", s2cfg["num_head"], True)
    return ln(x, p, pre + "norm.") @ p[pre + "to_logits.weight"].T + p[pre + "to_logits.bias"]


def sample_step(ids, mask_ratio, context, topk, temperature, noise, p, cfg, s2cfg, decode=True):
    """generate.py:159-181 (noise: uniform(0,1) tensor shaped like the logits)"""
    mask_id = cfg["n_embed"]
    table = torch.cat([p["vqgan.quantize.embedding.weight"], p["mask_token"]])
    logits = cond_transformer(table[ids], context, p, s2cfg)
    val, ind = logits.topk(topk, dim=-1)
    filt = torch.full_like(logits, float("-inf")).scatter_(2, ind, val)
    g = -torch.log((-torch.log(noise.clamp(min=1e-20))).clamp(min=1e-20))
    pred = (filt / max(temperature, 1e-10) + g).argmax(-1)
    img = None
    if decode:
        vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
        img = vqgan_decode(F.normalize(vq_p["quantize.embedding.weight"][pred], dim=-1), vq_p, cfg)
    is_mask = ids == mask_id
    ids = torch.where(is_mask, pred, ids)
    score = (1 - logits.softmax(-1).gather(2, pred[..., None]))[..., 0].masked_fill(~is_mask, -1e5)
    nm = max(int(mask_ratio * ids.shape[1]), 1)
    ids = ids.scatter(1, score.topk(nm, dim=-1).indices, mask_id)
    return ids, img, logits


def mask_schedule(ratio):
    return math.cos(math.pi / 2. * ratio)
